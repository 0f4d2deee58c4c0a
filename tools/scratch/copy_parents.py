#!/usr/bin/env python3
"""which top-level ops own the copy / fill kernels of a training step (GPU box): torch.profiler events, kernels matched to their CPU op's ancestor chain"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dimsum_amd import gemm
from dimsum_amd.train import build_training, train_step
from dimsum_amd.transport import create_transport
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = True
gemm.set_policy("f16s")
model = bench.build_model("DiM-L/2", dev, 256)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(64, 4, 32, 32, device=dev, generator=g); y = torch.randint(0, 1000, (64,), device=dev, generator=g)
ddp, ema, opt = build_training(model.train(), dev, 1e-4, 1, [0])
tr = create_transport("GVP", "velocity")
for _ in range(3):
    train_step(ddp, ema, opt, tr, x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(ddp, ema, opt, tr, x, y)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    ks = [k for k in ev.kernels if any(s in k.name for s in ("copyBuffer", "direct_copy", "FillFunctor", "fillBuffer", "Memcpy", "Memset"))]
    if not ks:
        continue
    chain, p = [ev.name], ev.cpu_parent
    while p is not None:
        chain.append(p.name)
        p = p.cpu_parent
    key = " < ".join(chain[:5]) + "  " + str(ev.input_shapes)[:60]
    agg[key][0] += len(ks)
    agg[key][1] += sum(k.duration for k in ks)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n:5d} {t / 1e3:7.3f} ms  {k[:230]}")
