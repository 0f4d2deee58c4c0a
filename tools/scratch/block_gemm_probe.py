#!/usr/bin/env python3
"""Every GEMM launch of one DiMBlockCombined(1024) forward+backward: aten op, input shapes, device time, calling frame (GPU box)."""
import collections
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
torch.backends.cuda.matmul.allow_tf32 = True
dev = torch.device("cuda", 0)
model, hidden = bench.build_block("DiM-L/2", dev)
g = torch.Generator(device=dev).manual_seed(0)
B, L = 256, 256
hs = torch.randn(B, L, hidden, device=dev, generator=g).requires_grad_()
res = torch.randn(B, L, hidden, device=dev, generator=g).requires_grad_()
cond = torch.randn(B, hidden, device=dev, generator=g).requires_grad_()
dy = torch.randn(B, L, hidden, device=dev, generator=g)


def step():
    for p_ in model.parameters():
        p_.grad = None
    hs.grad = res.grad = cond.grad = None
    out, res_out = model(hs, res, cond)
    torch.autograd.backward((out, res_out), (dy, dy))


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name in ("aten::mm", "aten::bmm", "aten::addmm") and ev.device_time_total > 0:
        rows.append((ev.device_time_total, ev.name, str(ev.input_shapes)))
tot = sum(r[0] for r in rows)
print(f"library GEMMs: {tot / 1e3:.2f} ms")
for t, n, s in sorted(rows, reverse=True):
    if t > 20:
        print(f"{t / 1e3:7.3f} ms {n:10s} {s}")
kern = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kern[ev.name[:90]][0] += 1
        kern[ev.name[:90]][1] += ev.device_time_total
print("kernels:")
for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{v[1] / 1e3:7.3f} ms x{v[0]:3d} {k}")
