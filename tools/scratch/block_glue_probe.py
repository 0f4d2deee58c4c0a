#!/usr/bin/env python3
"""Which torch-native kernels does one DiMBlockCombined(1024) forward+backward launch, and from where? (GPU box)
torch.profiler with stacks, grouped by (aten op, input shapes, innermost dimsum_amd frame)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"))
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
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith("aten::") or ev.name in ("aten::mm", "aten::bmm", "aten::addmm", "aten::matmul", "aten::linear"):
        continue
    if ev.cpu_children and any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue          # count leaves only
    frame = next((f for f in (ev.stack or []) if "dimsum_amd" in f), (ev.stack or ["?"])[0] if ev.stack else "?")
    key = (ev.name, str(ev.input_shapes)[:90], frame.split("dimsum_amd/")[-1][:70])
    agg[key][0] += 1
    agg[key][1] += ev.device_time_total
tot = sum(v[1] for v in agg.values())
print(f"torch-native leaf ops: {tot / 1e3:.2f} ms of device time")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / 1e3:7.3f} ms x{v[0]:3d}  {k[0]:22s} {k[1]:92s} {k[2]}")
