#!/usr/bin/env python3
"""autograd node census of one DiM-L/2 training loss (GPU box): which C++ backward nodes (slice / select / copy / expand ...) sit in the graph"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dimsum_amd import gemm
from dimsum_amd.transport import create_transport
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = True
gemm.set_policy("f16s")
model = bench.build_model("DiM-L/2", dev, 256).train()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(64, 4, 32, 32, device=dev, generator=g); y = torch.randint(0, 1000, (64,), device=dev, generator=g)
loss = create_transport("GVP", "velocity").training_losses(model, x, dict(y=y))["loss"].mean()
seen, todo, cnt = set(), [loss.grad_fn], collections.Counter()
while todo:
    n = todo.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    cnt[type(n).__name__] += 1
    todo.extend(f for f, _ in n.next_functions)
print(", ".join(f"{k} {v}" for k, v in cnt.most_common(80)))
