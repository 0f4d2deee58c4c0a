#!/usr/bin/env python3
"""Where do the torch-native elementwise / reduction launches of one DiMBlockCombined(1024) forward+backward come from? (GPU box)
A TorchDispatchMode logs every aten op that is not a matmul with its input shapes and the innermost dimsum_amd frame of the Python stack
(autograd-engine nodes without Python frames show as <autograd>)."""
import collections
import importlib.util
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

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
SKIP = ("mm", "bmm", "addmm", "view", "_unsafe_view", "t", "transpose", "detach", "alias", "as_strided", "slice", "select", "expand", "unsqueeze", "squeeze",
        "permute", "reshape", "split", "chunk", "unbind", "empty", "empty_like", "empty_strided", "new_empty", "zeros", "new_zeros", "_reshape_alias", "unsafe_split", "split_with_sizes", "unsafe_chunk", "narrow", "is_same_size", "stride", "sym_size", "sym_stride", "size")
log = collections.defaultdict(int)


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in SKIP:
            big = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor) and a.numel() >= 1 << 20]
            if not big and isinstance(args[0] if args else None, (list, tuple)):
                big = [tuple(a.shape) for a in args[0] if isinstance(a, torch.Tensor) and a.numel() >= 1 << 20]
            if big:
                fr = [f for f in traceback.extract_stack() if "dimsum_amd" in f.filename]
                where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].name}" if fr else "<autograd>"
                log[(name, str(big)[:80], where)] += 1
        return func(*args, **(kwargs or {}))


def step():
    for p_ in model.parameters():
        p_.grad = None
    hs.grad = res.grad = cond.grad = None
    out, res_out = model(hs, res, cond)
    torch.autograd.backward((out, res_out), (dy, dy))


step()
with Log():
    step()
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print(f"x{v:2d} {k[0]:18s} {k[1]:82s} {k[2]}")
