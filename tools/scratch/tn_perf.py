#!/usr/bin/env python3
"""TN (weight-gradient) products of the DiM-L/2 block at batch 256: the library's batched bf16 TN GEMM + sum (gemm.mm_tn before round 4's
kernel) against native.gemm_tn, interleaved on one box. (GPU box)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dimsum_amd import gemm, native


def lib_tn(a, b):
    R, N = a.shape
    K = b.shape[1]
    s = gemm._slices(R, N, K)
    if s == 1:
        return torch.mm(a.t(), b, out_dtype=torch.float32)
    return torch.bmm(a.view(s, R // s, N).transpose(1, 2), b.view(s, R // s, K), out_dtype=torch.float32).sum(0)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    return ms[len(ms) // 2]


M = 65536
for name, P, Q in (("dW12", 8192, 1024), ("dW3", 1024, 4096), ("dWproj", 1024, 1024), ("dWqkv", 1536, 512)):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.randn((3 * M, P), device="cuda", generator=g)).to(torch.bfloat16)
    b = (torch.randn((3 * M, Q), device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    ref = lib_tn(a, b)
    row = {"shape": name, "R": 3 * M, "P": P, "Q": Q, "library_ms": t(lambda: lib_tn(a, b)), "library_slices": gemm._slices(3 * M, P, Q)}
    for s in (1, 2, 4, 8, 16):
        if (P // 256) * (Q // 256) * s > 2048:
            continue
        got = native.gemm_tn(a, b, splits=s)
        row[f"tn_s{s}_ms"] = t(lambda: native.gemm_tn(a, b, splits=s))
        row[f"tn_s{s}_relerr_vs_lib"] = ((got - ref).abs().max() / ref.abs().max()).item()
    row["auto_splits"] = native.gemm_tn_splits(3 * M, P, Q)
    flops = 2.0 * 3 * M * P * Q
    best = min(v for k, v in row.items() if k.startswith("tn_s") and k.endswith("_ms"))
    row["library_TF"], row["tn_best_TF"] = flops / row["library_ms"] / 1e9, flops / best / 1e9
    print(json.dumps(row), flush=True)
    del a, b
