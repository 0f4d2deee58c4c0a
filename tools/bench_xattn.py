#!/usr/bin/env python3
"""Micro-benchmark of the cross-attention fusion forward kernel at the model's launch shapes (GPU box):
DiM-L/2 (256 latents, 256 tokens, 8 heads x 64) and DiM-XL/2 at 512 px (64 latents, 1024 tokens, 8 heads x 72)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dimsum_amd import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--L", type=int, default=256)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--hd", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--exact", action="store_true", help="exact fp32 MFMA kernel instead of the split-bf16 one")
    ap.add_argument("--bwd", action="store_true", help="time the backward pair (xattn_bwd_dq + xattn_bwd_dkv) instead of the forward")
    ap.add_argument("--split3", action="store_true", help="write the output as the split-bf16 operand image of the proj Linear")
    ap.add_argument("--f16", action="store_true", help="the single-product fp16 kernel as the headline forward runs it: q | k | v as the scaled fp16 of the "
                                                       "qkv GEMMs' F16_QKV epilogue, the output as the scaled-fp16 image of proj (csrc/xattn_fusion_f16.hip)")
    a = ap.parse_args()
    W = 3 * a.heads * a.hd
    g = torch.Generator(device="cuda").manual_seed(0)
    q1, q2 = torch.randn(a.B, a.L, W, device="cuda", generator=g), torch.randn(a.B, a.L, W, device="cuda", generator=g)
    b1, b2 = torch.randn(W, device="cuda", generator=g), torch.randn(W, device="cuda", generator=g)
    f = lambda: native.xattn_fusion_fwd(q1, q2, a.heads, bias1=b1, bias2=b2, split_bf16=not a.exact, split3=a.split3)
    if a.f16:
        from dimsum_amd import gemm
        C = a.heads * a.hd
        xs = [torch.randn(a.B * a.L, C, device="cuda", generator=g) for _ in range(2)]
        ws = [torch.randn(3 * C, C, device="cuda", generator=g) * C ** -0.5 for _ in range(2)]
        imgs = [native.rows_f16s(x) for x in xs]
        bound = gemm.attn_kv_bound(ws[0], b1, ws[1], b2)
        q16 = [gemm.qkv_f16s(i, w, b_, a.L, bound[2 * n:2 * n + 2]).view(a.B, a.L, 3 * C) for n, (i, w, b_) in enumerate(zip(imgs, ws, (b1, b2)))]
        sc = (imgs[0].inv.reshape(a.B, a.L), imgs[1].inv.reshape(a.B, a.L), bound)
        f = lambda: native.xattn_fusion_fwd(q16[0], q16[1], a.heads, split3="f16s", f16s=sc)
    if a.bwd:
        out, lse = native.xattn_fusion_fwd(q1, q2, a.heads, need_lse=True, bias1=b1, bias2=b2, split_bf16=not a.exact)
        dout = torch.randn(out.shape, device="cuda", generator=g)
        f = lambda: native.xattn_fusion_bwd(q1, q2, out, lse, dout, a.heads, bias1=b1, bias2=b2, split_bf16=not a.exact, f16=a.f16)      # --f16: ONE fp16 product (precision 2)
    for _ in range(3):
        f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        f()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
    flop = 2 * 4 * a.B * a.heads * a.L * a.L * a.hd * (3.5 if a.bwd else 1.0)      # backward: 3 + 4 GEMM-equivalents against the forward's 2
    med = ms[len(ms) // 2]
    print(json.dumps({"kernel": ("xattn_bwd" if a.bwd else "xattn_fwd") + ("_f16" if a.f16 else "") + ("" if not a.exact else "_exact") + ("_split3" if a.split3 else ""), "shape": [a.B, a.L, a.heads, a.hd], "ms_median": med, "ms_min": ms[0],
                      "GFLOP": flop / 1e9, "TFLOPs_equivalent": flop / med / 1e9}))


if __name__ == "__main__":
    main()
