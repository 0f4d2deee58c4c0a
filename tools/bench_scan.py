#!/usr/bin/env python3
"""Micro-benchmark of the selective-scan forward (default) or backward (--bwd) kernel at the BASELINE config-2/3 shape
(GPU box). The operands have the layouts MambaInnerFn produces (d-major delta / out / dout, z = half of xz)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dimsum_amd import native  # noqa: E402


def scan_bytes(B, D, L, N, G=1, s=4):
    # 5 B D L s + 2 B G N L s + B D ceil(L/2048) 2N 4 + (D N + 2 D) 4
    return 5 * B * D * L * s + 2 * B * G * N * L * s + B * D * ((L + 2047) // 2048) * 2 * N * 4 + (D * N + 2 * D) * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--D", type=int, default=1024)
    ap.add_argument("--L", type=int, default=256)
    ap.add_argument("--N", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--dmajor", action="store_true", help="u and z d-major too, exactly like inside MambaInnerFn (xz = in_proj GEMM view)")
    ap.add_argument("--contig", action="store_true", help="every activation (u, delta, z, out, dout) contiguous (B, D, L): the layout probe "
                                                          "against MambaInnerFn's d-major tensors (channel stride B * L)")
    ap.add_argument("--bwd", action="store_true", help="time selective_scan_bwd (with the saved states of the forward)")
    ap.add_argument("--no-ckpt", action="store_true", help="--bwd without saved states: the kernel pair of the reference-shaped call")
    ap.add_argument("--no-out-z", action="store_true", help="--bwd without the out_z recompute (what dimsum_amd.ops requests: it keeps the forward's out_z)")
    ap.add_argument("--pad", type=int, default=0, help="--dmajor: extra elements in the channel stride (B*L + pad): probes L2 / HBM channel camping "
                                                       "of power-of-two strides")
    ap.add_argument("--variant", type=int, default=-1, help="ask for a forward kernel: lanes per channel 1 (64 channels per wave) / 2 / 4 / 16 (0 or -1: automatic)")
    ap.add_argument("--infer", action="store_true", help="forward as the model calls it at inference: no `out` / `x` stores (1.082 GB at the default shape)")
    ap.add_argument("--train-fwd", action="store_true", help="time the forward's training variant (also stores the states the backward consumes)")
    ap.add_argument("--z16", action="store_true", help="--dt-fused with out_z as block-scaled fp16 (the launch of the headline forward: out_proj's operand image)")
    ap.add_argument("--dt-fused", action="store_true", help="forward with dt_proj inside the kernel (implies --infer): delta = W_dt x_dbl[:32] per tile on the matrix cores")
    a = ap.parse_args()
    from dimsum_amd import _lib
    native._scan_fwd_variant = max(a.variant, 0)          # per-call field of the C ABI (dimsum_ssm_ext_t.kernel_variant)
    dt = getattr(torch, a.dtype)
    B, D, L, N = a.B, a.D, a.L, a.N
    dev = "cuda"
    xz = torch.randn(B, 2 * D, L, device=dev, dtype=dt)
    u = torch.randn(B, D, L, device=dev, dtype=dt)
    def dm(t):          # (D, B, L) -> d-major (B, D, L) view with channel stride B * L + pad
        if a.pad == 0:
            return t.permute(1, 0, 2)
        buf = torch.empty(t.shape[0], B * L + a.pad, device=dev, dtype=t.dtype)
        v = buf[:, :B * L].view(t.shape[0], B, L)
        v.copy_(t)
        return v.permute(1, 0, 2)
    delta = dm((0.5 * torch.rand(D, B, L, device=dev)).to(dt))   # d-major like MambaInnerFn
    A = -0.5 * torch.rand(D, N, device=dev)
    Bm, Cm = torch.randn(B, 1, N, L, device=dev, dtype=dt), torch.randn(B, 1, N, L, device=dev, dtype=dt)
    Dv, bias = torch.randn(D, device=dev), 0.5 * torch.rand(D, device=dev)
    z = xz.chunk(2, 1)[1]
    if a.dmajor:
        xz = dm(torch.randn(2 * D, B, L, device=dev, dtype=dt))                      # strides (L, B L, 1)
        z = xz.chunk(2, 1)[1]
        u = dm(u.permute(1, 0, 2).contiguous())                                      # conv_out = empty_like(x): d-major
    if a.contig:
        delta = delta.contiguous()
        xz = torch.randn(B, 2 * D, L, device=dev, dtype=dt)
        z = xz.chunk(2, 1)[1]
        u = torch.randn(B, D, L, device=dev, dtype=dt)
        dm = lambda t: t.permute(1, 0, 2).contiguous()      # noqa: E731  (dout below)
    if a.bwd:
        out, x, out_z, ckpt = native.selective_scan_fwd(u, delta, A, Bm, Cm, Dv, z, bias, True, need_ckpt=True)
        dout = dm(torch.randn(D, B, L, device=dev).to(dt))
        dxz = torch.empty_like(xz)
        dz = dxz.chunk(2, 1)[1]
        ck = None if a.no_ckpt else ckpt

        def call():
            return native.selective_scan_bwd(u, delta, A, Bm, Cm, Dv, z, bias, dout, x, out, dz, True, not a.no_out_z, ckpt=ck)
        s_ = u.element_size()       # (8 + out_z recompute) B D L s + 2 B N L (s + 4) + x
        nbytes = (8 if a.no_out_z else 9) * B * D * L * s_ + 2 * B * N * L * (s_ + 4) + B * D * ((L + 2047) // 2048) * 2 * N * 4 + (D * N + 2 * D) * 4
    elif a.dt_fused or a.z16:
        R = 32
        dt_w, dt_xt = torch.randn(D, R, device=dev) * 0.1, torch.randn(R + 2 * N, B * L, device=dev)[:R]

        def call():
            return native.selective_scan_fwd(u, None, A, Bm, Cm, Dv, z, bias, True, need_out=False, need_x=False, dt_proj=(dt_w, dt_xt), **({"out_z_f16": True} if a.z16 else {}))
        nbytes = 3 * B * D * L * 4 + 2 * B * N * L * 4 + (D * N + 2 * D) * 4 + (B * L * R + D * R) * 4 - ((B * D * L * 2 - B * D * L // 2048 * 4) if a.z16 else 0)
    else:
        def call():
            return native.selective_scan_fwd(u, delta, A, Bm, Cm, Dv, z, bias, True, need_out=not a.infer, need_x=not a.infer, need_ckpt=a.train_fwd)
        nbytes = scan_bytes(B, D, L, N, 1, u.element_size()) - ((B * D * L * u.element_size() + B * D * ((L + 2047) // 2048) * 2 * N * 4) if a.infer else 0)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
    for s, e in evs:
        s.record()
        call()
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    med = ms[len(ms) // 2]
    # this box's own copy bandwidth, same process (fractions normalised per box)
    ca, cb = torch.empty(1 << 28, device="cuda"), torch.empty(1 << 28, device="cuda")
    cb.copy_(ca)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(5):
        cb.copy_(ca)
    c1.record()
    torch.cuda.synchronize()
    copy_gbps = 5 * 2 * 4 * (1 << 28) / (c0.elapsed_time(c1) * 1e-3) / 1e9
    P = _lib.SsmParams()
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups, P.n_chunks = B, D, L, N, 1, (L + 2047) // 2048
    _lib.attach_ext(P, _lib.SsmExt).kernel_variant = native._scan_fwd_variant
    print(json.dumps({"kernel": "bwd" if a.bwd else "fwd", "fwd_variant": _lib.load().dimsum_ssm_scan_fwd_variant(P), "shape": [B, D, L, N], "dtype": a.dtype, "ms_median": med, "ms_min": ms[0], "algorithmic_GB": nbytes / 1e9,
                      "GBps": nbytes / med / 1e6, "frac_of_8TBps": nbytes / med / 1e6 / 8000,
                      "box_copy_GBps": copy_gbps, "frac_of_box_copy": nbytes / med / 1e6 / copy_gbps, "timed_by": "HIP events, this process"}))


if __name__ == "__main__":
    main()
