"""norm forward at the headline's two launch shapes: the prenorm (x + residual -> y, residual_out; all fp32) and the norm_2 pass (x + bias, RMS, modulate ->
scaled-fp16 image + residual_out)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dimsum_amd import native
def t(f, n=20):
    for _ in range(3): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2] * 1e3
g = torch.Generator(device="cuda").manual_seed(0)
for rows, N, L in ((65536, 1024, 256), (65536, 1152, 1024), (16384, 1024, 256)):
    x, res = torch.randn(rows, N, device="cuda", generator=g), torch.randn(rows, N, device="cuda", generator=g)
    w, xb = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    sc, sh = 0.1 * torch.randn(rows // L, N, device="cuda", generator=g), torch.randn(rows // L, N, device="cuda", generator=g)
    a = t(lambda: native.layer_norm_fwd(x, w, None, 1e-5, res, residual_dtype=torch.float32, is_rms_norm=True))
    b = t(lambda: native.layer_norm_fwd(x, w, None, 1e-5, is_rms_norm=True, x_bias=xb, mod_scale=sc, mod_shift=sh, rows_per_batch=L, split3="f16s"))
    mb = rows * N * 4 / 1e6
    print(f"({rows}, {N}): prenorm {a:.1f} us = {4 * mb / a / 1e6:.2f} TB/s | norm_2 + image {b:.1f} us = {2.5 * mb / b / 1e6:.2f} TB/s")
