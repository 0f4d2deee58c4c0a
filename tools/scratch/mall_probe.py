#!/usr/bin/env python3
"""Does a producer GEMM -> gated-GeLU consumer pair run faster when it is cut into row chunks whose intermediate
(rows x 8192 fp32) fits the 256 MB Infinity Cache? (GPU box) Times the w12 GEMM + gated-GeLU pass of DiM-L/2 at batch 256:
whole (2.1 GB intermediate) vs chunked over rows."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dimsum_amd import native

torch.backends.cuda.matmul.allow_tf32 = True
M, K, H = 256 * 256, 1024, 4096
x = torch.randn(M, K, device="cuda")
W = torch.randn(2 * H, K, device="cuda") * 0.02
bias = torch.randn(2 * H, device="cuda")


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


y = torch.empty(M, 2 * H, device="cuda")
h = torch.empty(M, H, device="cuda")


def whole():
    torch.mm(x, W.t(), out=y)
    return native.gated_gelu_fwd(y, bias)


print("whole: gemm %.3f ms, gelu %.3f ms, both %.3f ms" % (timed(lambda: torch.mm(x, W.t(), out=y)), timed(lambda: native.gated_gelu_fwd(y, bias)), timed(whole)))
for rows in (2048, 4096, 8192, 16384):
    yc = torch.empty(rows, 2 * H, device="cuda")

    def chunked():
        outs = []
        for r0 in range(0, M, rows):
            torch.mm(x[r0:r0 + rows], W.t(), out=yc)
            outs.append(native.gated_gelu_fwd(yc, bias))
        return outs
    print("chunks of %5d rows (%.0f MB intermediate): %.3f ms" % (rows, rows * 2 * H * 4 / 1e6, timed(chunked, 5)))
