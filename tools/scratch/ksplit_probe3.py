#!/usr/bin/env python3
"""weight-gradient GEMMs (reduction over the 65536 rows): fp32 library split path vs one bf16 GEMM over row-stacked images. GPU box."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = True
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def img(x, left):
    hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
    return torch.cat([hi, hi, lo] if left else [hi, lo, hi], 1).contiguous()
for (M, K1, K2) in [(65536, 8192, 1024), (65536, 1024, 4096), (65536, 1024, 1024)]:
    a, b = torch.randn(M, K1, device="cuda"), torch.randn(M, K2, device="cuda")
    t0 = timeit(lambda: torch.mm(a.t(), b))
    a3, b3 = img(a, False).view(3 * M, K1), img(b, True).view(3 * M, K2)
    t1 = timeit(lambda: torch.mm(a3.t(), b3, out_dtype=torch.float32))
    ref = torch.mm(a.t().double(), b.double())
    e0 = ((torch.mm(a.t(), b) - ref).abs().max() / ref.abs().max()).item(); e1 = ((torch.mm(a3.t(), b3, out_dtype=torch.float32) - ref).abs().max() / ref.abs().max()).item()
    print((M, K1, K2), "dW fp32-split %.3f ms | images %.3f ms | err %.1e / %.1e" % (t0, t1, e0, e1), flush=True)
    # input gradient: dX = dY (M, K1) @ W (K1, K2)
    w = torch.randn(K1, K2, device="cuda")
    t2 = timeit(lambda: torch.mm(a, w))
    a3l, wt3 = img(a, False), img(w.t().contiguous(), True)
    t3 = timeit(lambda: torch.mm(a3l, wt3.t(), out_dtype=torch.float32))
    print("      dX fp32-split %.3f ms | images %.3f ms" % (t2, t3), flush=True)
