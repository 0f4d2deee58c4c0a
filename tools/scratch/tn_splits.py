"""dW-shaped TN products with row factors: time (GEMM + the sum over the partial results) against the number of reduction ranges"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dimsum_amd import native
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (16384, 65536):
    for name, P, Q in (("dW12", 8192, 1024), ("dW3", 1024, 4096), ("dWqkv", 1536, 512), ("dWproj", 1024, 1024)):
        a, b = native.rows_f16s(torch.randn(M, P, device="cuda")), native.rows_f16s(torch.randn(M, Q, device="cuda"))
        rs = native.row_factors(a.inv, b.inv)
        auto = native.gemm_tn_splits(M, P, Q)
        row = [f"{name} M={M} tiles={(P // 256) * (Q // 256)} auto={auto}"]
        for s in (1, 2, 4, 8, 16):
            if M // s > 16384 or M // s < 2048: row.append(f"s{s}: -"); continue
            row.append(f"s{s}: {t(lambda: native.gemm_tn(a.data, b.data, splits=s, row_scales=rs)):6.1f}")
        print("  ".join(row))
