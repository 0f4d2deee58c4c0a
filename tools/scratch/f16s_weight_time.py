import torch, sys, os
sys.path.insert(0, os.getcwd())
from dimsum_amd import native
w = torch.randn(8192, 1024, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("rows_f16s 8192x1024 us: plain", t(lambda: native.rows_f16s(w)), "with l1", t(lambda: native.rows_f16s(w, want_l1=True)))
a, l1 = native.rows_f16s(w, want_l1=True)
print("l1 ok", torch.allclose(l1, w.abs().sum(1).max().reshape(1)))
