"""tools/scratch/gated_parts.py (GPU box): the w12 + gate launch under whatever library DIMSUM_HIP_LIB names: persistent and one tile per workgroup"""
import json, sys, torch
sys.path.insert(0, ".")
from dimsum_amd import native
def t(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
M, F, K = 65536, 4096, 1024
torch.manual_seed(0)
x = native.rows_f16s(torch.randn(M, K, device="cuda")); w, l1 = native.rows_f16s(torch.randn(2 * F, K, device="cuda") * K ** -0.5, want_l1=True)
b12 = torch.randn(2 * F, device="cuda") * 0.1
kw = dict(scales=(x.inv, w.inv), bias=b12, epilogue="gated_f16", gate_bound=torch.cat([l1 * (1 + 2.0 ** -10), b12.abs().max().reshape(1)]).contiguous())
r = {}
for tag, tv in (("one_tile_per_wg", 513), ("persistent", 514)):
    r[tag] = round(sorted(t(lambda: native.gemm_nt(x.data, w.data, tune=(tv, 0, 0), **kw)) for _ in range(5))[2], 4)
print(json.dumps(r))
