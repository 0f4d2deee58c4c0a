import sys, torch
sys.path.insert(0, ".")
from dimsum_amd import native
torch.manual_seed(0)
M, F, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(M, K, device="cuda"); w = torch.randn(2 * F, K, device="cuda") * K ** -0.5
a = native.rows_f16s(x); b, l1 = native.rows_f16s(w, want_l1=True)
b12 = torch.randn(2 * F, device="cuda") * 0.1
kw = dict(scales=(a.inv, b.inv), bias=b12, epilogue="gated_f16", gate_bound=torch.cat([l1 * (1 + 2.0 ** -10), b12.abs().max().reshape(1)]).contiguous())
r0 = native.gemm_nt(a.data, b.data, tune=(513, 0, 0), **kw)
r1 = native.gemm_nt(a.data, b.data, tune=(514, 0, 0), **kw)
torch.cuda.synchronize()
d = (r0.data != r1.data)
print("mismatch", d.sum().item(), "of", d.numel(), "inv equal", torch.equal(r0.inv, r1.inv))
if d.any():
    rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
    print("rows", rows[:8].tolist(), "...", rows[-4:].tolist(), len(rows), "cols", cols[:8].tolist(), "...", cols[-4:].tolist(), len(cols))
    tm = (d.view(M // 256, 256, F // 128, 128).any(3).any(1)).nonzero()
    print("bad tiles (m, n):", tm[:20].tolist(), len(tm))
    i = d.nonzero()[0]; print("first", i.tolist(), r0.data[i[0], i[1]].item(), r1.data[i[0], i[1]].item())
    # pattern inside the first bad tile
    t = tm[0]; blk = d[t[0]*256:(t[0]+1)*256, t[1]*128:(t[1]+1)*128]
    print("rows in tile", blk.any(1).nonzero().flatten().tolist()[:40]); print("cols in tile", blk.any(0).nonzero().flatten().tolist()[:40])
