"""token_transform (kind none) with and without the per-(batch, channel) reductions at the training step's shapes: what the atomics cost"""
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
for B, L, C in ((64, 256, 512), (64, 256, 1024), (256, 256, 512), (256, 256, 1024)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x, w = torch.randn(B, L, C, device="cuda", generator=g), torch.randn(B, L, C, device="cuda", generator=g)
    sc = torch.randn(B, C, device="cuda", generator=g)
    plain = t(lambda: native.token_transform(x, "none", True, scale=sc))
    red = t(lambda: native.token_transform(x, "none", True, scale=sc, w=w))
    red3 = t(lambda: native.token_transform(x, "none", True, scale=sc, w=w, want_wsum=True, want_tsum=True))
    ronly = t(lambda: native.token_transform(x, "none", True, w=w, want_y=False, want_wsum=True))
    mb = B * L * C * 4 / 1e6
    print(f"({B},{L},{C}) {mb:.0f} MB per tensor: y only {plain:.1f} us | y + wdot {red:.1f} | y + 3 sums {red3:.1f} | sums only {ronly:.1f}")
