import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location('tx', 'tests/test_xattn_gpu.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from dimsum_amd import native
from dimsum_amd.utils.tf32_emulation import round_tf32
for (B, L, heads, hd, sa) in [(2, 256, 8, 64, False), (1, 1024, 8, 72, False), (2, 100, 4, 24, False), (2, 256, 16, 64, True)]:
    W = 3 * heads * hd
    gen = torch.Generator().manual_seed(11 * L + hd)
    q1 = torch.randn(B, L, W, generator=gen); q2 = None if sa else torch.randn(B, L, W, generator=gen)
    b1 = torch.randn(W, generator=gen); b2 = None if sa else torch.randn(W, generator=gen)
    dout = torch.randn(B, L, (1 if sa else 2) * heads * hd, generator=gen) * 1e-3
    ref = m._attn_backward_f64(q1, q2, b1, b2, dout, heads)
    emu = m._attn_backward_f64(q1, q2, b1, b2, dout, heads, rnd=round_tf32)
    c = lambda t: None if t is None else t.cuda()
    out, lse = native.xattn_fusion_fwd(c(q1), c(q2), heads, need_lse=True, bias1=c(b1), bias2=c(b2), split_bf16=True)
    for f16 in (True, False):
        got = native.xattn_fusion_bwd(c(q1), c(q2), out, lse, c(dout), heads, bias1=c(b1), bias2=c(b2), f16=f16)
        for name, g, r_, e in zip(("dqkv1", "dqkv2"), got, ref, emu):
            if r_ is None: continue
            g = g.double().cpu()
            for part, sl in zip(("dq", "dk", "dv"), (slice(0, W // 3), slice(W // 3, 2 * W // 3), slice(2 * W // 3, W))):
                eg, ee, sc = (g[..., sl] - r_[..., sl]), (e[..., sl] - r_[..., sl]), r_[..., sl].abs().max().item()
                print((B, L, heads, hd, sa), "f16" if f16 else "split", name, part, "max %.2e emu %.2e | rms %.2e emu %.2e | scale %.2e" % (eg.abs().max().item(), ee.abs().max().item(), eg.pow(2).mean().sqrt().item(), ee.pow(2).mean().sqrt().item(), sc))
