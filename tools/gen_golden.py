#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by importing the REFERENCE (read-only, /root/reference) with the
shims of tools/ref_shim.py. Runs in the build container only; the fixtures (data: inputs + expected outputs) are
committed, the reference never travels. Re-run:  python tools/gen_golden.py [--only NAME ...]

Every fixture records which reference function produced it (file:line) in its `__doc__` field.
Inputs follow the reference tests' distributions (mamba/tests/ops/test_selective_scan.py:62-95,
causal-conv1d/tests/test_causal_conv1d.py:39-49).
"""
import argparse
import hashlib
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
warnings.filterwarnings("ignore", category=FutureWarning)

import ref_shim  # noqa: E402
from procedural import procedural_fill, seeded, toy_denoiser  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
T = torch.from_numpy


def save(name, doc, **arrs):
    path = os.path.join(OUT, name + ".npz")
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(path, __doc__=np.array(doc), **conv)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.0f} KiB")


# ---------------------------------------------------------------------------------------------------------------
def gen_scan(ns):
    """selective_scan_ref fwd + autograd grads (selective_scan_interface.py:104-171)."""
    cases = {
        # name: (B, D, L, N, has_D, has_z, has_bias, softplus, groups)
        "scan_main": (2, 72, 256, 16, True, True, True, True, 1),
        "scan_long": (1, 8, 4096, 16, True, True, True, True, 1),
        "scan_odd": (2, 5, 151, 8, True, True, True, True, 1),
        "scan_plain": (2, 4, 64, 8, False, False, False, False, 1),
        "scan_nosoftplus_z": (2, 4, 128, 8, True, True, False, False, 1),
        "scan_groups2": (2, 4, 128, 8, True, True, True, True, 2),
    }
    for name, (B, D, L, N, has_D, has_z, has_b, sp, G) in cases.items():
        torch.manual_seed(0)
        A = (-0.5 * torch.rand(D, N)).requires_grad_()
        Bm = torch.randn(B, G, N, L, requires_grad=True)
        Cm = torch.randn(B, G, N, L, requires_grad=True)
        Dv = torch.randn(D, requires_grad=True) if has_D else None
        z = torch.randn(B, D, L, requires_grad=True) if has_z else None
        db = (0.5 * torch.rand(D)).requires_grad_() if has_b else None
        u = torch.randn(B, D, L, requires_grad=True)
        delta = (0.5 * torch.rand(B, D, L)).requires_grad_()
        out, last = ns.ssi.selective_scan_ref(u, delta, A, Bm, Cm, Dv, z, db, sp, return_last_state=True)
        g = torch.randn_like(out)
        out.backward(g)
        # `out` before gating (the CUDA kernel's first output) = ref without z
        with torch.no_grad():
            y = ns.ssi.selective_scan_ref(u, delta, A, Bm, Cm, Dv, None, db, sp)
        arrs = dict(u=u, delta=delta, A=A, B=Bm, C=Cm, dout=g, out=out, y=y, last_state=last,
                    du=u.grad, ddelta=delta.grad, dA=A.grad, dB=Bm.grad, dC=Cm.grad,
                    softplus=np.array(sp))
        if has_D:
            arrs.update(D=Dv, dD=Dv.grad)
        if has_z:
            arrs.update(z=z, dz=z.grad)
        if has_b:
            arrs.update(delta_bias=db, ddelta_bias=db.grad)
        save(name, "selective_scan_ref fwd + autograd (mamba/mamba_ssm/ops/selective_scan_interface.py:104-171); "
             "distributions of mamba/tests/ops/test_selective_scan.py:62-95", **arrs)


def gen_conv(ns):
    """causal_conv1d_ref fwd + autograd (causal_conv1d_interface.py:48-64)."""
    cases = {
        "conv_L8_w4_silu": (2, 72, 8, 4, True, True),
        "conv_L151_w4_silu": (2, 72, 151, 4, True, True),
        "conv_L256_w4_silu": (2, 72, 256, 4, True, True),
        "conv_L1134_w4_silu": (2, 8, 1134, 4, True, True),
        "conv_L256_w3_nosilu": (2, 8, 256, 3, False, True),
        "conv_L64_w2_nobias": (2, 8, 64, 2, True, False),
        "conv_L4096_w4_silu": (1, 4, 4096, 4, True, True),
    }
    for name, (B, D, L, W, silu, has_bias) in cases.items():
        torch.manual_seed(0)
        # strided view, like test_causal_conv1d.py:39-49 and like x = xz.chunk(2,1)[0]
        xz = torch.randn(B, 2 * D, L)
        x = xz[:, :D].detach().clone().requires_grad_()
        w = torch.randn(D, W, requires_grad=True)
        b = torch.randn(D, requires_grad=True) if has_bias else None
        out = ns.cci.causal_conv1d_ref(x, w, b, "silu" if silu else None)
        g = torch.randn_like(out)
        out.backward(g)
        arrs = dict(x=x, weight=w, dout=g, out=out, dx=x.grad, dweight=w.grad, silu=np.array(silu))
        if has_bias:
            arrs.update(bias=b, dbias=b.grad)
        save(name, "causal_conv1d_ref fwd + autograd (causal-conv1d/causal_conv1d/causal_conv1d_interface.py:48-64)",
             **arrs)


def gen_norm(ns):
    """rms_norm_ref / layer_norm_ref (ops/triton/layernorm.py:19-45), upcast=True, prenorm."""
    for name, (M, N, has_res, is_rms) in {
        "rmsnorm_prenorm_res": (24, 1024, True, True),
        "rmsnorm_prenorm_nores": (33, 384, False, True),
        "rmsnorm_odd": (7, 200, True, True),
        "layernorm_prenorm_res": (16, 256, True, False),
    }.items():
        torch.manual_seed(0)
        x = torch.randn(M, N, requires_grad=True)
        res = torch.randn(M, N, requires_grad=True) if has_res else None
        w = (1 + 0.1 * torch.randn(N)).requires_grad_()
        b = (0.1 * torch.randn(N)).requires_grad_() if not is_rms else None
        fn = ns.ln.rms_norm_ref if is_rms else ns.ln.layer_norm_ref
        y, res_out = fn(x, w, b, residual=res, eps=1e-5, prenorm=True, upcast=True)
        gy, gr = torch.randn_like(y), torch.randn_like(res_out)
        (y * gy).sum().add((res_out * gr).sum()).backward()
        arrs = dict(x=x, weight=w, y=y, res_out=res_out, dy=gy, dres_out=gr, dx=x.grad, dweight=w.grad,
                    eps=np.array(1e-5))
        if has_res:
            arrs.update(residual=res, dresidual=res.grad)
        if b is not None:
            arrs.update(bias=b, dbias=b.grad)
        save(name, "rms_norm_ref/layer_norm_ref prenorm, upcast (mamba/mamba_ssm/ops/triton/layernorm.py:19-45)", **arrs)


def gen_perm(ns):
    """SCAN_ZOO tables + inverses (dimsum/scanning_orders.py:7-253,419-423) and block-level reorders."""
    so = ns.scanning_orders
    arrs, shas = {}, {}
    for kind in ("sweep", "zigma", "jpeg"):
        for N in (4, 8, 16, 32):
            paths = np.stack(so.SCAN_ZOO[kind](N)).astype(np.int64)
            inv = np.stack([so.reverse_permut_np(p) for p in paths]).astype(np.int64)
            arrs[f"{kind}{N}"] = paths.astype(np.int16)
            arrs[f"{kind}{N}_inv"] = inv.astype(np.int16)
            shas[f"{kind}{N}"] = hashlib.sha256(paths.tobytes()).hexdigest()[:16]
    for k, v in shas.items():
        arrs["sha_" + k] = np.array(v)
    # local_scan / local_reverse (scanning_orders.py:347-367,393-416) as index tables
    for (H, w) in ((4, 2), (16, 4), (32, 8), (8, 2)):
        L = H * H
        ids = torch.arange(L, dtype=torch.float32).view(1, L, 1)
        for cf in (False, True):
            p = so.local_scan(ids.clone(), w=w, H=H, W=H, column_first=cf).view(-1).long()
            back = so.local_reverse(p.view(1, L, 1).float(), w=w, H=H, W=H, column_first=cf).view(-1).long()
            assert torch.equal(back, torch.arange(L))
            arrs[f"local_H{H}_w{w}_{'col' if cf else 'row'}"] = p.numpy().astype(np.int16)
    save("perm_tables", "SCAN_ZOO[...](N), reverse_permut_np, local_scan (dimsum/scanning_orders.py)", **arrs)


def _block_order(ns, reverse, transpose, cont, H):
    """Order seen by the mixer inside DiMBlockRaw.forward (models_dim.py:1496-1507)."""
    from einops import rearrange
    L = H * H
    hs = torch.arange(L, dtype=torch.float32).view(1, L, 1)
    if transpose:
        hs = rearrange(hs, "n (h w) c -> n (w h) c", h=H, w=H)
    if cont:
        hs = rearrange(hs.clone(), "n (w h) c -> n c w h", h=H, w=H)
        hs[:, :, 1::2] = hs[:, :, 1::2].flip(-1)
        hs = rearrange(hs, "n c w h -> n (w h) c", h=H, w=H)
    if reverse:
        hs = hs.flip(1)
    return hs.view(-1).long().numpy()


def gen_block_orders(ns):
    arrs = {}
    for H in (4, 16, 32):
        for r in (0, 1):
            for t in (0, 1):
                for c in (0, 1):
                    arrs[f"H{H}_r{r}_t{t}_c{c}"] = _block_order(ns, r, t, c, H).astype(np.int16)
    save("block_orders", "token order seen by the mixer, DiMBlockRaw.forward (dimsum/models_dim.py:1496-1507)", **arrs)


def gen_wavelet_dct(ns):
    md = ns.models_dim
    blk = md.WaveDiMBlock.__new__(md.WaveDiMBlock)
    torch.nn.Module.__init__(blk)
    blk.num_wavelet_lv = 2
    blk.dwt = ns.wavelet_layer.DWT_2D("haar")
    blk.idwt = ns.wavelet_layer.IDWT_2D("haar")
    arrs = {}
    for (H, C) in ((16, 8), (32, 4), (4, 3)):
        x = T(seeded((2, H * H, C), 11 + H)).requires_grad_()
        y = blk._dwt_fast(x)
        g = T(seeded(tuple(y.shape), 12 + H))
        y.backward(g)
        xr = blk._idwt_fast(y.detach())
        y2 = T(seeded((2, H * H, C), 13 + H)).requires_grad_()
        xi = blk._idwt_fast(y2)
        gi = T(seeded(tuple(xi.shape), 14 + H))
        xi.backward(gi)
        arrs.update({f"H{H}_x": x, f"H{H}_dwt": y, f"H{H}_dwt_g": g, f"H{H}_dwt_dx": x.grad, f"H{H}_roundtrip": xr,
                     f"H{H}_y2": y2, f"H{H}_idwt": xi, f"H{H}_idwt_g": gi, f"H{H}_idwt_dy": y2.grad})
    # the equivalent 16x16 per-4x4-block matrix: apply to unit impulses
    eye = torch.zeros(16, 16, 1)
    for i in range(16):
        eye[i, i, 0] = 1.0
    arrs["haar4x4_matrix"] = blk._dwt_fast(eye)[:, :, 0].t().contiguous()  # out = M @ in  (index = 4*row+col in tile)
    save("haar", "WaveDiMBlock._dwt_fast/_idwt_fast (dimsum/models_dim.py:572-604) over DWT_2D/IDWT_2D "
         "(dimsum/wavelet_layer.py:7-115)", **arrs)

    # DCT: dct_conv + rearrange / rearrange + idct_conv (models_dim.py:876-882, 919-928)
    from einops import rearrange
    arrs = {}
    for (H, C) in ((16, 8), (32, 4)):
        dct = ns.dct_layer.init_dct_kernel(C, 4, 4)
        idct = torch.nn.Sequential(ns.dct_layer.init_idct_kernel(C, 4, 4), torch.nn.PixelShuffle(4))
        x = T(seeded((2, H * H, C), 21 + H))
        h = rearrange(x, "b (h w) d -> b d h w", h=H)
        h = dct(h)
        h = rearrange(h, "b (c p1 p2) h w -> b (h p1 w p2) c", c=C, p1=4).contiguous()
        h2 = rearrange(h, "b (h p1 w p2) c -> b (c p1 p2) h w", c=C, p1=4, p2=4, h=H // 4).contiguous()
        back = rearrange(idct(h2), "b c h w -> b (h w) c")
        arrs.update({f"H{H}_x": x, f"H{H}_dct": h, f"H{H}_roundtrip": back})
        if H == 16:
            arrs["dct_weight_1ch"] = dct.weight[:16].detach()
            arrs["idct_weight_1ch"] = idct[0].weight[:16].detach()
    save("dct", "DCTBlock 4x4 DCT/IDCT path (dimsum/models_dim.py:876-882,919-928; dimsum/dct_layer.py:6-84)", **arrs)


def gen_fusion(ns):
    torch.manual_seed(0)
    for name, (B, L, dim) in {"fusion_128": (2, 256, 128), "fusion_hd24": (2, 64, 384)}.items():
        m = ns.attention_fusion.CrossAttentionFusion(dim, num_heads=8, qkv_bias=True, swap_k=False)
        procedural_fill(m, seed=5)
        x1 = T(seeded((B, L, dim // 2), 31)).requires_grad_()
        x2 = T(seeded((B, L, dim // 2), 32)).requires_grad_()
        y = m(x1, x2)
        g = T(seeded(tuple(y.shape), 33))
        y.backward(g)
        save(name, "CrossAttentionFusion.forward (dimsum/attention_fusion.py:61-84), procedural weights seed 5",
             x1=x1, x2=x2, y=y, dy=g, dx1=x1.grad, dx2=x2.grad,
             **{"g_" + k: v.grad for k, v in m.named_parameters()})


def gen_inner(ns):
    """mamba_inner_ref (selective_scan_interface.py:1455-1500) fwd + autograd, routed to the reference's own refs."""
    torch.manual_seed(0)
    B, Dm, L, N, W = 2, 48, 96, 16, 4
    D = 2 * Dm
    R = 3
    xz = torch.randn(B, 2 * D, L, requires_grad=True)
    p = dict(
        conv_w=torch.randn(D, 1, W) * 0.5, conv_b=torch.randn(D) * 0.1,
        x_proj_w=torch.randn(R + 2 * N, D) / D ** 0.5, dt_proj_w=torch.randn(D, R) / R ** 0.5,
        out_proj_w=torch.randn(Dm, D) / D ** 0.5, A=-torch.exp(torch.log(torch.arange(1, N + 1).float()).repeat(D, 1)
                                                           + 0.1 * torch.randn(D, N)),
        Dv=1 + 0.1 * torch.randn(D), dt_bias=torch.randn(D) * 0.5 - 4.0,
    )
    for v in p.values():
        v.requires_grad_()
    out = ns.ssi.mamba_inner_ref(xz, p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["out_proj_w"], None,
                                 p["A"], None, None, p["Dv"], delta_bias=p["dt_bias"], delta_softplus=True)
    g = torch.randn_like(out)
    out.backward(g)
    save("mamba_inner", "mamba_inner_ref fwd+autograd (mamba/mamba_ssm/ops/selective_scan_interface.py:1455-1500)",
         xz=xz, out=out, dout=g, dxz=xz.grad, **{k: v for k, v in p.items()},
         **{"g_" + k: v.grad for k, v in p.items()})


def gen_mixer(ns):
    """CondMamba / Mamba slow path (mamba_simple.py:562-701) incl. the zigzag gather semantics of :627-657."""
    for name, kw in {
        "condmamba_none": dict(cls="CondMamba", scan_type="none"),
        "mamba_none": dict(cls="Mamba", scan_type="none"),
        "condmamba_zigma8": dict(cls="CondMamba", scan_type="zigma_8"),
        "condmamba_v2": dict(cls="CondMamba", scan_type="v2"),
    }.items():
        d_model, L, B, H = 32, 64, 2, 8
        extra = {}
        if kw["scan_type"].startswith("zigma"):
            so = ns.scanning_orders
            paths = so.SCAN_ZOO["zigma"](H)[:8]
            extra["zigzag_paths"] = torch.stack([T(p) for p in paths])
            extra["zigzag_paths_reverse"] = torch.stack([T(so.reverse_permut_np(p)) for p in paths])
        cls = getattr(ns.ms, kw["cls"])
        ckw = dict(layer_idx=3, scan_type=kw["scan_type"], **extra)
        if kw["cls"] == "CondMamba":
            ckw["d_cond"] = 48
        m = cls(d_model, **ckw)
        procedural_fill(m, seed=7)
        x = T(seeded((B, L, d_model), 41)).requires_grad_()
        c = T(seeded((B, 48), 42))
        if kw["scan_type"] == "v2":    # the reference's own fast-path code (:593-625) over its *_ref ops
            ref_shim.route_no_out_proj_to_refs(ns)
            assert m.use_fast_path
        else:
            m.use_fast_path = False
            ref_shim.slow_path(m)      # wraps the zigzag gather around the slow path (mamba_simple.py:627-657)
        y = m(x, c) if kw["cls"] == "CondMamba" else m(x)
        g = T(seeded(tuple(y.shape), 43))
        y.backward(g)
        save(name, "CondMamba/Mamba.forward slow path (mamba/mamba_ssm/modules/mamba_simple.py:562-701); zigzag per "
             ":627-657; procedural weights seed 7", x=x, c=c, y=y, dy=g, dx=x.grad,
             **{"g_" + k: v.grad for k, v in m.named_parameters() if v.grad is not None})


def _mk_block(ns, hidden, reverse, transpose, cont=False, fourier=False):
    md = ns.models_dim
    blk = md.create_block(hidden, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                          layer_idx=1, scan_type="none", block_type="combined_fourier" if fourier else "combined",
                          reverse=reverse, transpose=transpose, cond_mamba=True, scanning_continuity=cont,
                          use_gated_mlp=True, block_kwargs={},
                          block_kwargs2=_jpeg2_kwargs(ns, 16, 2) if fourier else {})
    return ref_shim.slow_path(blk)


def _jpeg2_kwargs(ns, grid, depth):
    so = ns.scanning_orders
    zz = so.SCAN_ZOO["jpeg"](grid)[:2]
    rev = [so.reverse_permut_np(x) for x in zz]
    return dict(zigzag_paths=torch.cat([T(x)[None] for x in zz] * depth, 0),
                zigzag_paths_reverse=torch.cat([T(x)[None] for x in rev] * depth, 0), scan_type="jpeg_2")


def gen_block(ns):
    """DiMBlockCombined fwd + input grads for the 4 (reverse, transpose) sweep orders (+continuity)."""
    hidden, B, L = 128, 2, 256
    arrs = {}
    x = T(seeded((B, L, hidden), 51))
    res = T(seeded((B, L, hidden), 52))
    c = T(seeded((B, hidden), 53))
    gy = T(seeded((B, L, hidden), 54))
    gr = T(seeded((B, L, hidden), 55))
    arrs.update(x=x, residual=res, c=c, dy=gy, dres=gr)
    for (r, t, cont) in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)):
        blk = _mk_block(ns, hidden, bool(r), bool(t), bool(cont))
        procedural_fill(blk, seed=9)
        xi, ri, ci = x.clone().requires_grad_(), res.clone().requires_grad_(), c.clone().requires_grad_()
        y, ro = blk(xi, ri, ci)
        ((y * gy).sum() + (ro * gr).sum()).backward()
        tag = f"r{r}t{t}c{cont}"
        arrs.update({f"{tag}_y": y, f"{tag}_res_out": ro, f"{tag}_dx": xi.grad, f"{tag}_dres": ri.grad,
                     f"{tag}_dc": ci.grad})
    save("block_combined", "DiMBlockCombined.forward (dimsum/models_dim.py:1055-1117) via create_block(:2001-2160), "
         "hidden 128, procedural weights seed 9", **arrs)
    # the same at DiM-S/2's width (hidden 384: fusion head_dim 24, the smallest the MFMA attention kernels are built for),
    # one batch row, the richest flag combination -- so that a GPU block test runs every HIP kernel incl. attention
    hidden, B = 384, 1
    x, res, c = T(seeded((B, L, hidden), 56)), T(seeded((B, L, hidden), 57)), T(seeded((B, hidden), 58))
    gy, gr = T(seeded((B, L, hidden), 59)), T(seeded((B, L, hidden), 60))
    blk = _mk_block(ns, hidden, True, True, True)
    procedural_fill(blk, seed=9)
    xi, ri, ci = x.clone().requires_grad_(), res.clone().requires_grad_(), c.clone().requires_grad_()
    y, ro = blk(xi, ri, ci)
    ((y * gy).sum() + (ro * gr).sum()).backward()
    f16 = lambda t: t.detach().numpy().astype(np.float32)
    save("block_combined_384", "DiMBlockCombined.forward + input grads (dimsum/models_dim.py:1055-1117), hidden 384, reverse / transpose / "
         "continuity all on, procedural weights seed 9; inputs = seeded(56..60)", y=f16(y), res_out=f16(ro), dx=f16(xi.grad), dres=f16(ri.grad),
         dc=f16(ci.grad), g_qkv1=f16(blk.proj.qkv1.weight.grad), g_A_log=f16(blk.spatial_mamba.mixer.A_log.grad),
         g_norm2=f16(blk.norm_2.weight.grad))


def gen_block_1024(ns):
    """BASELINE configs[2]'s block at ITS width: DiMBlockCombined(1024) (DiM-L/2: mixers d_model 512 / D 1024 / R 32, fusion
    head_dim 64, MLP 1024 -> 8192 -> 4096 -> 1024), reverse + transpose on, continuity off (published), batch 2."""
    hidden, B, L = 1024, 2, 256
    x, res, c = T(seeded((B, L, hidden), 91)), T(seeded((B, L, hidden), 92)), T(seeded((B, hidden), 93))
    gy, gr = T(seeded((B, L, hidden), 94)), T(seeded((B, L, hidden), 95))
    blk = _mk_block(ns, hidden, True, True, False)
    procedural_fill(blk, seed=9)
    xi, ri, ci = x.clone().requires_grad_(), res.clone().requires_grad_(), c.clone().requires_grad_()
    y, ro = blk(xi, ri, ci)
    ((y * gy).sum() + (ro * gr).sum()).backward()
    f = lambda t: t.detach().numpy().astype(np.float32)
    sm, fm = blk.spatial_mamba.mixer, blk.freq_mamba.mixer
    save("block_combined_1024", "DiMBlockCombined.forward + input grads + parameter grads (dimsum/models_dim.py:974-1117 via create_block "
         ":2001-2160), hidden 1024, reverse / transpose on, continuity off, batch 2, procedural weights seed 9; inputs = seeded(91..95)",
         y=f(y), res_out=f(ro), dx=f(xi.grad), dres=f(ri.grad), dc=f(ci.grad),
         g_qkv1_rows64=f(blk.proj.qkv1.weight.grad[:64]), g_qkv2_bias=f(blk.proj.qkv2.bias.grad), g_proj_bias=f(blk.proj.proj.bias.grad),
         g_A_log=f(sm.A_log.grad), g_D_freq=f(fm.D.grad), g_x_proj=f(sm.x_proj.weight.grad), g_dt_bias_freq=f(fm.dt_proj.bias.grad),
         g_conv1d=f(sm.conv1d.weight.grad), g_in_proj_rows64=f(fm.in_proj.weight.grad[:64]), g_out_proj_rows64=f(sm.out_proj.weight.grad[:64]),
         g_norm2=f(blk.norm_2.weight.grad), g_norm=f(blk.norm.weight.grad), g_w12_bias=f(blk.mlp.w12.bias.grad),
         g_w3_rows16=f(blk.mlp.w3.weight.grad[:16]), g_adaLN_bias=f(blk.adaLN_modulation[1].bias.grad))


def _mk_model(ns, name, **over):
    md = ns.models_dim
    kw = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, learn_sigma=False,
              scan_type="none", pe_type="ape", block_type="combined", cond_mamba=True, scanning_continuity=False,
              enable_fourier_layers=False, drop_path=0.0, rms_norm=True, fused_add_norm=True, learnable_pe=True,
              use_final_norm=False, use_attn_every_k_layers=4, use_gated_mlp=True)
    kw.update(over)
    if name == "tiny":
        m = md.DiM(depth=4, hidden_size=64, patch_size=2, **kw)
    elif name == "S/2":
        m = md.DiM(depth=12, hidden_size=384, patch_size=2, **kw)   # SURVEY finding 6: DiT-S analogy
    else:
        m = md.DiM_models[name](**kw)
    return ref_shim.slow_path(m).eval()


def gen_models(ns, which, only_tags=()):
    if "tiny" in which:
        ref_shim.allow_zigzag_through_dim(ns)
        for tag, over in {"tiny": {}, "tiny_cont": dict(scanning_continuity=True),
                          "tiny_fourier": dict(block_type="combined_fourier"),
                          "tiny_final_norm": dict(use_final_norm=True, num_classes=10),
                          "tiny_zigma8": dict(scan_type="zigma_8"), "tiny_jpeg8": dict(scan_type="jpeg_8"),
                          "tiny_sweep8": dict(scan_type="sweep_8")}.items():
            if only_tags and tag not in only_tags:
                continue
            m = _mk_model(ns, "tiny", **over)
            procedural_fill(m, seed=3)
            x = T(seeded((2, 4, 32, 32), 61)).requires_grad_()
            t = T(seeded((2,), 62, kind="uniform"))
            y = torch.tensor([3, 7])
            out = m(x, t, y)
            g = T(seeded(tuple(out.shape), 63))
            out.backward(g)
            arrs = dict(x=x, t=t, y=y, out=out, dout=g, dx=x.grad, n_keys=np.array(len(m.state_dict())),
                        keys=np.array(sorted(m.state_dict().keys())))
            if tag == "tiny":
                with torch.no_grad():
                    x4 = T(seeded((4, 4, 32, 32), 64))
                    t4 = T(seeded((4,), 65, kind="uniform"))
                    y4 = torch.tensor([3, 7, 1000, 1000])
                    arrs.update(cfg_x=x4, cfg_t=t4, cfg_y=y4, cfg_out=m.forward_with_cfg(x4, t4, y4, cfg_scale=1.4),
                                adacfg_out=m.forward_with_adacfg(x4, t4, y4, cfg_scale=3.8, scale_pow=4.0),
                                out_nolabel=m(x4, t4, None))
            save("model_" + tag, "DiM.forward (dimsum/models_dim.py:1796-1884), depth 4 hidden 64, "
                 "published flags (scripts/train.sh), procedural weights seed 3", **arrs)
    for name, tag, B in (("S/2", "model_S2", 4), ("DiM-L/2", "model_L2", 2), ("DiM-XL/2", "model_XL2_512", 1),
                         ("DiM-XL/2", "model_XL2_512_zigma8", 1)):
        if tag not in which:
            continue
        over = dict(img_resolution=64) if "512" in tag else {}
        if tag.endswith("zigma8"):      # BASELINE configs[4]: 8-way zigzag scanning orders inside the mixers
            ref_shim.allow_zigzag_through_dim(ns)
            over["scan_type"] = "zigma_8"
        m = _mk_model(ns, name, **over)
        procedural_fill(m, seed=3)
        R = 64 if "512" in tag else 32
        x = T(seeded((B, 4, R, R), 71))
        t = T(seeded((B,), 72, kind="uniform"))
        y = torch.arange(B) * 37 % 1000
        with torch.no_grad():
            out = m(x, t, y)
        nparam = sum(p.numel() for p in m.parameters())
        save(tag, f"DiM.forward {name} (dimsum/models_dim.py:1796-1884,2163-2236), procedural weights seed 3; "
             "inputs = seeded(71/72)", t=t, y=y, out=out, n_params=np.array(nparam),
             n_keys=np.array(len(m.state_dict())), keys=np.array(sorted(m.state_dict().keys())),
             shapes=np.array([str(tuple(v.shape)) for _, v in sorted(m.state_dict().items())]))
        del m


TRANSPORT_PATHS = ("GVP", "Linear", "VP")
TRANSPORT_PREDS = ("velocity", "noise", "score")
# explicit end-point margins: with the defaults the reference leaves sample_eps = None for VP and for noise / score
# prediction (transport/__init__.py:50-55 tests `train_eps is None` after assigning it), which its samplers cannot use
TRANSPORT_EPS = dict(train_eps=1e-3, sample_eps=2e-3)
SDE_CASES = (  # (sampling_method, diffusion_form, diffusion_norm, last_step, last_step_size)
    ("Euler", "SBDM", 1.0, "Mean", 0.04), ("Euler", "SBDM", 1.0, None, 0.04), ("Euler", "SBDM", 1.0, "Tweedie", 0.04),
    ("Euler", "SBDM", 1.0, "Euler", -1), ("Heun", "SBDM", 1.0, "Mean", 0.04), ("Heun", "SBDM", 1.0, "Tweedie", -1),
    ("Heun", "sigma", 0.7, "Euler", 0.04), ("Heun", "constant", 0.5, None, 0.04), ("Euler", "constant", 0.5, "Mean", 0.04),
    ("Euler", "sigma", 1.0, "Mean", 0.04), ("Euler", "linear", 1.0, "Mean", 0.04), ("Euler", "decreasing", 1.0, "Mean", 0.04),
    ("Euler", "increasing-decreasing", 1.0, "Tweedie", 0.04), ("Euler", "log", 1.0, "Euler", 0.04), ("Euler", "none", 1.0, "Mean", 0.04),
)


def gen_transport(_ns=None):
    """The reference's flow-matching harness around a closed-form denoiser (tests/golden/procedural.py:toy_denoiser):
    Transport.training_losses (dimsum/transport/transport.py:127-164) for {GVP, Linear, VP} x {velocity, noise, score} x
    loss weights; Sampler.sample_sde (:286-341; integrators.py:5-73) for Euler-Maruyama / Heun x diffusion forms x last
    steps; Sampler.sample_ode (:343-386) on the fixed Euler grid (through ref_shim's declared
    stand-in for torchdiffeq's euler). Random draws come from the CPU generator after torch.manual_seed(seed) exactly as
    the reference draws them (randn_like(x1) then rand(B); randn(x.size()) per SDE step)."""
    tp = ref_shim.load_transport()
    arrs = {}
    x1 = T(seeded((5, 3, 4, 4), 81))
    y = torch.tensor([3, 0, 7, 1, 5])
    arrs.update(loss_x1=x1, y=y)
    for pt in TRANSPORT_PATHS:
        for pred in TRANSPORT_PREDS:
            for lw in ((None,) if pred == "velocity" else (None, "velocity", "likelihood")):
                tr = tp.create_transport(pt, pred, lw, **TRANSPORT_EPS)
                seen = {}

                def model(xt, t, y=None):
                    seen["xt"], seen["t"] = xt, t
                    return toy_denoiser(xt, t, y)

                seed = 1000 + len(arrs)
                torch.manual_seed(seed)
                terms = tr.training_losses(model, x1, dict(y=y))
                tag = f"loss_{pt}_{pred}_{lw}"
                arrs.update({tag + "_seed": np.array(seed), tag + "_t": seen["t"], tag + "_xt": seen["xt"],
                             tag + "_loss": terms["loss"], tag + "_pred": terms["pred"]})
    # logit-normal time sampling (transport.py:116-121)
    tr = tp.create_transport("GVP", "velocity", None, t_sample_mode="logitnormal")
    torch.manual_seed(77)
    seen = {}

    def model(xt, t, y=None):
        seen["t"] = t
        return toy_denoiser(xt, t, y)

    terms = tr.training_losses(model, x1, dict(y=y))
    arrs.update(loss_logitnormal_t=seen["t"], loss_logitnormal_loss=terms["loss"])

    z = T(seeded((3, 3, 4, 4), 82))
    yz = torch.tensor([2, 9, 4])
    arrs.update(z=z, yz=yz)
    for pt in TRANSPORT_PATHS:
        for pred in TRANSPORT_PREDS:
            tr = tp.create_transport(pt, pred, **TRANSPORT_EPS)
            # velocity prediction on GVP / Linear: create_transport forces both margins to 0 (__init__.py:56-58), with which the
            # SBDM diffusion (alpha' / alpha at t = 0) is infinite; the SDE fixtures set the margin on the object instead
            tr.train_eps, tr.sample_eps = TRANSPORT_EPS["train_eps"], TRANSPORT_EPS["sample_eps"]
            smp = tp.Sampler(tr)
            for i, (method, form, norm, last, lss) in enumerate(SDE_CASES):
                torch.manual_seed(2000 + i)
                xs = smp.sample_sde(sampling_method=method, diffusion_form=form, diffusion_norm=norm, last_step=last,
                                    last_step_size=lss, num_steps=10)(z, toy_denoiser, y=yz)
                arrs[f"sde_{pt}_{pred}_{i}_last"] = xs[-1]
                arrs[f"sde_{pt}_{pred}_{i}_prev"] = xs[-2]
                arrs[f"sde_{pt}_{pred}_{i}_len"] = np.array(len(xs))
            # (reverse=True cannot run in the reference: check_interval returns t0 > t1 and integrators.py:90 asserts)
            traj = smp.sample_ode(sampling_method="euler", num_steps=9)(z, toy_denoiser, y=yz)
            arrs[f"ode_{pt}_{pred}"] = traj[-1]
            arrs[f"ode_{pt}_{pred}_mid"] = traj[4]
            # likelihood ODE (transport.py:388-443): data -> prior with Hutchinson's divergence estimate, (x, delta_logp) tuple state;
            # under no_grad like its callers (the reference sets x.requires_grad on the solver's state, which must be a leaf)
            torch.manual_seed(3000)
            with torch.no_grad():
                logp, zprior = smp.sample_ode_likelihood(sampling_method="euler", num_steps=9)(z, toy_denoiser, y=yz)
            arrs[f"lik_{pt}_{pred}_logp"], arrs[f"lik_{pt}_{pred}_z"] = logp, zprior
    arrs["sde_cases"] = np.array([repr(c) for c in SDE_CASES])
    save("transport", "dimsum/transport: Transport.training_losses (transport.py:127-164), Sampler.sample_sde (:286-341, "
         "integrators.py:5-73), Sampler.sample_ode on the euler grid (:343-386, integrators.py:98-111 with the declared torchdiffeq "
         "euler stand-in of tools/ref_shim.py:load_transport), path.py:21-246; denoiser = tests/golden/procedural.py:toy_denoiser", **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    ns = ref_shim.load()
    torch.set_num_threads(8)
    steps = {
        "scan": lambda: gen_scan(ns), "conv": lambda: gen_conv(ns), "norm": lambda: gen_norm(ns),
        "perm": lambda: gen_perm(ns), "orders": lambda: gen_block_orders(ns), "wavelet": lambda: gen_wavelet_dct(ns),
        "fusion": lambda: gen_fusion(ns), "inner": lambda: gen_inner(ns), "mixer": lambda: gen_mixer(ns),
        "block": lambda: gen_block(ns), "tiny": lambda: gen_models(ns, {"tiny"}),
        "S2": lambda: gen_models(ns, {"model_S2"}), "L2": lambda: gen_models(ns, {"model_L2"}),
        "XL2": lambda: gen_models(ns, {"model_XL2_512"}),
        "zigzag": lambda: gen_models(ns, {"tiny"}, only_tags=("tiny_zigma8", "tiny_jpeg8", "tiny_sweep8")),
        "XL2zigzag": lambda: gen_models(ns, {"model_XL2_512_zigma8"}),
        "transport": lambda: gen_transport(ns), "block1024": lambda: gen_block_1024(ns),
    }
    for k, fn in steps.items():
        if args.only is None or k in args.only:
            print(f"[{k}]")
            fn()


if __name__ == "__main__":
    main()
