"""GPU parity: HIP causal conv1d and fused add+norm (through the C ABI) vs reference goldens and the CPU oracle.
Tolerances: conv fwd/dx rtol 3e-4 + atol 1e-3, weights 1e-3/1e-3 (causal-conv1d/tests/test_causal_conv1d.py:31-33 uses the
same for fp32); we actually hold tighter bounds stated inline. Norm: rtol 1e-4 + atol 1e-5."""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden

pytestmark = pytest.mark.gpu

CONV_CASES = ["conv_L8_w4_silu", "conv_L151_w4_silu", "conv_L256_w4_silu", "conv_L1134_w4_silu", "conv_L256_w3_nosilu",
              "conv_L64_w2_nobias", "conv_L4096_w4_silu"]


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _opt(g, k):
    return g[k] if k in g.files else None


@pytest.mark.parametrize("name", CONV_CASES)
def test_conv_vs_golden(name):
    from dimsum_amd import native
    g = golden(name)
    silu = bool(g["silu"])
    x, w, b, dout = _t(g["x"]), _t(g["weight"]), _t(_opt(g, "bias")), _t(g["dout"])
    out = native.causal_conv1d_fwd(x, w, b, silu)
    assert_close(out.cpu().numpy(), g["out"], 1e-5, 2e-6, "out", scale_atol=1e-6)
    dx, dw, db = native.causal_conv1d_bwd(x, w, b, dout, None, silu)
    assert_close(dx.cpu().numpy(), g["dx"], 1e-5, 2e-6, "dx", scale_atol=1e-6)
    assert_close(dw.cpu().numpy(), g["dweight"], 1e-4, 0, "dweight", scale_atol=1e-5)
    if b is not None:
        assert_close(db.cpu().numpy(), g["dbias"], 1e-4, 0, "dbias", scale_atol=1e-5)


@pytest.mark.parametrize("B,D,L,W", [(4, 96, 256, 4), (2, 33, 1000, 4), (3, 5, 7, 2), (2, 16, 260, 3)])
def test_conv_strided_views_like_mamba(B, D, L, W):
    """x = xz.chunk(2,1)[0] (batch stride 2DL); dx written into the first half of a caller-owned dxz
    (selective_scan_interface.py:834, 933-934, 985-987)."""
    from dimsum_amd import native
    from oracle import c_ops
    gen = torch.Generator().manual_seed(B + D + L)
    xz = torch.randn(B, 2 * D, L, generator=gen)
    w, b = torch.randn(D, W, generator=gen), torch.randn(D, generator=gen)
    dout = torch.randn(B, D, L, generator=gen)
    xz_g = xz.cuda()
    x_g = xz_g.chunk(2, 1)[0]
    out = native.causal_conv1d_fwd(x_g, w.cuda(), b.cuda(), True)
    ref = c_ops.causal_conv1d_fwd(xz[:, :D].numpy(), w.numpy(), b.numpy(), True)
    assert_close(out.cpu().numpy(), ref, 1e-5, 2e-6, "out")
    dxz = torch.full_like(xz_g, float("nan"))
    dx_view = dxz.chunk(2, 1)[0]
    dx, dw, db = native.causal_conv1d_bwd(x_g, w.cuda(), b.cuda(), dout.cuda(), dx_view, True)
    rdx, rdw, rdb = c_ops.causal_conv1d_bwd(xz[:, :D].numpy(), w.numpy(), b.numpy(), dout.numpy(), True)
    assert dx.data_ptr() == dx_view.data_ptr()
    assert_close(dxz[:, :D].cpu().numpy(), rdx, 1e-5, 2e-6, "dx")
    assert torch.isnan(dxz[:, D:]).all(), "the z half of dxz must not be touched"
    assert_close(dw.cpu().numpy(), rdw, 1e-4, 0, "dw", scale_atol=1e-5)
    assert_close(db.cpu().numpy(), rdb, 1e-4, 0, "db", scale_atol=1e-5)


def test_conv_cond_alias_semantics():
    """_fwd_cond returns init_x overwritten with the plain conv result (SURVEY finding 1)."""
    from dimsum_amd import native
    x = torch.randn(2, 8, 64, device="cuda")
    w, b = torch.randn(8, 4, device="cuda"), torch.randn(8, device="cuda")
    init = torch.randn(2, 8, 64, device="cuda")
    plain = native.causal_conv1d_fwd(x, w, b, True)
    out = native.causal_conv1d_fwd_cond(x, w, b, True, init)
    assert out.data_ptr() == init.data_ptr() and torch.equal(out, plain)


def test_conv_half_dtypes():
    from dimsum_amd import native
    from oracle import c_ops
    for dt, rtol, atol in ((torch.bfloat16, 1e-2, 5e-2), (torch.float16, 3e-3, 5e-3)):   # test_causal_conv1d.py:32-35
        x = torch.randn(2, 64, 512).to(dt)
        w, b = torch.randn(64, 4), torch.randn(64)
        out = native.causal_conv1d_fwd(x.cuda(), w.cuda(), b.cuda(), True)
        ref = c_ops.causal_conv1d_fwd(x.float().numpy(), w.numpy(), b.numpy(), True)
        assert out.dtype == dt
        assert_close(out.float().cpu().numpy(), ref, rtol, atol, str(dt))


def test_conv_determinism():
    """out and dx bit-identical across launches; dw/db within 1e-4 (atomics) -- test_causal_conv1d.py:120-180."""
    from dimsum_amd import native
    x, w, b = torch.randn(8, 256, 512, device="cuda"), torch.randn(256, 4, device="cuda"), torch.randn(256, device="cuda")
    dout = torch.randn_like(x)
    o0 = native.causal_conv1d_fwd(x, w, b, True)
    dx0, dw0, db0 = native.causal_conv1d_bwd(x, w, b, dout, None, True)
    for _ in range(20):
        assert torch.equal(native.causal_conv1d_fwd(x, w, b, True), o0)
        dx, dw, db = native.causal_conv1d_bwd(x, w, b, dout, None, True)
        assert torch.equal(dx, dx0)
        assert torch.allclose(dw, dw0, rtol=1e-4, atol=1e-4) and torch.allclose(db, db0, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["rmsnorm_prenorm_res", "rmsnorm_prenorm_nores", "rmsnorm_odd", "layernorm_prenorm_res"])
def test_norm_vs_golden(name):
    from dimsum_amd import native
    g = golden(name)
    is_rms = name.startswith("rms")
    x, w, b, res = _t(g["x"]), _t(g["weight"]), _t(_opt(g, "bias")), _t(_opt(g, "residual"))
    y, mean, rstd, ro = native.layer_norm_fwd(x, w, b, float(g["eps"]), residual=res, residual_dtype=torch.float32, is_rms_norm=is_rms)
    assert_close(y.cpu().numpy(), g["y"], 1e-5, 1e-5, "y")
    assert np.array_equal(ro.cpu().numpy(), g["res_out"]), "residual_out is the fp32 sum, bit for bit"
    dx, dw, db, dres_in = native.layer_norm_bwd(_t(g["dy"]), ro, w, b, float(g["eps"]), mean, rstd, dresidual=_t(g["dres_out"]),
                                                has_residual=res is not None, is_rms_norm=is_rms)
    assert_close(dx.cpu().numpy(), g["dx"], 1e-4, 1e-5, "dx")
    if res is not None:
        assert_close(dres_in.cpu().numpy(), g["dresidual"], 1e-4, 1e-5, "dresidual")
    assert_close(dw.cpu().numpy(), g["dweight"], 1e-4, 0, "dweight", scale_atol=1e-5)
    if b is not None:
        assert_close(db.cpu().numpy(), g["dbias"], 1e-4, 0, "dbias", scale_atol=1e-5)


@pytest.mark.parametrize("M,N", [(1000, 1024), (257, 1152), (64, 384), (3, 2048), (5, 100)])
def test_norm_vs_oracle(M, N):
    from dimsum_amd import native
    from oracle import c_ops
    gen = torch.Generator().manual_seed(M + N)
    x, res = torch.randn(M, N, generator=gen), torch.randn(M, N, generator=gen)
    w = 1 + 0.1 * torch.randn(N, generator=gen)
    dy, dro = torch.randn(M, N, generator=gen), torch.randn(M, N, generator=gen)
    y, mean, rstd, ro = native.layer_norm_fwd(x.cuda(), w.cuda(), None, 1e-5, residual=res.cuda(), is_rms_norm=True)
    y_ref, ro_ref, _, rstd_ref = c_ops.norm_fwd(x.numpy(), w.numpy(), None, res.numpy(), 1e-5, True)
    assert_close(y.cpu().numpy(), y_ref, 1e-5, 1e-5, "y")
    assert np.array_equal(ro.cpu().numpy(), ro_ref)
    assert_close(rstd.cpu().numpy(), rstd_ref, 1e-6, 0, "rstd")
    dx, dw, _, _ = native.layer_norm_bwd(dy.cuda(), ro, w.cuda(), None, 1e-5, mean, rstd, dresidual=dro.cuda(), has_residual=True, is_rms_norm=True)
    dr_ref, dw_ref, _ = c_ops.norm_bwd(ro_ref, w.numpy(), dy.numpy(), dro.numpy(), 1e-5, True)
    assert_close(dx.cpu().numpy(), dr_ref, 1e-4, 1e-5, "dx")
    assert_close(dw.cpu().numpy(), dw_ref, 1e-4, 0, "dw", scale_atol=1e-5)


@pytest.mark.parametrize("M,N,L", [(6 * 64, 1024, 64), (4 * 16, 384, 16), (3 * 8, 72, 8)])
def test_fused_bias_add_rmsnorm_modulate(M, N, L):
    """h' = x + x_bias + residual; y = RMSNorm(h') * w * (1 + scale[b]) + shift[b] in ONE pass vs the composition.
    The residual stream must be bit-identical to the separate adds; y within fp32 roundoff (rtol 2e-6 + 2e-6 max)."""
    from dimsum_amd import native
    g = torch.Generator().manual_seed(M + N)
    x, res = torch.randn(M, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    w, xb = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), torch.randn(N, generator=g).cuda()
    mods = torch.randn(M // L, 3 * N, generator=g).cuda()
    shift, scale = mods[:, :N], mods[:, N:2 * N]
    y, _, rstd, hnew = native.layer_norm_fwd(x, w, None, 1e-5, residual=res, is_rms_norm=True, x_bias=xb, mod_scale=scale, mod_shift=shift,
                                             rows_per_batch=L)
    h_ref = (x + xb) + res
    assert torch.equal(hnew, h_ref)
    n_ref = h_ref * torch.rsqrt(h_ref.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    y_ref = (n_ref.view(M // L, L, N) * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)).view(M, N)
    assert_close(y.cpu().numpy(), y_ref.cpu().numpy(), 2e-6, 0, "y", scale_atol=2e-6)
