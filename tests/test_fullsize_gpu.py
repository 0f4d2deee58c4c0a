"""Full-size oracle comparisons (not only properties): every streaming kernel of the hot path at BASELINE's launch shapes --
config 2/3 (256, 1024, 256, 16) and config 5 (64, 1152, 1024, 16; the scan also at the stress shape (16, 1152, 4096, 16)), operands in the layouts MambaInnerFn produces (d-major
u / delta / z / dout) -- compared with the C / numpy oracle on three batch rows (first, middle, last; all channels, so the
last channel tile and the largest in-tile offsets are covered). The ops are independent per batch row, so the oracle only
computes those rows; batch-summed outputs (dA, dD, dweight ...) are pinned by the small-shape tests.
Also: the host guards of the 32-bit in-tile byte offsets of the scan kernels (DIMSUM_ERR_STRIDE)."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu

SHAPES = [(256, 1024, 256, 16), (64, 1152, 1024, 16)]
SCAN_SHAPES = SHAPES + [(16, 1152, 4096, 16)]     # + the long-sequence stress shape: one-lane-per-state forward, two 2048-step chunks


def rows_of(B):
    return [0, B // 2, B - 1]


def dmajor(gen, B, D, L, scale=1.0, uniform=False):
    """(B, D, L) tensor with strides (L, B L, 1), like the outputs of the fused-transpose GEMMs (mamba_simple.py:578-582)"""
    t = (torch.rand if uniform else torch.randn)(D, B, L, device="cuda", generator=gen)
    return (t * scale).permute(1, 0, 2)


def f(t):
    return np.ascontiguousarray(t.detach().float().cpu().numpy())


def scan_tol(L):
    return dict(rtol=2e-4, atol=0.0, scale_atol=1e-5) if L <= 512 else dict(rtol=6e-4, atol=0.0, scale_atol=1e-4)


@pytest.mark.parametrize("B,D,L,N", SCAN_SHAPES)
def test_scan_fwd_bwd_rows_vs_oracle(B, D, L, N):
    from dimsum_amd import native
    from oracle import c_ops
    g = torch.Generator(device="cuda").manual_seed(B + L)
    u, z, dout = dmajor(g, B, D, L), dmajor(g, B, D, L), dmajor(g, B, D, L)
    delta = dmajor(g, B, D, L, 0.5, uniform=True)
    A = -0.5 * torch.rand(D, N, device="cuda", generator=g)
    Bm, Cm = torch.randn(B, 1, N, L, device="cuda", generator=g), torch.randn(B, 1, N, L, device="cuda", generator=g)
    Dv, bias = torch.randn(D, device="cuda", generator=g), 0.5 * torch.rand(D, device="cuda", generator=g)
    out, x, out_z, ckpt = native.selective_scan_fwd(u, delta, A, Bm, Cm, Dv, z, bias, True, need_ckpt=True)
    dz = torch.empty_like(z)
    res = native.selective_scan_bwd(u, delta, A, Bm, Cm, Dv, z, bias, dout, x, out, dz, True, True, ckpt=ckpt)
    du, ddelta, dA, dB, dC = res[:5]
    # inference variant of the forward (no saved states) must give the same out_z
    out_z2 = native.selective_scan_fwd(u, delta, A, Bm, Cm, Dv, z, bias, True)[2]
    assert torch.equal(out_z2, out_z)
    r = rows_of(B)
    y_ref, oz_ref, x_ref = c_ops.selective_scan_fwd(f(u[r]), f(delta[r]), f(A), f(Bm[r]), f(Cm[r]), f(Dv), f(z[r]), f(bias), True)
    t = scan_tol(L)
    assert_close(f(out[r]), y_ref, what="out", **t)
    assert_close(f(out_z[r]), oz_ref, what="out_z", **t)
    assert_close(f(x[r]), x_ref, what="x (chunk states)", **t)
    gr = c_ops.selective_scan_bwd(f(u[r]), f(delta[r]), f(A), f(Bm[r]), f(Cm[r]), f(Dv), f(z[r]), f(bias), True, f(dout[r]))
    bt = dict(rtol=1e-3, atol=0.0, scale_atol=2e-4) if L <= 512 else dict(rtol=2e-3, atol=0.0, scale_atol=5e-4)
    assert_close(f(du[r]), gr["du"], what="du", **bt)
    assert_close(f(ddelta[r]), gr["ddelta"], what="ddelta", **bt)
    assert_close(f(dz[r]), gr["dz"], what="dz", **bt)
    assert_close(f(dB[r]), gr["dB"].reshape(f(dB[r]).shape), what="dB", **bt)
    assert_close(f(dC[r]), gr["dC"].reshape(f(dC[r]).shape), what="dC", **bt)


@pytest.mark.parametrize("B", [256, 128])
def test_headline_scan_launch_fused_dt_proj_fp16_out_z_rows_vs_oracle(B):
    """The launch bench.py's headline times (DiM-L/2 inference under the scaled-fp16 policy, batch 256; 128 per GPU in the sampling leg):
    dt_proj fused into the scan + block-scaled fp16 out_z, at the DISPATCH THE LIBRARY CHOOSES (no forced variant: 4096 / 2048 waves ->
    the 64-channel kernel), operands in MambaInnerFn's layouts (x_proj written r-major: its first R rows feed the in-scan dt_proj, the
    next 2N are B and C as (b, 1, N, l) views), decoded through the scale table and compared with the C oracle on three batch rows.
    The oracle's delta is the float64 product W_dt x_dbl[:R] rounded to fp32 (selective_scan_interface.py:840-841)."""
    from dimsum_amd import native
    from oracle import c_ops
    D, L, N, R = 1024, 256, 16, 32
    g = torch.Generator(device="cuda").manual_seed(1000 + B)
    u, z = dmajor(g, B, D, L), dmajor(g, B, D, L)
    x_dbl_t = torch.randn(R + 2 * N, B * L, device="cuda", generator=g)
    x_dbl_t[:R] *= 0.5
    dt_w = (torch.rand(D, R, device="cuda", generator=g) * 2 - 1) * R ** -0.5           # dt_proj's own init range (mamba_simple.py:494-499)
    A = -0.5 * torch.rand(D, N, device="cuda", generator=g) - 0.05
    Bm = x_dbl_t[R:R + N].view(N, B, L).permute(1, 0, 2).unsqueeze(1)
    Cm = x_dbl_t[R + N:].view(N, B, L).permute(1, 0, 2).unsqueeze(1)
    Dv, bias = torch.randn(D, device="cuda", generator=g), 0.5 * torch.rand(D, device="cuda", generator=g)
    assert native._scan_fwd_variant == 0 and native.scan_fwd_kernel_for(B, D, L, N) == 1
    assert native.scan_dt_proj_supported(u, z, A, dt_w, x_dbl_t[:R]) and native.scan_out_z_f16_supported(u, z, A, 1)
    out, x, (img, inv) = native.selective_scan_fwd(u, None, A, Bm, Cm, Dv, z, bias, True, need_out=False, need_x=False,
                                                   dt_proj=(dt_w, x_dbl_t[:R]), out_z_f16=True)
    assert out is None and x is None and img.shape == (D, B * L) and img.dtype == torch.float16 and inv.shape == (B * L // 32, D // 64)
    r = rows_of(B)
    delta = (dt_w.double() @ x_dbl_t[:R].double()).float().view(D, B, L).permute(1, 0, 2)
    _, oz_ref, _ = c_ops.selective_scan_fwd(f(u[r]), f(delta[r]), f(A), f(Bm[r]), f(Cm[r]), f(Dv), f(z[r]), f(bias), True)
    # decode: value = fp16 * table[token / 32, channel / 64]
    scale = inv.repeat_interleave(32, 0).repeat_interleave(64, 1).t()                      # (D, B L)
    dec = (img.float() * scale).view(D, B, L).permute(1, 0, 2)
    # tolerance: the fp32 scan's (scan_tol) + the in-scan dt_proj's three bf16 products (2e-5 of |delta|, test_scan_gpu.py) + half an fp16
    # ulp: 2^-11 relative for every element within 2^-14 of its block's maximum, 2^-25 of that maximum below
    t = scan_tol(L)
    got, want = f(dec[r]), oz_ref
    bmax = np.abs(want.reshape(3, D // 64, 64, L // 32, 32)).max(axis=(2, 4), keepdims=True)
    bmax = np.broadcast_to(bmax, (3, D // 64, 64, L // 32, 32)).reshape(3, D, L)
    tol = (t["rtol"] + 2.0 ** -11) * np.abs(want) + t["scale_atol"] * np.abs(want).max() + 2.0 ** -24 * bmax
    err = np.abs(got - want)
    assert (err <= tol).all(), f"max violation {np.max(err - tol):.3e}; max err {err.max():.3e} of {np.abs(want).max():.3e}"
    # and the same launch without the two fusions (GEMM-fed delta, fp32 out_z) agrees with the decoded image block by block
    _, _, oz32 = native.selective_scan_fwd(u[r], delta[r].contiguous(), A, Bm[r].contiguous(), Cm[r].contiguous(), Dv, z[r], bias, True, need_out=False, need_x=False)
    assert_close(got, f(oz32), rtol=2.0 ** -10, atol=0.0, what="fused launch vs the fp32 launch", scale_atol=2e-5)


@pytest.mark.parametrize("B,D,L,N", SHAPES)
def test_conv1d_fwd_bwd_rows_vs_oracle(B, D, L, N):
    from dimsum_amd import native
    from oracle import c_ops
    g = torch.Generator(device="cuda").manual_seed(2 * B + L)
    xz = torch.randn(2 * D, B, L, device="cuda", generator=g).permute(1, 0, 2)       # in_proj output: d-major
    x = xz[:, :D]
    w, b = torch.randn(D, 4, device="cuda", generator=g), torch.randn(D, device="cuda", generator=g)
    dout = dmajor(g, B, D, L)
    out = native.causal_conv1d_fwd(x, w, b, True)
    assert out.stride() == x.stride()
    dx, dw, db = native.causal_conv1d_bwd(x, w, b, dout, None, True)
    r = rows_of(B)
    assert_close(f(out[r]), c_ops.causal_conv1d_fwd(f(x[r]), f(w), f(b), True), 2e-5, 0, "out", scale_atol=2e-6)
    rdx, _, _ = c_ops.causal_conv1d_bwd(f(x[r]), f(w), f(b), f(dout[r]), True)
    assert_close(f(dx[r]), rdx, 1e-4, 0, "dx", scale_atol=1e-5)


@pytest.mark.parametrize("B,L,H", [(256, 256, 1024), (64, 1024, 1152)])
def test_prenorm_rows_vs_oracle(B, L, H):
    from dimsum_amd import native
    from oracle import c_ops
    g = torch.Generator(device="cuda").manual_seed(B + H)
    x, res = torch.randn(B * L, H, device="cuda", generator=g), torch.randn(B * L, H, device="cuda", generator=g)
    w = 1 + 0.1 * torch.randn(H, device="cuda", generator=g)
    y, _, rstd, res_out = native.layer_norm_fwd(x, w, None, 1e-5, residual=res, is_rms_norm=True)
    dy, dres = torch.randn(B * L, H, device="cuda", generator=g), torch.randn(B * L, H, device="cuda", generator=g)
    dx, dw, _, _ = native.layer_norm_bwd(dy, res_out, w, None, 1e-5, None, rstd, dresidual=dres, has_residual=True, is_rms_norm=True)
    rows = torch.cat([torch.arange(b * L, (b + 1) * L) for b in rows_of(B)]).cuda()
    y_ref, ro_ref, _, _ = c_ops.norm_fwd(f(x[rows]), f(w), None, f(res[rows]), 1e-5, True)
    assert_close(f(y[rows]), y_ref, 2e-5, 0, "y", scale_atol=2e-6)
    assert np.array_equal(f(res_out[rows]), ro_ref)                                  # x + residual: one fp32 add, bit-exact
    dr_ref, _, _ = c_ops.norm_bwd(ro_ref, f(w), f(dy[rows]), f(dres[rows]), 1e-5, True)
    assert_close(f(dx[rows]), dr_ref, 1e-4, 0, "dx", scale_atol=1e-5)


@pytest.mark.parametrize("B,L,C,reverse,transpose", [(256, 256, 512, True, False), (64, 1024, 576, False, True)])
def test_haar_pre_post_rows_vs_oracle(B, L, C, reverse, transpose):
    """the two fused token passes around the frequency mixer (WaveDiMBlock.forward, models_dim.py:656-705) on a half-width
    channel slice of the block activations (token stride = hidden): pre = modulate(P(DWT(x))), post = x + IDWT(P^-1(gate m))"""
    from dimsum_amd import scanning_orders as so
    from dimsum_amd.ops import token_ops
    from oracle import np_ops
    g = torch.Generator(device="cuda").manual_seed(L + C)
    H = int(L ** 0.5)
    hs = torch.randn(B, L, 2 * C, device="cuda", generator=g)
    x = hs[:, :, C:]                                                                  # the freq branch's half (not contiguous)
    shift, scale, gate = (0.3 * torch.randn(B, C, device="cuda", generator=g) for _ in range(3))
    tab = so.compose(so.local_scan_table(H, H // 4, column_first=transpose), so.block_order_table(H, reverse, False, False))
    inv = so.reverse_permut_np(tab)
    table = {"inv32": torch.as_tensor(inv.astype(np.int32), device="cuda")}
    pre = token_ops.pre_mixer(x, "haar", table, shift, scale)
    m = torch.randn(B, L, C, device="cuda", generator=g)
    post = token_ops.post_mixer(x, m, gate, "haar", table)
    r = rows_of(B)
    t = np_ops.haar_dwt_tokens(f(x[r]))[:, tab]
    pre_ref = t * (1 + f(scale[r])[:, None]) + f(shift[r])[:, None]
    assert_close(f(pre[r]), pre_ref, 2e-5, 0, "pre_mixer", scale_atol=2e-6)
    back = np.empty_like(t)
    back[:, tab] = f(m[r]) * f(gate[r])[:, None]
    post_ref = f(x[r]) + np_ops.haar_idwt_tokens(back)
    assert_close(f(post[r]), post_ref, 2e-5, 0, "post_mixer", scale_atol=2e-6)


@pytest.mark.parametrize("M,H", [(256 * 256, 4096), (64 * 1024, 4608)])
def test_gated_gelu_rows_vs_oracle(M, H):
    from dimsum_amd import native
    from oracle import np_ops
    g = torch.Generator(device="cuda").manual_seed(H)
    x12, bias = torch.randn(M, 2 * H, device="cuda", generator=g), torch.randn(2 * H, device="cuda", generator=g)
    h = native.gated_gelu_fwd(x12, bias)
    rows = torch.tensor([0, 1, M // 2, M - 2, M - 1], device="cuda")
    assert_close(f(h[rows]), np_ops.gated_gelu(f(x12[rows] + bias)), 2e-5, 0, "gated gelu", scale_atol=2e-6)


def test_batch_reduced_gradients_at_full_reduction_depth():
    """dA, dD, ddelta_bias of the scan, dweight / dbias of the conv, dweight of the norm: sums over batch x sequence that the row-wise
    tests above cannot see at depth. Config 2/3's batch (256) and sequence (256) on a 64-channel slice, EVERY batch row through the
    oracle: 65536 positions per reduced element, accumulated on the GPU through fp32 atomics / register partials in another order."""
    from dimsum_amd import native
    from oracle import c_ops
    B, D, L, N = 256, 64, 256, 16
    g = torch.Generator(device="cuda").manual_seed(77)
    u, z, dout = dmajor(g, B, D, L), dmajor(g, B, D, L), dmajor(g, B, D, L)
    delta = dmajor(g, B, D, L, 0.5, uniform=True)
    A = -0.5 * torch.rand(D, N, device="cuda", generator=g)
    Bm, Cm = torch.randn(B, 1, N, L, device="cuda", generator=g), torch.randn(B, 1, N, L, device="cuda", generator=g)
    Dv, bias = torch.randn(D, device="cuda", generator=g), 0.5 * torch.rand(D, device="cuda", generator=g)
    out, x, out_z, ckpt = native.selective_scan_fwd(u, delta, A, Bm, Cm, Dv, z, bias, True, need_ckpt=True)
    dz = torch.empty_like(z)
    res = native.selective_scan_bwd(u, delta, A, Bm, Cm, Dv, z, bias, dout, x, out, dz, True, True, ckpt=ckpt)
    dA, dD, ddb = res[2], res[5], res[6]
    gr = c_ops.selective_scan_bwd(f(u), f(delta), f(A), f(Bm), f(Cm), f(Dv), f(z), f(bias), True, f(dout))
    # the sums have ~65536 terms of mixed sign: tolerance relative to the largest element of each reduced tensor
    assert_close(f(dA), gr["dA"], what="dA (sum over 256 x 256 positions)", rtol=1e-3, atol=0.0, scale_atol=2e-4)
    assert_close(f(dD), gr["dD"], what="dD", rtol=1e-3, atol=0.0, scale_atol=2e-4)
    assert_close(f(ddb), gr["ddelta_bias"], what="ddelta_bias", rtol=1e-3, atol=0.0, scale_atol=2e-4)
    # conv1d (width 4, SiLU): dweight (D, 4), dbias (D)
    xz = torch.randn(2 * D, B, L, device="cuda", generator=g).permute(1, 0, 2)
    xc = xz[:, :D]
    w, b = torch.randn(D, 4, device="cuda", generator=g), torch.randn(D, device="cuda", generator=g)
    dx, dw, db = native.causal_conv1d_bwd(xc, w, b, dout, None, True)
    rdx, rdw, rdb = c_ops.causal_conv1d_bwd(f(xc), f(w), f(b), f(dout), True)
    assert_close(f(dw), rdw, what="conv dweight (sum over 256 x 256 positions)", rtol=1e-3, atol=0.0, scale_atol=2e-4)
    assert_close(f(db), rdb, what="conv dbias", rtol=1e-3, atol=0.0, scale_atol=2e-4)
    assert_close(f(dx), rdx, 1e-4, 0, "conv dx", scale_atol=1e-5)
    # RMSNorm dweight over 65536 rows (H = 64 columns of a prenorm)
    M, H = B * L, 64
    xr, rr = torch.randn(M, H, device="cuda", generator=g), torch.randn(M, H, device="cuda", generator=g)
    wn = 1 + 0.1 * torch.randn(H, device="cuda", generator=g)
    _, _, rstd, res_out = native.layer_norm_fwd(xr, wn, None, 1e-5, residual=rr, is_rms_norm=True)
    dy, dres = torch.randn(M, H, device="cuda", generator=g), torch.randn(M, H, device="cuda", generator=g)
    dxn, dwn, _, _ = native.layer_norm_bwd(dy, res_out, wn, None, 1e-5, None, rstd, dresidual=dres, has_residual=True, is_rms_norm=True)
    dr_ref, dw_ref, _ = c_ops.norm_bwd(f(res_out), f(wn), f(dy), f(dres), 1e-5, True)
    assert_close(f(dwn), dw_ref, what="norm dweight (sum over 65536 rows)", rtol=1e-3, atol=0.0, scale_atol=2e-4)
    assert_close(f(dxn), dr_ref, 1e-4, 0, "norm dx", scale_atol=1e-5)


def test_scan_32bit_offset_guards():
    """the scan kernels address inside a tile with one 32-bit byte offset per lane: a channel stride for which
    (channels per wave) x stride + seqlen does not fit must be refused by the host (DIMSUM_ERR_STRIDE = 4), forward and
    backward, for every tensor -- and a stride just below the limit must be accepted by the checks."""
    from dimsum_amd import _lib, native
    lib = _lib.load()
    B, D, L, N = 1, 64, 64, 16
    g = torch.Generator(device="cuda").manual_seed(0)
    u, delta, z = (torch.randn(B, D, L, device="cuda", generator=g) for _ in range(3))
    A = -torch.rand(D, N, device="cuda", generator=g)
    Bm, Cm = torch.randn(B, 1, N, L, device="cuda", generator=g), torch.randn(B, 1, N, L, device="cuda", generator=g)
    out, x, out_z = native.selective_scan_fwd(u, delta, A, Bm, Cm, None, z, None, True)
    stream = torch.cuda.current_stream().cuda_stream
    too_big = (1 << 32) // (4 * 64)            # 64 rows x stride x 4 bytes reaches 2^32
    for field in ("u_d_stride", "delta_d_stride", "z_d_stride", "out_d_stride", "out_z_d_stride"):
        P = _lib.SsmParams()
        native._fill_ssm(P, u, delta, A, Bm, Cm, None, z, None, True, out, x, out_z)
        assert lib.dimsum_ssm_scan_fwd(P, stream) == 0
        setattr(P, field, too_big)
        assert lib.dimsum_ssm_scan_fwd(P, stream) == 4, field
    P = _lib.SsmParams()
    native._fill_ssm(P, u, delta, A, Bm, Cm, None, z, None, True, out, x, out_z)
    P.B_dstate_stride = (1 << 32) // (4 * N)
    assert lib.dimsum_ssm_scan_fwd(P, stream) == 4
    # backward: the same guard on its own tensors
    dout = torch.randn(B, D, L, device="cuda", generator=g)
    du, ddelta, dz = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    dA, dB, dC = torch.zeros_like(A), torch.empty(B, 1, N, L, device="cuda"), torch.empty(B, 1, N, L, device="cuda")
    ws = torch.empty(lib.dimsum_ssm_scan_bwd_workspace_bytes(B, D, L, N, 1) // 4 + 4, device="cuda")
    for field in (None, "dout_d_stride", "du_d_stride", "ddelta_d_stride", "dz_d_stride"):
        Q = _lib.SsmBwdParams()
        native._fill_ssm(Q.fwd, u, delta, A, Bm, Cm, None, z, None, True, out, x, None)
        Q.dout_batch_stride, Q.dout_d_stride = dout.stride(0), dout.stride(1)
        Q.dA_d_stride, Q.dA_dstate_stride = dA.stride(0), dA.stride(1)
        Q.dB_batch_stride, Q.dB_group_stride, Q.dB_dstate_stride = dB.stride(0), dB.stride(1), dB.stride(2)
        Q.dC_batch_stride, Q.dC_group_stride, Q.dC_dstate_stride = dC.stride(0), dC.stride(1), dC.stride(2)
        Q.du_batch_stride, Q.du_d_stride = du.stride(0), du.stride(1)
        Q.ddelta_batch_stride, Q.ddelta_d_stride = ddelta.stride(0), ddelta.stride(1)
        Q.dz_batch_stride, Q.dz_d_stride = dz.stride(0), dz.stride(1)
        Q.dout_ptr, Q.dA_ptr, Q.dB_ptr, Q.dC_ptr = dout.data_ptr(), dA.data_ptr(), dB.data_ptr(), dC.data_ptr()
        Q.du_ptr, Q.dz_ptr, Q.ddelta_ptr = du.data_ptr(), dz.data_ptr(), ddelta.data_ptr()
        Q.workspace_ptr, Q.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        if field is None:
            assert lib.dimsum_ssm_scan_bwd(Q, stream) == 0
        else:
            setattr(Q, field, (1 << 32) // (4 * 16))         # 16 channels per wave in the backward
            assert lib.dimsum_ssm_scan_bwd(Q, stream) == 4, field
    torch.cuda.synchronize()
