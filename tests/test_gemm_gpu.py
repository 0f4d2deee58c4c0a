"""The hand-written NT GEMM (csrc/gemm_nt_kernel.hpp, dimsum_gemm_nt) through the C ABI: plain fp32 output against float64 products of
the same 16-bit operands, ragged N, biases, and the gated-GeLU epilogues (dimsum/mlp.py:66-70) against the float64 expression and
against the unfused pair (library GEMM + csrc/token_transform.hip gated GeLU pass) they replace."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (256, 256, 192), (512, 384, 320), (768, 1152, 1152), (256, 132, 128), (512, 4, 256),
                                    (1024, 2048, 3072)])
def test_plain_product_vs_float64(M, N, K, dtype):
    """fp32 accumulation of exact 16-bit products: the only error is the accumulation order (<= 2e-6 of max |C| here); repeated
    launches are bit-identical (a race in the LDS-DMA pipeline would show as run-to-run differences)"""
    from dimsum_amd import native
    dt = getattr(torch, dtype)
    a, b = _rnd((M, K), dt, 1), _rnd((N, K), dt, 2)
    ref = a.double() @ b.double().t()
    got = native.gemm_nt(a, b)
    assert got.shape == (M, N) and got.dtype == torch.float32
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6 * max(1.0, (K / 1024) ** 0.5) + 1e-7, err
    for _ in range(3):
        assert torch.equal(native.gemm_nt(a, b), got)


def test_transpose_detecting_operands():
    """A = one-hot rows against an asymmetric B: C must be exactly the selected rows of B^T (a swapped fragment or epilogue mapping fails)"""
    from dimsum_amd import native
    M, N, K = 256, 512, 256
    a = torch.zeros((M, K), device="cuda", dtype=torch.bfloat16)
    idx = (torch.arange(M, device="cuda") * 7 + 3) % K
    a[torch.arange(M, device="cuda"), idx] = 1.0
    b = ((torch.arange(N, device="cuda")[:, None] * 3 + torch.arange(K, device="cuda")[None, :] * 5) % 251).to(torch.bfloat16)
    got = native.gemm_nt(a, b)
    assert torch.equal(got, b.float()[:, idx].t().contiguous())


def test_bias_and_strided_rows():
    """row strides larger than the row length on all three matrices (views into wider buffers), per-column bias"""
    from dimsum_amd import native
    M, N, K = 512, 384, 256
    abuf, bbuf = _rnd((M, K + 64), torch.bfloat16, 3), _rnd((N, K + 8), torch.bfloat16, 4)
    a, b = abuf[:, :K], bbuf[:, :K]
    bias = _rnd((N,), torch.float32, 5)
    out = torch.full((M, N + 16), 7.0, device="cuda")
    native.gemm_nt(a, b, bias=bias, out=out[:, :N])
    ref = a.double() @ b.double().t() + bias.double()
    assert (out[:, :N].double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6
    assert torch.all(out[:, N:] == 7.0)


@pytest.mark.parametrize("M,F,H,with_bias", [(512, 256, 128, True), (512, 1536, 384, False), (1024, 4608, 1152, True), (256, 136, 192, True)])
def test_gated_gelu_epilogues(M, F, H, with_bias):
    """w12 + bias + gelu_tanh(x1) * x2 as one kernel: the split-bf16 image decodes (hi + lo) to the float64 expression within the
    split-image error (2e-5 of max |h|, tests/test_split3_gpu.py's bound), both hi copies are identical, and the result is as close
    to float64 as the unfused pair it replaces; the fp16 image carries the TF32-class operand rounding"""
    from dimsum_amd import native
    x = _rnd((M, H), torch.float32, 4)
    w12 = _rnd((2 * F, H), torch.float32, 5, scale=H ** -0.5)
    bias = _rnd((2 * F,), torch.float32, 6, scale=0.1) if with_bias else None
    x3, w3 = native.split3_rows(x, left=True), native.split3_rows(w12, left=False)
    got = native.gemm_nt(x3, w3, bias=bias, epilogue="gated_split3")
    assert got.shape == (M, 3 * F) and got.dtype == torch.bfloat16
    x12 = x.double() @ w12.double().t() + (0 if bias is None else bias.double())
    ref = torch.nn.functional.gelu(x12[:, :F], approximate="tanh") * x12[:, F:]
    hi, hi2, lo = got[:, :F], got[:, F:2 * F], got[:, 2 * F:]
    assert torch.equal(hi, hi2)
    scale = ref.abs().max().item()
    err = ((hi.double() + lo.double()) - ref).abs().max().item() / scale
    unf = native.gated_gelu_fwd(torch.mm(x3, w3.t(), out_dtype=torch.float32), bias, split3=True)
    uerr = ((unf[:, :F].double() + unf[:, 2 * F:].double()) - ref).abs().max().item() / scale
    assert err < 2e-5 and err < 1.5 * uerr + 1e-6, (err, uerr)
    # training forward: the same image, bit for bit, plus the bias-free accumulators [x1 | x2] the backward's adjoint reads
    got2, kept = native.gemm_nt(x3, w3, bias=bias, epilogue="gated_split3", keep_x12=True)
    assert torch.equal(got2, got)
    assert kept.shape == (M, 2 * F) and kept.dtype == torch.float32
    assert torch.equal(kept, native.gemm_nt(x3, w3))
    x16, w16 = x.half(), w12.half()
    g16 = native.gemm_nt(x16, w16, bias=bias, epilogue="gated_f16", out_scale=8.0)
    x12h = x16.double() @ w16.double().t() + (0 if bias is None else bias.double())
    refh = torch.nn.functional.gelu(x12h[:, :F], approximate="tanh") * x12h[:, F:]
    assert (g16.double() / 8.0 - refh).abs().max().item() / refh.abs().max().item() < 1e-3


def test_rejects_what_the_tiling_cannot_take():
    from dimsum_amd import native
    a, b = _rnd((255, 128), torch.bfloat16, 1), _rnd((256, 128), torch.bfloat16, 2)
    assert not native.gemm_nt_supported(a, b)
    with pytest.raises(RuntimeError):
        native.gemm_nt(a, b)
    assert not native.gemm_nt_supported(_rnd((256, 64), torch.bfloat16, 1), _rnd((256, 64), torch.bfloat16, 2))      # K < 128
    assert not native.gemm_nt_supported(_rnd((256, 128), torch.float32, 1), _rnd((256, 128), torch.float32, 2))


def test_full_size_w12_against_library():
    """BASELINE config 2's w12 launch (65536 x 8192 x 3072 split images): equal to the library's bf16 GEMM within accumulation order"""
    from dimsum_amd import native
    a, b = _rnd((65536, 3072), torch.bfloat16, 1), _rnd((8192, 3072), torch.bfloat16, 2, scale=3072 ** -0.5)
    got = native.gemm_nt(a, b)
    lib = torch.mm(a, b.t(), out_dtype=torch.float32)
    assert (got - lib).abs().max().item() <= 2e-5 * lib.abs().max().item()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("with_gate,with_bias", [(True, True), (True, False), (False, True)])
def test_gate_residual_epilogue(dtype, with_gate, with_bias):
    """out = residual + gate[row // rows_per_batch] * (a b^T + bias): the residual tail of a block (models_dim.py:1107-1113) in the
    epilogue of its last Linear, against the float64 expression; ragged N; the residual may be the output buffer's neighbour view"""
    from dimsum_amd import native
    dt = getattr(torch, dtype)
    B, L, N, K = 3, 256, 392, 320
    M = B * L
    a, b = _rnd((M, K), dt, 1), _rnd((N, K), dt, 2, scale=K ** -0.5)
    res = _rnd((M, N), torch.float32, 3)
    gate = _rnd((B, 3 * N), torch.float32, 4)[:, N:2 * N] if with_gate else None        # a chunk of the adaLN output: row stride 3 N
    bias = _rnd((N,), torch.float32, 5) if with_bias else None
    got = native.gemm_nt(a, b, bias=bias, residual=res, gate=gate, rows_per_batch=L if with_gate else None)
    y = a.double() @ b.double().t() + (0 if bias is None else bias.double())
    if gate is not None:
        y = (y.view(B, L, N) * gate.double().unsqueeze(1)).view(M, N)
    ref = res.double() + y
    assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6


def test_mlp_tail_in_the_epilogue_matches_the_separate_pass(monkeypatch):
    """GatedMLP.forward_deferred(x3=..., residual=..., gate=...) == token_ops.gate_residual(residual, mlp(x), gate, b3), split-bf16 and
    scaled-fp16 images"""
    from dimsum_amd import gemm, native
    from dimsum_amd.mlp import GatedMLP
    from dimsum_amd.ops import token_ops
    import torch.nn.functional as F
    torch.manual_seed(0)
    B, L, H = 2, 256, 384
    mlp = GatedMLP(H, 4 * H, act_layer=lambda: torch.nn.GELU(approximate="tanh")).cuda()
    x, res, gate = _rnd((B, L, H), torch.float32, 1), _rnd((B, L, H), torch.float32, 2), _rnd((B, H), torch.float32, 3)
    with torch.no_grad():
        for img in (native.split3_rows(x.reshape(B * L, H), left=True), native.rows_f16s(x.reshape(B * L, H))):
            m, mb = mlp.forward_deferred(x, x3=img)
            want = token_ops.gate_residual(res, m, gate, mb)
            got, none = mlp.forward_deferred(x, x3=img, residual=res, gate=gate)
            assert none is None and got.shape == want.shape
            assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("R,P,Q,splits", [(128, 256, 256, 1), (4096, 512, 256, 4), (12288, 256, 768, None), (3 * 2048, 1536, 512, 2)])
def test_gemm_tn_against_float64(dtype, R, P, Q, splits):
    """the weight-gradient shape C = A^T B (reduction over the rows of both operands) on the transposing-read variant of the kernel:
    float64 products of the same 16-bit operands, strided operand views (a column range of a wider matrix), bit-repeatable launches"""
    from dimsum_amd import native
    dt = getattr(torch, dtype)
    a = _rnd((R, P + 64), dt, 11)[:, 64:]                     # row stride P + 64, 128-byte aligned start
    b = _rnd((R, Q), dt, 12, scale=R ** -0.5)
    assert native.gemm_tn_supported(a, b)
    got = native.gemm_tn(a, b, splits=splits)
    ref = a.double().t() @ b.double()
    assert got.shape == (P, Q) and got.dtype == torch.float32
    assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6
    assert torch.equal(got, native.gemm_tn(a, b, splits=splits))


def test_gemm_tn_one_hot_rows_pin_the_operand_layout():
    """A = one-hot rows: C[p, :] = the sum of the B rows whose A row selects column p -- any mix-up of the transposing read's lane / row
    mapping or of the staging swizzle shows as a wrong row, not as a rounding difference"""
    from dimsum_amd import native
    R, P, Q = 1024, 256, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    sel = torch.randint(0, P, (R,), device="cuda", generator=g)
    a = torch.zeros((R, P), device="cuda", dtype=torch.bfloat16)
    a[torch.arange(R, device="cuda"), sel] = 1.0
    b = torch.randint(-8, 9, (R, Q), device="cuda", generator=g).to(torch.bfloat16)      # small integers: every sum is exact
    ref = torch.zeros((P, Q), device="cuda", dtype=torch.float32).index_add_(0, sel, b.float())
    assert torch.equal(native.gemm_tn(a, b, splits=1), ref)
    assert torch.equal(native.gemm_tn(a, b, splits=4), ref)
    assert torch.equal(native.gemm_tn(b, a, splits=2), ref.t())


def test_gemm_tn_rejects_what_it_cannot_take():
    from dimsum_amd import native
    assert not native.gemm_tn_supported(_rnd((128, 255), torch.bfloat16, 1), _rnd((128, 256), torch.bfloat16, 2))
    assert not native.gemm_tn_supported(_rnd((96, 256), torch.bfloat16, 1), _rnd((96, 256), torch.bfloat16, 2))
    with pytest.raises(RuntimeError):
        native.gemm_tn(_rnd((256, 256), torch.bfloat16, 1), _rnd((256, 256), torch.bfloat16, 2), splits=3)


def test_out_proj_from_the_scan_planes():
    """MambaInnerFn's out_proj at inference: the scan writes out_z as its split-bf16 pair of d-major planes (bit for bit the split of the
    fp32 out_z it writes otherwise), split3_rows_t stacks the transposed weight as [hi; lo; hi], and gemm_tn(alias_rows = d_inner) reads
    the pair as [hi; hi; lo]: the product is the fp32-class out_z^T W^T (3 bf16 products), against float64"""
    from dimsum_amd import native
    B, D, L, N, Q = 4, 192, 256, 16, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    rn = lambda *s: torch.randn(s, device="cuda", generator=g)
    xz = rn(2 * D, B * L)                                            # d-major like the in_proj output
    u, z = (t.view(D, B, L).permute(1, 0, 2) for t in (xz[:D], xz[D:]))
    delta = (rn(D, B * L) * 0.5).view(D, B, L).permute(1, 0, 2)
    A = -torch.rand(D, N, device="cuda", generator=g) - 0.5
    Bm, Cm = rn(B, 1, N, L), rn(B, 1, N, L)
    Dv, bias = rn(D), rn(D) * 0.1
    _, _, out_z = native.selective_scan_fwd(u, delta, A, Bm, Cm, Dv, z, bias, True)
    _, _, planes = native.selective_scan_fwd(u, delta, A, Bm, Cm, Dv, z, bias, True, out_z_planes=True)
    assert planes.shape == (2 * D, B * L) and planes.dtype == torch.bfloat16
    ref_rows = out_z.permute(1, 0, 2).reshape(D, B * L)             # (d, tokens) fp32
    hi = ref_rows.to(torch.bfloat16)
    lo = (ref_rows - hi.float()).to(torch.bfloat16)
    assert torch.equal(planes[:D], hi) and torch.equal(planes[D:], lo)
    w = rn(Q, D) * D ** -0.5
    wt = native.split3_rows_t(w)
    whi = w.t().contiguous().to(torch.bfloat16)
    wlo = (w.t().contiguous() - whi.float()).to(torch.bfloat16)
    assert torch.equal(wt, torch.cat([whi, wlo, whi], 0))
    y = native.gemm_tn(planes, wt, alias_rows=D)
    ref = ref_rows.double().t() @ w.double().t()
    assert y.shape == (B * L, Q)
    assert (y.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-5
    lib = torch.nn.functional.linear(out_z.transpose(1, 2).reshape(B * L, D), w)        # what the host layer called before (fp32 operands)
    assert (y - lib).abs().max().item() / ref.abs().max().item() < 2e-5


@pytest.mark.parametrize("M,F,H", [(512, 256, 128), (256, 1536, 384)])
def test_pair_image_is_the_same_product_bit_for_bit(M, F, H):
    """the gated epilogue's h image as the pair [hi | lo] (a third less to store) and the w3 GEMM reading it as [hi | hi | lo]
    (a_alias_rows): the same tiles in the same order -- the result equals the three-piece path bit for bit; a consumer without the
    kernel expands the pair"""
    from dimsum_amd import native
    x = _rnd((M, H), torch.float32, 4)
    w12 = _rnd((2 * F, H), torch.float32, 5, scale=H ** -0.5)
    w3 = _rnd((H, F), torch.float32, 7, scale=F ** -0.5)
    bias = _rnd((2 * F,), torch.float32, 6, scale=0.1)
    x3, w12i, w3i = native.split3_rows(x, left=True), native.split3_rows(w12, left=False), native.split3_rows(w3, left=False)
    h3 = native.gemm_nt(x3, w12i, bias=bias, epilogue="gated_split3")
    hp = native.gemm_nt(x3, w12i, bias=bias, epilogue="gated_split3", pair_out=True)
    assert isinstance(hp, native.PairImage) and hp.data.shape == (M, 2 * F)
    assert torch.equal(hp.data[:, :F], h3[:, :F]) and torch.equal(hp.data[:, F:], h3[:, 2 * F:])
    assert torch.equal(hp.image3(), h3)
    y3 = native.gemm_nt(h3, w3i)
    yp = native.gemm_nt(hp, w3i)
    assert torch.equal(yp, y3)
    res, gate = _rnd((M, H), torch.float32, 8), _rnd((M // 256, H), torch.float32, 9)
    assert torch.equal(native.gemm_nt(hp, w3i, residual=res, gate=gate, rows_per_batch=256), native.gemm_nt(h3, w3i, residual=res, gate=gate, rows_per_batch=256))
    assert not native.gemm_nt_supported(native.PairImage(hp.data[:, :F + 32]), w3i)
    # the pair on the right (in_proj: weight image x activation pair -> d-major output)
    wl = native.split3_rows(_rnd((256, F), torch.float32, 10, scale=F ** -0.5), left=False)        # (256, 3F) as the LEFT operand
    assert torch.equal(native.gemm_nt(wl, hp), native.gemm_nt(wl, h3))


def test_training_pair_images_are_the_same_products():
    """the gradient-side pair forms: a pair read in WEIGHT order [hi | lo | hi] by the NT kernel (dx = dy_w . (W^T)image) and two pairs as the
    operands of the TN kernel (dW = dy_w^T x3, three piece ranges) -- against the three-piece images they replace: the NT product bit for
    bit (same tiles, same order), the TN product to fp32 summation order"""
    from dimsum_amd import native
    M, N, K = 2048, 512, 256
    dy, x = _rnd((M, N), torch.float32, 21), _rnd((M, K), torch.float32, 22)
    wt = native.split3_rows(_rnd((K, N), torch.float32, 23, scale=N ** -0.5), left=True)          # (K, 3N): the left-order image of W^T
    dy3, dyp = native.split3_rows(dy, left=False), native.split3_rows(dy, left="pair")
    assert isinstance(dyp, native.PairImage) and torch.equal(dyp.data[:, :N], dy3[:, :N]) and torch.equal(dyp.data[:, N:], dy3[:, N:2 * N])
    assert torch.equal(native.gemm_nt(dyp, wt, weight_order=True), native.gemm_nt(dy3, wt))
    x3, xp = native.split3_rows(x, left=True), native.split3_rows(x, left="pair")
    ref = native.gemm_tn(dy3.view(3 * M, N), x3.view(3 * M, K))
    for splits in (None, 1, 2):
        got = native.gemm_tn_pairs(dyp, xp, splits=splits)
        assert got.shape == (N, K)
        assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    f64 = dy.double().t() @ x.double()
    assert (native.gemm_tn_pairs(dyp, xp).double() - f64).abs().max().item() / f64.abs().max().item() < 2e-5
    # the gated-GeLU adjoint's pair output = hi | lo of its three-piece image
    H = 256
    x12, dh, b = _rnd((M, 2 * H), torch.float32, 24), _rnd((M, H), torch.float32, 25), _rnd((2 * H,), torch.float32, 26, scale=0.1)
    d3, db3 = native.gated_gelu_bwd(x12, b, dh, split3=True)
    dp, dbp = native.gated_gelu_bwd(x12, b, dh, split3="pair")
    assert torch.equal(dp.data[:, :2 * H], d3[:, :2 * H]) and torch.equal(dp.data[:, 2 * H:], d3[:, 2 * H:4 * H]) and torch.allclose(db3, dbp, rtol=1e-4, atol=1e-4)


def test_alias_parameters_are_checked_at_the_c_abi():
    """dimsum_gemm_nt / dimsum_gemm_tn reject alias / pair parameters that do not describe a [hi | lo] pair of whole 64-column pieces"""
    import ctypes
    from dimsum_amd import _lib, native
    lib = _lib.load()
    a, b = _rnd((256, 2 * 192), torch.bfloat16, 1), _rnd((256, 3 * 192), torch.bfloat16, 2)
    c = torch.empty((256, 256), device="cuda", dtype=torch.float32)

    def params(**kw):
        P = _lib.GemmParams()
        P.m, P.n, P.k = 256, 256, 3 * 192
        P.operand_dtype, P.epilogue, P.out_scale = native._DT[torch.bfloat16], _lib.GEMM_EPI_F32, 1.0
        P.lda, P.ldb, P.ldc = a.stride(0), b.stride(0), 256
        P.a_ptr, P.b_ptr, P.c_ptr = a.data_ptr(), b.data_ptr(), c.data_ptr()
        X = _lib.attach_ext(P, _lib.GemmExt)
        for k, v in kw.items():
            setattr(X if hasattr(X, k) else P, k, v)
        return P

    s = torch.cuda.current_stream().cuda_stream
    assert lib.dimsum_gemm_nt(ctypes.byref(params(a_alias_rows=192)), s) == 0
    assert lib.dimsum_gemm_nt(ctypes.byref(params(a_alias_rows=128)), s) != 0                 # k != 3 x alias
    assert lib.dimsum_gemm_nt(ctypes.byref(params(a_alias_rows=192, k=3 * 192 - 64)), s) != 0
    assert lib.dimsum_gemm_nt(ctypes.byref(params(a_alias_weight_order=1)), s) != 0            # weight order without an alias
    assert lib.dimsum_gemm_nt(ctypes.byref(params(b_alias_rows=192, epilogue=_lib.GEMM_EPI_F32_BIAS)), s) != 0
    assert lib.dimsum_gemm_tn(ctypes.byref(params(b_alias_rows=192)), 1, 0, s) != 0
    assert lib.dimsum_gemm_tn(ctypes.byref(params(m=256, n=256, k=256, lda=2 * 256, ldb=2 * 256, tn_pair_a_cols=256, tn_pair_b_cols=256)), 2, 256 * 256, s) != 0   # splits % 3
    torch.cuda.synchronize()


# ---- the 128 x 256-tile variant (kVarM128: 4-wave workgroups, two per CU, 80-KB ring; scaled-fp16 operands with short K) ----------------
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (256, 256, 192), (512, 384, 320), (768, 1152, 512), (256, 132, 256), (512, 4, 128),
                                    (1024, 2048, 1024), (2048, 512, 2048)])
def test_m128_tiles_are_bit_identical_to_the_256_row_tiles(M, N, K):
    """the tile shape changes which workgroup owns an output element, not the order in which its products are accumulated (K tiles in
    order, the same MFMA shape): fp16 products on 128-row tiles (tune 512) == on 256-row tiles (tune 513), bit for bit, for every ring
    phase count (K / 64 = 2, 3, 5, 8, 16, 32: prologue-only, run-down and steady loops), ragged N, with and without row scales and bias;
    both against float64; repeated launches identical (a race in the ring would show as run-to-run differences)"""
    from dimsum_amd import native
    a, b = _rnd((M, K), torch.float16, 1), _rnd((N, K), torch.float16, 2)
    ref = a.double() @ b.double().t()
    small, big = native.gemm_nt(a, b, tune=(512, 0, 0)), native.gemm_nt(a, b, tune=(513, 0, 0))
    assert torch.equal(small, big)
    assert (small.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6 * max(1.0, (K / 1024) ** 0.5) + 1e-7
    for _ in range(3):
        assert torch.equal(native.gemm_nt(a, b, tune=(512, 0, 0)), small)
    sa = torch.exp2(torch.randint(-20, 20, (M,), device="cuda").float())
    sb = torch.exp2(torch.randint(-20, 20, (N,), device="cuda").float())
    bias = _rnd((N,), torch.float32, 3)
    s1 = native.gemm_nt(a, b, bias=bias, scales=(sa, sb), tune=(512, 0, 0))
    s2 = native.gemm_nt(a, b, bias=bias, scales=(sa, sb), tune=(513, 0, 0))
    assert torch.equal(s1, s2)
    want = ref * sa.double()[:, None] * sb.double()[None, :] + bias.double()
    assert ((s1.double() - want).abs() <= 3e-6 * (ref.abs().max().item() * sa.double()[:, None] * sb.double()[None, :]) + 1e-6 * bias.abs().max().item()).all()


def test_m128_one_hot_rows_land_where_they_belong():
    """A = one-hot rows against an asymmetric B on the 128-row tiles: C must be exactly the selected rows of B^T"""
    from dimsum_amd import native
    M, N, K = 768, 512, 256
    a = torch.zeros((M, K), device="cuda", dtype=torch.float16)
    idx = (torch.arange(M, device="cuda") * 7 + 3) % K
    a[torch.arange(M, device="cuda"), idx] = 1.0
    b = ((torch.arange(N, device="cuda")[:, None] * 3 + torch.arange(K, device="cuda")[None, :] * 5) % 251).to(torch.float16)
    got = native.gemm_nt(a, b, tune=(512, 0, 0))
    assert torch.equal(got, b.float()[:, idx].t().contiguous())


@pytest.mark.parametrize("with_gate", [True, False])
def test_m128_epilogues_match_the_256_row_tiles(with_gate):
    """the residual tail and the gated-GeLU fp16 epilogue (bound-derived per-row scales) on both tile shapes: identical bits"""
    from dimsum_amd import native
    B, L, N, K = 3, 256, 392, 320
    M = B * L
    a, b = _rnd((M, K), torch.float16, 1), _rnd((N, K), torch.float16, 2, scale=K ** -0.5)
    sa, sb = torch.exp2(torch.randint(-6, 6, (M,), device="cuda").float()), torch.exp2(torch.randint(-6, 6, (N,), device="cuda").float())
    res, bias = _rnd((M, N), torch.float32, 3), _rnd((N,), torch.float32, 4)
    gate = _rnd((B, N), torch.float32, 5) if with_gate else None
    kw = dict(bias=bias, residual=res, gate=gate, rows_per_batch=L if with_gate else None, scales=(sa, sb))
    o1, o2 = native.gemm_nt(a, b, tune=(512, 0, 0), **kw), native.gemm_nt(a, b, tune=(513, 0, 0), **kw)
    assert torch.equal(o1, o2)
    y = (a.double() @ b.double().t()) * sa.double()[:, None] * sb.double()[None, :] + bias.double()
    ref = res.double() + (y if gate is None else (y.view(B, L, N) * gate.double()[:, None, :]).view(M, N))
    assert (o1.double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6
    # gated GeLU -> scaled-fp16 h image
    F, H = 384, 256
    x = native.rows_f16s(_rnd((M, H), torch.float32, 6) * torch.logspace(-2, 2, M, device="cuda")[:, None])
    w12 = _rnd((2 * F, H), torch.float32, 7, scale=H ** -0.5)
    b12 = _rnd((2 * F,), torch.float32, 8, scale=0.1)
    w16, l1 = native.rows_f16s(w12, want_l1=True)
    bound = torch.cat([l1 * (1.0 + 2.0 ** -10), b12.abs().max().reshape(1)]).contiguous()
    kw = dict(bias=b12, epilogue="gated_f16", scales=(x.inv, w16.inv), gate_bound=bound)
    h1, h2 = native.gemm_nt(x.data, w16.data, tune=(512, 0, 0), **kw), native.gemm_nt(x.data, w16.data, tune=(513, 0, 0), **kw)
    assert torch.equal(h1.data.view(torch.int16), h2.data.view(torch.int16)) and torch.equal(h1.inv, h2.inv)
    x12 = x.float().double() @ w16.float().double().t() + b12.double()
    href = torch.nn.functional.gelu(x12[:, :F], approximate="tanh") * x12[:, F:]
    assert ((h1.float().double() - href).abs() / href.abs().amax(-1, keepdim=True)).max().item() < 2e-3


@pytest.mark.parametrize("dtype,tune", [("float16", 512), ("float16", 513), ("bfloat16", 513)])
@pytest.mark.parametrize("M,D,N,K,seq,width,with_bias", [(512, 256, 512, 128, 256, 4, True), (1024, 512, 768, 256, 128, 3, False), (512, 512, 1024, 512, 64, 2, True)])
def test_conv_epilogue_is_the_gemm_followed_by_the_conv_kernel(dtype, tune, M, D, N, K, seq, width, with_bias):
    """DIMSUM_GEMM_EPI_F32_CONV (in_proj + causal_conv1d_fn of a Mamba mixer, mamba_simple.py / selective_scan_interface.py:616): rows [0, D) of
    the d-major product carry silu(conv + bias) along their columns in sequences of `seq` tokens, the other rows the plain product -- against
    the plain GEMM followed by the stand-alone conv kernel on the (batch, D, seq) view of those rows (the same fp32 formula: differences at the
    level of the fused multiply-adds' order); both tile shapes, scaled-fp16 and bf16 operands, widths 2-4, several sequences per tile"""
    from dimsum_amd import native
    dt = getattr(torch, dtype)
    a, b = _rnd((M, K), dt, 1, K ** -0.5), _rnd((N, K), dt, 2)
    cw, cb = _rnd((D, width), torch.float32, 3), (_rnd((D,), torch.float32, 4) if with_bias else None)
    kw = {}
    if dtype == "float16":
        kw["scales"] = (torch.exp2(torch.randint(-4, 4, (M,), device="cuda").float()), torch.exp2(torch.randint(-4, 4, (N,), device="cuda").float()))
    plain = native.gemm_nt(a, b, tune=(tune, 0, 0), **kw)
    got = native.gemm_nt(a, b, tune=(tune, 0, 0), conv=(cw, cb, seq), **kw)
    assert torch.equal(got[D:], plain[D:])
    x = plain[:D].view(D, N // seq, seq).permute(1, 0, 2)                       # (batch, D, seq) d-major view, like MambaInnerFn's x
    ref = native.causal_conv1d_fwd(x, cw, cb, True).permute(1, 0, 2).reshape(D, N)
    assert (got[:D] - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() + 1e-7
    assert torch.equal(native.gemm_nt(a, b, tune=(tune, 0, 0), conv=(cw, cb, seq), **kw), got)


@pytest.mark.parametrize("K,M,N", [(128, 256, 256), (1024, 2048, 512), (320, 512, 256)])
def test_tn_product_of_block_scaled_fp16_rows(K, M, N):
    """dimsum_gemm_tn with a_block_inv_ptr: A (K, M) float16 whose (64-row, 32-column) blocks carry their own power-of-two scales (what the scan's
    fp16 out_z is), rebased to one scale per 32-column group as it is read, times a per-column-scaled B (K, N): against the float64 product of
    the decoded operands (exact: every factor is a power of two and nothing leaves fp16's normal range here), and against the float64 product of
    the ORIGINAL fp32 data at the TF32 class (10-bit mantissas); block magnitudes spread over 2^-12 .. 2^12"""
    from dimsum_amd import native
    from dimsum_amd.utils.tf32_emulation import round_tf32
    g = torch.Generator(device="cuda").manual_seed(K + M)
    x = torch.randn(K, M, device="cuda", generator=g)
    mag = torch.exp2(torch.randint(-12, 13, (K // 64, M // 32), device="cuda", generator=g).float())
    x = x * mag.repeat_interleave(64, 0).repeat_interleave(32, 1)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    # the producer's side: one scale per block from the block's maximum (f16s_scales: the maximum lands in [2^14, 2^15))
    bmax = x.view(K // 64, 64, M // 32, 32).abs().amax((1, 3))
    sc = torch.exp2(14 - torch.floor(torch.log2(bmax)))
    a16 = (x * sc.repeat_interleave(64, 0).repeat_interleave(32, 1)).half()
    inv = (1.0 / sc).t().contiguous()                                   # (M / 32, K / 64)
    sa_g = inv.amax(1)
    rebase = (inv / sa_g[:, None]).half()
    wimg = native.rows_f16s(w)
    b16 = wimg.data.t().contiguous()                                    # (K, N) float16, column n scaled by 2^s_n
    got = native.gemm_tn(a16, b16, scales=(inv, wimg.inv))
    a_dec = a16.double() * rebase.double().t().repeat_interleave(64, 0).repeat_interleave(32, 1)
    exact = (a_dec.t() @ b16.double()) * sa_g.repeat_interleave(32).double()[:, None] * wimg.inv.double()[None, :]
    row = exact.abs().amax(1, keepdim=True)
    assert ((got.double() - exact).abs() / row).max().item() < 3e-6
    ref = x.double().t() @ w.double().t()
    tf = round_tf32(x).double().t() @ round_tf32(w).double().t()
    e, et = (got.double() - ref).abs() / ref.abs().amax(1, keepdim=True), (tf - ref).abs() / ref.abs().amax(1, keepdim=True)
    assert e.max().item() <= 1.1 * et.max().item() + 1e-6 and e.pow(2).mean().sqrt().item() <= 1.1 * et.pow(2).mean().sqrt().item() + 1e-7, (e.max().item(), et.max().item())
    assert torch.equal(native.gemm_tn(a16, b16, scales=(inv, wimg.inv)), got)
    with pytest.raises(Exception):
        native.gemm_tn(a16, b16, scales=(inv[:, :-1].contiguous(), wimg.inv))


@pytest.mark.parametrize("M,F,K", [(16384, 1024, 256), (12800, 768, 512), (33280, 256, 1024)])
def test_persistent_stream_is_the_one_tile_per_workgroup_kernel_bit_for_bit(M, F, K):
    """kVarPersist (csrc/gemm_nt_kernel.hpp): one workgroup per CU walks the tile list as ONE stream of K tiles -- the last two K tiles of an
    output tile stage the first two of the next, the gated epilogue stages h in the 32 KB behind the ring while those DMAs are in flight.
    Same arithmetic in the same order as the one-tile-per-workgroup kernel (tune 513): identical bits, for the launch heuristics' own choice
    (tune None) and the forced one (514); 512 / 300 / 260 tiles (whole rounds, a partial second round, four workgroups with a second tile),
    K / 64 = 4, 8, 16; the residual-tail epilogue's persistent build (A / B only) too."""
    from dimsum_amd import native
    x = native.rows_f16s(_rnd((M, K), torch.float32, 1) * torch.logspace(-1, 1, M, device="cuda")[:, None])
    w16, l1 = native.rows_f16s(_rnd((2 * F, K), torch.float32, 2, scale=K ** -0.5), want_l1=True)
    b12 = _rnd((2 * F,), torch.float32, 3, scale=0.1)
    bound = torch.cat([l1 * (1.0 + 2.0 ** -10), b12.abs().max().reshape(1)]).contiguous()
    kw = dict(bias=b12, epilogue="gated_f16", scales=(x.inv, w16.inv), gate_bound=bound)
    plain = native.gemm_nt(x.data, w16.data, tune=(513, 0, 0), **kw)
    for tune in (None, (514, 0, 0)):
        got = native.gemm_nt(x.data, w16.data, tune=tune, **kw)
        assert torch.equal(got.data.view(torch.int16), plain.data.view(torch.int16)) and torch.equal(got.inv, plain.inv)
    x12 = x.float().double() @ w16.float().double().t() + b12.double()
    href = torch.nn.functional.gelu(x12[:, :F], approximate="tanh") * x12[:, F:]
    assert ((plain.float().double() - href).abs() / href.abs().amax(-1, keepdim=True)).max().item() < 2e-3
    # residual tail: out = res + gate * (x w^T + b) with N = 2 F columns
    N = 2 * F
    if N % 256 == 0:
        res, gate = _rnd((M, N), torch.float32, 4), _rnd((M // 256, N), torch.float32, 5)
        kw = dict(bias=b12, residual=res, gate=gate, rows_per_batch=256, scales=(x.inv, w16.inv))
        assert torch.equal(native.gemm_nt(x.data, w16.data, tune=(514, 0, 0), **kw), native.gemm_nt(x.data, w16.data, tune=(513, 0, 0), **kw))
