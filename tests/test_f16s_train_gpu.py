"""Training on the single-product carrier (policy "f16s" under autograd): the reference trains under TF32 (dimsum/train.py:20-21 sets
allow_tf32 before the training loop: forward, input-gradient and weight-gradient GEMMs all round their operands to 10 mantissa bits and
accumulate in fp32). Here every such GEMM of the block takes scaled-fp16 operand images and ONE fp16 MFMA product per element:
  * the pieces against float64 and against the emulated-TF32 arithmetic of the same product (never less accurate than TF32),
  * a DiMBlockCombined(1024) forward + backward against the REFERENCE golden (fwd, dx / dres / dc, 15 parameter gradients), with the
    emulated-TF32 run of the same block (attention core included) as the yardstick: error <= 1.1 x its rms / 2 x its maximum per checked tensor."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tf32(x):
    from dimsum_amd.utils.tf32_emulation import round_tf32
    return round_tf32(x)


def _errs(got, ref):
    e = (got.double() - ref).abs()
    return e.max().item(), e.pow(2).mean().sqrt().item()


@pytest.fixture
def f16s_train(monkeypatch):
    from dimsum_amd import gemm
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    gemm.set_policy("f16s")
    yield
    gemm.set_policy("default")


@pytest.mark.parametrize("M,P,Q,adversarial", [(2048, 256, 256, False), (4096, 512, 256, True), (65536, 256, 512, False), (32768, 256, 256, True)])
def test_weight_gradient_product_with_row_factors_vs_tf32(M, P, Q, adversarial):
    """dW = dy^T x from two scaled-fp16 images whose row scales become per-reduction-row factors inside the TN kernel (dimsum_gemm_ext_t.k_scale_ptr):
    against the float64 product, never less accurate than the same product with TF32-rounded operands. adversarial: row magnitudes of BOTH
    operands log-uniform over 2^-30 .. 1 (independent draws: the per-row products span 2^-60 .. 1), element magnitudes inside a row log-uniform
    over 2^-12 .. 1, a block of all-zero rows, one row 2^20 above everything else. Several reduction ranges (65536 rows: ranges of <= 16384)."""
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(M + P)
    dy, x = torch.randn(M, P, device="cuda", generator=g), torch.randn(M, Q, device="cuda", generator=g)
    if adversarial:
        for t in (dy, x):
            t *= torch.exp2(-12 * torch.rand(t.shape, device="cuda", generator=g))
            t *= torch.exp2(-30 * torch.rand(M, 1, device="cuda", generator=g))
        dy[100:164] = 0
        x[300:310] = 0
        dy[7] *= 2.0 ** 20
    dy16, x16 = native.rows_f16s(dy), native.rows_f16s(x)
    fac, cs = native.row_factors(dy16.inv, x16.inv)
    assert fac.dtype == torch.float16 and fac.max().item() == 1.0 and torch.all((fac == 0) | (torch.frexp(fac.float())[0] == 0.5))
    got = native.gemm_tn(dy16.data, x16.data, row_scales=(fac, cs))
    again = native.gemm_tn(dy16.data, x16.data, row_scales=(fac, cs))
    assert torch.equal(got, again)
    ref = dy.double().t() @ x.double()
    tf = (_tf32(dy).double().t() @ _tf32(x).double())                     # TF32 operand rounding, exact accumulation: the floor of what TF32 can do
    e_max, e_rms = _errs(got, ref)
    t_max, t_rms = _errs(tf.float(), ref)
    scale = ref.abs().max().item()
    print(f"dW ({M} rows{' adversarial' if adversarial else ''}): f16s {e_max / scale:.2e} / {e_rms / scale:.2e}, TF32 operands {t_max / scale:.2e} / {t_rms / scale:.2e}")
    assert e_max <= 1.1 * t_max + 2.0 ** -22 * scale and e_rms <= 1.1 * t_rms + 2.0 ** -24 * scale
    # the same product with the factors formed INSIDE the kernel from the row scales (ext->k_inv_a_ptr / k_inv_b_ptr: what training runs -- no factor
    # launch in front of the GEMM): every range normalised by its own maximum. Same bars; bit-identical to the table route where the ranges' maxima
    # coincide with the tensor's (plain data: all ranges hold a row at the top scale)
    inside = native.gemm_tn(dy16.data, x16.data, row_invs=(dy16.inv, x16.inv))
    assert torch.equal(inside, native.gemm_tn(dy16.data, x16.data, row_invs=(dy16.inv, x16.inv)))
    i_max, i_rms = _errs(inside, ref)
    assert i_max <= 1.1 * t_max + 2.0 ** -22 * scale and i_rms <= 1.1 * t_rms + 2.0 ** -24 * scale
    if not adversarial:
        assert (inside - got).abs().max().item() <= 2.0 ** -20 * scale
    # the order of the factors inside a fragment: a product whose only non-zero reduction row is r picks dy[r] x[r]^T, for rows of every residue mod 64
    for r in (0, 1, 5, 18, 23, 33, 47, 63, 64 + 38, M - 1):
        dz, xz = torch.zeros_like(dy), torch.zeros_like(x)
        dz[r], xz[r] = dy[r], x[r]
        if dz[r].abs().max() == 0 or xz[r].abs().max() == 0:
            continue
        a16, b16 = native.rows_f16s(dz), native.rows_f16s(xz)
        # every other row keeps a factor (its scale is the zero row's 2^126 clamp): give the live row a small one so that a misplaced factor shows
        one = native.gemm_tn(a16.data, b16.data, row_scales=native.row_factors(a16.inv, b16.inv))
        want = torch.outer(dz[r].double(), xz[r].double())
        assert (one.double() - want).abs().max().item() <= 2.0 ** -9 * want.abs().max().item(), r
        one = native.gemm_tn(a16.data, b16.data, row_invs=(a16.inv, b16.inv))
        assert (one.double() - want).abs().max().item() <= 2.0 ** -9 * want.abs().max().item(), ("in-kernel factors", r)


def test_row_factor_limits_are_refused():
    from dimsum_amd import native
    a, b = torch.randn(32768, 256, device="cuda").half(), torch.randn(32768, 256, device="cuda").half()
    fac, cs = torch.ones(32768, device="cuda", dtype=torch.float16), torch.ones(1, device="cuda")
    with pytest.raises(RuntimeError):
        native.gemm_tn(a, b, splits=1, row_scales=(fac, cs))              # one range of 32768 rows: the factors do not fit the 32 KB behind the ring
    out = native.gemm_tn(a, b, row_scales=(fac, cs))                       # (the host layer cuts the reduction itself)
    ref = a.double().t() @ b.double()
    assert (out.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("rows,H", [(512, 4096), (300, 1024), (64, 4608), (130, 512)])
def test_gated_gelu_adjoint_as_scaled_fp16_image(rows, H):
    """dimsum_gated_gelu_bwd_f16s: the image decodes to the fp32 kernel's dx12 to half an fp16 ulp of each row's maximum, the scales are the exact
    row maxima's powers of two, d bias agrees; rows that are not a multiple of the workgroup's 128, H = 4.5 x 1024 (DiM-XL/2)"""
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(rows + H)
    x12, bias = torch.randn(rows, 2 * H, device="cuda", generator=g), 0.1 * torch.randn(2 * H, device="cuda", generator=g)
    dh = torch.randn(rows, H, device="cuda", generator=g) * torch.exp2(-20 * torch.rand(rows, 1, device="cuda", generator=g))
    dh[3] = 0
    ref, db_ref = native.gated_gelu_bwd(x12, bias, dh)
    img, db = native.gated_gelu_bwd(x12, bias, dh, split3="f16s")
    assert img.data.dtype == torch.float16 and img.data.shape == (rows, 2 * H) and img.inv.shape == (rows,)
    rmax = ref.abs().amax(1)
    top = img.data.float().abs().amax(1)
    live = rmax > 0
    assert torch.all(top[live] >= 2.0 ** 14) and torch.all(top < 2.0 ** 15)
    assert torch.all(torch.frexp(img.inv)[0] == 0.5)
    err = (img.float() - ref).abs().amax(1)
    assert torch.all(err <= 2.0 ** -11 * rmax * 1.001 + 1e-37), (err / rmax.clamp_min(1e-30)).max().item()
    assert torch.all(img.data[3] == 0)
    assert (db - db_ref).abs().max().item() <= 1e-5 * db_ref.abs().max().item() + 1e-6


def test_training_gate_epilogue_keeps_x12_in_true_units():
    """gated_f16 + keep_x12 over scaled-fp16 operands (the training forward of w12): the stored fp32 [x1 | x2] is the bias-free product in TRUE
    units (accumulator x row scale x column scale) -- bit for bit the plain fp32-output launch of the same operands -- and the h image is the
    inference launch's"""
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(11)
    M, K, F = 512, 256, 384
    x = torch.randn(M, K, device="cuda", generator=g) * torch.exp2(8 * torch.rand(M, 1, device="cuda", generator=g) - 4)
    w, b = torch.randn(2 * F, K, device="cuda", generator=g) * K ** -0.5, 0.1 * torch.randn(2 * F, device="cuda", generator=g)
    x16 = native.rows_f16s(x)
    w16, l1 = native.rows_f16s(w, want_l1=True)
    bound = torch.cat([l1 * gemm._K10, b.abs().max().reshape(1)]).contiguous()
    h_ref = native.gemm_nt(x16.data, w16.data, bias=b, epilogue="gated_f16", scales=(x16.inv, w16.inv), gate_bound=bound, tune=(513, 0))
    h16, x12 = native.gemm_nt(x16.data, w16.data, bias=b, epilogue="gated_f16", scales=(x16.inv, w16.inv), gate_bound=bound, keep_x12=True)
    assert torch.equal(h16.data, h_ref.data) and torch.equal(h16.inv, h_ref.inv)
    plain = native.gemm_nt(x16.data, w16.data, scales=(x16.inv, w16.inv), tune=(513, 0))
    assert torch.equal(x12, plain)


def _mlp(H, Fh):
    from dimsum_amd.mlp import GatedMLP
    torch.manual_seed(5)
    m = GatedMLP(in_features=H, hidden_features=Fh, act_layer=lambda: torch.nn.GELU(approximate="tanh")).cuda()
    return m


def test_gated_mlp_forward_backward_on_one_product_vs_tf32(f16s_train):
    """mlp(modulate(normed)) under autograd, policy "f16s": output, d normed / d shift / d scale and the four parameter gradients against the
    float64 evaluation of the same function; per tensor, error <= 1.25 x max / 1.1 x rms of the emulated-TF32 evaluation's
    (every matmul of forward AND backward on operands rounded to 10 mantissa bits)"""
    from dimsum_amd import gemm
    from dimsum_amd.mlp import _ModGatedMlpF16sFn, mod_gated_mlp_images
    from dimsum_amd.utils.tf32_emulation import emulated_tf32
    B, L, H, Fh = 8, 256, 256, 512
    mlp = _mlp(H, Fh)
    g = torch.Generator(device="cuda").manual_seed(3)
    normed = torch.randn(B, L, H, device="cuda", generator=g)
    shift, scale = 0.3 * torch.randn(B, H, device="cuda", generator=g), 0.3 * torch.randn(B, H, device="cuda", generator=g)
    dout = torch.randn(B, L, H, device="cuda", generator=g)

    def run(fn, dtype=torch.float32):
        ins = [t.detach().clone().to(dtype).requires_grad_() for t in (normed, shift, scale)]
        ps = [p.detach().clone().to(dtype).requires_grad_() for p in (mlp.w12.weight, mlp.w12.bias, mlp.w3.weight)]
        out = fn(ins, ps)
        out.backward(dout.to(dtype))
        return [out.detach()] + [t.grad for t in ins + ps]

    def plain(ins, ps):
        h = ins[0] * (1 + ins[2].unsqueeze(1)) + ins[1].unsqueeze(1)
        x1, x2 = torch.nn.functional.linear(h, ps[0], ps[1]).chunk(2, -1)
        return torch.nn.functional.linear(torch.nn.functional.gelu(x1, approximate="tanh") * x2, ps[2])

    seen = []
    real = _ModGatedMlpF16sFn.apply

    def ours(ins, ps):
        seen.append(gemm.split3_train_enabled(ins[0], ps[0]))
        return real(ins[0], ins[1], ins[2], ps[0], ps[1], ps[2])
    got = run(ours)
    assert seen == ["f16s"]
    ref = run(plain, torch.float64)
    gemm.set_policy("default")
    with emulated_tf32():
        emu = run(plain)
    for name, a, e, r in zip(("out", "d normed", "d shift", "d scale", "d w12", "d b12", "d w3"), got, emu, ref):
        (am, ar), (em, er) = _errs(a, r), _errs(e, r)
        s = r.abs().max().item()
        print(f"{name}: f16s {am / s:.2e} / {ar / s:.2e}   emulated TF32 {em / s:.2e} / {er / s:.2e}")
        assert am <= 1.25 * em + 1e-7 * s and ar <= 1.1 * er + 1e-8 * s, name
    # ... and the module-level switch takes this path under the policy
    gemm.set_policy("f16s")
    m, mb = mod_gated_mlp_images(mlp, normed.requires_grad_(), shift, scale)
    assert m.grad_fn is not None and "F16s" in type(m.grad_fn).__name__


def test_linear_forward_backward_on_one_product_vs_tf32(f16s_train):
    """gemm.linear under autograd, policy "f16s" (qkv / proj of the fusion): y, dx, dW vs float64, within the emulated-TF32 errors; rows with
    very different magnitudes in x and in dy"""
    from dimsum_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(9)
    M, K, N = 2048, 512, 768
    x = torch.randn(M, K, device="cuda", generator=g) * torch.exp2(16 * torch.rand(M, 1, device="cuda", generator=g) - 8)
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    dy = torch.randn(M, N, device="cuda", generator=g) * torch.exp2(16 * torch.rand(M, 1, device="cuda", generator=g) - 8)
    xs, ws = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = gemm.linear(xs, ws)
    assert "F16s" in type(y.grad_fn).__name__
    y.backward(dy)
    xd, wd, dyd = x.double(), w.double(), dy.double()
    for name, got, ref, tf in (("y", y.detach(), xd @ wd.t(), _tf32(x).double() @ _tf32(w).double().t()),
                               ("dx", xs.grad, dyd @ wd, _tf32(dy).double() @ _tf32(w).double()),
                               ("dW", ws.grad, dyd.t() @ xd, _tf32(dy).double().t() @ _tf32(x).double())):
        (am, ar), (tm, tr) = _errs(got, ref), _errs(tf.float(), ref)
        s = ref.abs().max().item()
        print(f"{name}: f16s {am / s:.2e} / {ar / s:.2e}   TF32 operands {tm / s:.2e} / {tr / s:.2e}")
        assert am <= 1.1 * tm + 2.0 ** -22 * s and ar <= 1.1 * tr + 2.0 ** -24 * s, name


def test_block_combined_1024_fwd_bwd_on_one_product_vs_reference_golden(f16s_train):
    """BASELINE configs[2]'s block under the single-product TRAINING carrier against the reference golden (forward, dx / dres / dc and 15 parameter
    gradients from the reference block in exact fp32 on the CPU): every tensor within the north star's 1e-3 (of its largest element) and not
    further from the golden than 1.1 x the rms / 2 x the maximum error of the emulated-TF32 run of the same block -- the reference's own
    training arithmetic (forward and backward matmuls, the attention core's included, on operands rounded to 10 mantissa bits)."""
    from dimsum_amd import gemm, utils
    from dimsum_amd.utils.tf32_emulation import emulated_tf32
    from test_model_cpu import check_block_1024
    before = utils.torch_path_counts()
    ours = {}
    check_block_1024("cuda", None, None, collect=ours)
    assert utils.torch_path_counts() == before
    gemm.set_policy("default")
    emu = {}
    with emulated_tf32():
        check_block_1024("cuda", None, None, collect=emu)
    worst = 0.0
    for key in ours:
        got, ref = ours[key]
        e1 = np.abs(got.astype(np.float64) - ref)
        e2 = np.abs(emu[key][0].astype(np.float64) - ref)
        s = np.abs(ref).max()
        print(f"{key:20s} f16s {e1.max() / s:.2e} / {np.sqrt((e1 ** 2).mean()) / s:.2e}   emulated TF32 {e2.max() / s:.2e} / {np.sqrt((e2 ** 2).mean()) / s:.2e}")
        # the north star's 1e-3 -- or, for the few tensors where the reference's own TF32 arithmetic does not hold 1e-3 against its exact-fp32 golden
        # (g_A_log, g_x_proj, g_in_proj: emulated TF32 1.2e-3 .. 1.4e-3), no worse than that arithmetic. The rms error is the robust statistic
        # (<= 1.1 x the emulated run's); the maximum is ONE element of up to 4 M and moves by +-30 % between two roundings of the same
        # arithmetic (the emulated run's own g_A_log maximum: 1.41e-3 with the exact attention core, 1.24e-3 with the rounded one) -- 2 x.
        assert e1.max() <= max(1e-3 * s, 2.0 * e2.max()), key
        assert np.sqrt((e1 ** 2).mean()) <= 1.1 * np.sqrt((e2 ** 2).mean()) + 2e-6 * s, key
        worst = max(worst, e1.max() / s)
    print("worst relative-to-max error under the single-product training carrier:", worst)


@pytest.mark.parametrize("P,R,Q", [(256, 2048, 256), (2048, 65536, 512), (1024, 16384, 512)])
def test_mixed_layout_weight_gradient_product_vs_tf32(P, R, Q):
    """dimsum_gemm_nn: C = A B with A (P, R) rows contiguous along the reduction (a d-major activation: channels x tokens, one scale per channel)
    and B (R, Q) rows over it (token-major, one scale per token -> per-reduction-row factors): d in_proj.weight = dxz x and d out_proj.weight^T =
    out_z dout of a Mamba mixer (selective_scan_interface.py:954-981). Against float64, never less accurate than TF32-rounded operands; token
    magnitudes over 2^-12 .. 1; the long-row image kernel (one workgroup per row) against the per-wave one."""
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(P + R)
    a = torch.randn(P, R, device="cuda", generator=g) * torch.exp2(6 * torch.rand(P, 1, device="cuda", generator=g) - 3)
    b = torch.randn(R, Q, device="cuda", generator=g) * torch.exp2(-12 * torch.rand(R, 1, device="cuda", generator=g))
    a16, b16 = native.rows_f16s(a), native.rows_f16s(b)
    if R > 8192:        # the long-row kernel wrote a16: the same image as the generic two-pass kernel on a column slice
        ref16 = native.rows_f16s(a[:, :4096].contiguous())
        wide = a.abs().amax(1) == a[:, :4096].abs().amax(1)
        assert wide.any() and torch.equal(a16.data[wide][:, :4096], ref16.data[wide]) and torch.equal(a16.inv[wide], ref16.inv[wide])
    got = native.gemm_nn(a16.data, a16.inv, b16.data, b16.inv)
    assert torch.equal(got, native.gemm_nn(a16.data, a16.inv, b16.data, b16.inv))
    ref = a.double() @ b.double()
    tf = _tf32(a).double() @ _tf32(b).double()
    (em, er), (tm, tr) = _errs(got, ref), _errs(tf.float(), ref)
    s = ref.abs().max().item()
    print(f"NN ({P} x {R} x {Q}): f16s {em / s:.2e} / {er / s:.2e}, TF32 operands {tm / s:.2e} / {tr / s:.2e}")
    assert em <= 1.1 * tm + 2.0 ** -22 * s and er <= 1.1 * tr + 2.0 ** -24 * s
    # a single live reduction row of every residue class picks a[:, r] b[r, :]
    for r in (0, 3, 9, 20, 37, 62, 64 + 45, R - 1):
        az, bz = torch.zeros_like(a), torch.zeros_like(b)
        az[:, r], bz[r] = a[:, r], b[r]
        x16, y16 = native.rows_f16s(az), native.rows_f16s(bz)
        one = native.gemm_nn(x16.data, x16.inv, y16.data, y16.inv)
        want = torch.outer(az[:, r].double(), bz[r].double())
        assert (one.double() - want).abs().max().item() <= 2.0 ** -9 * want.abs().max().item(), r
