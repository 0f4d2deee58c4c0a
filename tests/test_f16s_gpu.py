"""Scaled-fp16 operand images ("f16s", csrc/common.hpp): the single-product carrier of the reference's TF32 matmul policy
(dimsum/train.py:20-21). A row travels as fp16(row * 2^s) + the exact inverse scale; one v_mfma_f32_16x16x32_f16 product per element
with fp32 accumulation is then the TF32 arithmetic itself (10-bit mantissas, fp32 sums), with the range taken care of by construction.
Checked: the images against torch's own fp16 rounding, the producers against the stand-alone converter, the products against float64
next to an emulated-TF32 product (dimsum_amd/utils/tf32_emulation.py), adversarial ranges, and blocks / models against exact fp32 next to
the emulated-TF32 run of the same module."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _expected_image(x):
    m = x.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(m.double().clamp_min(2.0 ** -112))).clamp(-112, 127)
    scale = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=x.device), 14 - e).float()
    return (x * scale).half(), (1.0 / scale.double()).float().squeeze(-1)


def test_rows_f16s_bit_exact_and_in_range():
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(67, 200, device="cuda", generator=g) * torch.logspace(-30, 30, 67, device="cuda")[:, None]
    x[5] = 0.0
    x[6, 3] = 1e4 * x[6].abs().max()                      # an outlier
    buf = torch.zeros(67, 208, device="cuda")
    buf[:, :200] = x
    for src in (x, buf[:, :200]):
        img, l1 = native.rows_f16s(src, want_l1=True)
        want, inv = _expected_image(src)
        assert img.data.dtype == torch.float16 and tuple(img.data.shape) == (67, 200)
        assert torch.equal(img.data.view(torch.int16), want.view(torch.int16)) and torch.equal(img.inv, inv)
        assert torch.isfinite(img.data).all()
        top = img.data.float().abs().amax(-1)
        assert ((top >= 2.0 ** 14) & (top < 2.0 ** 15))[src.abs().amax(-1) > 0].all()
        assert abs(l1.item() - src.abs().sum(-1).max().item()) <= 1e-5 * l1.item()
    # every element at or above 2^-28 of its row's maximum keeps fp16's full significand: relative error <= 2^-11
    img = native.rows_f16s(x)
    back = img.float()
    big = x.abs() >= x.abs().amax(-1, keepdim=True) * 2.0 ** -28
    assert ((back - x).abs()[big] <= x.abs()[big] * 2.0 ** -11).all()


def test_norm_and_token_passes_write_the_image_of_their_fp32_output():
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, L = 96, 384, 16
    x, res = torch.randn(M, N, device="cuda", generator=g) * torch.logspace(-4, 4, M, device="cuda")[:, None], torch.randn(M, N, device="cuda", generator=g)
    w, xb = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    sc, sh = 0.1 * torch.randn(M // L, N, device="cuda", generator=g), torch.randn(M // L, N, device="cuda", generator=g)
    kw = dict(residual=res, is_rms_norm=True, x_bias=xb, mod_scale=sc, mod_shift=sh, rows_per_batch=L)
    y, _, rstd, r = native.layer_norm_fwd(x, w, None, 1e-5, **kw)
    yi, _, rstd2, r2 = native.layer_norm_fwd(x, w, None, 1e-5, split3="f16s", **kw)
    want = native.rows_f16s(y)
    assert torch.equal(yi.data.view(torch.int16), want.data.view(torch.int16)) and torch.equal(yi.inv, want.inv)
    assert torch.equal(rstd, rstd2) and torch.equal(r, r2)
    B, L, C = 3, 64, 40
    for kind in ("none", "haar", "dct"):
        wide = torch.randn(B, L, 2 * C, device="cuda", generator=g) * torch.logspace(-3, 3, L, device="cuda")[None, :, None]
        xs, rs = wide[:, :, :C], wide[:, :, C:]
        sc, sh = 0.1 * torch.randn(B, C, device="cuda", generator=g), torch.randn(B, C, device="cuda", generator=g)
        perm = torch.randperm(L, device="cuda", generator=g).to(torch.int32)
        for fwd, kw in ((True, dict(out_index=perm, scale=sc, shift=sh)), (False, dict(in_index=perm, gate=sc, residual=rs))):
            y = native.token_transform(xs, kind, fwd, **kw)
            yi = native.token_transform(xs, kind, fwd, split3="f16s", **kw)
            want = native.rows_f16s(y.reshape(B * L, C))
            assert torch.equal(yi.data.reshape(B * L, C).view(torch.int16), want.data.view(torch.int16)), (kind, fwd)
            assert torch.equal(yi.inv.reshape(-1), want.inv)
    # full width of a DiM-L/2 branch (512 channels) and of a whole block (1024)
    for C in (512, 1024):
        xs = torch.randn(2, 256, C, device="cuda", generator=g)
        sc, sh = 0.1 * torch.randn(2, C, device="cuda", generator=g), torch.randn(2, C, device="cuda", generator=g)
        y = native.token_transform(xs, "haar", True, scale=sc, shift=sh)
        yi = native.token_transform(xs, "haar", True, scale=sc, shift=sh, split3="f16s")
        want = native.rows_f16s(y.reshape(512, C))
        assert torch.equal(yi.data.reshape(512, C).view(torch.int16), want.data.view(torch.int16)) and torch.equal(yi.inv.reshape(-1), want.inv)


def _errs(got, ref):
    d = (got.double() - ref).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("M,K,N", [(512, 256, 384), (1024, 1024, 2048)])
@pytest.mark.parametrize("adversarial", [False, True])
def test_product_is_never_less_accurate_than_tf32(M, K, N, adversarial):
    """x W^T on scaled-fp16 images against float64, next to the emulated TF32 product of the same operands: the same 10-bit
    mantissas, so the same error (<= 1.05 x in max and rms). adversarial: rows scaled by 10^-6 .. 10^6 plus one 10^4 outlier per row --
    what plain fp16 operands cannot carry (overflow / flush) and per-row scales absorb exactly."""
    from dimsum_amd.utils.tf32_emulation import round_tf32
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(5)
    x, w = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    if adversarial:
        x = x * torch.logspace(-6, 6, M, device="cuda")[:, None]
        x[torch.arange(M), torch.randint(0, K, (M,), device="cuda", generator=g)] *= 1e4
        w = w * torch.logspace(-3, 3, N, device="cuda")[:, None]
    ref = x.double() @ w.double().t()
    row = ref.abs().amax(-1, keepdim=True)                                        # errors are judged per output row (its own magnitude)
    got = gemm._nt_f16s(native.rows_f16s(x), native.rows_f16s(w))
    tf = (round_tf32(x).double() @ round_tf32(w).double().t()).float()           # exact products of TF32 operands, then one fp32 rounding
    e_got, e_tf = ((got.double() - ref).abs() / row), ((tf.double() - ref).abs() / row)
    assert torch.isfinite(got).all()
    assert e_got.max().item() <= 1.05 * e_tf.max().item() + 1e-6, (e_got.max().item(), e_tf.max().item())
    assert e_got.pow(2).mean().sqrt().item() <= 1.05 * e_tf.pow(2).mean().sqrt().item() + 1e-7
    if adversarial:                                                               # ... where unscaled fp16 operands fail outright
        plain = torch.mm(x.half(), w.half().t(), out_dtype=torch.float32)
        assert not torch.isfinite(plain).all() or ((plain.double() - ref).abs() / row).max().item() > 10 * e_got.max().item()


@pytest.mark.parametrize("M,K,N", [(512, 256, 384), (1024, 1024, 2048)])
def test_product_with_wide_ranges_inside_a_row_vs_tf32(M, K, N):
    """the per-row scale cannot help WITHIN a row: an element below 2^-28 of its row's maximum lands in fp16's subnormal range (absolute
    error <= 2^-39 of the row maximum instead of TF32's 2^-11 of the element). Element magnitudes log-uniform over 2^-30 .. 1 inside every
    row of x AND of w (independent draws), plus: a row of x at 1e-38 (fp32's own subnormal edge: the scale clamp), a row with one 2^32
    outlier over O(1) entries (every other element subnormal after scaling), an all-zero row. The product against float64, judged per
    output row like above: still <= 1.05 x the emulated-TF32 error -- the lost elements carry 2^-28 of the row's largest term."""
    from dimsum_amd.utils.tf32_emulation import round_tf32
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(11)
    def wide(R, C, s):
        mag = torch.exp2(-30.0 * torch.rand(R, C, device="cuda", generator=g))
        return torch.randn(R, C, device="cuda", generator=g).sign() * mag * (1.0 + torch.rand(R, C, device="cuda", generator=g)) * s
    x, w = wide(M, K, 1.0), wide(N, K, K ** -0.5)
    x[3] = torch.randn(K, device="cuda", generator=g) * 1e-38
    x[4] = torch.randn(K, device="cuda", generator=g)
    x[4, 7] = 2.0 ** 32
    x[5] = 0.0
    ref = x.double() @ w.double().t()
    row = ref.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    xi = native.rows_f16s(x)
    assert xi.data[4].float().abs().max().item() >= 2.0 ** 14                                                      # the outlier owns the row's scale
    assert (xi.data[4, 8:].float().abs() < 2.0 ** -14).all()                                                       # ... everything else is subnormal fp16
    got = gemm._nt_f16s(xi, native.rows_f16s(w))
    tf = (round_tf32(x).double() @ round_tf32(w).double().t()).float()
    assert torch.isfinite(got).all() and (got[5] == 0).all()
    e_got, e_tf = (got.double() - ref).abs() / row, (tf.double() - ref).abs() / row
    keep = torch.ones(M, dtype=torch.bool, device="cuda")
    keep[3] = False          # (the 1e-38 row: its fp32 OUTPUT is subnormal -- judged on its own below)
    assert e_got[keep].max().item() <= 1.05 * e_tf[keep].max().item() + 1e-6, (e_got[keep].max().item(), e_tf[keep].max().item())
    assert e_got[keep].pow(2).mean().sqrt().item() <= 1.05 * e_tf[keep].pow(2).mean().sqrt().item() + 1e-7
    # per row too: no single row is worse than 1.5 x the TF32 error of the worst row of its own kind
    assert (e_got[keep].amax(-1) <= 1.5 * e_tf[keep].amax(-1).max() + 1e-6).all()
    # the tiny row: 1e-38 inputs, outputs ~1e-38 (fp32 subnormal steps of 1.4e-45 = 1e-7 relative): TF32-class relative to the row's own size
    assert e_got[3].max().item() <= 2.0 ** -9, e_got[3].max().item()


def test_where_the_row_scale_ends():
    """the documented limit of the construction: if the weights cancel a row's large elements EXACTLY (here: zero columns
    where x is large), the output is made of elements that sit below 2^-28 of the row maximum, whose image keeps an ABSOLUTE error
    <= 2^-39 max|x_r| per element -- the product's error is then bounded by K 2^-39 max|x_r| max|w|, not by TF32's relative bound.
    Activations out of RMSNorm / modulate / Haar / DCT passes have row-internal ranges of 2^10 .. 2^15 (measured on DiM-L/2: the model
    tests), 13 octaves short of this regime."""
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(12)
    M, K, N = 256, 256, 256
    x = torch.randn(M, K, device="cuda", generator=g) * 2.0 ** -31
    x[:, :8] = torch.randn(M, 8, device="cuda", generator=g)            # 8 large columns ...
    w = torch.randn(N, K, device="cuda", generator=g)
    w[:, :8] = 0.0                                                      # ... that the weights ignore
    ref = x.double() @ w.double().t()
    got = gemm._nt_f16s(native.rows_f16s(x), native.rows_f16s(w))
    bound = K * 2.0 ** -39 * x.abs().amax(-1, keepdim=True).double() * w.abs().max().item() + 2.0 ** -10 * (x.abs().double() @ w.abs().double().t())
    assert ((got.double() - ref).abs() <= bound).all()


def test_gated_mlp_on_scaled_images_vs_tf32():
    """w12 + bias + gelu_tanh * gate -> h image (per-row scale from the bound, no row reduction) -> w3: against float64 next to the
    emulated-TF32 evaluation of the same MLP; with rows of very different magnitude and outliers"""
    from dimsum_amd.utils.tf32_emulation import round_tf32
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(6)
    M, H, F = 1024, 512, 2048
    x = torch.randn(M, H, device="cuda", generator=g) * torch.logspace(-3, 3, M, device="cuda")[:, None]
    x[torch.arange(M), torch.randint(0, H, (M,), device="cuda", generator=g)] *= 50.0
    w12, b12 = torch.randn(2 * F, H, device="cuda", generator=g) * H ** -0.5, 0.1 * torch.randn(2 * F, device="cuda", generator=g)
    w3 = torch.randn(H, F, device="cuda", generator=g) * F ** -0.5

    def mlp64(xx, a, b):
        x12 = xx.double() @ a.double().t() + b12.double()
        h = torch.nn.functional.gelu(x12[:, :F], approximate="tanh") * x12[:, F:]
        return h, h @ b.double().t()
    h_ref, y_ref = mlp64(x, w12, w3)
    himg = gemm.gated_mlp_hidden_split3(native.rows_f16s(x), w12, b12)
    assert isinstance(himg, native.F16Image) and torch.isfinite(himg.data).all()
    assert himg.data.float().abs().max().item() < 2.0 ** 15                       # the bound-derived scale keeps every row in range
    y = gemm.linear_split3(himg, w3)
    # emulated TF32: both GEMMs on operands rounded to 10 mantissa bits, everything else in float64
    x12 = round_tf32(x).double() @ round_tf32(w12).double().t() + b12.double()
    h_tf = (torch.nn.functional.gelu(x12[:, :F], approximate="tanh") * x12[:, F:]).float()
    y_tf = round_tf32(h_tf).double() @ round_tf32(w3).double().t()
    row = y_ref.abs().amax(-1, keepdim=True)
    e, e_tf = (y.double() - y_ref).abs() / row, (y_tf - y_ref).abs() / row
    assert e.max().item() <= 1.1 * e_tf.max().item() + 1e-6, (e.max().item(), e_tf.max().item())
    assert e.pow(2).mean().sqrt().item() <= 1.1 * e_tf.pow(2).mean().sqrt().item() + 1e-7
    hrow = h_ref.abs().amax(-1, keepdim=True)
    assert ((himg.float().double() - h_ref).abs() / hrow).max().item() <= 1.1 * ((h_tf.double() - h_ref).abs() / hrow).max().item() + 1e-6


def _block_1024():
    from procedural import procedural_fill, seeded
    from dimsum_amd.models_dim import create_block
    blk = create_block(1024, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=True, transpose=True, cond_mamba=True,
                       scanning_continuity=True, use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    T = torch.from_numpy
    x, res, cc = (T(seeded(sh, sd)).cuda() for sh, sd in (((2, 256, 1024), 56), ((2, 256, 1024), 57), ((2, 1024), 58)))
    return blk.cuda().eval(), (x, res, cc)


def test_block_under_f16s_policy_vs_emulated_tf32(monkeypatch):
    """DiMBlockCombined(1024), inference: the deviation of the f16s policy from the exact-fp32 forward is not larger than the deviation
    of the emulated-TF32 forward (the reference's own arithmetic) -- and the block still meets the reference golden's tolerance"""
    from dimsum_amd.utils.tf32_emulation import emulated_tf32
    from dimsum_amd import gemm
    blk, args = _block_1024()
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    old = torch.backends.cuda.matmul.allow_tf32
    try:
        with torch.no_grad():
            torch.backends.cuda.matmul.allow_tf32 = False
            ref = blk(*args)[0].double()
            with emulated_tf32():
                tf = blk(*args)[0]
            torch.backends.cuda.matmul.allow_tf32 = True
            three = blk(*args)[0]
            gemm.set_policy("f16s")
            one = blk(*args)[0]
    finally:
        gemm.set_policy("default")
        torch.backends.cuda.matmul.allow_tf32 = old
    scale = ref.abs().max().item()
    (m1, r1), (mt, rt), (m3, r3) = _errs(one, ref), _errs(tf, ref), _errs(three, ref)
    print(f"block_1024: max / rms deviation from exact fp32 over max|y|: f16s {m1 / scale:.2e} / {r1 / scale:.2e}, emulated TF32 {mt / scale:.2e} / {rt / scale:.2e}, "
          f"3-product {m3 / scale:.2e} / {r3 / scale:.2e}")
    assert not torch.equal(one, three)
    assert m1 <= 1.25 * mt and r1 <= 1.1 * rt, ((m1, r1), (mt, rt))
    assert m1 / scale < 1e-3


@pytest.mark.parametrize("name,image_size,B", [("DiM-L/2", 256, 32), ("DiM-XL/2", 512, 8)])
def test_model_under_f16s_policy_vs_emulated_tf32(name, image_size, B):
    """the whole denoiser (reference init, zero tensors re-drawn): f16s deviation from exact fp32 <= emulated-TF32 deviation"""
    from dimsum_amd.utils.tf32_emulation import emulated_tf32
    from dimsum_amd import gemm
    from dimsum_amd.create_model import create_model, published_config
    from dimsum_amd.utils import rerandomize_zeros
    torch.manual_seed(0)
    m = create_model(published_config(model=name, image_size=image_size))
    rerandomize_zeros(m, std=0.02, seed=0)
    m = m.cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(0)
    r = image_size // 8                                      # B x tokens = 8192 rows: the image carriers are on at their default threshold
    x, t = torch.randn(B, 4, r, r, device="cuda", generator=gen), torch.rand(B, device="cuda", generator=gen)
    y = torch.randint(0, 1000, (B,), device="cuda", generator=gen)
    old = torch.backends.cuda.matmul.allow_tf32
    try:
        with torch.no_grad():
            torch.backends.cuda.matmul.allow_tf32 = False
            ref = m(x, t, y).double()
            with emulated_tf32():
                tf = m(x, t, y)
            torch.backends.cuda.matmul.allow_tf32 = True
            three = m(x, t, y)
            gemm.set_policy("f16s")
            one = m(x, t, y)
    finally:
        gemm.set_policy("default")
        torch.backends.cuda.matmul.allow_tf32 = old
    scale = ref.abs().max().item()
    (m1, r1), (mt, rt), (m3, r3) = _errs(one, ref), _errs(tf, ref), _errs(three, ref)
    print(f"{name}: max / rms deviation from exact fp32 over max|out|: f16s {m1 / scale:.2e} / {r1 / scale:.2e}, emulated TF32 {mt / scale:.2e} / {rt / scale:.2e}, "
          f"3-product {m3 / scale:.2e} / {r3 / scale:.2e}")
    assert m1 <= 1.25 * mt and r1 <= 1.1 * rt, ((m1, r1), (mt, rt))
    assert m1 / scale < 1e-3


@pytest.mark.parametrize("hd,heads,L,self_attn,fp16_qkv", [(64, 8, 256, False, False), (64, 4, 200, False, False), (72, 4, 320, False, False), (24, 4, 256, True, False),
                                                            (64, 16, 256, True, False), (64, 8, 256, False, True), (72, 8, 512, False, True), (48, 8, 256, True, True),
                                                            (64, 16, 256, True, True), (48, 8, 256, False, True)])
def test_attention_single_product_vs_tf32(hd, heads, L, self_attn, fp16_qkv):
    """the fp16 single-product attention kernel (csrc/xattn_fusion_f16.hip) against float64 SDPA math, next to the emulated-TF32
    evaluation (q, k, v and p rounded to 10 mantissa bits, fp32 products): the deviation is of the same size; the scaled-fp16 image
    output decodes to the fp32 output within fp16's significand; rows of very different magnitude (10^-3 .. 10^3) per token"""
    from dimsum_amd.utils.tf32_emulation import round_tf32
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(7)
    B, C = 2, heads * hd
    mag = torch.logspace(-3, 3, B * L, device="cuda").reshape(B, L, 1)
    xs = [torch.randn(B, L, C, device="cuda", generator=g) * mag for _ in range(2)]
    ws = [torch.randn(3 * C, C, device="cuda", generator=g) * C ** -0.5 for _ in range(2)]
    bs = [0.1 * torch.randn(3 * C, device="cuda", generator=g) for _ in range(2)]
    imgs = [native.rows_f16s(x.reshape(B * L, C)) for x in xs]
    qkvs = [gemm.linear_split3(i, w).view(B, L, 3 * C) for i, w in zip(imgs, ws)]
    if self_attn:
        bound = gemm.attn_kv_bound(ws[0], bs[0])
        args = (qkvs[0], None, heads)
        kw = dict(bias1=bs[0], f16s=(imgs[0].inv.reshape(B, L), None, bound))
    else:
        bound = gemm.attn_kv_bound(ws[0], bs[0], ws[1], bs[1])
        args = (qkvs[0], qkvs[1], heads)
        kw = dict(bias1=bs[0], bias2=bs[1], f16s=(imgs[0].inv.reshape(B, L), imgs[1].inv.reshape(B, L), bound))
    if fp16_qkv:
        # q | k | v as the scaled fp16 the qkv GEMM's F16_QKV epilogue writes (biases included): the attention kernel's fp16-input variant
        q16 = [gemm.qkv_f16s(i, w, b_, L, bound[2 * n:2 * n + 2]) for n, (i, w, b_) in enumerate(zip(imgs, ws, bs)) if n == 0 or not self_attn]
        assert all(q is not None and q.dtype == torch.float16 for q in q16)
        # the epilogue's numbers: fp16((x W^T + b) 2^s) with s from the row's (q) / the batch element's (k, v) bound -- decoded, they are the
        # fp32 GEMM output within fp16's significand, and no element overflows
        for n, q in enumerate(q16):
            full = (qkvs[n].reshape(B * L, 3 * C) + bs[n]).double()
            xinv = imgs[n].inv.double()
            def sc(bnd):                        # the power of two f16s_scales derives from a bound (csrc/common.hpp)
                return torch.exp2(14 - torch.floor(torch.log2(bnd)))
            row = sc(2.0 * (32768.0 * xinv * bound[2 * n].double() + bound[2 * n + 1].double()))
            per_b = sc(2.0 * (32768.0 * xinv.reshape(B, L).amax(1) * bound[2 * n].double() + bound[2 * n + 1].double())).repeat_interleave(L)
            dec = torch.cat([q[:, :C].double() / row[:, None], q[:, C:].double() / per_b[:, None]], 1)
            assert torch.isfinite(q).all() and q.float().abs().max().item() < 2.0 ** 15
            big = full.abs() > full.abs().amax(-1, keepdim=True) * 2.0 ** -12
            assert ((dec - full).abs()[big] <= full.abs()[big] * 2.0 ** -10).all()
        args = (q16[0].view(B, L, 3 * C), None if self_attn else q16[1].view(B, L, 3 * C), heads)
        kw = dict(f16s=kw["f16s"])
    out = native.xattn_fusion_fwd(*args, **kw)
    img = native.xattn_fusion_fwd(*args, split3="f16s", **kw)

    def sdpa(q, k, v, rnd):
        q, k, v = (t.reshape(B, L, heads, hd).transpose(1, 2) for t in (q, k, v))
        if rnd:
            s = (round_tf32(q.float()).double() @ round_tf32(k.float()).double().transpose(-1, -2)) * hd ** -0.5
            pm = torch.softmax(s, -1)
            o = round_tf32(pm.float()).double() @ round_tf32(v.float()).double()
        else:
            o = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, -1) @ v
        return o.transpose(1, 2).reshape(B, L, C)

    def ref(rnd):
        (q1, k1, v1), (q2, k2, v2) = ((qk.double() + b.double()).split(C, -1) for qk, b in zip(qkvs, bs))
        if self_attn:
            return sdpa(q1, k1, v1, rnd)
        return torch.cat([sdpa(q1, k2, v2, rnd), sdpa(q2, k1, v1, rnd)], -1)
    r64, rtf = ref(False), ref(True)
    row = r64.abs().amax(-1, keepdim=True)
    e, etf = (out.double() - r64).abs() / row, (rtf - r64).abs() / row
    assert torch.isfinite(out).all()
    print(f"attention hd {hd} heads {heads} L {L} self {self_attn} fp16_qkv {fp16_qkv}: max / rms error over the row maximum {e.max().item():.2e} / "
          f"{e.pow(2).mean().sqrt().item():.2e}, emulated TF32 {etf.max().item():.2e} / {etf.pow(2).mean().sqrt().item():.2e}")
    assert e.max().item() <= 1.5 * etf.max().item() + 1e-6, (e.max().item(), etf.max().item())
    assert e.pow(2).mean().sqrt().item() <= 1.25 * etf.pow(2).mean().sqrt().item() + 1e-7, (e.pow(2).mean().sqrt().item(), etf.pow(2).mean().sqrt().item())
    assert isinstance(img, native.F16Image) and torch.isfinite(img.data).all() and img.data.float().abs().max().item() < 2.0 ** 15
    assert ((img.float().double() - out.double()).abs() / row).max().item() <= 2.0 ** -10
    # the split-bf16 kernel on the same inputs agrees to the TF32 class
    three = native.xattn_fusion_fwd(qkvs[0], None if self_attn else qkvs[1], heads, bias1=bs[0], bias2=None if self_attn else bs[1], split_bf16=True)
    assert ((out - three).abs() / row.float()).max().item() <= 3 * etf.max().item() + 1e-6


def test_multi_job_conversion_is_the_single_conversion_bit_for_bit():
    """dimsum_rows_f16s_multi (one launch for many weights: gemm.forward_scope) against dimsum_rows_f16s job by job: images, inverse
    scales, L1 bounds (with their factor) and the reduce-only jobs of bias vectors; more jobs than one launch holds (24)"""
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(3)
    shapes = [(2048, 512), (1536, 512), (1024, 1024), (8192, 1024), (1024, 4096), (48, 1024), (7, 260), (300, 72)] * 4      # 32 jobs
    ws = [torch.randn(r, c, device="cuda", generator=g) * (10.0 ** (i % 5 - 2)) for i, (r, c) in enumerate(shapes)]
    buf = torch.randn(64, 520, device="cuda", generator=g)
    ws.append(buf[:, 4:516])                                            # a strided view (16-byte aligned rows)
    bias = torch.randn(8192, device="cuda", generator=g)
    jobs = [(w, True, 2 * i, 2 * i + 1, 1.0 + 2.0 ** -10) for i, w in enumerate(ws)] + [(bias, False, None, 2 * len(ws), 1.0)]
    images, scal = native.rows_f16s_multi(jobs)
    for i, w in enumerate(ws):
        ref, l1 = native.rows_f16s(w, want_l1=True)
        assert torch.equal(images[i].data.view(torch.int16), ref.data.view(torch.int16)) and torch.equal(images[i].inv, ref.inv), i
        assert scal[2 * i].item() == (l1 * (1.0 + 2.0 ** -10)).item() and scal[2 * i + 1].item() == w.abs().max().item(), i
    assert images[-1] is None and scal[2 * len(ws)].item() == bias.abs().max().item()


def test_forward_scope_changes_launch_counts_not_bits(monkeypatch):
    """DiM under the headline policy: with the per-forward weight scope (one multi-job launch for all images and bounds) the output is
    bit-identical to per-call conversions, no single conversion is left, and the images are rebuilt on the next forward (a `.data` update
    in between is seen)"""
    from dimsum_amd import gemm, native
    from dimsum_amd.create_model import create_model, published_config
    from dimsum_amd.utils import rerandomize_zeros
    torch.manual_seed(0)
    m = create_model(published_config(model="DiM-L/2", image_size=256))
    rerandomize_zeros(m, std=0.02, seed=0)
    m = m.cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(0)
    x, t = torch.randn(32, 4, 32, 32, device="cuda", generator=gen), torch.rand(32, device="cuda", generator=gen)
    y = torch.randint(0, 1000, (32,), device="cuda", generator=gen)
    singles, multis = [], []
    real1, realm = native.rows_f16s, native.rows_f16s_multi
    monkeypatch.setattr(native, "rows_f16s", lambda *a, **k: (singles.append(1), real1(*a, **k))[1])
    monkeypatch.setattr(native, "rows_f16s_multi", lambda jobs, **k: (multis.append(len(jobs)), realm(jobs, **k))[1])
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    gemm.set_policy("f16s")
    try:
        with torch.no_grad():
            monkeypatch.setenv("DIMSUM_FORWARD_SCOPE", "0")
            ref = m(x, t, y)
            n_single = len(singles)
            assert n_single > 100 and not multis
            monkeypatch.delenv("DIMSUM_FORWARD_SCOPE")
            got = m(x, t, y)
            assert len(singles) == n_single and len(multis) == 1 and multis[0] >= 7 * 16, (len(singles) - n_single, multis)
            assert torch.equal(got, ref)
            w = m.blocks[3].mlp.w3.weight
            w.data.mul_(1.5)                                   # (no version bump: exactly what a cached image would miss)
            changed = m(x, t, y)
            assert len(multis) == 2 and not torch.equal(changed, ref)
            w.data.div_(1.5)
            assert torch.equal(m(x, t, y), ref)
            # fallbacks (what the multi-job kernel does not take drops to the lazy per-weight path, never an error): a swapped Linear is seen (the plan
            # is keyed by the parameters' identity); a bias-free trailing gated entry still finds its slots
            old_lin = m.blocks[2].mlp.w3
            new_lin = torch.nn.Linear(old_lin.in_features, old_lin.out_features, bias=True).cuda()
            with torch.no_grad():
                new_lin.weight.copy_(old_lin.weight * 0.5)
                new_lin.bias.copy_(old_lin.bias)
            m.blocks[2].mlp.w3 = new_lin
            swapped = m(x, t, y)
            assert not torch.equal(swapped, ref)
            m.blocks[2].mlp.w3 = old_lin
            assert torch.equal(m(x, t, y), ref)
            last = m.blocks[-1].mlp.w12
            keep = last.bias
            last.bias = None
            no_bias = m(x, t, y)                                # (a gated entry without its bias job: the bound tensor must still be two slots long)
            monkeypatch.setenv("DIMSUM_FORWARD_SCOPE", "0")
            assert torch.equal(no_bias, m(x, t, y))
            monkeypatch.delenv("DIMSUM_FORWARD_SCOPE")
            last.bias = keep
    finally:
        gemm.set_policy("default")


@pytest.mark.parametrize("B,L,d_model,d_inner", [(4, 256, 256, 512), (2, 1024, 512, 1024)])
def test_mamba_inner_out_proj_as_one_fp16_product_vs_tf32(B, L, d_model, d_inner, monkeypatch):
    """MambaInnerFn's inference forward under the scaled-fp16 policy with out_proj on the scan's block-scaled fp16 out_z (gemm.out_proj_f16:
    selective_scan_fwd(out_z_f16) + gemm_tn(scales = (block table, ..))) against the same call with out_proj on the library's fp32 GEMM
    (DIMSUM_OUT_PROJ_F16=0) and against the float64 product of the fp32 kernel's out_z: no further from it than 1.1 x what rounding both
    operands to TF32 costs (the reference's arithmetic for this Linear, selective_scan_interface.py:954-981 under train.py:20-21)"""
    from dimsum_amd import gemm, native
    from dimsum_amd.ops import selective_scan_interface as ssi
    g = torch.Generator().manual_seed(L + d_inner)
    N, R = 16, 32
    dev = "cuda"
    r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(dev)
    xz = r(B, 2 * d_inner, L)
    xz[:, d_inner:, : L // 2] *= 30.0                                  # token groups of very different magnitude
    conv_w, conv_b = r(d_inner, 1, 4, k=0.5), r(d_inner, k=0.1)
    x_w, dt_w, out_w = r(R + 2 * N, d_inner, k=d_inner ** -0.5), r(d_inner, R, k=R ** -0.5), r(d_model, d_inner, k=d_inner ** -0.5)
    A = (-0.5 * torch.rand(d_inner, N, generator=g) - 0.05).to(dev)
    Dv, dt_b = r(d_inner), (0.5 * torch.rand(d_inner, generator=g)).to(dev)
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "256")
    old_tf32, old_policy, old_var = torch.backends.cuda.matmul.allow_tf32, gemm.get_policy(), native._scan_fwd_variant
    torch.backends.cuda.matmul.allow_tf32 = True
    gemm.set_policy("f16s")
    native._scan_fwd_variant = 1
    try:
        with torch.no_grad():
            run = lambda: ssi.mamba_inner_fn(xz, conv_w, conv_b, x_w, dt_w, out_w, None, A, None, None, Dv, dt_b, delta_softplus=True)
            calls = []
            real = native.gemm_tn
            monkeypatch.setattr(native, "gemm_tn", lambda *a, **k: (calls.append(k), real(*a, **k))[1])
            got = run()
            assert len(calls) == 1 and calls[0]["scales"][0].dim() == 2, "out_proj did not take the fp16 path"
            monkeypatch.setenv("DIMSUM_OUT_PROJ_F16", "0")
            lib = run()
            assert len(calls) == 1
            out_z = ssi.mamba_inner_fn_no_out_proj(xz, conv_w, conv_b, x_w, dt_w, A, None, None, Dv, dt_b, delta_softplus=True)     # (B, d_inner, L) fp32
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old_tf32
        gemm.set_policy(old_policy)
        native._scan_fwd_variant = old_var
    a = out_z.transpose(1, 2).reshape(B * L, d_inner)
    exact = (a.double() @ out_w.double().t()).view(B, L, d_model)
    tf = lambda t: ((t.view(torch.int32) + 0x1000) & ~0x1FFF).view(torch.float32)
    tf32 = (tf(a.contiguous()).double() @ tf(out_w).double().t()).view(B, L, d_model)
    e16, e32 = (got.double() - exact), (tf32 - exact)
    assert e16.abs().max() <= 1.1 * e32.abs().max() and e16.pow(2).mean().sqrt() <= 1.1 * e32.pow(2).mean().sqrt(), \
        (e16.abs().max().item(), e32.abs().max().item(), e16.pow(2).mean().sqrt().item(), e32.pow(2).mean().sqrt().item())
    assert (lib.double() - exact).abs().max() <= 1.1 * e32.abs().max()


@pytest.mark.parametrize("D,M,N", [(1152, 2048, 576), (128, 256, 132), (1024, 4096, 512)])
def test_block_scaled_image_of_a_d_major_matrix_and_ragged_tn(D, M, N):
    """dimsum_rows_block_f16s: the conversion pass that gives launches of the state-split scan kernels the block-scaled fp16 out_z the 64-channel kernel
    writes itself (64 channels x 32 tokens per scale): decoded, it is the input to half an fp16 ulp of each block's maximum, scales are the block
    maxima's powers of two, an all-zero block decodes to zeros; and out_proj over it -- dimsum_gemm_tn with the table and a right operand whose
    columns (d_model = 576 at DiM-XL/2, not a multiple of the kernel's 256-column tiles) are zero-padded behind the slice -- against float64, within
    the emulated-TF32 product's error"""
    from dimsum_amd import gemm, native
    from dimsum_amd.utils.tf32_emulation import round_tf32
    g = torch.Generator(device="cuda").manual_seed(D + N)
    x = torch.randn(D, M, device="cuda", generator=g) * torch.exp2(8 * torch.rand(1, M, device="cuda", generator=g) - 4)      # token magnitudes over 2^-4 .. 2^4
    x[:64, :32] = 0
    x[64:128, 32:64] *= 1e-9
    img, tab = native.rows_block_f16s(x)
    assert img.shape == (D, M) and img.dtype == torch.float16 and tab.shape == (M // 32, D // 64)
    blocks = lambda t: t.reshape(D // 64, 64, M // 32, 32).permute(2, 0, 1, 3)
    dec = blocks(img.float()) * tab[:, :, None, None]
    want = blocks(x)
    bmax = want.abs().amax((2, 3))
    assert torch.all((dec - want).abs().amax((2, 3)) <= 2.0 ** -11 * bmax * 1.001 + 1e-37)
    assert torch.all(torch.frexp(tab)[0] == 0.5) and torch.all(dec[0, 0] == 0)
    top = blocks(img.float()).abs().amax((2, 3))
    assert torch.all(top[bmax > 0] >= 2.0 ** 14) and torch.all(top <= 2.0 ** 15)          # (a maximum just below 2^15 may round up to it)
    w = torch.randn(N, D, device="cuda", generator=g) * D ** -0.5
    if M % 256 == 0 and D >= 128:
        got = gemm.out_proj_f16(img, tab, w)                                  # (M, N)
        ref = x.double().t() @ w.double().t()
        tf = round_tf32(x).double().t() @ round_tf32(w).double().t()
        e, et = (got.double() - ref).abs(), (tf - ref).abs()
        s_ = ref.abs().max().item()
        print(f"out_proj over the converted image ({D} x {M} x {N}): f16s {e.max().item() / s_:.2e} / {e.pow(2).mean().sqrt().item() / s_:.2e}, "
              f"TF32 operands {et.max().item() / s_:.2e} / {et.pow(2).mean().sqrt().item() / s_:.2e}")
        assert got.shape == (M, N) and e.max().item() <= 1.1 * et.max().item() + 2.0 ** -22 * s_ and e.pow(2).mean().sqrt().item() <= 1.1 * et.pow(2).mean().sqrt().item() + 2.0 ** -24 * s_
