"""Split-bf16 operand images (csrc/operand_split.hip, gemm.py "split3"): the producer kernels write the left operand of a Linear
as [hi | hi | lo] bfloat16 rows and ONE plain bf16 library GEMM against the weight image [hi | lo | hi] forms the three
products hipBLASLt's fp32-under-allow_tf32 path forms. Checked: the images bit for bit against torch's own bf16 rounding, the
fused producers against the stand-alone converter, the product against float64, and the model with and without the carrier."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(x):
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return hi, lo


@pytest.mark.parametrize("left", [True, False])
def test_split3_rows_bit_exact(left):
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(37, 72, device="cuda", generator=g) * torch.logspace(-6, 6, 72, device="cuda")
    buf = torch.zeros(37, 80, device="cuda")
    buf[:, :72] = x                                   # a row stride that is not the width
    for src in (x, buf[:, :72]):
        hi, lo = _images(src)
        want = torch.cat([hi, hi, lo] if left else [hi, lo, hi], 1)
        got = native.split3_rows(src, left=left)
        assert got.dtype == torch.bfloat16 and tuple(got.shape) == (37, 216)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    # hi + lo reproduces x to 2^-16 relative
    hi, lo = _images(x)
    assert ((hi.float() + lo.float() - x).abs() <= x.abs() * 2.0 ** -16 + 1e-37).all()


def test_split3_rows_rejects_bad_layouts():
    from dimsum_amd import native
    with pytest.raises(RuntimeError):
        native.split3_rows(torch.zeros(4, 6, device="cuda"), left=True)             # K % 4
    with pytest.raises(RuntimeError):
        native.split3_rows(torch.zeros(4, 8, device="cuda", dtype=torch.float16), left=True)
    assert tuple(native.split3_rows(torch.zeros(0, 8, device="cuda"), left=True).shape) == (0, 24)


def test_norm_split3_output_matches_converter():
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(1)
    M, N, L = 96, 384, 16
    x, res = torch.randn(M, N, device="cuda", generator=g), torch.randn(M, N, device="cuda", generator=g)
    w, xb = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    sc, sh = 0.1 * torch.randn(M // L, N, device="cuda", generator=g), torch.randn(M // L, N, device="cuda", generator=g)
    kw = dict(residual=res, is_rms_norm=True, x_bias=xb, mod_scale=sc, mod_shift=sh, rows_per_batch=L)
    y, _, rstd, r = native.layer_norm_fwd(x, w, None, 1e-5, **kw)
    y3, _, rstd3, r3 = native.layer_norm_fwd(x, w, None, 1e-5, split3=True, **kw)
    assert y3.dtype == torch.bfloat16 and tuple(y3.shape) == (M, 3 * N)
    assert torch.equal(y3.view(torch.int16), native.split3_rows(y, left=True).view(torch.int16))
    assert torch.equal(rstd, rstd3) and torch.equal(r, r3)
    # LayerNorm with bias, no modulation
    b = torch.randn(N, device="cuda", generator=g)
    y, *_ = native.layer_norm_fwd(x, w, b, 1e-5)
    y3, *_ = native.layer_norm_fwd(x, w, b, 1e-5, split3=True)
    assert torch.equal(y3.view(torch.int16), native.split3_rows(y, left=True).view(torch.int16))


def test_gated_gelu_split3_output_is_the_image_of_the_fp32_output():
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(2)
    x12, bias = 2 * torch.randn(3, 50, 2 * 136, device="cuda", generator=g), torch.randn(2 * 136, device="cuda", generator=g)
    for b in (bias, None):
        h = native.gated_gelu_fwd(x12, b)
        h3 = native.gated_gelu_fwd(x12, b, split3=True)
        assert tuple(h3.shape) == (3, 50, 3 * 136) and h3.dtype == torch.bfloat16
        # (the two instantiations of the kernel may contract gelu(a) * g differently: the image is checked against h to one fp32
        # ulp + the 2^-16 of the split, and its two hi copies against each other bit for bit)
        hi, hi2, lo = h3[..., :136], h3[..., 136:272], h3[..., 272:]
        assert torch.equal(hi.contiguous().view(torch.int16), hi2.contiguous().view(torch.int16))
        assert ((hi.float() + lo.float() - h).abs() <= h.abs() * (2.0 ** -16 + 2.0 ** -22) + 1e-30).all()
        assert ((hi.float() - h).abs() <= h.abs() * 2.0 ** -8).all()


@pytest.mark.parametrize("kind", ["none", "haar", "dct"])
def test_token_transform_split3_output_matches_converter(kind):
    """pre- and post-mixer passes (modulate / gate + residual, token gather, 4x4 transforms) writing the operand image directly"""
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(3)
    B, L, C = 3, 64, 40
    wide = torch.randn(B, L, 2 * C, device="cuda", generator=g)
    x, res = wide[:, :, :C], wide[:, :, C:]                     # channel slices of a wider tensor, like x1 / x2 of a combined block
    sc, sh = 0.1 * torch.randn(B, C, device="cuda", generator=g), torch.randn(B, C, device="cuda", generator=g)
    perm = torch.randperm(L, device="cuda", generator=g).to(torch.int32)
    kw_pre = dict(out_index=perm, scale=sc, shift=sh)
    kw_post = dict(in_index=perm, gate=sc, residual=res)
    for fwd, kw in ((True, kw_pre), (False, kw_post)):
        y = native.token_transform(x, kind, fwd, **kw)
        y3 = native.token_transform(x, kind, fwd, split3=True, **kw)
        assert y3.dtype == torch.bfloat16 and tuple(y3.shape) == (B, L, 3 * C)
        assert torch.equal(y3.reshape(B * L, -1).view(torch.int16), native.split3_rows(y.reshape(B * L, C), left=True).view(torch.int16))


@pytest.mark.parametrize("hd,self_attn", [(24, False), (64, False), (72, False), (64, True)])
def test_xattn_split3_output_matches_converter(hd, self_attn):
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(hd)
    B, L, H = 2, 200, 4
    W = 3 * H * hd
    q1 = torch.randn(B, L, W, device="cuda", generator=g)
    q2 = None if self_attn else torch.randn(B, L, W, device="cuda", generator=g)
    b1 = torch.randn(W, device="cuda", generator=g)
    b2 = None if self_attn else torch.randn(W, device="cuda", generator=g)
    o = native.xattn_fusion_fwd(q1, q2, H, bias1=b1, bias2=b2, split_bf16=True)
    o3 = native.xattn_fusion_fwd(q1, q2, H, bias1=b1, bias2=b2, split_bf16=True, split3=True)
    width = o.shape[-1]
    assert o3.dtype == torch.bfloat16 and tuple(o3.shape) == (B, L, 3 * width)
    assert torch.equal(o3.reshape(B * L, -1).view(torch.int16), native.split3_rows(o.reshape(B * L, width), left=True).view(torch.int16))
    with pytest.raises(RuntimeError):
        native.xattn_fusion_fwd(q1, q2, H, bias1=b1, bias2=b2, split_bf16=False, split3=True)      # exact-fp32 kernel has no image output


@pytest.mark.parametrize("M,K,N", [(512, 1024, 768), (4096, 384, 1536), (100, 72, 40)])
def test_linear_split3_is_an_fp32_class_product(M, K, N):
    """max error of the 3-product GEMM against float64: 2e-5 of max|y| (the dropped lo.lo term and the bf16 rounding of lo are
    ~2^-16 relative per product; plain bf16 operands would be ~4e-3) -- and no worse than 1.5x the library's own fp32 path + 1e-6"""
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x, w = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    ref = x.double() @ w.double().t()
    y = gemm.linear_split3(native.split3_rows(x, left=True), w)
    assert y.dtype == torch.float32 and tuple(y.shape) == (M, N)
    scale = ref.abs().max().item()
    err = (y - ref).abs().max().item() / scale
    assert err < 2e-5, err
    old = torch.backends.cuda.matmul.allow_tf32
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        err_lib = (torch.nn.functional.linear(x, w) - ref).abs().max().item() / scale
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
    assert err <= 1.5 * err_lib + 1e-6, (err, err_lib)


def test_block_with_and_without_split3_carrier(monkeypatch):
    """DiMBlockCombined(384) inference forward under allow_tf32 against the reference block golden, with the split3 carrier
    (norm -> w12 -> gated GeLU -> w3 on operand images) and with fp32 operands: the same tolerance as the training-mode golden
    test; and the two against each other (both are 3-product results: 2e-5 of max|y|)."""
    from conftest import assert_close, golden
    from procedural import procedural_fill, seeded
    T = torch.from_numpy
    from dimsum_amd.models_dim import create_block
    g = golden("block_combined_384")
    blk = create_block(384, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=True, transpose=True, cond_mamba=True,
                       scanning_continuity=True, use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    blk = blk.cuda()
    x, res, cc = (T(seeded(sh, sd)).cuda() for sh, sd in (((1, 256, 384), 56), ((1, 256, 384), 57), ((1, 384), 58)))
    old = torch.backends.cuda.matmul.allow_tf32
    outs = {}
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")      # 256 rows here: below the default threshold of the carrier
        for flag in ("1", "0"):
            monkeypatch.setenv("DIMSUM_SPLIT3", flag)
            with torch.no_grad():
                outs[flag] = blk(x, res, cc)[0]
            assert_close(outs[flag].cpu().numpy(), g["y"], what=f"y (DIMSUM_SPLIT3={flag})", rtol=2e-4, atol=0.0, scale_atol=2e-5)
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
    assert not torch.equal(outs["1"], outs["0"])          # the carrier really ran (another summation order)
    err = (outs["1"] - outs["0"]).abs().max().item() / outs["0"].abs().max().item()
    assert err < 2e-5, err


def test_gated_gelu_bwd_split3_output_matches_converter():
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(5)
    x12, bias = 2 * torch.randn(150, 2 * 136, device="cuda", generator=g), torch.randn(2 * 136, device="cuda", generator=g)
    dh = torch.randn(150, 136, device="cuda", generator=g)
    for b in (bias, None):
        dx, db = native.gated_gelu_bwd(x12, b, dh)
        dx3, db3 = native.gated_gelu_bwd(x12, b, dh, split3=True)
        assert dx3.dtype == torch.bfloat16 and tuple(dx3.shape) == (150, 6 * 136)
        assert torch.equal(dx3.view(torch.int16), native.split3_rows(dx, left=False).view(torch.int16))      # weight order [hi | lo | hi]
        if b is not None:
            assert torch.allclose(db, db3, rtol=1e-5, atol=1e-5)                                            # atomics: order differs


def test_training_mlp_on_operand_images_matches_fp32_operands(monkeypatch):
    """_ModGatedMlpImagesFn (forward + backward GEMMs on images) against the fp32-operand path of the same policy: the module
    output and the gradients of the input, the modulation and all four parameters: 5e-5 of each tensor's max (both 3-product)"""
    from dimsum_amd import models_dim
    from dimsum_amd.mlp import GatedMLP
    from dimsum_amd.ops import token_ops
    torch.manual_seed(1)
    H, B, L = 256, 4, 96
    mlp = GatedMLP(in_features=H, hidden_features=4 * H, act_layer=models_dim._approx_gelu, drop=0).cuda()
    for prm in mlp.parameters():
        torch.nn.init.normal_(prm, std=0.05)
    x = torch.randn(B, L, H, device="cuda")
    normed0, shift0, scale0, gate0 = torch.randn(B, L, H, device="cuda"), torch.randn(B, H, device="cuda"), 0.1 * torch.randn(B, H, device="cuda"), torch.randn(B, H, device="cuda")
    dout = torch.randn(B, L, H, device="cuda")
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DIMSUM_SPLIT3_TRAIN", flag)
        leaves = [t.clone().requires_grad_() for t in (normed0, shift0, scale0, gate0)]
        mlp.zero_grad()
        out = models_dim._mlp_tail(mlp, x, *leaves[:3], leaves[3])
        out.backward(dout)
        res[flag] = [out.detach()] + [t.grad for t in leaves] + [p.grad.clone() for p in mlp.parameters()]
    assert not torch.equal(res["1"][0], res["0"][0])              # the image path really ran
    names = ["out", "d normed", "d shift", "d scale", "d gate"] + ["d " + n for n, _ in mlp.named_parameters()]
    for n, a, b in zip(names, res["1"], res["0"]):
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err < 5e-5, (n, err)


@pytest.mark.parametrize("train_images", ["1", "0"])
def test_block_training_golden_with_operand_images(monkeypatch, train_images):
    """the reference block golden (forward, input and parameter gradients) under allow_tf32 with the training MLP on operand
    images (and without): same tolerances as tests/test_model_gpu.py::test_block_combined_384_fwd_bwd_all_hip"""
    from test_model_cpu import check_block_384
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    monkeypatch.setenv("DIMSUM_SPLIT3_TRAIN", train_images)
    check_block_384("cuda", dict(rtol=2e-4, atol=0.0, scale_atol=2e-5), dict(rtol=5e-4, atol=0.0, scale_atol=5e-5))


def test_linear_on_images_under_autograd(monkeypatch):
    """gemm.linear under autograd (training policy): y, dx, dW on operand images against float64"""
    from dimsum_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(7)
    M, K, N = 1024, 384, 640
    x0, w0 = torch.randn(4, M // 4, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    dy = torch.randn(4, M // 4, N, device="cuda", generator=g)
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    x, w = x0.clone().requires_grad_(), w0.clone().requires_grad_()
    y = gemm.linear(x, w)
    assert type(y.grad_fn).__name__ == "_LinearImagesFnBackward"
    y.backward(dy)
    xd, wd = x0.double().requires_grad_(), w0.double().requires_grad_()
    yd = torch.nn.functional.linear(xd, wd)
    yd.backward(dy.double())
    for name, a, b in (("y", y, yd), ("dx", x.grad, xd.grad), ("dW", w.grad, wd.grad)):
        err = (a.double() - b).abs().max().item() / b.abs().max().item()
        assert err < 2e-5, (name, err)
    monkeypatch.setenv("DIMSUM_SPLIT3_TRAIN", "0")
    assert type(gemm.linear(x, w).grad_fn).__name__ != "_LinearImagesFnBackward"


def test_sliced_weight_gradient_products():
    """gemm.mm_tn / mm_nn_rows (a long reduction run as a batched GEMM over row slices + a sum when the output is small) and the
    in_proj autograd function built on them, against float64"""
    from dimsum_amd import gemm
    g = torch.Generator(device="cuda").manual_seed(11)
    R, N, K = 32768, 384, 96
    a, b = torch.randn(R, N, device="cuda", generator=g), torch.randn(R, K, device="cuda", generator=g)
    assert gemm._slices(R, N, K) > 1 and gemm._slices(2048, N, K) == 1 and gemm._slices(R, 8192, 4096) == 1
    ref = a.double().t() @ b.double()
    for got in (gemm.mm_tn(a, b), gemm.mm_nn_rows(a.t().contiguous(), b)):
        assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-6
    a16, b16 = a.bfloat16(), b.bfloat16()
    got = gemm.mm_tn(a16, b16, out_dtype=torch.float32)
    ref16 = a16.double().t() @ b16.double()
    assert got.dtype == torch.float32 and (got.double() - ref16).abs().max().item() / ref16.abs().max().item() < 2e-6
    # in_proj: W @ x^T with the weight gradient through the sliced product
    w0, x0 = torch.randn(N, K, device="cuda", generator=g), torch.randn(R, K, device="cuda", generator=g)
    dy = torch.randn(N, R, device="cuda", generator=g)
    w, x = w0.clone().requires_grad_(), x0.clone().requires_grad_()
    y = gemm.matmul_wx(w, x.t())
    assert type(y.grad_fn).__name__ == "_MatmulWxFnBackward"
    y.backward(dy)
    wd, xd = w0.double().requires_grad_(), x0.double().requires_grad_()
    (wd @ xd.t()).backward(dy.double())
    for name, p, q in (("y", y, wd @ xd.t()), ("dW", w.grad, wd.grad), ("dx", x.grad, xd.grad)):
        assert (p.double() - q).abs().max().item() / q.abs().max().item() < 2e-6, name


def test_block_training_at_a_size_where_everything_is_active(monkeypatch):
    """DiMBlockCombined(384) forward+backward on 32 x 256 = 8192 rows (the default threshold of the operand images; the weight
    gradients run as sliced reductions): output, input gradients and every parameter gradient against the same block with fp32
    operands and unsliced products -- 1e-4 of each tensor's max (summation order and 3-product rounding only)"""
    from procedural import procedural_fill
    from dimsum_amd import gemm
    from dimsum_amd.models_dim import create_block
    blk = create_block(384, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=True, transpose=True, cond_mamba=True,
                       scanning_continuity=True, use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    blk = blk.cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x0, r0, c0 = (torch.randn(*sh, device="cuda", generator=g) for sh in ((32, 256, 384), (32, 256, 384), (32, 384)))
    dy, dr = torch.randn(32, 256, 384, device="cuda", generator=g), torch.randn(32, 256, 384, device="cuda", generator=g)
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    res = {}
    for mode in ("new", "old"):
        if mode == "old":
            monkeypatch.setenv("DIMSUM_SPLIT3", "0")
            monkeypatch.setattr(gemm, "_slices", lambda rows, n, k: 1)
        x, r, c = (t.clone().requires_grad_() for t in (x0, r0, c0))
        blk.zero_grad()
        y, ro = blk(x, r, c)
        ((y * dy).sum() + (ro * dr).sum()).backward()
        res[mode] = [y.detach(), x.grad, r.grad, c.grad] + [p.grad.clone() for p in blk.parameters() if p.grad is not None]
    names = ["y", "dx", "dres", "dc"] + ["d " + n for n, p in blk.named_parameters() if p.grad is not None]     # (cond_proj is a graph edge only)
    assert len(res["new"]) == len(res["old"]) == len(names) > 30
    assert not torch.equal(res["new"][0], res["old"][0])
    for n, a, b in zip(names, res["new"], res["old"]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-4 * scale + 1e-12, (n, (a - b).abs().max().item(), scale)


def test_frozen_weights_scope_caches_weight_images(monkeypatch):
    """gemm.frozen_weights(): inside the scope a weight's split image is built once (same object on the second call, bit-identical
    product); outside every call converts afresh, so an in-place update through .data is seen; the cache dies with the scope."""
    from dimsum_amd import gemm, native
    g = torch.Generator(device="cuda").manual_seed(11)
    w = torch.randn(256, 128, device="cuda", generator=g)
    x = torch.randn(512, 128, device="cuda", generator=g)
    x3 = native.split3_rows(x, left=True)
    ref = gemm.linear_split3(x3, w)
    assert gemm.weight_image(w) is not gemm.weight_image(w)                   # no scope: nothing is kept
    with gemm.frozen_weights():
        a = gemm.weight_image(w)
        assert gemm.weight_image(w) is a
        assert torch.equal(gemm.linear_split3(x3, w), ref)
        with gemm.frozen_weights():                                            # nested scopes share the outer cache
            assert gemm.weight_image(w) is a
        assert gemm.weight_image(w) is a
    w.data.mul_(2.0)                                                           # an update the version counter does not see
    assert torch.equal(gemm.linear_split3(x3, w), 2.0 * ref)                   # ... is seen outside the scope
    assert gemm._tls.frozen is None


def test_sample_batch_builds_each_weight_image_once(monkeypatch):
    """sample_batch opens the frozen-weights scope around its NFE loop: the converter runs once per weight and batch, not once
    per evaluation, and the samples are bit-identical to the uncached run"""
    from procedural import procedural_fill
    from dimsum_amd import gemm, native
    from dimsum_amd.models_dim import DiM
    from dimsum_amd.sample_ddp import sample_batch
    kw = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, scan_type="none", pe_type="ape",
              block_type="combined", cond_mamba=True, rms_norm=True, fused_add_norm=True, learnable_pe=True, use_attn_every_k_layers=4)
    m = DiM(depth=2, hidden_size=384, patch_size=2, **kw).eval()
    procedural_fill(m, seed=3)
    m = m.cuda()
    gen = torch.Generator(device="cuda").manual_seed(0)
    z, y = torch.randn(4, 4, 32, 32, device="cuda", generator=gen), torch.tensor([1, 2, 3, 4], device="cuda")
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    old = torch.backends.cuda.matmul.allow_tf32
    calls = {"w": 0}
    real = native.split3_rows

    def counting(t, left=True):
        calls["w"] += (not left)
        return real(t, left=left)

    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        monkeypatch.setattr(native, "split3_rows", counting)
        a = sample_batch(m, z, y, num_steps=3, gather=False)
        per_batch = calls["w"]
        calls["w"] = 0

        @__import__("contextlib").contextmanager
        def no_scope():
            yield
        monkeypatch.setattr(gemm, "frozen_weights", no_scope)
        b = sample_batch(m, z, y, num_steps=3, gather=False)
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
    assert per_batch > 0 and calls["w"] == 3 * per_batch, (per_batch, calls["w"])
    assert torch.equal(a, b)


def test_mamba_out_proj_from_the_scan_planes_matches_the_fp32_operand_path(monkeypatch):
    """Mamba inference under allow_tf32 with out_proj on the scan's split-bf16 out_z planes (DIMSUM_OUT_PROJ_PLANES=1: the scan epilogue
    writes the operand image, gemm_tn reads it transposed) against the library's fp32 GEMM on the fp32 out_z: both are 3-product
    results (2e-5 of max|y|); the default (auto) takes the planes only where the scan runs a state-split kernel"""
    from dimsum_amd import native
    from dimsum_amd.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(256, d_state=16, expand=2).cuda().eval()
    x = torch.randn(4, 256, 256, device="cuda")
    old = torch.backends.cuda.matmul.allow_tf32
    outs = {}
    calls = []
    real = native.gemm_tn
    monkeypatch.setattr(native, "gemm_tn", lambda *a, **k: (calls.append(k.get("alias_rows", 0)), real(*a, **k))[1])
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
        for flag in ("1", "0", "auto"):
            monkeypatch.setenv("DIMSUM_OUT_PROJ_PLANES", flag)
            n = len(calls)
            with torch.no_grad():
                outs[flag] = m(x)
            took = len(calls) > n
            assert took == (flag == "1" or (flag == "auto" and native.scan_fwd_kernel_for(4, 512, 256, 16) != 1)), (flag, took)
        y = m(x)                                   # under autograd: never the planes
        assert y.requires_grad
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
    assert calls and all(c == 512 for c in calls)
    ref = outs["0"]
    assert not torch.equal(outs["1"], ref)
    assert (outs["1"] - ref).abs().max().item() / ref.abs().max().item() < 2e-5


def test_producers_write_the_pair_image_as_hi_lo_of_their_three_piece_image():
    """split3="pair": every producer of a LEFT operand image (norm pass, token passes with and without a transform, attention fusion) writes
    [hi | lo] -- the first and the last third of the [hi | hi | lo] image it writes otherwise, bit for bit; PairImage.image3() restores it"""
    from dimsum_amd import native
    g = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *s: torch.randn(s, device="cuda", generator=g)

    def check(pair, three, C):
        assert isinstance(pair, native.PairImage) and pair.data.shape[-1] == 2 * C and three.shape[-1] == 3 * C
        assert torch.equal(pair.data[..., :C], three[..., :C]) and torch.equal(pair.data[..., C:], three[..., 2 * C:])
        assert torch.equal(pair.image3(), three)

    M, N, L = 96, 384, 16
    x, res, w, xb = rn(M, N), rn(M, N), torch.rand(N, device="cuda", generator=g) + 0.5, rn(N)
    kw = dict(residual=res, is_rms_norm=True, x_bias=xb, mod_scale=0.1 * rn(M // L, N), mod_shift=rn(M // L, N), rows_per_batch=L)
    check(native.layer_norm_fwd(x, w, None, 1e-5, split3="pair", **kw)[0], native.layer_norm_fwd(x, w, None, 1e-5, split3=True, **kw)[0], N)
    B, L, C = 3, 64, 40
    wide = rn(B, L, 2 * C)
    xt, rest = wide[:, :, :C], wide[:, :, C:]
    sc, sh = 0.1 * rn(B, C), rn(B, C)
    perm = torch.randperm(L, device="cuda", generator=g).to(torch.int32)
    for kind in ("none", "haar"):
        for fwd, kw in ((True, dict(out_index=perm, scale=sc, shift=sh)), (False, dict(in_index=perm, gate=sc, residual=rest))):
            check(native.token_transform(xt, kind, fwd, split3="pair", **kw), native.token_transform(xt, kind, fwd, split3=True, **kw), C)
    for hd, self_attn in ((64, False), (72, False), (24, True)):
        H = 4
        q1 = rn(2, 200, 3 * H * hd)
        q2 = None if self_attn else rn(2, 200, 3 * H * hd)
        b1 = rn(3 * H * hd)
        b2 = None if self_attn else rn(3 * H * hd)
        three = native.xattn_fusion_fwd(q1, q2, H, bias1=b1, bias2=b2, split_bf16=True, split3=True)
        check(native.xattn_fusion_fwd(q1, q2, H, bias1=b1, bias2=b2, split_bf16=True, split3="pair"), three, three.shape[-1] // 3)


def test_out_proj_planes_predicate_never_raises_on_shapes_the_kernels_refuse(monkeypatch):
    """DIMSUM_OUT_PROJ_PLANES=1 on a mixer whose d_inner is 64 (the TN GEMM needs two 64-row reduction tiles per piece: d_inner >= 128):
    the predicate must send it down the fp32 out_z + library GEMM path instead of raising from native.gemm_tn's checks"""
    from dimsum_amd import native
    from dimsum_amd.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    small = Mamba(32, d_state=16, expand=2).cuda().eval()          # d_inner 64; out_proj (32, 64): also no 256-row panel
    x = torch.randn(4, 256, 32, device="cuda")
    calls = []
    real = native.gemm_tn
    monkeypatch.setattr(native, "gemm_tn", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    monkeypatch.setenv("DIMSUM_OUT_PROJ_PLANES", "1")
    with torch.no_grad():
        y1 = small(x)
    assert not calls
    monkeypatch.setenv("DIMSUM_OUT_PROJ_PLANES", "0")
    with torch.no_grad():
        y0 = small(x)
    assert torch.equal(y0, y1)
    from dimsum_amd import gemm
    monkeypatch.setenv("DIMSUM_OUT_PROJ_PLANES", "1")
    w64 = torch.randn(256, 64, device="cuda")                       # the advisor's case proper: out_proj (256, 64) with 8192 rows
    assert not gemm.out_proj_planes_enabled(torch.randn(4, 128, 2048, device="cuda"), w64, 8192, scan_kernel=4)
    w128 = torch.randn(256, 128, device="cuda")
    assert gemm.out_proj_planes_enabled(torch.randn(4, 256, 2048, device="cuda"), w128, 8192, scan_kernel=4)
    assert not gemm.out_proj_planes_enabled(torch.randn(4, 256, 2050, device="cuda")[:, :, 1:2049], w128, 8192, scan_kernel=4)   # misaligned rows
