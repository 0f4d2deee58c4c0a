"""GPU parity: MFMA cross-attention fusion core (through the C ABI) vs the numpy oracle (float64 softmax attention)
and vs the reference CrossAttentionFusion goldens. fp32 MFMA is exact-fp32 arithmetic: rtol 2e-5 + 2e-6 * max|ref|."""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden
from procedural import procedural_fill

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L,heads,hd", [(2, 256, 8, 64), (1, 1024, 8, 72), (3, 256, 8, 24), (2, 256, 8, 48), (2, 100, 4, 32), (1, 64, 8, 64)])
def test_core_vs_oracle(B, L, heads, hd):
    from dimsum_amd import native
    from oracle import np_ops
    gen = torch.Generator().manual_seed(L + hd)
    qkv1, qkv2 = torch.randn(B, L, 3 * heads * hd, generator=gen), torch.randn(B, L, 3 * heads * hd, generator=gen)
    out, lse = native.xattn_fusion_fwd(qkv1.cuda(), qkv2.cuda(), heads, need_lse=True)
    ref = np_ops.xattn_fusion_core(qkv1.numpy(), qkv2.numpy(), heads)
    assert_close(out.cpu().numpy(), ref, 2e-5, 0, "out", scale_atol=2e-6)
    # lse of direction 0, head 0 against a direct computation
    q = qkv1[:, :, :hd].double().numpy()
    k = qkv2[:, :, heads * hd: heads * hd + hd].double().numpy()
    sc = np.einsum("bid,bjd->bij", q, k) * hd ** -0.5
    ref_lse = np.log(np.exp(sc - sc.max(-1, keepdims=True)).sum(-1)) + sc.max(-1)
    assert_close(lse[:, 0, 0].cpu().numpy(), ref_lse, 1e-5, 1e-5, "lse")


@pytest.mark.parametrize("name,dim", [("fusion_128", 128), ("fusion_hd24", 384)])
def test_module_vs_golden(name, dim, monkeypatch):
    """CrossAttentionFusion end to end (qkv GEMMs + MFMA core + proj) against the reference module's output.
    fusion_hd24 (head_dim 24) runs the MFMA kernels; fusion_128 has head_dim 8, for which no kernel is built: it checks the
    module's explicit torch-SDPA opt-in path (DIMSUM_ALLOW_TORCH_SDPA) and that the call is counted."""
    from dimsum_amd.attention_fusion import CrossAttentionFusion
    g = golden(name)
    from dimsum_amd import utils
    if name == "fusion_128":
        monkeypatch.setenv("DIMSUM_ALLOW_TORCH_SDPA", "1")
    m = CrossAttentionFusion(dim, num_heads=8, qkv_bias=True, swap_k=False)
    procedural_fill(m, seed=5)
    m = m.cuda().eval()
    x1, x2 = torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda()
    before = sum(utils.torch_path_counts().values())
    with torch.no_grad():
        y = m(x1, x2)
    assert sum(utils.torch_path_counts().values()) - before == (1 if name == "fusion_128" else 0)
    assert_close(y.cpu().numpy(), g["y"], 1e-4, 0, "y (fused MFMA path)", scale_atol=1e-5)
    # autograd path (MFMA forward + the two MFMA backward kernels) gives the same forward and the reference gradients
    x1r, x2r = x1.clone().requires_grad_(), x2.clone().requires_grad_()
    y2 = m(x1r, x2r)
    y2.backward(torch.from_numpy(g["dy"]).cuda())
    assert_close(y2.detach().cpu().numpy(), g["y"], 1e-4, 0, "y (autograd path)", scale_atol=1e-5)
    assert_close(x1r.grad.cpu().numpy(), g["dx1"], 5e-4, 0, "dx1", scale_atol=5e-5)
    assert_close(x2r.grad.cpu().numpy(), g["dx2"], 5e-4, 0, "dx2", scale_atol=5e-5)


def test_module_on_scaled_fp16_images_vs_reference_golden(monkeypatch):
    """CrossAttentionFusion the way a block calls it under the headline policy "f16s" (forward_deferred(images=True): scaled-fp16 images of
    x1 / x2 in, qkv GEMMs + the single-product fp16 attention kernel + proj, all with ONE fp16 MFMA product per element) against the
    REFERENCE module's output (golden fusion_hd24, head_dim 24). Tolerance: 1e-3 of max|ref| (north star); the emulated-TF32 evaluation
    of the same module (operands of every matmul and of QK^T / PV rounded to 10 mantissa bits) is measured next to it and must not be
    beaten by more than 1.25 x."""
    from dimsum_amd import gemm, native
    from dimsum_amd.attention_fusion import CrossAttentionFusion
    from dimsum_amd.utils.tf32_emulation import round_tf32
    g = golden("fusion_hd24")
    m = CrossAttentionFusion(384, num_heads=8, qkv_bias=True, swap_k=False)
    procedural_fill(m, seed=5)
    m = m.cuda().eval()
    x1, x2 = torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda()
    B, N, C = x1.shape
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    gemm.set_policy("f16s")
    try:
        with torch.no_grad():
            i1, i2 = (native.rows_f16s(x.reshape(B * N, C)).reshape(B, N, C) for x in (x1, x2))
            assert m.takes_images(x1)
            y, b = m.forward_deferred(i1, i2, images=True)
            y = y + b
    finally:
        gemm.set_policy("default")
    ref = torch.from_numpy(g["y"]).cuda().double()
    assert_close(y.cpu().numpy(), g["y"], 1e-3, 0, "y (f16s images)", scale_atol=1e-3)
    # emulated TF32 of the same module in float64 arithmetic over rounded operands
    r = lambda t: round_tf32(t.float()).double()
    H, hd = m.num_heads, m.head_dim
    def qkv(x, lin):
        return (r(x) @ r(lin.weight).t() + lin.bias.double()).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    (q1, k1, v1), (q2, k2, v2) = qkv(x1, m.qkv1), qkv(x2, m.qkv2)
    def sdpa(q, k, v):
        p = torch.softmax((r(q) @ r(k).transpose(-1, -2)) * hd ** -0.5, -1)
        return (r(p) @ r(v)).transpose(1, 2).reshape(B, N, H * hd)
    fused = torch.cat([sdpa(q1, k2, v2), sdpa(q2, k1, v1)], -1)
    y_tf = r(fused) @ r(m.proj.weight).t() + m.proj.bias.double()
    e1, et = (y.double() - ref).abs(), (y_tf - ref).abs()
    scale = ref.abs().max().item()
    print(f"fusion_hd24 vs reference golden, max / rms over max|y|: f16s {e1.max().item() / scale:.2e} / {e1.pow(2).mean().sqrt().item() / scale:.2e}, "
          f"emulated TF32 {et.max().item() / scale:.2e} / {et.pow(2).mean().sqrt().item() / scale:.2e}")
    assert e1.max().item() <= 1.5 * et.max().item() and e1.pow(2).mean().sqrt().item() <= 1.25 * et.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("L,heads,hd", [(256, 8, 64), (100, 4, 24), (64, 8, 72)])
def test_in_kernel_qkv_bias(L, heads, hd):
    """the qkv Linear biases added inside the kernel == attention on (qkv + bias): the adds are the same fp32 operations,
    so the two results must agree to the last bit."""
    from dimsum_amd import native
    gen = torch.Generator().manual_seed(L * hd)
    W = 3 * heads * hd
    qkv1, qkv2 = torch.randn(2, L, W, generator=gen).cuda(), torch.randn(2, L, W, generator=gen).cuda()
    b1, b2 = torch.randn(W, generator=gen).cuda(), torch.randn(W, generator=gen).cuda()
    a = native.xattn_fusion_fwd(qkv1, qkv2, heads, bias1=b1, bias2=b2)
    b = native.xattn_fusion_fwd(qkv1 + b1, qkv2 + b2, heads)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,L,heads,hd,bias", [(2, 256, 8, 64, True), (1, 1024, 8, 72, False), (2, 100, 4, 24, True), (1, 64, 8, 48, False),
                                               (2, 72, 2, 32, True)])
def test_core_backward_vs_torch_sdpa(B, L, heads, hd, bias):
    """dqkv1 / dqkv2 (and the bias gradients) of the MFMA backward kernels vs torch autograd through fp32 SDPA math.
    fp32 MFMA with recomputed probabilities: rtol 1e-4 + 1e-5 max|ref|."""
    from dimsum_amd.attention_fusion import _XattnCoreFn
    W = 3 * heads * hd
    gen = torch.Generator().manual_seed(L + hd)
    base = [torch.randn(B, L, W, generator=gen), torch.randn(B, L, W, generator=gen)]
    bs = [torch.randn(W, generator=gen), torch.randn(W, generator=gen)] if bias else [None, None]
    dout = torch.randn(B, L, 2 * heads * hd, generator=gen).cuda()
    res = []
    for fused in (True, False):
        q1, q2 = (t.clone().cuda().requires_grad_() for t in base)
        b1, b2 = ((t.clone().cuda().requires_grad_() if t is not None else None) for t in bs)
        if fused:
            out = _XattnCoreFn.apply(q1, q2, b1, b2, heads)
        else:
            def split(t, bb):
                t = t if bb is None else t + bb
                return t.reshape(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
            (qa, ka, va), (qb, kb, vb) = split(q1, b1), split(q2, b2)
            from torch.nn.attention import SDPBackend, sdpa_kernel
            with sdpa_kernel(SDPBackend.MATH):
                x12 = torch.nn.functional.scaled_dot_product_attention(qa, kb, vb)
                x21 = torch.nn.functional.scaled_dot_product_attention(qb, ka, va)
            out = torch.cat((x12.transpose(1, 2).reshape(B, L, -1), x21.transpose(1, 2).reshape(B, L, -1)), dim=-1)
        out.backward(dout)
        res.append([out.detach(), q1.grad, q2.grad] + ([b1.grad, b2.grad] if bias else []))
    for name, a, b in zip(("out", "dqkv1", "dqkv2", "dbias1", "dbias2"), res[0], res[1]):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-4, 0, name, scale_atol=1e-5 if not name.startswith("dbias") else 5e-5)


@pytest.mark.parametrize("B,L,heads,hd", [(2, 256, 16, 64), (1, 100, 4, 24), (2, 64, 6, 48)])
def test_self_attention_mode_fwd_bwd(B, L, heads, hd):
    """n_dirs = 1 (the shared DiTBlock's attention): forward and backward vs fp32 SDPA math incl. the in-kernel qkv bias."""
    from dimsum_amd.attention_fusion import _XattnCoreFn
    from torch.nn.attention import SDPBackend, sdpa_kernel
    W = 3 * heads * hd
    gen = torch.Generator().manual_seed(L + heads)
    base, bias0 = torch.randn(B, L, W, generator=gen), torch.randn(W, generator=gen)
    dout = torch.randn(B, L, heads * hd, generator=gen).cuda()
    res = []
    for fused in (True, False):
        qkv, bias = base.clone().cuda().requires_grad_(), bias0.clone().cuda().requires_grad_()
        if fused:
            out = _XattnCoreFn.apply(qkv, None, bias, None, heads)
        else:
            q, k, v = (qkv + bias).reshape(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
            with sdpa_kernel(SDPBackend.MATH):
                out = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, L, -1)
        out.backward(dout)
        res.append((out.detach(), qkv.grad, bias.grad))
    for name, a, b in zip(("out", "dqkv", "dbias"), res[0], res[1]):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-4, 0, name, scale_atol=1e-5 if name != "dbias" else 5e-5)


@pytest.mark.parametrize("B,L,heads,hd", [(2, 256, 8, 64), (1, 1024, 8, 72), (3, 256, 8, 24), (2, 256, 8, 48), (2, 100, 4, 32), (1, 64, 8, 64)])
def test_split_bf16_core_vs_oracle(B, L, heads, hd):
    """precision = 1: the QK^T / PV contractions on bf16 MFMA with every fp32 operand split into hi + lo (3 products,
    fp32 accumulation). Tolerance rtol 5e-5 + 2e-5 * max|ref| -- the dropped lo.lo terms are 2^-16 relative per product;
    20x tighter than real TF32 (the reference's own matmul policy) would need, and far inside the north star's 1e-3."""
    from dimsum_amd import native
    from oracle import np_ops
    gen = torch.Generator().manual_seed(L + hd)
    qkv1, qkv2 = torch.randn(B, L, 3 * heads * hd, generator=gen), torch.randn(B, L, 3 * heads * hd, generator=gen)
    ref = np_ops.xattn_fusion_core(qkv1.numpy(), qkv2.numpy(), heads)
    out = native.xattn_fusion_fwd(qkv1.cuda(), qkv2.cuda(), heads, split_bf16=True)
    assert_close(out.cpu().numpy(), ref, 5e-5, 0, "out (split-bf16)", scale_atol=2e-5)
    exact = native.xattn_fusion_fwd(qkv1.cuda(), qkv2.cuda(), heads, split_bf16=False)
    rms = ((out - exact).pow(2).mean().sqrt() / exact.pow(2).mean().sqrt()).item()
    assert rms < 1e-5, rms
    # biases ride along the same way, and the self-attention mode shares the kernel
    W = 3 * heads * hd
    b1, b2 = torch.randn(W, generator=gen).cuda(), torch.randn(W, generator=gen).cuda()
    a = native.xattn_fusion_fwd(qkv1.cuda(), qkv2.cuda(), heads, bias1=b1, bias2=b2, split_bf16=True)
    r = native.xattn_fusion_fwd(qkv1.cuda() + b1, qkv2.cuda() + b2, heads, split_bf16=False)
    assert_close(a.cpu().numpy(), r.cpu().numpy(), 5e-5, 0, "out (split-bf16, in-kernel bias)", scale_atol=2e-5)
    sa = native.xattn_fusion_fwd(qkv1.cuda(), None, heads, split_bf16=True)
    sr = native.xattn_fusion_fwd(qkv1.cuda(), None, heads, split_bf16=False)
    assert_close(sa.cpu().numpy(), sr.cpu().numpy(), 5e-5, 0, "self-attention (split-bf16)", scale_atol=2e-5)


def test_split_bf16_follows_matmul_policy():
    """the default precision follows torch.backends.cuda.matmul.allow_tf32 (the reference's policy switch, train.py:20-21)"""
    from dimsum_amd import native
    gen = torch.Generator().manual_seed(1)
    qkv1, qkv2 = torch.randn(1, 128, 3 * 8 * 64, generator=gen).cuda(), torch.randn(1, 128, 3 * 8 * 64, generator=gen).cuda()
    old = torch.backends.cuda.matmul.allow_tf32
    try:
        torch.backends.cuda.matmul.allow_tf32 = False
        assert torch.equal(native.xattn_fusion_fwd(qkv1, qkv2, 8), native.xattn_fusion_fwd(qkv1, qkv2, 8, split_bf16=False))
        torch.backends.cuda.matmul.allow_tf32 = True
        assert torch.equal(native.xattn_fusion_fwd(qkv1, qkv2, 8), native.xattn_fusion_fwd(qkv1, qkv2, 8, split_bf16=True))
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old


@pytest.mark.parametrize("B,L,heads,hd,self_attn", [(2, 256, 8, 64, False), (1, 1024, 8, 72, False), (2, 100, 4, 24, False), (1, 64, 8, 48, False),
                                                    (2, 72, 2, 32, False), (2, 256, 16, 64, True), (1, 100, 4, 24, True)])
def test_split_bf16_backward_vs_exact(B, L, heads, hd, self_attn):
    """precision = 1 in the two backward kernels: dqkv of the split-bf16 MFMA kernels vs the exact-fp32 MFMA kernels (which
    test_core_backward_vs_torch_sdpa pins to torch autograd). rtol 1e-4 + 3e-5 * max|ref|, rms deviation < 2e-5."""
    from dimsum_amd import native
    W = 3 * heads * hd
    gen = torch.Generator().manual_seed(L + hd)
    qkv1 = torch.randn(B, L, W, generator=gen).cuda()
    qkv2 = None if self_attn else torch.randn(B, L, W, generator=gen).cuda()
    b1 = torch.randn(W, generator=gen).cuda()
    b2 = None if self_attn else torch.randn(W, generator=gen).cuda()
    nd = 1 if self_attn else 2
    dout = torch.randn(B, L, nd * heads * hd, generator=gen).cuda()
    out, lse = native.xattn_fusion_fwd(qkv1, qkv2, heads, need_lse=True, bias1=b1, bias2=b2, split_bf16=False)
    ref = native.xattn_fusion_bwd(qkv1, qkv2, out, lse, dout, heads, bias1=b1, bias2=b2, split_bf16=False)
    got = native.xattn_fusion_bwd(qkv1, qkv2, out, lse, dout, heads, bias1=b1, bias2=b2, split_bf16=True)
    for name, a, b in zip(("dqkv1", "dqkv2"), got, ref):
        if b is None:
            assert a is None
            continue
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-4, 0, name, scale_atol=3e-5)
        rms = ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
        assert rms < 2e-5, (name, rms)


@pytest.mark.parametrize("B,L,heads,hd,bias", [(2, 256, 8, 64, True), (1, 1024, 8, 72, False), (2, 100, 4, 24, True), (2, 256, 8, 48, True),
                                               (1, 200, 4, 32, True), (1, 328, 8, 64, False)])      # ragged key / query blocks of the 2-tile kernels
def test_split_bf16_backward_vs_sdpa_math_float64(B, L, heads, hd, bias):
    """The backward that training and `bench.py --mode block` run under the reference's allow_tf32 policy (precision = 1,
    split-bf16 MFMA in xattn_bwd_dq_split / xattn_bwd_dkv_split) against an independent oracle: torch autograd through
    float64 SDPA math on the CPU (attention_fusion.py:72-79 written out). rtol 2e-4 + 4e-5 * max|ref| -- the split drops
    the lo*lo terms (2^-16 relative per product) in each of the 7 GEMM-equivalents; real TF32 would need ~1e-3."""
    from dimsum_amd import native
    from torch.nn.attention import SDPBackend, sdpa_kernel
    W = 3 * heads * hd
    gen = torch.Generator().manual_seed(7 * L + hd)
    q1, q2 = torch.randn(B, L, W, generator=gen), torch.randn(B, L, W, generator=gen)
    b1, b2 = (torch.randn(W, generator=gen), torch.randn(W, generator=gen)) if bias else (None, None)
    dout = torch.randn(B, L, 2 * heads * hd, generator=gen)
    # oracle (float64, CPU)
    r1, r2 = q1.double().requires_grad_(), q2.double().requires_grad_()

    def split(t, bb):
        t = t if bb is None else t + bb.double()
        return t.reshape(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
    (qa, ka, va), (qb, kb, vb) = split(r1, b1), split(r2, b2)
    with sdpa_kernel(SDPBackend.MATH):
        x12 = torch.nn.functional.scaled_dot_product_attention(qa, kb, vb)
        x21 = torch.nn.functional.scaled_dot_product_attention(qb, ka, va)
    ref = torch.cat((x12.transpose(1, 2).reshape(B, L, -1), x21.transpose(1, 2).reshape(B, L, -1)), dim=-1)
    ref.backward(dout.double())
    # HIP, split-bf16 forward and backward
    c = lambda t: None if t is None else t.cuda()
    out, lse = native.xattn_fusion_fwd(c(q1), c(q2), heads, need_lse=True, bias1=c(b1), bias2=c(b2), split_bf16=True)
    d1, d2 = native.xattn_fusion_bwd(c(q1), c(q2), out, lse, c(dout), heads, bias1=c(b1), bias2=c(b2), split_bf16=True)
    assert_close(out.cpu().numpy(), ref.detach().numpy(), 5e-5, 0, "out", scale_atol=2e-5)
    assert_close(d1.cpu().numpy(), r1.grad.numpy(), 2e-4, 0, "dqkv1", scale_atol=4e-5)
    assert_close(d2.cpu().numpy(), r2.grad.numpy(), 2e-4, 0, "dqkv2", scale_atol=4e-5)


@pytest.mark.parametrize("split", [False, True])
def test_linearity_in_v_at_full_size(split):
    """Size-independent property at the BASELINE config-2 shape (batch 256, 256 tokens, 8 heads x 64): for fixed q, k the
    fusion core is linear in v:  core(q, k, 2 v1 - 3 v2) == 2 core(q, k, v1) - 3 core(q, k, v2). Exact-fp32 MFMA: 2e-5 of
    the scale; split-bf16 (hi + lo of a sum is not the sum of the parts): 5e-5."""
    from dimsum_amd import native
    B, L, heads, hd = 256, 256, 8, 64
    C = heads * hd
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda: torch.randn(B, L, 3 * C, device="cuda", generator=g)
    a1, a2, b1, b2 = mk(), mk(), mk(), mk()
    a2[:, :, :2 * C], b2[:, :, :2 * C] = a1[:, :, :2 * C], b1[:, :, :2 * C]          # same q, k; different v
    f = lambda x, y: native.xattn_fusion_fwd(x, y, heads, split_bf16=split)
    mix1, mix2 = a1.clone(), b1.clone()
    mix1[:, :, 2 * C:], mix2[:, :, 2 * C:] = 2 * a1[:, :, 2 * C:] - 3 * a2[:, :, 2 * C:], 2 * b1[:, :, 2 * C:] - 3 * b2[:, :, 2 * C:]
    lhs = f(mix1, mix2)
    rhs = 2 * f(a1, b1) - 3 * f(a2, b2)
    err, scale = (lhs - rhs).abs().max().item(), rhs.abs().max().item()
    assert err <= (5e-5 if split else 2e-5) * scale, (err, scale)


# ---- the backward pair on the single-product fp16 carrier (precision 2; the "f16s" policy's training arithmetic) -----------------------
def _attn_backward_f64(q1, q2, b1, b2, dout, heads, rnd=None):
    """dqkv1, dqkv2 of the fusion core written out matmul by matmul in float64 (attention_fusion.py:64-79 and its adjoint); `rnd` rounds the
    operands of every matmul first (the reference's allow_tf32 arithmetic when rnd = round-to-TF32)."""
    B, L, W = q1.shape
    hd = W // (3 * heads)
    r = (lambda t: t) if rnd is None else (lambda t: rnd(t.float()).double())

    def split(t, bb):
        t = t.double() if bb is None else t.double() + bb.double()
        return t.reshape(B, L, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)        # (B, heads, L, hd) each
    (qa, ka, va), (qb, kb, vb) = split(q1, b1), split(q2 if q2 is not None else q1, b2 if q2 is not None else b1)
    dirs = [(qa, kb, vb), (qb, ka, va)] if q2 is not None else [(qa, ka, va)]
    C = heads * hd
    grads = []
    for d, (q, k, v) in enumerate(dirs):
        do = dout.double()[..., d * C:(d + 1) * C].reshape(B, L, heads, hd).permute(0, 2, 1, 3)
        S = r(q) @ r(k).transpose(-1, -2) * hd ** -0.5
        P = torch.softmax(S, dim=-1)
        O = r(P) @ r(v)
        dP = r(do) @ r(v).transpose(-1, -2)
        dS = P * (dP - (do * O).sum(-1, keepdim=True))
        grads.append((r(dS) @ r(k) * hd ** -0.5, r(dS).transpose(-1, -2) @ r(q) * hd ** -0.5, r(P).transpose(-1, -2) @ r(do)))
    rows = lambda t: t.permute(0, 2, 1, 3).reshape(B, L, C)
    if q2 is None:
        return torch.cat([rows(g) for g in grads[0]], dim=-1), None
    (dqa, dkb, dvb), (dqb, dka, dva) = grads
    return torch.cat((rows(dqa), rows(dka), rows(dva)), dim=-1), torch.cat((rows(dqb), rows(dkb), rows(dvb)), dim=-1)


@pytest.mark.parametrize("B,L,heads,hd,self_attn", [(2, 256, 8, 64, False), (1, 1024, 8, 72, False), (2, 100, 4, 24, False), (2, 256, 8, 48, False),
                                                    (1, 200, 4, 32, False), (1, 328, 8, 64, False), (2, 256, 16, 64, True), (1, 100, 4, 24, True)])
def test_fp16_backward_vs_float64_and_emulated_tf32(B, L, heads, hd, self_attn):
    """precision = 2 in the two backward kernels (ONE fp16 MFMA product per element, dout rows scaled by exact powers of two) against float64
    math on the CPU, with the reference's own arithmetic as the yardstick: the same math with every matmul operand rounded to TF32
    (train.py:20-21 turns TF32 on for the backward matmuls too; rounded to NEAREST -- the kinder reading of that hardware). Per tensor:
    rms error <= 1.5 x and max error <= 2.5 x the emulated-TF32 run's (measured: rms 1.0 - 1.4 x, max 0.75 - 1.85 x: the kernels take P from
    the forward's exact log-sum-exp, the emulation re-normalises its rounded scores), and within 5e-3 of the scale in absolute terms (emulated TF32 itself: up to 3.2e-3 on dq at 1024 tokens)."""
    from dimsum_amd import native
    from dimsum_amd.utils.tf32_emulation import round_tf32
    W = 3 * heads * hd
    gen = torch.Generator().manual_seed(11 * L + hd)
    q1 = torch.randn(B, L, W, generator=gen)
    q2 = None if self_attn else torch.randn(B, L, W, generator=gen)
    b1 = torch.randn(W, generator=gen)
    b2 = None if self_attn else torch.randn(W, generator=gen)
    dout = torch.randn(B, L, (1 if self_attn else 2) * heads * hd, generator=gen) * 1e-3
    ref = _attn_backward_f64(q1, q2, b1, b2, dout, heads)
    emu = _attn_backward_f64(q1, q2, b1, b2, dout, heads, rnd=round_tf32)
    c = lambda t: None if t is None else t.cuda()
    out, lse = native.xattn_fusion_fwd(c(q1), c(q2), heads, need_lse=True, bias1=c(b1), bias2=c(b2), split_bf16=True)
    got = native.xattn_fusion_bwd(c(q1), c(q2), out, lse, c(dout), heads, bias1=c(b1), bias2=c(b2), f16=True)
    for name, g, r_, e in zip(("dqkv1", "dqkv2"), got, ref, emu):
        if r_ is None:
            assert g is None
            continue
        g = g.double().cpu()
        assert torch.isfinite(g).all(), name
        for part, sl in zip(("dq", "dk", "dv"), (slice(0, W // 3), slice(W // 3, 2 * W // 3), slice(2 * W // 3, W))):
            eg, ee, sc = (g[..., sl] - r_[..., sl]), (e[..., sl] - r_[..., sl]), r_[..., sl].abs().max().item()
            assert eg.abs().max().item() <= 2.5 * ee.abs().max().item() + 1e-6 * sc, (name, part, eg.abs().max().item(), ee.abs().max().item(), sc)
            assert eg.pow(2).mean().sqrt().item() <= 1.5 * ee.pow(2).mean().sqrt().item() + 1e-7 * sc, (name, part, eg.pow(2).mean().sqrt().item(), ee.pow(2).mean().sqrt().item())
            assert eg.abs().max().item() <= 5e-3 * sc, (name, part, eg.abs().max().item(), sc)


@pytest.mark.parametrize("hd,L", [(64, 256), (72, 160), (24, 100)])
def test_fp16_backward_takes_gradient_rows_of_any_magnitude(hd, L):
    """What the row scales are for. (a) dout rows spread over 2^-60 .. 1 (and some exactly zero): every dq row keeps its own accuracy -- within
    3 x the row's error under emulated TF32 (the reference's arithmetic, whose 8-bit exponent covers any row) + 5e-4 of the row's maximum;
    a carrier without row scales would flush the small rows to zero, an error of the whole row. dk / dv, sums over rows of all magnitudes,
    are held to the same yardstick per (batch, head). (b) Scaling dout by a power of two scales every result by exactly that power, bit
    for bit, from 2^-100 to 2^+60: the kernels' arithmetic does not depend on the gradient's magnitude. (c) dout = 0 -> zeros."""
    from dimsum_amd import native
    from dimsum_amd.utils.tf32_emulation import round_tf32
    B, heads = 2, 4
    W = 3 * heads * hd
    gen = torch.Generator().manual_seed(hd + L)
    q1, q2 = torch.randn(B, L, W, generator=gen), torch.randn(B, L, W, generator=gen)
    dout = torch.randn(B, L, 2 * heads * hd, generator=gen)
    expo = torch.randint(-60, 1, (B, L, 2 * heads, 1), generator=gen).float()
    live = (torch.rand(B, L, 2 * heads, 1, generator=gen) > 0.1).float()
    dout = (dout.view(B, L, 2 * heads, hd) * torch.exp2(expo) * live).view(B, L, -1)
    ref = _attn_backward_f64(q1, q2, None, None, dout, heads)
    emu = _attn_backward_f64(q1, q2, None, None, dout, heads, rnd=round_tf32)
    out, lse = native.xattn_fusion_fwd(q1.cuda(), q2.cuda(), heads, need_lse=True, split_bf16=True)
    got = native.xattn_fusion_bwd(q1.cuda(), q2.cuda(), out, lse, dout.cuda(), heads, f16=True)
    C = heads * hd
    for name, g, r_, e_ in zip(("dqkv1", "dqkv2"), got, ref, emu):
        g = g.double().cpu()
        assert torch.isfinite(g).all(), name
        dq, rq, eq = (t[..., :C].reshape(B, L, heads, hd) for t in (g, r_, e_))
        rowmax = rq.abs().amax(-1)
        err, err_emu = (dq - rq).abs().amax(-1), (eq - rq).abs().amax(-1)
        bad = err > 3 * err_emu + 5e-4 * rowmax
        assert not bad.any(), (name, "dq rows", int(bad.sum()), (err / rowmax.clamp_min(1e-300))[bad].max().item())
        assert (rowmax > 0).sum() > 0.8 * rowmax.numel() and (rowmax[rowmax > 0].max() / rowmax[rowmax > 0].min()) > 2.0 ** 40      # the rows do span the range
        for part, sl in (("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
            # per (batch, head): the sums over queries are accurate relative to the head's own largest entry
            a, b, e = (t[..., sl].reshape(B, L, heads, hd) for t in (g, r_, e_))
            sc = b.abs().amax((1, 3))
            err, err_emu = (a - b).abs().amax((1, 3)), (e - b).abs().amax((1, 3))
            assert (err <= 3 * err_emu + 5e-4 * sc).all(), (name, part, (err / sc).max().item(), (err_emu / sc).max().item())
    for k in (-100, 60):
        s = 2.0 ** k
        scaled = native.xattn_fusion_bwd(q1.cuda(), q2.cuda(), out, lse, (dout * s).cuda(), heads, f16=True)
        for a, b in zip(scaled, got):
            keep = (b.abs() * s > 1e-30) & (b.abs() * s < 1e30)       # away from fp32's own under / overflow
            assert torch.equal(a[keep], (b * s)[keep]), k
    zero = native.xattn_fusion_bwd(q1.cuda(), q2.cuda(), out, lse, torch.zeros_like(dout).cuda(), heads, f16=True)
    assert all((z == 0).all() for z in zero)


def test_fp16_backward_follows_the_training_policy():
    """_XattnCoreFn picks the carrier from the matmul policy: three split-bf16 products by default, ONE fp16 product under gemm.set_policy("f16s")
    (DIMSUM_F16S_TRAIN=0 keeps the three products)"""
    from dimsum_amd import gemm, native
    from dimsum_amd.attention_fusion import _XattnCoreFn
    gen = torch.Generator().manual_seed(3)
    W = 3 * 8 * 64
    base1, base2 = torch.randn(2, 128, W, generator=gen).cuda(), torch.randn(2, 128, W, generator=gen).cuda()
    dout = torch.randn(2, 128, 2 * 8 * 64, generator=gen).cuda()
    seen = []
    real = native.xattn_fusion_bwd
    old_tf32 = torch.backends.cuda.matmul.allow_tf32

    def spy(*a, **k):
        seen.append(bool(k.get("f16")))
        return real(*a, **k)
    native.xattn_fusion_bwd = spy
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        for policy, want in (("default", False), ("f16s", True)):
            gemm.set_policy(policy)
            a, b = base1.clone().requires_grad_(), base2.clone().requires_grad_()
            _XattnCoreFn.apply(a, b, None, None, 8).backward(dout)
            assert seen[-1] is want, (policy, seen)
    finally:
        native.xattn_fusion_bwd = real
        gemm.set_policy("default")
        torch.backends.cuda.matmul.allow_tf32 = old_tf32
