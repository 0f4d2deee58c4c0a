"""GPU parity of the composed denoiser (every hot op through the C ABI of libdimsum_hip.so) vs goldens captured from
the reference model with identical procedural weights (tests/golden/procedural.py).
Tolerance: the north star's 1e-3 relative; we assert rtol 1e-3 + 1e-4 * max|ref| on whole-model outputs (16-24 layers of
fp32 GEMMs whose summation order differs between hipBLASLt and the CPU BLAS that produced the goldens) and
rtol 2e-4 + 2e-5 * max|ref| on single blocks."""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden
from procedural import procedural_fill, seeded

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _published(**over):
    kw = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, learn_sigma=False, scan_type="none",
              pe_type="ape", block_type="combined", cond_mamba=True, scanning_continuity=False, enable_fourier_layers=False,
              drop_path=0.0, rms_norm=True, fused_add_norm=True, learnable_pe=True, use_final_norm=False,
              use_attn_every_k_layers=4, use_gated_mlp=True)
    kw.update(over)
    return kw


@pytest.fixture(autouse=True)
def _fp32_matmul():
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")


# the hidden-64 / hidden-128 fixtures have head_dim 4 / 8: their attention core is torch SDPA by explicit opt-in (conftest);
# everything else in them -- and ALL of the zoo-size tests -- runs on libdimsum_hip.so
tiny = pytest.mark.usefixtures("allow_torch_sdpa")


@tiny
@pytest.mark.parametrize("r,t,c", [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)])
def test_block_combined_forward(r, t, c):
    from dimsum_amd.models_dim import create_block
    g = golden("block_combined")
    blk = create_block(128, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=bool(r), transpose=bool(t), cond_mamba=True,
                       scanning_continuity=bool(c), use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    blk = blk.cuda().eval()
    with torch.no_grad():
        y, ro = blk(T(g["x"]).cuda(), T(g["residual"]).cuda(), T(g["c"]).cuda())
    tag = f"r{r}t{t}c{c}"
    assert_close(y.cpu().numpy(), g[f"{tag}_y"], 2e-4, 0, "y", scale_atol=2e-5)
    assert np.array_equal(ro.cpu().numpy(), g[f"{tag}_res_out"])


@tiny
@pytest.mark.parametrize("tag,over,fold", [("tiny", {}, "1"), ("tiny_cont", dict(scanning_continuity=True), "1"),
                                           ("tiny_fourier", dict(block_type="combined_fourier"), "1"),
                                           ("tiny_fourier", dict(block_type="combined_fourier"), "0"),
                                           ("tiny_final_norm", dict(use_final_norm=True, num_classes=10), "1"),
                                           ("tiny_zigma8", dict(scan_type="zigma_8"), "1"), ("tiny_zigma8", dict(scan_type="zigma_8"), "0"),
                                           ("tiny_jpeg8", dict(scan_type="jpeg_8"), "1"), ("tiny_jpeg8", dict(scan_type="jpeg_8"), "0"),
                                           ("tiny_sweep8", dict(scan_type="sweep_8"), "1"), ("tiny_sweep8", dict(scan_type="sweep_8"), "0")])
def test_tiny_models_forward(tag, over, fold, monkeypatch):
    """`fold`: DIMSUM_FOLD_ZIGZAG -- 1 composes the mixers' zigzag gathers (mamba_simple.py:627-657) into the enclosing
    block's token tables, 0 lets the mixers gather by themselves; both against the reference golden."""
    from dimsum_amd.models_dim import DiM
    monkeypatch.setenv("DIMSUM_FOLD_ZIGZAG", fold)
    g = golden("model_" + tag)
    m = DiM(depth=4, hidden_size=64, patch_size=2, **_published(**over))
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(T(g["x"]).cuda(), T(g["t"]).cuda(), T(g["y"]).cuda())
        assert_close(out.cpu().numpy(), g["out"], 2e-4, 0, "out", scale_atol=2e-5)
        if tag == "tiny":
            x4, t4, y4 = T(g["cfg_x"]).cuda(), T(g["cfg_t"]).cuda(), T(g["cfg_y"]).cuda()
            assert_close(m.forward_with_cfg(x4, t4, y4, cfg_scale=1.4).cpu().numpy(), g["cfg_out"], 2e-4, 0, "cfg", scale_atol=2e-5)
            assert_close(m(x4, t4, None).cpu().numpy(), g["out_nolabel"], 2e-4, 0, "nolabel", scale_atol=2e-5)
            assert_close(m.forward_with_adacfg(x4, t4, y4, cfg_scale=3.8, scale_pow=4.0).cpu().numpy(), g["adacfg_out"], 2e-4, 0,
                         "adacfg", scale_atol=2e-5)     # models_dim.py:1904-1924
    zz = [mm for mm in m.modules() if getattr(mm, "zigzag_paths", None) is not None and mm._is_zigzag()]
    if "scan_type" in over or tag == "tiny_fourier":
        assert len(zz) > 0 and all(getattr(mm, "_zigzag_folded", False) == (fold == "1") for mm in zz)


@pytest.mark.parametrize("name,tag,B,R,over,fold", [("DiM-S/2", "model_S2", 4, 32, {}, "1"), ("DiM-L/2", "model_L2", 2, 32, {}, "1"),
                                                    ("DiM-XL/2", "model_XL2_512", 1, 64, {}, "1"),
                                                    ("DiM-XL/2", "model_XL2_512_zigma8", 1, 64, dict(scan_type="zigma_8"), "1"),
                                                    ("DiM-XL/2", "model_XL2_512_zigma8", 1, 64, dict(scan_type="zigma_8"), "0")])
def test_zoo_forward(name, tag, B, R, over, fold, monkeypatch):
    """BASELINE configs 1/2/5 shapes: S/2 (L=256), L/2 (L=256, the headline model), XL/2 at 512 px (L=1024), and
    configs[4] proper: XL/2 at 512 px with the 8-way zigzag scanning orders inside the mixers (mamba_simple.py:627-657)."""
    from dimsum_amd.models_dim import DiM_models
    monkeypatch.setenv("DIMSUM_FOLD_ZIGZAG", fold)
    g = golden(tag)
    m = DiM_models[name](**_published(img_resolution=R, **over))
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    with torch.no_grad():
        out = m(T(seeded((B, 4, R, R), 71)).cuda(), T(g["t"]).cuda(), T(g["y"]).cuda())
    assert_close(out.cpu().numpy(), g["out"], 1e-3, 0, "out", scale_atol=1e-4)


@pytest.mark.parametrize("name,tag,B,R,over", [("DiM-L/2", "model_L2", 2, 32, {}), ("DiM-XL/2", "model_XL2_512", 1, 64, {}),
                                               ("DiM-XL/2", "model_XL2_512_zigma8", 1, 64, dict(scan_type="zigma_8"))])
def test_zoo_forward_under_the_reference_matmul_policy(name, tag, B, R, over):
    """the configuration bench.py times: torch.backends.cuda.matmul.allow_tf32 = True (dimsum/train.py:20-21), i.e. hipBLASLt's
    split-bf16 GEMMs AND the split-bf16 MFMA attention kernels -- against the same reference goldens with the same tolerance
    as the exact-fp32 run (north star: 1e-3)."""
    from dimsum_amd.models_dim import DiM_models
    g = golden(tag)
    m = DiM_models[name](**_published(img_resolution=R, **over))
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    old = torch.backends.cuda.matmul.allow_tf32
    try:
        torch.backends.cuda.matmul.allow_tf32 = True
        with torch.no_grad():
            out = m(T(seeded((B, 4, R, R), 71)).cuda(), T(g["t"]).cuda(), T(g["y"]).cuda())
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
    assert_close(out.cpu().numpy(), g["out"], 1e-3, 0, "out (allow_tf32 policy)", scale_atol=1e-4)


# ---- the scaled-fp16 single-product policy ("f16s", bench.py's headline arithmetic since round 5) against the REFERENCE goldens ----------
@pytest.fixture
def f16s_policy(monkeypatch):
    """policy "f16s" with its image carriers forced on (the golden batches have 512 / 1024 rows, below the 8192-row default threshold):
    exactly the kernels bench.py's headline runs at batch 256 -- scaled-fp16 images out of the norm / token / attention producers,
    ONE fp16 MFMA product per element in in_proj / qkv / proj / w12 + gate / w3 and in the attention's QK^T / PV"""
    from dimsum_amd import gemm
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    gemm.set_policy("f16s")
    yield
    gemm.set_policy("default")


def _count_f16s_products(monkeypatch):
    """spy: how many GEMM launches took scaled-fp16 operands (scales=...) -- the policy must really be the path under test"""
    from dimsum_amd import native
    seen = {"f16s": 0, "other": 0}
    real = native.gemm_nt

    def spy(a, b, **kw):
        seen["f16s" if kw.get("scales") is not None else "other"] += 1
        return real(a, b, **kw)
    monkeypatch.setattr(native, "gemm_nt", spy)
    return seen


@pytest.mark.parametrize("name,tag,B,R,over", [("DiM-L/2", "model_L2", 2, 32, {}), ("DiM-XL/2", "model_XL2_512", 1, 64, {}),
                                               ("DiM-XL/2", "model_XL2_512_zigma8", 1, 64, dict(scan_type="zigma_8"))])
def test_zoo_forward_under_the_f16s_policy_vs_reference_goldens(name, tag, B, R, over, f16s_policy, monkeypatch):
    """the headline arithmetic (TF32-equivalent single product; dimsum/train.py:20-21 is what it stands for) against the goldens the
    REFERENCE model produced in exact fp32 on the CPU. Tolerance: |err| <= 1e-3 |ref| + 5e-4 max|ref| -- half the north star's 1e-3.
    The fp32-class rows above hold 1e-4 max|ref|; 16-28 layers of 10-bit-mantissa products cannot (measured here: f16s 2.2e-4 / 3.0e-4 /
    3.6e-4 of max|ref| for the three models), and neither can the reference's own TF32 arithmetic: the emulated-TF32 forward of the same
    model (every torch matmul on operands rounded to 10 mantissa bits, attention exact) is run against the same golden and the f16s
    error must not exceed 1.25 x its maximum / 1.1 x its rms."""
    from dimsum_amd import gemm
    from dimsum_amd.models_dim import DiM_models
    from dimsum_amd.utils.tf32_emulation import emulated_tf32
    g = golden(tag)
    m = DiM_models[name](**_published(img_resolution=R, **over))
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    seen = _count_f16s_products(monkeypatch)
    args = (T(seeded((B, 4, R, R), 71)).cuda(), T(g["t"]).cuda(), T(g["y"]).cuda())
    with torch.no_grad():
        out = m(*args)
    depth = len(m.blocks)
    assert seen["f16s"] >= 4 * depth, seen             # in_proj x 2, w12, w3 per block (+ qkv x 2, proj in every block with a fusion)
    assert_close(out.cpu().numpy(), g["out"], 1e-3, 0, "out (f16s policy)", scale_atol=5e-4)
    gemm.set_policy("default")
    with torch.no_grad(), emulated_tf32():
        out_tf = m(*args)
    ref = T(g["out"]).cuda().double()
    e1, et, scale = (out.double() - ref).abs(), (out_tf.double() - ref).abs(), ref.abs().max().item()
    print(f"{tag} vs reference golden, max / rms over max|out|: f16s {e1.max().item() / scale:.2e} / {e1.pow(2).mean().sqrt().item() / scale:.2e}, "
          f"emulated TF32 {et.max().item() / scale:.2e} / {et.pow(2).mean().sqrt().item() / scale:.2e}")
    assert e1.max().item() <= 1.25 * et.max().item() and e1.pow(2).mean().sqrt().item() <= 1.1 * et.pow(2).mean().sqrt().item()


class _LaunchSpy:
    """records which of the inference launch routes a forward took: the scan's fused options, the TN out_proj over a block-scale table,
    the GEMM epilogues and the kernel family the library picked for each gemm_nt call (native.gemm_kernel_log)"""

    def __init__(self, monkeypatch):
        from dimsum_amd import native
        self.scan, self.tn_tables, self.conv_done = [], 0, 0
        real_scan, real_tn = native.selective_scan_fwd, native.gemm_tn

        def scan(*a, **kw):
            self.scan.append((kw.get("dt_proj") is not None, bool(kw.get("out_z_f16")), native.scan_fwd_kernel_for(*a[0].shape, a[2].shape[1], a[3].shape[1])))
            return real_scan(*a, **kw)

        def tn(a, b, **kw):
            sc = kw.get("scales")
            self.tn_tables += int(sc is not None and sc[0].dim() == 2)
            return real_tn(a, b, **kw)
        monkeypatch.setattr(native, "selective_scan_fwd", scan)
        monkeypatch.setattr(native, "gemm_tn", tn)


@pytest.mark.parametrize("name,tag,B,R,over", [("DiM-L/2", "model_L2", 128, 32, {}), ("DiM-XL/2", "model_XL2_512_zigma8", 64, 64, dict(scan_type="zigma_8"))])
def test_as_run_launch_mix_at_bench_batch_sizes_vs_reference_goldens(name, tag, B, R, over, monkeypatch):
    """The EXACT configuration bench.py times, against the reference goldens: policy "f16s" with its DEFAULT thresholds (no forced carriers,
    no forced kernel variant) at the batch sizes of the sampling leg (128 latents per GPU of DiM-L/2) and of BASELINE's config 5 (64 latents
    of DiM-XL/2 at 512 px with the 8-way zigzag orders). The golden latents (2 resp. 1, from the reference model in exact fp32) sit at the first
    and last rows of the batch, random latents in between (every latent is independent through the denoiser). A spy asserts the launch mix:
    at DiM-L/2 every mixer takes the 64-channel scan with dt_proj fused and block-scaled fp16 out_z, out_proj the TN product over that
    table, in_proj carries the conv in its epilogue, qkv leaves as scaled fp16, w12 + gate runs as the persistent stream; at XL/2 (1152
    waves: the 4-lanes-per-channel scan) the scan keeps the reference-shaped launch. Tolerance: the zoo's f16s bound."""
    from dimsum_amd import _lib, gemm, native
    from dimsum_amd.models_dim import DiM_models
    monkeypatch.delenv("DIMSUM_SPLIT3_MIN_ROWS", raising=False)
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", True)
    g = golden(tag)
    m = DiM_models[name](**_published(img_resolution=R, **over))
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    n_gold = g["out"].shape[0]
    gold_rows = [0, B - 1][:n_gold]
    gen = torch.Generator().manual_seed(4242)
    x = torch.randn(B, 4, R, R, generator=gen)
    t, y = torch.rand(B, generator=gen), torch.randint(0, 1000, (B,), generator=gen)
    x[gold_rows] = T(seeded((n_gold, 4, R, R), 71))
    t[gold_rows], y[gold_rows] = T(g["t"]).to(t.dtype), T(g["y"]).to(y.dtype)
    spy = _LaunchSpy(monkeypatch)
    gemm.set_policy("f16s")
    try:
        assert native._scan_fwd_variant == 0
        with torch.no_grad(), native.gemm_kernel_log() as log:
            out = m(x.cuda(), t.cuda(), y.cuda())
    finally:
        gemm.set_policy("default")
    depth = len(m.blocks)
    mixers = sum(1 for mm in m.modules() if hasattr(mm, "x_proj") and hasattr(mm, "dt_proj"))
    assert len(spy.scan) == mixers >= 2 * depth
    by = {}
    for epi, kern in log:
        by.setdefault(epi, []).append(kern)
    assert len(by.get("gated_f16", [])) >= depth and all(k == 2 for k in by["gated_f16"]), by.get("gated_f16")     # w12 + gate: persistent stream
    assert len(by.get("f16_qkv", [])) >= 2 * depth, {k: len(v) for k, v in by.items()}
    if name == "DiM-L/2":
        assert all(s == (True, True, 1) for s in spy.scan), spy.scan
        assert spy.tn_tables == mixers and len(by.get("f32_conv", [])) == mixers, {k: len(v) for k, v in by.items()}
    else:
        # the 4-lanes-per-channel scan writes fp32 out_z; out_proj still runs as ONE fp16 product: a conversion pass builds the block-scaled image and
        # the TN kernel takes d_model = 576 through zero-padded weight rows (gemm.out_proj_f16_convert_enabled)
        assert all(s == (False, False, 4) for s in spy.scan), spy.scan
        assert spy.tn_tables == mixers
    assert_close(out[gold_rows].cpu().numpy(), g["out"], 1e-3, 0, f"{tag} rows {gold_rows} of a batch of {B} (f16s, default thresholds)", scale_atol=5e-4)
    assert torch.isfinite(out).all()


def test_block_combined_1024_forward_under_the_f16s_policy_vs_reference_golden(f16s_policy, monkeypatch):
    """configs[2]'s block at DiM-L/2's width against the reference block golden, inference forward under the headline policy. Tolerance:
    the north star's 1e-3 of max|ref| -- ONE block under any 10-bit-mantissa arithmetic sits at 4-5e-4 (the emulated-TF32 run of this
    block: tests/test_f16s_gpu.py), which the 2e-5 of the fp32-class rows cannot hold; the residual stream stays bit-exact."""
    from dimsum_amd import utils
    from dimsum_amd.models_dim import create_block
    from dimsum_amd.utils.tf32_emulation import emulated_tf32
    g = golden("block_combined_1024")
    blk = create_block(1024, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=True, transpose=True, cond_mamba=True,
                       scanning_continuity=False, use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    blk = blk.cuda().eval()
    sh = (2, 256, 1024)
    x, res, cc = (T(seeded(s_, sd)).cuda() for s_, sd in ((sh, 91), (sh, 92), ((2, 1024), 93)))
    before = utils.torch_path_counts()
    seen = _count_f16s_products(monkeypatch)
    with torch.no_grad():
        y, ro = blk(x, res, cc)
    assert seen["f16s"] >= 7 and utils.torch_path_counts() == before, seen
    assert np.array_equal(ro.cpu().numpy(), g["res_out"])
    assert_close(y.cpu().numpy(), g["y"], 1e-3, 0, "y (f16s policy)", scale_atol=1e-3)
    # ... and not further from the reference golden than the emulated-TF32 run of the same block (the reference's own arithmetic)
    from dimsum_amd import gemm
    gemm.set_policy("default")
    torch.backends.cuda.matmul.allow_tf32 = False
    with torch.no_grad(), emulated_tf32():
        y_tf = blk(x, res, cc)[0]
    ref = T(g["y"]).cuda().double()
    e1, et = (y.double() - ref).abs(), (y_tf.double() - ref).abs()
    print(f"block_1024 vs reference golden, max / rms over max|y|: f16s {e1.max().item() / ref.abs().max().item():.2e} / "
          f"{e1.pow(2).mean().sqrt().item() / ref.abs().max().item():.2e}, emulated TF32 {et.max().item() / ref.abs().max().item():.2e} / "
          f"{et.pow(2).mean().sqrt().item() / ref.abs().max().item():.2e}")
    assert e1.max().item() <= 1.25 * et.max().item() and e1.pow(2).mean().sqrt().item() <= 1.1 * et.pow(2).mean().sqrt().item()


# ---- forward + backward through autograd on the GPU (BASELINE config 3 path) ---------------------------------------------
def test_mamba_inner_fn_fwd_bwd_gpu():
    from dimsum_amd.ops import mamba_inner_fn
    g = golden("mamba_inner")
    names = ["conv_w", "conv_b", "x_proj_w", "dt_proj_w", "out_proj_w", "A", "Dv", "dt_bias"]
    p = {k: T(g[k]).cuda().requires_grad_() for k in names}
    xz = T(g["xz"]).cuda().requires_grad_()
    out = mamba_inner_fn(xz, p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["out_proj_w"], None, p["A"], None, None,
                         p["Dv"], delta_bias=p["dt_bias"], delta_softplus=True)
    out.backward(T(g["dout"]).cuda())
    assert_close(out.detach().cpu().numpy(), g["out"], 2e-4, 0, "out", scale_atol=2e-5)
    assert_close(xz.grad.cpu().numpy(), g["dxz"], 5e-4, 0, "dxz", scale_atol=5e-5)
    for k in names:
        assert_close(p[k].grad.cpu().numpy(), g["g_" + k], 1e-3, 0, "g_" + k, scale_atol=2e-4)


def test_mamba_inner_checkpoint_levels_are_bit_identical(monkeypatch):
    """checkpoint_lvl 0 (the default here: conv_out and delta stay resident between forward and backward) against the reference's hard-coded
    level 1 (selective_scan_interface.py:588: both recomputed in the backward): output and every gradient bit for bit."""
    from dimsum_amd.ops import mamba_inner_fn
    g = golden("mamba_inner")
    names = ["conv_w", "conv_b", "x_proj_w", "dt_proj_w", "out_proj_w", "A", "Dv", "dt_bias"]
    res = []
    for lvl in ("0", "1"):
        monkeypatch.setenv("DIMSUM_MAMBA_CHECKPOINT_LVL", lvl)
        p = {k: T(g[k]).cuda().requires_grad_() for k in names}
        xz = T(g["xz"]).cuda().requires_grad_()
        out = mamba_inner_fn(xz, p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["out_proj_w"], None, p["A"], None, None,
                             p["Dv"], delta_bias=p["dt_bias"], delta_softplus=True)
        out.backward(T(g["dout"]).cuda())
        res.append([out.detach(), xz.grad] + [p[k].grad for k in names])
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("tf32", [False, True])
def test_block_combined_384_fwd_bwd_all_hip(tf32):
    """BASELINE configs[2]'s block at a width whose attention runs on the MFMA kernels (hidden 384, head_dim 24): forward,
    input gradients and parameter gradients vs the reference block golden, nothing on a torch-library path -- in exact fp32
    and under the reference's allow_tf32 policy (split-bf16 GEMMs + split-bf16 attention forward / backward kernels)."""
    from dimsum_amd import utils
    from test_model_cpu import check_block_384
    before = utils.torch_path_counts()
    old = torch.backends.cuda.matmul.allow_tf32
    try:
        torch.backends.cuda.matmul.allow_tf32 = tf32
        check_block_384("cuda", dict(rtol=2e-4, atol=0.0, scale_atol=2e-5), dict(rtol=5e-4, atol=0.0, scale_atol=5e-5))
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
    assert utils.torch_path_counts() == before


@pytest.mark.parametrize("policy", ["fp32", "tf32", "tf32_images"])
def test_block_combined_1024_fwd_bwd_all_hip(policy, monkeypatch):
    """BASELINE configs[2]'s block at config-3's width (hidden 1024, head_dim 64, batch 2) vs the reference golden: exact fp32;
    the reference's allow_tf32 policy (split-bf16 library GEMMs + split-bf16 attention forward / backward at head_dim 64); and
    the same with the split operand images forced on (512 rows is below their default threshold), i.e. exactly the kernels the
    block_fwdbwd leg of bench.py runs at batch 256."""
    from dimsum_amd import utils
    from test_model_cpu import check_block_1024
    before = utils.torch_path_counts()
    old = torch.backends.cuda.matmul.allow_tf32
    if policy == "tf32_images":
        monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")
    try:
        torch.backends.cuda.matmul.allow_tf32 = policy != "fp32"
        check_block_1024("cuda", dict(rtol=2e-4, atol=0.0, scale_atol=2e-5), dict(rtol=5e-4, atol=0.0, scale_atol=5e-5))
    finally:
        torch.backends.cuda.matmul.allow_tf32 = old
    assert utils.torch_path_counts() == before


def test_no_torch_library_attention_on_published_configs():
    """"no silent fallback": a published-config forward + backward never touches torch's SDPA (dimsum_amd.utils counts every
    such call), and a head size without an MFMA kernel raises instead of quietly running elsewhere"""
    from dimsum_amd import utils
    from dimsum_amd.attention_fusion import CrossAttentionFusion
    from dimsum_amd.models_dim import DiM
    before = utils.torch_path_counts()
    m = DiM(depth=4, hidden_size=384, patch_size=2, **_published())        # DiM-S/2's width: head_dim 24 in both attention flavours
    procedural_fill(m, seed=3)
    m = m.cuda()
    x = torch.randn(2, 4, 32, 32, device="cuda", requires_grad=True)
    m(x, torch.rand(2, device="cuda"), torch.tensor([1, 2], device="cuda")).sum().backward()
    assert utils.torch_path_counts() == before
    odd = CrossAttentionFusion(2 * 8 * 40, num_heads=8, qkv_bias=True).cuda()        # head_dim 40: no kernel
    with pytest.raises(RuntimeError, match="no HIP kernel"):
        odd(torch.randn(1, 16, 320, device="cuda"), torch.randn(1, 16, 320, device="cuda"))


@pytest.mark.parametrize("name", ["condmamba_none", "mamba_none", "condmamba_zigma8", "condmamba_v2"])
def test_mixer_modules_gpu(name):
    """Mamba / CondMamba on the HIP path: forward, dx and every parameter gradient vs the reference module goldens
    (mamba_simple.py:42-297 Mamba, :439-657 CondMamba incl. the zigzag gather -> inner fn -> inverse gather of :627-657,
    :593-625 the bidirectional v2 pair over mamba_inner_fn_no_out_proj_cond)."""
    from test_model_cpu import build_mixer, check_mixer
    g = golden(name)
    m = build_mixer(name).cuda()
    check_mixer(name, m, g, "cuda", dict(rtol=2e-4, atol=0.0, scale_atol=2e-5), dict(rtol=1e-3, atol=0, scale_atol=2e-4))
    if name == "condmamba_zigma8":
        assert not getattr(m, "_zigzag_folded", False)        # stand-alone mixer: gathers by itself


@tiny
@pytest.mark.parametrize("r,t,c", [(0, 0, 0), (1, 1, 1)])
def test_block_combined_fwd_bwd(r, t, c):
    from dimsum_amd.models_dim import create_block
    g = golden("block_combined")
    blk = create_block(128, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=bool(r), transpose=bool(t), cond_mamba=True,
                       scanning_continuity=bool(c), use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    blk = blk.cuda()
    x, res, cc = (T(g[k]).cuda().requires_grad_() for k in ("x", "residual", "c"))
    y, ro = blk(x, res, cc)
    ((y * T(g["dy"]).cuda()).sum() + (ro * T(g["dres"]).cuda()).sum()).backward()
    tag = f"r{r}t{t}c{c}"
    assert_close(y.detach().cpu().numpy(), g[f"{tag}_y"], 2e-4, 0, "y", scale_atol=2e-5)
    assert_close(x.grad.cpu().numpy(), g[f"{tag}_dx"], 5e-4, 0, "dx", scale_atol=5e-5)
    assert_close(res.grad.cpu().numpy(), g[f"{tag}_dres"], 5e-4, 0, "dres", scale_atol=5e-5)
    assert_close(cc.grad.cpu().numpy(), g[f"{tag}_dc"], 1e-3, 0, "dc", scale_atol=2e-4)


@tiny
def test_tiny_model_fwd_bwd():
    from dimsum_amd.models_dim import DiM
    g = golden("model_tiny")
    m = DiM(depth=4, hidden_size=64, patch_size=2, **_published())
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    x = T(g["x"]).cuda().requires_grad_()
    out = m(x, T(g["t"]).cuda(), T(g["y"]).cuda())
    out.backward(T(g["dout"]).cuda())
    assert_close(out.detach().cpu().numpy(), g["out"], 2e-4, 0, "out", scale_atol=2e-5)
    assert_close(x.grad.cpu().numpy(), g["dx"], 1e-3, 0, "dx", scale_atol=1e-4)


@tiny
@pytest.mark.gpu
@pytest.mark.parametrize("tf32", [False, True, "f16s"])
def test_hip_graph_replay_is_bit_identical(tf32, monkeypatch):
    """GraphedForward: the denoiser forward replayed from a captured hipGraph (every libdimsum_hip.so launch recorded on
    the capture stream) returns exactly the eager result, also for new input values and through the Euler sampler -- in exact
    fp32 and under allow_tf32, where the MLP GEMMs run on split operand images whose weight halves are rebuilt inside the graph."""
    import torch
    from dimsum_amd import gemm
    monkeypatch.setattr(torch.backends.cuda.matmul, "allow_tf32", bool(tf32))
    monkeypatch.setattr(gemm, "_policy", "f16s" if tf32 == "f16s" else "default")      # (the scaled-fp16 images and their scale tensors under capture)
    monkeypatch.setenv("DIMSUM_SPLIT3_MIN_ROWS", "0")          # 4 x 256 rows: below the carrier's default threshold
    from dimsum_amd.hip_graph import GraphedForward
    from dimsum_amd.models_dim import DiM
    from dimsum_amd.sample_ddp import sample_batch
    from procedural import procedural_fill
    kw = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, scan_type="none", pe_type="ape",
              block_type="combined", cond_mamba=True, rms_norm=True, fused_add_norm=True, learnable_pe=True,
              use_attn_every_k_layers=4)
    m = DiM(depth=4, hidden_size=64, patch_size=2, **kw).eval()
    procedural_fill(m, seed=3)
    m = m.cuda()
    g = GraphedForward(m)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(3):
        x, t = torch.randn(4, 4, 32, 32, device="cuda", generator=gen), torch.rand(4, device="cuda", generator=gen)
        y = torch.randint(0, 1000, (4,), device="cuda", generator=gen)
        with torch.no_grad():
            ref = m(x, t, y)
        assert torch.equal(g(x, t, y), ref)
    assert len(g.graphs) == 1
    # in-place parameter updates through .data (EMA, load_state_dict: no version bump) are seen by eager AND by the replay:
    # nothing derived from a parameter is cached on the host or frozen into the graph (A = -exp(A_log) is part of it)
    before = ref.clone()
    for mm in m.modules():
        if hasattr(mm, "A_log"):
            mm.A_log.data.add_(0.3)
            mm.in_proj.weight.data.mul_(1.01)
        if hasattr(mm, "w12"):
            mm.w12.weight.data.mul_(1.02)
    with torch.no_grad():
        ref2 = m(x, t, y)
    assert not torch.equal(ref2, before) and torch.equal(g(x, t, y), ref2)
    z, y = torch.randn(4, 4, 32, 32, device="cuda", generator=gen), torch.randint(0, 1000, (4,), device="cuda", generator=gen)
    a = sample_batch(m, z, y, num_steps=5, gather=False)
    graphs = {}
    b = sample_batch(m, z, y, num_steps=5, gather=False, hip_graph=graphs)
    assert torch.equal(a, b)
    # the SAME graphs dict across batches (sample_ddp.main with --hip-graph): the graph was captured inside the first batch's
    # frozen_weights() scope, which is closed now -- a replay must neither read weight images that died with that scope nor
    # keep the first capture's weights. Churn the allocator and update weights through .data in between.
    junk = [torch.randn(1 << 20, device="cuda") for _ in range(8)]
    del junk
    torch.cuda.empty_cache()
    for mm in m.modules():
        if hasattr(mm, "in_proj"):
            mm.in_proj.weight.data.mul_(0.97)
        if hasattr(mm, "w12"):
            mm.w12.weight.data.mul_(1.03)
    junk = [torch.full((1 << 18,), float("nan"), device="cuda") for _ in range(16)]     # whatever was freed is overwritten
    del junk
    a2 = sample_batch(m, z, y, num_steps=5, gather=False)
    b2 = sample_batch(m, z, y, num_steps=5, gather=False, hip_graph=graphs)
    assert not torch.equal(a2, a) and torch.equal(a2, b2)


@pytest.mark.gpu
def test_fp16_product_gemm_policy():
    """dimsum_amd.gemm "fp16" (opt-in, inference): fp16 operands + fp32 accumulation in the large Linears. DiM-L/2 on the
    bench inputs (reference init, zero tensors re-drawn): the deviation from the exact-fp32 forward must stay in the TF32 class (the reference's own
    matmul policy has 10 mantissa bits: ~5e-4 per product): max|dev| <= 2e-3 max|ref|, rms <= 5e-4."""
    from dimsum_amd import gemm
    from dimsum_amd.create_model import create_model, published_config
    torch.manual_seed(0)
    m = create_model(published_config(model="DiM-L/2", image_size=256))
    from dimsum_amd.utils import rerandomize_zeros
    rerandomize_zeros(m, std=0.02, seed=0)
    m = m.cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(0)
    x, t = torch.randn(4, 4, 32, 32, device="cuda", generator=gen), torch.rand(4, device="cuda", generator=gen)
    y = torch.randint(0, 1000, (4,), device="cuda", generator=gen)
    try:
        with torch.no_grad():
            ref = m(x, t, y)
            gemm.set_policy("fp16")
            got = m(x, t, y)
    finally:
        gemm.set_policy("default")
    dev = (got - ref)
    assert dev.abs().max().item() <= 2e-3 * ref.abs().max().item(), (dev.abs().max().item(), ref.abs().max().item())
    assert (dev.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item() <= 5e-4
    assert not torch.equal(got, ref)            # the policy really took another path


@pytest.mark.gpu
def test_zigzag_paths_folded_into_block_tables():
    """scan_type zigma_8: the mixers' token gathers (mamba_simple.py:627-657) composed into the blocks' permutation tables
    give bit-identical outputs to the mixers gathering by themselves (tokens only move; every per-token op is unchanged)."""
    import os
    import subprocess
    import sys
    code = ("import torch, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from dimsum_amd.models_dim import DiM; from procedural import procedural_fill; "
            "kw = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, scan_type='zigma_8', pe_type='ape', block_type='combined', "
            "cond_mamba=True, rms_norm=True, fused_add_norm=True, learnable_pe=True, use_attn_every_k_layers=4); "
            "m = DiM(depth=4, hidden_size=64, patch_size=2, **kw).eval(); procedural_fill(m, seed=3); m = m.cuda(); "
            "g = torch.Generator(device='cuda').manual_seed(0); x = torch.randn(3, 4, 32, 32, device='cuda', generator=g); "
            "t = torch.rand(3, device='cuda', generator=g); y = torch.tensor([1, 2, 3], device='cuda'); "
            "folded = [getattr(mm, '_zigzag_folded', False) for mm in m.modules() if hasattr(mm, 'zigzag_paths')]; "
            "out = m(x, t, y).detach(); folded = [getattr(mm, '_zigzag_folded', False) for mm in m.modules() if hasattr(mm, 'zigzag_paths') and mm.zigzag_paths is not None]; "
            "torch.save((out.cpu(), folded), sys.argv[1])")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for fold in ("1", "0"):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"dimsum_zz_{fold}_{os.getpid()}.pt")
        subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests", "golden")), path], check=True,
                       env=dict(os.environ, DIMSUM_FOLD_ZIGZAG=fold, DIMSUM_ALLOW_TORCH_SDPA="1"))
        res.append(torch.load(path))
        os.remove(path)
    (a, fa), (b, fb) = res
    assert len(fa) > 0 and all(fa) and not any(fb)
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_two_stream_branches_are_bit_identical():
    """two-stream branches (the inference default; DIMSUM_BRANCH_STREAMS=0 at import / `branch_streams(False)` keep one stream): the frequency branch of every combined
    block on a second HIP stream -- same kernels, same operands, same result, also when the forward is called twice in a row and
    from a non-default stream"""
    from dimsum_amd.models_dim import DiM, branch_streams
    m = DiM(depth=4, hidden_size=384, patch_size=2, **_published())
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(0)
    x, t = torch.randn(8, 4, 32, 32, device="cuda", generator=g), torch.rand(8, device="cuda", generator=g)
    y = torch.randint(0, 1000, (8,), device="cuda", generator=g)
    with torch.no_grad():
        with branch_streams(False):
            ref = m(x, t, y)
        assert all("_side_stream" not in blk.__dict__ for blk in m.blocks)
        with branch_streams(True):                             # (the default unless DIMSUM_BRANCH_STREAMS=0 was exported)
            a, b = m(x, t, y), m(x, t, y)
        assert any("_side_stream" in blk.__dict__ for blk in m.blocks)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), branch_streams(True):
            c = m(x, t, y)
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(a, ref) and torch.equal(b, ref) and torch.equal(c, ref)


@pytest.mark.gpu
def test_inference_without_out_and_x_stores_is_bit_identical(monkeypatch):
    """the mixers' inference scans skip the stores of the ungated `out` and of the chunk states `x` (NULL out_ptr / x_ptr in the C ABI:
    only a backward reads them; the default since round 5), DIMSUM_SCAN_INFER_STORES=1 keeps the reference interface's stores. Same
    kernel, same arithmetic: the model output is bit-identical; and under autograd the switch changes nothing (the backward needs both)."""
    from dimsum_amd import native
    from dimsum_amd.models_dim import DiM
    m = DiM(depth=4, hidden_size=384, patch_size=2, **_published())
    procedural_fill(m, seed=3)
    m = m.cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    x, t = torch.randn(4, 4, 32, 32, device="cuda", generator=g), torch.rand(4, device="cuda", generator=g)
    y = torch.randint(0, 1000, (4,), device="cuda", generator=g)
    seen = []
    real = native.selective_scan_fwd

    def spy(*a, need_out=True, need_x=True, **k):
        seen.append((need_out, need_x))
        return real(*a, need_out=need_out, need_x=need_x, **k)

    monkeypatch.setattr(native, "selective_scan_fwd", spy)
    with torch.no_grad():
        monkeypatch.setenv("DIMSUM_SCAN_INFER_STORES", "1")
        ref = m(x, t, y)
        assert seen and all(s == (True, True) for s in seen)          # the reference interface's stores
        seen.clear()
        monkeypatch.delenv("DIMSUM_SCAN_INFER_STORES")
        got = m(x, t, y)
        assert seen and all(s == (False, False) for s in seen)        # the default: nothing reads them at inference
    assert torch.equal(got, ref)
    seen.clear()
    xg = x.clone().requires_grad_()
    m(xg, t, y).sum().backward()
    assert seen and all(s == (True, True) for s in seen)              # training: the backward reads out and x
