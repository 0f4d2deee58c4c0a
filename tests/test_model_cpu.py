"""Host-logic parity on CPU: dimsum_amd modules (with the CPU oracle standing in for the HIP library, see
tests/oracle_backend.py) vs golden outputs captured from the reference modules with identical procedural weights.
Tolerance: rtol 2e-4 + 2e-5 * max|ref| (fp32 composition of ~100 ops; the oracle computes in double)."""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden
from oracle.torch_backend import cpu_oracle_backend
from procedural import procedural_fill, seeded

T = torch.from_numpy
TOL = dict(rtol=2e-4, atol=0.0, scale_atol=2e-5)


def _published(**over):
    kw = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, learn_sigma=False, scan_type="none",
              pe_type="ape", block_type="combined", cond_mamba=True, scanning_continuity=False, enable_fourier_layers=False,
              drop_path=0.0, rms_norm=True, fused_add_norm=True, learnable_pe=True, use_final_norm=False,
              use_attn_every_k_layers=4, use_gated_mlp=True)
    kw.update(over)
    return kw


def test_mamba_inner_fn_fwd_bwd():
    from dimsum_amd.ops import mamba_inner_fn
    g = golden("mamba_inner")
    names = ["conv_w", "conv_b", "x_proj_w", "dt_proj_w", "out_proj_w", "A", "Dv", "dt_bias"]
    p = {k: T(g[k]).clone().requires_grad_() for k in names}
    xz = T(g["xz"]).clone().requires_grad_()
    with cpu_oracle_backend():
        out = mamba_inner_fn(xz, p["conv_w"], p["conv_b"], p["x_proj_w"], p["dt_proj_w"], p["out_proj_w"], None, p["A"], None, None,
                             p["Dv"], delta_bias=p["dt_bias"], delta_softplus=True)
        out.backward(T(g["dout"]))
    assert_close(out.detach().numpy(), g["out"], what="out", **TOL)
    assert_close(xz.grad.numpy(), g["dxz"], what="dxz", **TOL)
    for k in names:
        assert_close(p[k].grad.numpy(), g["g_" + k], what="g_" + k, rtol=5e-4, atol=0, scale_atol=1e-4)


def build_mixer(name):
    """the mixer of golden `name` (tools/gen_golden.py::gen_mixer) with its procedural weights"""
    from dimsum_amd.modules.mamba_simple import CondMamba, Mamba
    from dimsum_amd import scanning_orders as so
    kw = dict(layer_idx=3, scan_type="none")
    if name == "condmamba_zigma8":
        paths = so.SCAN_ZOO["zigma"](8)[:8]
        kw.update(scan_type="zigma_8", zigzag_paths=torch.stack([T(p) for p in paths]),
                  zigzag_paths_reverse=torch.stack([T(so.reverse_permut_np(p)) for p in paths]))
    elif name == "condmamba_v2":
        kw.update(scan_type="v2")
    m = Mamba(32, **kw) if name == "mamba_none" else CondMamba(32, d_cond=48, **kw)
    procedural_fill(m, seed=7)
    return m


def check_mixer(name, m, g, dev, tol, gtol):
    x = T(g["x"]).clone().to(dev).requires_grad_()
    y = m(x) if name == "mamba_none" else m(x, T(g["c"]).to(dev))
    y.backward(T(g["dy"]).to(dev))
    assert_close(y.detach().cpu().numpy(), g["y"], what="y", **tol)
    assert_close(x.grad.cpu().numpy(), g["dx"], what="dx", **tol)
    checked = 0
    for k, v in m.named_parameters():
        if "g_" + k in g.files:
            assert_close(v.grad.cpu().numpy(), g["g_" + k], what=k, **gtol)
            checked += 1
        else:
            assert k.startswith("cond_proj") and (v.grad is None or not v.grad.any()), k   # SURVEY finding 1: dead parameter
    assert checked >= 9


MIXERS = ["condmamba_none", "mamba_none", "condmamba_zigma8", "condmamba_v2"]


@pytest.mark.parametrize("name", MIXERS)
def test_mixer_modules(name):
    """Mamba / CondMamba forward + dx + every parameter gradient vs the reference module (mamba_simple.py:42-297, 439-657;
    v2 :593-625 over the reference's own *_ref ops)."""
    g = golden(name)
    m = build_mixer(name)
    with cpu_oracle_backend():
        check_mixer(name, m, g, "cpu", TOL, dict(rtol=5e-4, atol=0, scale_atol=1e-4))


@pytest.mark.parametrize("r,t,c", [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)])
def test_block_combined(r, t, c):
    from dimsum_amd.models_dim import create_block
    g = golden("block_combined")
    blk = create_block(128, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=bool(r), transpose=bool(t), cond_mamba=True,
                       scanning_continuity=bool(c), use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    x, res, cc = (T(g[k]).clone().requires_grad_() for k in ("x", "residual", "c"))
    tag = f"r{r}t{t}c{c}"
    with cpu_oracle_backend():
        y, ro = blk(x, res, cc)
        ((y * T(g["dy"])).sum() + (ro * T(g["dres"])).sum()).backward()
    assert_close(y.detach().numpy(), g[f"{tag}_y"], what="y", **TOL)
    assert np.array_equal(ro.detach().numpy(), g[f"{tag}_res_out"])
    assert_close(x.grad.numpy(), g[f"{tag}_dx"], what="dx", **TOL)
    assert_close(res.grad.numpy(), g[f"{tag}_dres"], what="dres", **TOL)
    assert_close(cc.grad.numpy(), g[f"{tag}_dc"], what="dc", rtol=5e-4, atol=0, scale_atol=1e-4)


def check_block_384(dev, tol, gtol):
    """DiMBlockCombined at DiM-S/2's width (hidden 384: attention head_dim 24), all three order flags on: y, res_out, input
    gradients and three parameter gradients (qkv1 of the fusion, A_log of the spatial mixer, norm_2) vs the reference block"""
    from dimsum_amd.models_dim import create_block
    g = golden("block_combined_384")
    blk = create_block(384, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=True, transpose=True, cond_mamba=True,
                       scanning_continuity=True, use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    blk = blk.to(dev)
    x, res, cc = (T(seeded(sh, sd)).to(dev).requires_grad_() for sh, sd in (((1, 256, 384), 56), ((1, 256, 384), 57), ((1, 384), 58)))
    y, ro = blk(x, res, cc)
    ((y * T(seeded((1, 256, 384), 59)).to(dev)).sum() + (ro * T(seeded((1, 256, 384), 60)).to(dev)).sum()).backward()
    n = lambda t: t.detach().cpu().numpy()
    assert_close(n(y), g["y"], what="y", **tol)
    assert np.array_equal(n(ro), g["res_out"])
    assert_close(n(x.grad), g["dx"], what="dx", **gtol)
    assert_close(n(res.grad), g["dres"], what="dres", **gtol)
    assert_close(n(cc.grad), g["dc"], what="dc", rtol=1e-3, atol=0, scale_atol=2e-4)
    assert_close(n(blk.proj.qkv1.weight.grad), g["g_qkv1"], what="g qkv1", rtol=1e-3, atol=0, scale_atol=2e-4)
    assert_close(n(blk.spatial_mamba.mixer.A_log.grad), g["g_A_log"], what="g A_log", rtol=1e-3, atol=0, scale_atol=2e-4)
    assert_close(n(blk.norm_2.weight.grad), g["g_norm2"], what="g norm_2", rtol=1e-3, atol=0, scale_atol=2e-4)


def check_block_1024(dev, tol, gtol, collect=None):
    """BASELINE configs[2]'s block at its own width -- DiMBlockCombined(1024): fusion head_dim 64, mixers D 1024 / R 32, MLP 8192 --
    batch 2, reverse + transpose on: y, res_out, the three input gradients and 15 parameter gradients (every kernel family's
    backward feeds at least one: attention dq / dkv, scan bwd, conv1d bwd, norm bwd, gated-GeLU bwd, token-transform adjoints) vs
    the reference block (golden block_combined_1024, dimsum/models_dim.py:974-1117)."""
    from dimsum_amd.models_dim import create_block
    g = golden("block_combined_1024")
    blk = create_block(1024, norm_epsilon=1e-5, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=1,
                       scan_type="none", block_type="combined", reverse=True, transpose=True, cond_mamba=True,
                       scanning_continuity=False, use_gated_mlp=True)
    procedural_fill(blk, seed=9)
    blk = blk.to(dev)
    sh = (2, 256, 1024)
    x, res, cc = (T(seeded(s_, sd)).to(dev).requires_grad_() for s_, sd in ((sh, 91), (sh, 92), ((2, 1024), 93)))
    y, ro = blk(x, res, cc)
    ((y * T(seeded(sh, 94)).to(dev)).sum() + (ro * T(seeded(sh, 95)).to(dev)).sum()).backward()
    n = lambda t: t.detach().cpu().numpy()
    sm, fm = blk.spatial_mamba.mixer, blk.freq_mamba.mixer
    if collect is not None:          # (a policy comparison: hand every checked quantity to the caller next to its golden, assert nothing but the exact residual)
        assert np.array_equal(n(ro), g["res_out"])
        collect.update({"y": (n(y), g["y"]), "dx": (n(x.grad), g["dx"]), "dres": (n(res.grad), g["dres"]), "dc": (n(cc.grad), g["dc"])})
    else:
        assert_close(n(y), g["y"], what="y", **tol)
        assert np.array_equal(n(ro), g["res_out"])
        assert_close(n(x.grad), g["dx"], what="dx", **gtol)
        assert_close(n(res.grad), g["dres"], what="dres", **gtol)
    ptol = dict(rtol=1e-3, atol=0, scale_atol=2e-4)
    if collect is None:
        assert_close(n(cc.grad), g["dc"], what="dc", **ptol)
    for key, got in (("g_qkv1_rows64", blk.proj.qkv1.weight.grad[:64]), ("g_qkv2_bias", blk.proj.qkv2.bias.grad),
                     ("g_proj_bias", blk.proj.proj.bias.grad), ("g_A_log", sm.A_log.grad), ("g_D_freq", fm.D.grad),
                     ("g_x_proj", sm.x_proj.weight.grad), ("g_dt_bias_freq", fm.dt_proj.bias.grad), ("g_conv1d", sm.conv1d.weight.grad),
                     ("g_in_proj_rows64", fm.in_proj.weight.grad[:64]), ("g_out_proj_rows64", sm.out_proj.weight.grad[:64]),
                     ("g_norm2", blk.norm_2.weight.grad), ("g_norm", blk.norm.weight.grad), ("g_w12_bias", blk.mlp.w12.bias.grad),
                     ("g_w3_rows16", blk.mlp.w3.weight.grad[:16]), ("g_adaLN_bias", blk.adaLN_modulation[1].bias.grad)):
        if collect is not None:
            collect[key] = (n(got), g[key])
        else:
            assert_close(n(got), g[key], what=key, **ptol)


def test_block_combined_1024():
    with cpu_oracle_backend():
        check_block_1024("cpu", TOL, TOL)


def test_block_combined_384():
    with cpu_oracle_backend():
        check_block_384("cpu", TOL, TOL)


@pytest.mark.parametrize("tag,over", [("tiny", {}), ("tiny_cont", dict(scanning_continuity=True)),
                                      ("tiny_fourier", dict(block_type="combined_fourier")),
                                      ("tiny_final_norm", dict(use_final_norm=True, num_classes=10)),
                                      ("tiny_zigma8", dict(scan_type="zigma_8")), ("tiny_jpeg8", dict(scan_type="jpeg_8")),
                                      ("tiny_sweep8", dict(scan_type="sweep_8"))])
def test_tiny_models(tag, over):
    from dimsum_amd.models_dim import DiM
    g = golden("model_" + tag)
    m = DiM(depth=4, hidden_size=64, patch_size=2, **_published(**over)).eval()
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    procedural_fill(m, seed=3)
    x = T(g["x"]).clone().requires_grad_()
    with cpu_oracle_backend():
        out = m(x, T(g["t"]), T(g["y"]))
        out.backward(T(g["dout"]))
        assert_close(out.detach().numpy(), g["out"], what="out", **TOL)
        assert_close(x.grad.numpy(), g["dx"], what="dx", **TOL)
        if tag == "tiny":
            with torch.no_grad():
                x4, t4, y4 = T(g["cfg_x"]), T(g["cfg_t"]), T(g["cfg_y"])
                assert_close(m.forward_with_cfg(x4, t4, y4, cfg_scale=1.4).numpy(), g["cfg_out"], what="cfg", **TOL)
                assert_close(m.forward_with_adacfg(x4, t4, y4, cfg_scale=3.8, scale_pow=4.0).numpy(), g["adacfg_out"], what="adacfg", **TOL)
                assert_close(m(x4, t4, None).numpy(), g["out_nolabel"], what="no label", **TOL)


def test_S2_forward_and_state_dict_layout():
    """BASELINE config 1: DiM-S/2 (depth 12, hidden 384), 4x4x32x32 latents, CPU plumbing."""
    from dimsum_amd.models_dim import DiM_models
    g = golden("model_S2")
    m = DiM_models["DiM-S/2"](**_published()).eval()
    sd = m.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for _, v in sorted(sd.items())] == [str(s) for s in g["shapes"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    procedural_fill(m, seed=3)
    with cpu_oracle_backend(), torch.no_grad():
        out = m(T(seeded((4, 4, 32, 32), 71)), T(g["t"]), T(g["y"]))
    assert_close(out.numpy(), g["out"], what="out", **TOL)


def test_L2_state_dict_layout():
    """checkpoint wire format: 742 keys, 459.92 M parameters, same shapes -- built on the meta device."""
    from dimsum_amd.models_dim import DiM_models
    g = golden("model_L2")
    with torch.device("meta"):
        m = DiM_models["DiM-L/2"](**_published())
    sd = m.state_dict()
    assert len(sd) == 742 == int(g["n_keys"])
    assert sorted(sd.keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for _, v in sorted(sd.items())] == [str(s) for s in g["shapes"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 459916304


def test_condmamba_v2_bidirectional_runs_and_matches_two_one_directional_calls():
    """scan_type="v2" (mamba_simple.py:593-625): out_proj( fwd_branch(xz) + flip(bwd_branch(flip(xz))) ). The reference's
    path calls CUDA-only code, so there is no golden: the module is checked against the same composition written out
    with the one-directional op, and every live parameter must receive a gradient."""
    from dimsum_amd.modules.mamba_simple import CondMamba
    from dimsum_amd.ops import mamba_inner_fn_no_out_proj
    torch.manual_seed(0)
    m = CondMamba(d_model=16, d_state=4, d_conv=4, expand=2, layer_idx=0, scan_type="v2", d_cond=16)
    x, c = torch.randn(2, 8, 16, requires_grad=True), torch.randn(2, 16)
    with cpu_oracle_backend():
        y = m(x, c)
        y.sum().backward()
        with torch.no_grad():
            xz = (m.in_proj.weight @ x.reshape(16, 16).t()).view(64, 2, 8).permute(1, 0, 2)
            a = mamba_inner_fn_no_out_proj(xz, m.conv1d.weight, m.conv1d.bias, m.x_proj.weight, m.dt_proj.weight, -torch.exp(m.A_log),
                                           None, None, m.D, delta_bias=m.dt_proj.bias, delta_softplus=True)
            b = mamba_inner_fn_no_out_proj(xz.flip([-1]), m.conv1d_b.weight, m.conv1d_b.bias, m.x_proj_b.weight, m.dt_proj_b.weight,
                                           -torch.exp(m.A_b_log), None, None, m.D_b, delta_bias=m.dt_proj_b.bias, delta_softplus=True)
            ref = torch.nn.functional.linear((a + b.flip([-1])).transpose(1, 2), m.out_proj.weight)
    assert torch.allclose(y, ref, atol=1e-5)
    dead = {n for n, p in m.named_parameters() if p.grad is None or not p.grad.any()}
    assert dead <= {"cond_proj.weight", "cond_proj.bias"}, dead
    assert x.grad is not None and x.grad.abs().sum() > 0
