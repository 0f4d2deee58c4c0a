"""GPU tests of the sampler breadth around the denoiser: the same sampling run -- same model, same
initial noise, same labels -- on the HIP path and on the CPU through the oracle backend. The integrators are host logic
(dimsum_amd/transport); what these tests pin is that the denoiser evaluations they chain (5-40 of them, with and without
classifier-free guidance, fixed-grid and adaptive) stay within the model tolerance of the oracle on the GPU, and that the
adaptive solver takes the same step sequence on both. torchdiffeq itself is absent: its semantics are restated
(parity unpinned)."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from procedural import procedural_fill, seeded

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("allow_torch_sdpa")]      # hidden 64: head_dim 4 (conftest)
T = torch.from_numpy

KW = dict(img_resolution=32, in_channels=4, label_dropout=0.15, num_classes=1000, learn_sigma=False, scan_type="none", pe_type="ape",
          block_type="combined", cond_mamba=True, scanning_continuity=False, drop_path=0.0, rms_norm=True, fused_add_norm=True,
          learnable_pe=True, use_final_norm=False, use_attn_every_k_layers=4, use_gated_mlp=True)


def _model():
    from dimsum_amd.models_dim import DiM
    m = DiM(depth=4, hidden_size=64, patch_size=2, **KW).eval()
    procedural_fill(m, seed=3)
    return m


@pytest.fixture(autouse=True)
def _fp32_matmul():
    torch.backends.cuda.matmul.allow_tf32 = False


def _both(run):
    """run(model, device) on the GPU (HIP path) and on the CPU (oracle backend) -> (gpu result, cpu result)"""
    from oracle.torch_backend import cpu_oracle_backend
    with torch.no_grad():
        got = run(_model().cuda(), "cuda")
        with cpu_oracle_backend():
            ref = run(_model(), "cpu")
    return got, ref


@pytest.mark.parametrize("method,steps", [("euler", 8), ("heun2", 4), ("rk4", 2)])
@pytest.mark.parametrize("cfg", [None, 1.4])
def test_fixed_grid_sampling_hip_vs_cpu_oracle(method, steps, cfg):
    """sample_batch (sample_ddp.py:159-191 counterpart): fixed-grid ODE samplers, with and without classifier-free guidance
    (forward_with_cfg on the doubled batch, models_dim.py:1886-1902). rtol 1e-3 + 1e-4 max|ref| (the whole-model tolerance)."""
    from dimsum_amd.sample_ddp import sample_batch
    z, y = T(seeded((3, 4, 32, 32), 91)), torch.tensor([5, 17, 999])

    def run(m, dev):
        return sample_batch(m, z.to(dev), y.to(dev), num_steps=steps, sampling_method=method, cfg_scale=cfg, gather=False).cpu().numpy()
    got, ref = _both(run)
    assert got.shape == (3, 4, 32, 32) and np.isfinite(got).all()
    assert_close(got, ref, 1e-3, 0, f"{method} x{steps} cfg={cfg}", scale_atol=1e-4)


def test_dopri5_sampling_hip_vs_cpu_oracle():
    """the adaptive solver of the published eval recipe (scripts/eval.sh:73-95: dopri5, atol 1e-6, rtol 1e-3): same number of
    function evaluations on both devices (the step controller sees the same error norms to ~1e-6) and the same samples."""
    from dimsum_amd.transport import Sampler, create_transport
    z, y = T(seeded((2, 4, 32, 32), 92)), torch.tensor([3, 500])
    nfe = {}

    def run(m, dev):
        sampler = Sampler(create_transport("GVP", "velocity"))
        fn = sampler.sample_ode(sampling_method="dopri5", num_steps=2, atol=1e-6, rtol=1e-3)
        out = fn(z.to(dev), m.forward, return_trajectory=False, y=y.to(dev))
        nfe[dev] = fn.__self__.last_nfe
        return out.cpu().numpy()
    got, ref = _both(run)
    assert nfe["cuda"] == nfe["cpu"] and nfe["cpu"] >= 8, nfe
    assert_close(got, ref, 1e-3, 0, "dopri5", scale_atol=2e-4)


def test_adacfg_and_sde_drift_hip_vs_cpu_oracle():
    """forward_with_adacfg inside the Euler sampler (models_dim.py:1904-1924) and ONE Euler-Maruyama step of the SDE sampler with
    its noise increment fixed (transport.py:286-341: drift + w(t) * score, two denoiser evaluations): HIP vs CPU oracle."""
    from dimsum_amd.transport import Sampler, create_transport
    z, y = T(seeded((4, 4, 32, 32), 93)), torch.tensor([1, 2, 1000, 1000])

    def run_ada(m, dev):
        fn = Sampler(create_transport("GVP", "velocity")).sample_ode(sampling_method="euler", num_steps=5)
        return fn(z.to(dev), m.forward_with_adacfg, return_trajectory=False, y=y.to(dev), cfg_scale=3.8, scale_pow=4.0).cpu().numpy()
    got, ref = _both(run_ada)
    assert_close(got, ref, 1e-3, 0, "ada-cfg euler", scale_atol=1e-4)

    def run_sde(m, dev):
        sampler = Sampler(create_transport("GVP", "velocity"))
        torch.manual_seed(94)          # the noise is drawn from the CPU generator on either device (like the reference): same draws
        xs = sampler.sample_sde(sampling_method="Euler", diffusion_form="sigma", diffusion_norm=1.0, last_step="Mean",
                                last_step_size=0.04, num_steps=3)(z.to(dev), m.forward, y=y.to(dev))
        return xs[-1].cpu().numpy()
    got, ref = _both(run_sde)
    assert_close(got, ref, 2e-3, 0, "sde euler-maruyama + mean last step", scale_atol=2e-4)


def test_likelihood_ode_hip_vs_cpu_oracle(monkeypatch):
    """Sampler.sample_ode_likelihood (transport.py:388-443) around the real denoiser: every drift evaluation is a forward AND an
    input-gradient backward of the model through the HIP kernels, chained over 3 Euler steps from the data to the prior. Same
    Rademacher probes on both sides (drawn on the CPU generator and moved: the reference draws them on the state's device)."""
    from dimsum_amd.transport import Sampler, create_transport
    x, y = T(seeded((2, 4, 32, 32), 95)), torch.tensor([7, 400])
    real = torch.randint
    monkeypatch.setattr(torch, "randint", lambda *a, device=None, **k: real(*a, **k).to(device or "cpu"))

    def run(m, dev):
        torch.manual_seed(96)
        logp, z = Sampler(create_transport("GVP", "velocity")).sample_ode_likelihood(sampling_method="euler", num_steps=4)(
            x.to(dev), m.forward, y=y.to(dev))
        return logp.cpu().numpy(), z.cpu().numpy()
    (lp, z), (lp_ref, z_ref) = _both(run)
    assert np.isfinite(lp).all()
    assert_close(z, z_ref, 1e-3, 0, "likelihood: state at the prior end", scale_atol=1e-4)
    assert_close(lp, lp_ref, 2e-3, 0, "likelihood: log p", scale_atol=2e-4)
