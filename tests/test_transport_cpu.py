"""CPU tests of the flow-matching host logic: path plans vs closed forms, Euler stepping count / accuracy on an analytic
field, GVP training target, and the world_size-2 gloo sharding + all-gather of sample_ddp."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dimsum_amd.transport import Sampler, create_transport


def test_gvp_plan_matches_closed_form():
    tr = create_transport("GVP", "velocity")
    x0, x1 = torch.randn(5, 4, 8, 8), torch.randn(5, 4, 8, 8)
    t = torch.rand(5)
    _, xt, ut = tr.path_sampler.plan(t, x0, x1)
    tt = t.view(-1, 1, 1, 1)
    assert torch.allclose(xt, torch.sin(math.pi * tt / 2) * x1 + torch.cos(math.pi * tt / 2) * x0, atol=1e-6)
    assert torch.allclose(ut, math.pi / 2 * (torch.cos(math.pi * tt / 2) * x1 - torch.sin(math.pi * tt / 2) * x0), atol=1e-6)
    assert tr.check_interval(tr.train_eps, tr.sample_eps, eval=True) == (0, 1)


def test_training_loss_is_mse_to_target():
    torch.manual_seed(0)
    tr = create_transport("GVP", "velocity")
    x1 = torch.randn(6, 4, 8, 8)
    seen = {}

    def model(xt, t, y=None):
        seen["t"], seen["xt"] = t, xt
        return torch.zeros_like(xt)

    torch.manual_seed(1)
    terms = tr.training_losses(model, x1, dict(y=torch.zeros(6, dtype=torch.long)))
    torch.manual_seed(1)
    t, x0, _ = tr.sample(x1)
    _, xt, ut = tr.path_sampler.plan(t, x0, x1)
    assert torch.equal(seen["xt"], xt) and torch.equal(seen["t"], t)
    assert torch.allclose(terms["loss"], (ut ** 2).mean(dim=(1, 2, 3)))


@pytest.mark.parametrize("method,nfe_per_step,order", [("euler", 1, 1), ("heun2", 2, 2), ("rk4", 4, 4)])
def test_fixed_grid_steppers(method, nfe_per_step, order):
    """dx/dt = -x: x(1) = x(0) e^-1; NFE = (num_steps - 1) * nfe_per_step; error shrinks at the method's order."""
    calls = []

    def model(x, t, y=None):
        calls.append(float(t[0]))
        assert t.shape == (x.shape[0],)
        return -x

    sampler = Sampler(create_transport("Linear", "velocity"))
    x0 = torch.ones(3, 2, dtype=torch.float64)
    errs = []
    for n in (11, 21):
        calls.clear()
        traj = sampler.sample_ode(sampling_method=method, num_steps=n)(x0, model, y=None)
        assert traj.shape[0] == n and len(calls) == (n - 1) * nfe_per_step
        errs.append(abs(traj[-1][0, 0].item() - math.exp(-1)))
    assert errs[1] < errs[0] / (2 ** order) * 1.3
    if method == "euler":
        assert calls[0] == 0.0 and abs(calls[-1] - 0.95) < 1e-6          # evaluated at the LEFT end of every interval


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dimsum_amd.sample_ddp import sample_batch, shard_range
    lo, hi = shard_range(total, rank, world)
    torch.manual_seed(0 * world + rank)                                   # sample_ddp.py:64 seeding rule
    z = torch.randn(4, 2, 4, 4)
    y = torch.full((4,), rank)

    class Field(torch.nn.Module):                                         # analytic velocity field, label dependent
        in_channels, num_classes = 2, 10

        def forward(self, x, t, y=None):
            return -x * (1 + y.view(-1, 1, 1, 1).float())

    full = sample_batch(Field(), z, y, num_steps=20, world_size=world)
    local = sample_batch(Field(), z, y, num_steps=20, world_size=world, gather=False)
    out_q.put((rank, lo, hi, full, local))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_all_gather():
    world, total = 2, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, full0, loc0), (r1, lo1, hi1, full1, loc1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 5, 5, 9)                           # contiguous, disjoint, covering
    assert torch.equal(full0, full1) and full0.shape == (8, 2, 4, 4)      # every rank holds the same gathered tensor
    assert torch.equal(full0[:4], loc0) and torch.equal(full0[4:], loc1)  # rank-major order
    assert not torch.equal(loc0, loc1)                                    # rank-dependent seeds / labels


def test_dopri5_adaptive():
    """dx/dt = -x from 1: the dense output on the grid matches e^-t within the tolerances; tighter tolerances cost more
    function evaluations; every call passes t as ones(B) * t."""
    nfes = []
    for rtol, atol, bound in ((1e-3, 1e-6, 2e-3), (1e-7, 1e-9, 1e-6)):
        def model(x, t, y=None):
            assert t.shape == (x.shape[0],)
            return -x
        sampler = Sampler(create_transport("Linear", "velocity"))
        fn_owner = sampler.sample_ode(sampling_method="dopri5", num_steps=11, atol=atol, rtol=rtol)
        x0 = torch.ones(3, 2, dtype=torch.float64)
        traj = fn_owner(x0, model, y=None)
        assert traj.shape == (11, 3, 2)
        ts = torch.linspace(0, 1, 11, dtype=torch.float64)
        err = (traj[:, 0, 0] - torch.exp(-ts)).abs().max().item()
        assert err < bound, err
        nfes.append(fn_owner.__self__.last_nfe)
    assert nfes[0] < nfes[1] and nfes[0] >= 8            # 2 for the initial step + 6 per step


def test_dopri5_matches_rk4_on_a_nonlinear_field():
    def model(x, t, y=None):
        return torch.sin(3 * t).view(-1, 1) * x - 0.5 * x ** 3
    sampler = Sampler(create_transport("GVP", "velocity"))
    x0 = torch.linspace(-1, 1, 8, dtype=torch.float64).view(4, 2)
    ref = sampler.sample_ode(sampling_method="rk4", num_steps=2001)(x0, model, return_trajectory=False)
    got = sampler.sample_ode(sampling_method="dopri5", num_steps=2, atol=1e-9, rtol=1e-8)(x0, model, return_trajectory=False)
    assert torch.allclose(got, ref, atol=1e-7)


@pytest.mark.parametrize("method,last_step", [("Euler", "Mean"), ("Heun", "Mean"), ("Euler", "Tweedie"), ("Euler", "Euler")])
def test_sde_sampler_shapes_and_zero_noise_limit(method, last_step):
    """structure of sample_sde (transport.py:286-341): number of returned states, Heun halves the grid, and with the
    diffusion switched off (form 'none') the Euler-Maruyama path equals the fixed-grid Euler ODE path."""
    torch.manual_seed(0)
    sampler = Sampler(create_transport("Linear", "velocity"))

    def model(x, t, y=None):
        return -x

    x0 = torch.randn(5, 3)
    n = 20
    xs = sampler.sample_sde(sampling_method=method, diffusion_form="sigma", last_step=last_step, last_step_size=0.04, num_steps=n)(x0, model)
    assert len(xs) == (n if method == "Euler" else n // 2) and xs[-1].shape == x0.shape and torch.isfinite(xs[-1]).all()
    if method == "Euler" and last_step == "Euler":
        det = sampler.sample_sde(sampling_method="Euler", diffusion_form="none", last_step=None, num_steps=n)(x0, model)
        ode_traj = sampler.sample_ode(sampling_method="euler", num_steps=n)(x0, model)
        assert torch.allclose(det[-2], ode_traj[-1], atol=1e-6)


def test_dopri5_accepted_steps_never_shrink():
    """torchdiffeq 0.2.3 `_optimal_step_size`: for an accepted step (error ratio < 1) dfactor becomes 1, so the factor is
    min(ifactor, max(1, safety * ratio^-1/5)) -- h never shrinks after an acceptance. Hence every shrink in the sequence of
    attempted step sizes comes from a rejection (ratio > 1), whose factor lies in [dfactor, safety) = [0.2, 0.9)."""
    from dimsum_amd.transport.integrators import _Dopri5
    hs = []

    class Spy(_Dopri5):
        def _step(self, t, h, x, f0):
            hs.append(h)
            return super()._step(t, h, x, f0)

    f = lambda t, x: torch.stack([x[1], -25.0 * x[0]]) * (1 + 2 * t)       # oscillator speeding up: forces step changes
    Spy(f, atol=1e-7, rtol=1e-5).integrate(torch.tensor([1.0, 0.0], dtype=torch.float64), [0.0, 1.0], False)
    assert len(hs) > 10
    shrank = [(a, b) for a, b in zip(hs[:-1], hs[1:]) if b < a]
    assert all(0.2 - 1e-12 <= b / a < 0.9 for a, b in shrank), shrank
    assert any(b >= a for a, b in zip(hs[:-1], hs[1:]))


def test_dopri5_tuple_state_uses_the_mixed_norm():
    """torchdiffeq 0.2.3 `_check_inputs`: a tuple state is integrated flat but with `_mixed_norm` = max over the components' RMS norms.
    A state (x: 4096 slowly varying values, delta_logp: 1 fast-varying value): with ONE flat RMS the fast component's error is diluted
    by 1 / sqrt(4097) and the controller takes far larger steps; with the mixed norm it alone rejects them. Checked: (a) the norm itself,
    (b) the tuple path takes many more steps than the diluted norm would and tracks the fast component to its tolerance,
    (c) single-tensor states keep the plain RMS."""
    from dimsum_amd.transport.integrators import _Dopri5, _rms, ode
    v = torch.cat([torch.full((4096,), 1e-3), torch.tensor([5.0])])
    mixed = _Dopri5(lambda t, x: x, 1e-6, 1e-3, split_sizes=[4096, 1]).norm(v)
    assert abs(float(mixed) - 5.0) < 1e-6 and abs(float(_rms(v)) - (4096e-6 / 4097 + 25.0 / 4097) ** 0.5) < 1e-6

    def drift(state, t, model, **kw):          # x' = -0.01 x (slow), logp' = 40 cos(40 t) (fast: logp(t) = sin(40 t))
        x, lp = state
        return (-0.01 * x, 40.0 * torch.cos(40.0 * t[:1]).expand_as(lp))

    x0 = (torch.ones(1, 4096, dtype=torch.float64), torch.zeros(1, dtype=torch.float64))
    solver = ode(drift, t0=0.0, t1=1.0, sampler_type="dopri5", num_steps=2, atol=1e-7, rtol=1e-6)
    x1, lp1 = solver.sample(x0, None, return_trajectory=False)
    nfe_mixed = solver.last_nfe
    assert abs(float(lp1) - torch.sin(torch.tensor(40.0, dtype=torch.float64)).item()) < 1e-5

    def flat_drift(t, v):
        return torch.cat([-0.01 * v[:4096], 40.0 * torch.cos(torch.tensor(40.0 * t, dtype=torch.float64)).reshape(1)])
    diluted = _Dopri5(flat_drift, 1e-7, 1e-6)                                   # the same system under ONE flat RMS
    out = diluted.integrate(torch.cat([x0[0].reshape(-1), x0[1]]), [0.0, 1.0], False)
    assert nfe_mixed > 1.5 * diluted.nfe, (nfe_mixed, diluted.nfe)
    assert abs(float(lp1) - torch.sin(torch.tensor(40.0, dtype=torch.float64)).item()) < abs(float(out[-1]) - torch.sin(torch.tensor(40.0, dtype=torch.float64)).item())
    assert _Dopri5(lambda t, x: x, 1e-6, 1e-3).norm is _rms


def test_dopri5_tableau_against_scipy_rk45():
    """The Dormand-Prince 5(4) coefficients of the dopri5 restatement (torchdiffeq is absent: parity with IT stays unpinned)
    against an independent published implementation, scipy's RK45: nodes c, stage matrix A and the 5th-order weights b are the
    same numbers; the error weights are proportional -- torchdiffeq uses Shampine's embedded 4th-order weights
    (b4 = 1951/21600, 0, 22642/50085, 451/720, -12231/42400, 649/6300, 1/60), whose difference to b5 is -2/3 of the classic
    pair's that scipy carries."""
    import numpy as np
    from scipy.integrate._ivp.rk import RK45
    from dimsum_amd.transport import integrators as I
    assert np.allclose(np.asarray(I._DP_C[:5]), RK45.C[1:], rtol=0, atol=1e-15)
    for i, row in enumerate(I._DP_A[:5]):                      # stages 2..6
        assert np.allclose(np.asarray(row), RK45.A[i + 1, :len(row)], rtol=0, atol=1e-15), i
    assert np.allclose(np.asarray(I._DP_A[5][:6]), RK45.B, rtol=0, atol=1e-15)       # 7th stage = the 5th-order solution (FSAL)
    assert np.allclose(np.asarray(I._DP_B[:6]), RK45.B, rtol=0, atol=1e-15) and I._DP_B[6] == 0.0
    assert np.allclose(np.asarray(I._DP_E), -2.0 / 3.0 * RK45.E, rtol=0, atol=1e-15)
