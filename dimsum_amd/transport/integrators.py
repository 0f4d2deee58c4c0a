"""Fixed-grid ODE integrators for the probability-flow ODE dx/dt = f(t, x).

The reference delegates to torchdiffeq 0.2.3 `odeint(_fn, x, t, method=...)` (dimsum/transport/integrators.py:98-111);
torchdiffeq is a third-party dependency that is not part of the reference tree, so its stepping is restated here
from its published fixed-grid semantics: on the grid t = linspace(t0, t1, num_steps) each solver does one step per
interval, i.e. `num_steps - 1` steps -- "euler": x += dt f(t, x) (1 NFE/step), "midpoint" (2), "heun2" (2), "rk4" (4,
torchdiffeq's 3/8 rule), and the adaptive "dopri5" of the published eval recipe (scripts/eval.sh:73-95: atol 1e-6,
rtol 1e-3): Dormand-Prince 5(4) with torchdiffeq's controller (mixed-norm error ratio, step factor
clamp(0.9 ratio^-1/5, 0.2, 10), its initial-step heuristic) and its quartic dense output evaluated on the output grid.
PARITY UNPINNED: no reference test stores sampler outputs.

The stochastic samplers (`sde`: Euler-Maruyama and Heun on a fixed grid) restate dimsum/transport/integrators.py:5-73."""
import torch as th


def _euler(f, t, dt, x):
    return x + dt * f(t, x)


def _midpoint(f, t, dt, x):
    return x + dt * f(t + 0.5 * dt, x + 0.5 * dt * f(t, x))


def _heun2(f, t, dt, x):
    k1 = f(t, x)
    return x + 0.5 * dt * (k1 + f(t + dt, x + dt * k1))


def _rk4(f, t, dt, x):      # 3/8 rule, as torchdiffeq's fixed-grid rk4
    k1 = f(t, x)
    k2 = f(t + dt / 3, x + dt * k1 / 3)
    k3 = f(t + dt * 2 / 3, x + dt * (k2 - k1 / 3))
    k4 = f(t + dt, x + dt * (k1 - k2 + k3))
    return x + dt * (k1 + 3 * (k2 + k3) + k4) / 8


_STEPPERS = {"euler": (_euler, 1), "midpoint": (_midpoint, 2), "heun2": (_heun2, 2), "rk4": (_rk4, 4)}

# Dormand-Prince 5(4) tableau (the published coefficients; torchdiffeq _DORMAND_PRINCE_SHAMPINE_TABLEAU)
_DP_C = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_A = ((1 / 5,),
         (3 / 40, 9 / 40),
         (44 / 45, -56 / 15, 32 / 9),
         (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
         (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_B = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
_DP_E = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
         11 / 84 - 649 / 6300, -1 / 60)
_DP_MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


def _rms(x):
    return x.abs().pow(2).mean().sqrt()


class _Dopri5:
    """adaptive RK45 over the whole batch as ONE system (like odeint on a stacked state): the error norm is the RMS over
    all elements, so the step sequence -- and the number of function evaluations -- is shared by every latent.
    split_sizes: the flat state is a flattened TUPLE with these component sizes; the norm is then torchdiffeq 0.2.3's `_mixed_norm`
    (odeint's `_check_inputs` installs it for tuple input): the MAX over the components' RMS norms -- so that a small component (the
    likelihood ODE's B values of delta_logp next to 4096 B of x) keeps its own say in the step-size control."""

    def __init__(self, f, atol, rtol, safety=0.9, ifactor=10.0, dfactor=0.2, max_steps=2 ** 31 - 1, split_sizes=None):
        self.f, self.atol, self.rtol = f, atol, rtol
        self.safety, self.ifactor, self.dfactor, self.max_steps = safety, ifactor, dfactor, max_steps
        self.nfe = 0
        self.norm = _rms if not split_sizes else (lambda v: max(_rms(c) for c in v.split(list(split_sizes), dim=-1)))

    def _f(self, t, x):
        self.nfe += 1
        return self.f(t, x)

    def _initial_step(self, t0, x0, f0):
        scale = self.atol + x0.abs() * self.rtol
        d0, d1 = self.norm(x0 / scale), self.norm(f0 / scale)
        h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else float(0.01 * d0 / d1)
        f1 = self._f(t0 + h0, x0 + h0 * f0)
        d2 = float(self.norm((f1 - f0) / scale)) / h0
        h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(float(d1), d2)) ** (1.0 / 5)
        return min(100 * h0, h1)

    def _step(self, t, h, x, f0):
        k = [f0]
        for ci, row in zip(_DP_C, _DP_A):
            xi = x
            for a, kj in zip(row, k):
                if a != 0.0:
                    xi = xi + (h * a) * kj
            k.append(self._f(t + ci * h, xi))
        x1 = x
        for bcoef, kj in zip(_DP_B, k):
            if bcoef != 0.0:
                x1 = x1 + (h * bcoef) * kj
        err = sum((h * e) * kj for e, kj in zip(_DP_E, k) if e != 0.0)
        mid = x + sum((h * m) * kj for m, kj in zip(_DP_MID, k) if m != 0.0)
        return x1, k[-1], err, mid          # FSAL: k[-1] = f(t + h, x1)

    @staticmethod
    def _interp(x0, x1, mid, f0, f1, h, theta):
        """quartic through (x0, f0), mid, (x1, f1) at theta in [0, 1] (torchdiffeq _interp_fit / _interp_evaluate)"""
        a = 2 * h * (f1 - f0) - 8 * (x1 + x0) + 16 * mid
        b = h * (5 * f0 - 3 * f1) + 18 * x0 + 14 * x1 - 32 * mid
        c = h * (f1 - 4 * f0) - 11 * x0 - 5 * x1 + 16 * mid
        d = h * f0
        return (((a * theta + b) * theta + c) * theta + d) * theta + x0

    def integrate(self, x, ts, return_trajectory):
        t, out = ts[0], [x]
        f0 = self._f(t, x)
        h = self._initial_step(t, x, f0)
        nxt, steps = 1, 0
        while nxt < len(ts):
            assert steps < self.max_steps, "dopri5: max_num_steps exceeded"
            steps += 1
            x1, f1, err, mid = self._step(t, h, x, f0)
            tol = self.atol + self.rtol * th.maximum(x.abs(), x1.abs())
            ratio = float(self.norm(err / tol))
            if ratio <= 1.0:          # accept
                while nxt < len(ts) and ts[nxt] <= t + h:
                    theta = (ts[nxt] - t) / h
                    out.append(x1 if theta >= 1.0 else self._interp(x, x1, mid, f0, f1, h, theta))
                    nxt += 1
                t, x, f0 = t + h, x1, f1
            # torchdiffeq 0.2.3 _optimal_step_size: an accepted step (ratio < 1) never shrinks h (its dfactor becomes 1)
            if ratio == 0.0:
                factor = self.ifactor
            else:
                lo = 1.0 if ratio < 1.0 else self.dfactor
                factor = min(self.ifactor, max(lo, self.safety * ratio ** -0.2))
            h = h * factor
        return th.stack(out) if return_trajectory else out[-1]


class ode:
    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol=1e-6, rtol=1e-3):
        assert t0 < t1, "ODE sampler has to be in forward time"
        self.method = sampler_type.lower()
        if self.method not in _STEPPERS and self.method != "dopri5":
            raise NotImplementedError(f"sampling_method={sampler_type!r}: fixed-grid {sorted(_STEPPERS)} and adaptive dopri5 are implemented")
        self.drift, self.t, self.atol, self.rtol = drift, th.linspace(t0, t1, num_steps), atol, rtol
        self.stepper, self.nfe_per_step = _STEPPERS.get(self.method, (None, None))
        self.last_nfe = None      # function evaluations of the last sample() call

    @property
    def nfe(self):
        """function evaluations per sample() call (known in advance only for the fixed-grid methods)"""
        return (len(self.t) - 1) * self.nfe_per_step if self.stepper is not None else self.last_nfe

    def sample(self, x, model, return_trajectory=True, **model_kwargs):
        """-> stacked states at every grid point like odeint (index [-1] = the sample), or only the last state.
        A TUPLE state (the likelihood ODE's (x, delta_logp)) is integrated as one flat system, the way torchdiffeq handles
        tuples; dopri5's error norm is then the maximum over the components' RMS norms (torchdiffeq 0.2.3 `_mixed_norm`), and the state
        comes back as a tuple."""
        if isinstance(x, tuple):
            shapes, sizes = [c.shape for c in x], [c.numel() for c in x]
            batch, dev = x[0].size(0), x[0].device

            def unpack(v, lead=()):
                return tuple(p.reshape(lead + tuple(sh)) for p, sh in zip(v.split(sizes, dim=-1), shapes))

            def flat_drift(v, t, mdl, **kw):          # t arrives as ones(1) * t (the flat state has "batch" 1)
                out = self.drift(unpack(v[0]), th.ones(batch, device=dev) * t[0], mdl, **kw)
                return th.cat([o.reshape(-1) for o in out])[None]

            inner = ode(flat_drift, t0=0.0, t1=1.0, sampler_type=self.method, num_steps=2, atol=self.atol, rtol=self.rtol)
            inner.t = self.t
            inner._split_sizes = sizes
            out = inner.sample(th.cat([c.reshape(-1) for c in x])[None], model, return_trajectory=return_trajectory, **model_kwargs)
            self.last_nfe = inner.last_nfe
            return unpack(out[:, 0], (out.shape[0],)) if return_trajectory else unpack(out[0])
        ones = th.ones(x.size(0), device=x.device)

        def f(t, x):
            return self.drift(x, ones * t, model, **model_kwargs)        # t passed as ones(B) * t (integrators.py:103)

        ts = self.t.tolist()
        if self.stepper is None:
            solver = _Dopri5(f, self.atol, self.rtol, split_sizes=getattr(self, "_split_sizes", None))
            out = solver.integrate(x, ts, return_trajectory)
            self.last_nfe = solver.nfe
            return out
        self.last_nfe = self.nfe
        traj = [x] if return_trajectory else None
        for t_a, t_b in zip(ts[:-1], ts[1:]):
            x = self.stepper(f, t_a, t_b - t_a, x)
            if return_trajectory:
                traj.append(x)
        return th.stack(traj) if return_trajectory else x


class sde:
    """fixed-grid SDE solvers on t = linspace(t0, t1, num_steps): Euler-Maruyama (1 NFE / step) and Heun (2), the noise
    increment is sqrt(dt) N(0, I) and the diffusion enters as sqrt(2 w(t)) (integrators.py:5-73)."""

    def __init__(self, drift, diffusion, *, t0, t1, num_steps, sampler_type):
        assert t0 < t1, "SDE sampler has to be in forward time"
        if sampler_type not in ("Euler", "Heun"):
            raise NotImplementedError(f"SDE sampler {sampler_type!r} not implemented (Euler, Heun)")
        self.t = th.linspace(t0, t1, num_steps)
        self.dt = float(self.t[1] - self.t[0])
        self.drift, self.diffusion, self.sampler_type = drift, diffusion, sampler_type

    @staticmethod
    def _noise(x):
        """drawn from the CPU generator and moved to x's device, like the reference (integrators.py:28,37
        `th.randn(x.size()).to(x)`): the same `torch.manual_seed` gives the same samples on every device"""
        return th.randn(x.size()).to(x)

    def _euler_maruyama(self, x, t, model, **kw):
        dw = self._noise(x) * self.dt ** 0.5
        tt = th.ones(x.size(0), device=x.device, dtype=x.dtype) * t
        mean_x = x + self.drift(x, tt, model, **kw) * self.dt
        return mean_x + th.sqrt(2 * self.diffusion(x, tt)) * dw

    def _heun(self, x, t, model, **kw):
        dw = self._noise(x) * self.dt ** 0.5
        tt = th.ones(x.size(0), device=x.device, dtype=x.dtype) * t
        xhat = x + th.sqrt(2 * self.diffusion(x, tt)) * dw
        k1 = self.drift(xhat, tt, model, **kw)
        k2 = self.drift(xhat + self.dt * k1, tt + self.dt, model, **kw)
        return xhat + 0.5 * self.dt * (k1 + k2)

    @th.no_grad()
    def sample(self, init, model, **model_kwargs):
        """-> list of the states after every step (len(t) - 1 entries)"""
        step = self._euler_maruyama if self.sampler_type == "Euler" else self._heun
        x, out = init, []
        for ti in self.t[:-1].tolist():
            x = step(x, ti, model, **model_kwargs)
            out.append(x)
        return out
