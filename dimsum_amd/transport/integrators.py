"""Fixed-grid ODE integrators for the probability-flow ODE dx/dt = f(t, x).

The reference delegates to torchdiffeq 0.2.3 `odeint(_fn, x, t, method=...)` (dimsum/transport/integrators.py:98-111);
torchdiffeq is a third-party dependency that is not part of the reference tree, so its stepping is restated here
from its published fixed-grid semantics: on the grid t = linspace(t0, t1, num_steps) each solver does one step per
interval, i.e. `num_steps - 1` steps -- "euler": x += dt f(t, x) (1 NFE/step), "midpoint" (2), "heun2" (2), "rk4" (4,
torchdiffeq's 3/8 rule). PARITY UNPINNED: no reference test stores sampler outputs; the adaptive
"dopri5" of the published eval recipe is not implemented yet."""
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


class ode:
    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol=1e-6, rtol=1e-3):
        assert t0 < t1, "ODE sampler has to be in forward time"
        if sampler_type.lower() not in _STEPPERS:
            raise NotImplementedError(f"sampling_method={sampler_type!r}: fixed-grid {sorted(_STEPPERS)} are implemented")
        self.drift, self.t = drift, th.linspace(t0, t1, num_steps)
        self.stepper, self.nfe_per_step = _STEPPERS[sampler_type.lower()]

    @property
    def nfe(self):
        return (len(self.t) - 1) * self.nfe_per_step

    def sample(self, x, model, return_trajectory=True, **model_kwargs):
        """-> stacked states at every grid point like odeint (index [-1] = the sample), or only the last state."""
        ones = th.ones(x.size(0), device=x.device)

        def f(t, x):
            return self.drift(x, ones * t, model, **model_kwargs)        # t passed as ones(B) * t (integrators.py:103)

        ts = self.t.tolist()
        traj = [x] if return_trajectory else None
        for t_a, t_b in zip(ts[:-1], ts[1:]):
            x = self.stepper(f, t_a, t_b - t_a, x)
            if return_trajectory:
                traj.append(x)
        return th.stack(traj) if return_trajectory else x
