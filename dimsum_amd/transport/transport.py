"""Flow-matching transport: training loss, ODE / SDE samplers and the likelihood ODE around the denoiser
(dimsum/transport/transport.py). The DCT blurring of the path (never enabled by the published configs) is left out."""
import enum
import math

import torch as th

from . import path
from .integrators import ode, sde


class ModelType(enum.Enum):
    NOISE = enum.auto()
    SCORE = enum.auto()
    VELOCITY = enum.auto()


class PathType(enum.Enum):
    LINEAR = enum.auto()
    GVP = enum.auto()
    VP = enum.auto()


class WeightType(enum.Enum):
    NONE = enum.auto()
    VELOCITY = enum.auto()
    LIKELIHOOD = enum.auto()


def mean_flat(x):
    return th.mean(x, dim=list(range(1, x.dim())))


class Transport:
    def __init__(self, *, model_type, path_type, loss_type, train_eps, sample_eps, path_args={}, t_sample_mode="uniform"):
        plans = {PathType.LINEAR: path.ICPlan, PathType.GVP: path.GVPCPlan, PathType.VP: path.VPCPlan}
        self.loss_type, self.model_type = loss_type, model_type
        self.path_sampler = plans[path_type](**path_args)
        self.train_eps, self.sample_eps, self.t_sample_mode = train_eps, sample_eps, t_sample_mode

    def prior_logp(self, z):
        """log density of the standard normal prior, per batch element (transport.py:69-77)"""
        n = z[0].numel()
        return -n / 2.0 * math.log(2 * math.pi) - z.reshape(z.shape[0], -1).pow(2).sum(1) / 2.0

    def check_interval(self, train_eps, sample_eps, *, diffusion_form="SBDM", sde=False, reverse=False, eval=False, last_step_size=0.0):
        """integration interval (transport.py:79-107)"""
        t0, t1 = 0, 1
        eps = train_eps if not eval else sample_eps
        if type(self.path_sampler) is path.VPCPlan:
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        elif type(self.path_sampler) in (path.ICPlan, path.GVPCPlan) and (self.model_type != ModelType.VELOCITY or sde):
            t0 = eps if (diffusion_form == "SBDM" and sde) or self.model_type != ModelType.VELOCITY else 0
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        return (1 - t0, 1 - t1) if reverse else (t0, t1)

    def sample(self, x1):
        """x0 ~ N(0, I), t ~ U(t0, t1) or logit-normal (transport.py:109-125)"""
        x0 = th.randn_like(x1)
        t0, t1 = self.check_interval(self.train_eps, self.sample_eps)
        if self.t_sample_mode == "logitnormal":
            t = th.sigmoid(th.randn((x1.shape[0],)) - 0.5) * (t1 - t0) + t0
        else:
            t = th.rand((x1.shape[0],)) * (t1 - t0) + t0
        return t.to(x1), x0, x1

    def training_losses(self, model, x1, model_kwargs=None):
        """mean((model(x_t, t) - u_t)^2) for velocity prediction (transport.py:127-164)"""
        model_kwargs = model_kwargs or {}
        t, x0, x1 = self.sample(x1)
        t, xt, ut = self.path_sampler.plan(t, x0, x1)
        out = model(xt, t, **model_kwargs)
        assert out.size() == xt.size()
        terms = {"pred": out}
        if self.model_type == ModelType.VELOCITY:
            terms["loss"] = mean_flat((out - ut) ** 2)
            return terms
        _, drift_var = self.path_sampler.compute_drift(xt, t)
        sigma_t, _ = self.path_sampler.compute_sigma_t(path.expand_t_like_x(t, xt))
        weight = {WeightType.VELOCITY: (drift_var / sigma_t) ** 2, WeightType.LIKELIHOOD: drift_var / (sigma_t ** 2),
                  WeightType.NONE: 1}[self.loss_type]
        if self.model_type == ModelType.NOISE:
            terms["loss"] = mean_flat(weight * ((out - x0) ** 2))
        else:
            terms["loss"] = mean_flat(weight * ((out * sigma_t + x0) ** 2))
        return terms

    def get_drift(self):
        """drift of the probability-flow ODE (transport.py:166-197)"""
        def score_ode(x, t, model, **kw):
            mean, var = self.path_sampler.compute_drift(x, t)
            return -mean + var * model(x, t, **kw)

        def noise_ode(x, t, model, **kw):
            mean, var = self.path_sampler.compute_drift(x, t)
            sigma_t, _ = self.path_sampler.compute_sigma_t(path.expand_t_like_x(t, x))
            return -mean + var * (model(x, t, **kw) / -sigma_t)

        def velocity_ode(x, t, model, **kw):
            return model(x, t, **kw)

        fn = {ModelType.NOISE: noise_ode, ModelType.SCORE: score_ode, ModelType.VELOCITY: velocity_ode}[self.model_type]

        def body_fn(x, t, model, **kw):
            out = fn(x, t, model, **kw)
            assert out.shape == x.shape, "Output shape from ODE solver must match input shape"
            return out

        return body_fn


    def get_score(self):
        """score of x_t = alpha_t x + sigma_t eps from the model's prediction (transport.py:199-219)"""
        ps = self.path_sampler
        if self.model_type == ModelType.NOISE:
            return lambda x, t, model, **kw: model(x, t, **kw) / -ps.compute_sigma_t(path.expand_t_like_x(t, x))[0]
        if self.model_type == ModelType.SCORE:
            return lambda x, t, model, **kw: model(x, t, **kw)
        return lambda x, t, model, **kw: ps.get_score_from_velocity(model(x, t, **kw), x, t)


class Sampler:
    def __init__(self, transport):
        self.transport = transport
        self.drift = transport.get_drift()
        self.score = transport.get_score()

    def sample_ode(self, *, sampling_method="euler", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False):
        """-> sample_fn(x, model, **model_kwargs) returning the trajectory (transport.py:343-386)"""
        if reverse:
            drift = lambda x, t, model, **kw: self.drift(x, th.ones_like(t) * (1 - t), model, **kw)  # noqa: E731
        else:
            drift = self.drift
        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, sde=False, eval=True,
                                               reverse=reverse, last_step_size=0.0)
        return ode(drift=drift, t0=t0, t1=t1, sampler_type=sampling_method, num_steps=num_steps, atol=atol, rtol=rtol).sample

    def sample_ode_likelihood(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3):
        """-> fn(x, model, **model_kwargs) = (log p(x), z): integrates the probability-flow ODE from the data x (t = 1) back to the
        prior (t = 0) together with the change of log density, whose divergence term is Hutchinson's estimator with one
        Rademacher probe per drift evaluation (transport.py:388-443). The state is the pair (x, delta_logp), which the solver
        treats as ONE flat system (integrators.ode). One denoiser evaluation (forward + input gradient) per drift evaluation;
        the reference evaluates the model a second time for the same drift value."""
        def likelihood_drift(state, t, model, **kw):
            x, _ = state
            eps = th.randint(2, x.size(), dtype=th.float, device=x.device) * 2 - 1
            t = th.ones_like(t) * (1 - t)
            with th.enable_grad():
                x = x.detach().requires_grad_(True)
                d = self.drift(x, t, model, **kw)
                grad = th.autograd.grad(th.sum(d * eps), x)[0]
            return -d.detach(), th.sum(grad * eps, dim=tuple(range(1, x.dim())))

        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, sde=False, eval=True,
                                               reverse=False, last_step_size=0.0)
        solver = ode(drift=likelihood_drift, t0=t0, t1=t1, sampler_type=sampling_method, num_steps=num_steps, atol=atol, rtol=rtol)

        @th.no_grad()
        def _sample(x, model, **model_kwargs):
            z, delta_logp = solver.sample((x, th.zeros(x.size(0)).to(x)), model, return_trajectory=False, **model_kwargs)
            return self.transport.prior_logp(z) - delta_logp, z

        return _sample

    def sample_sde(self, *, sampling_method="Euler", diffusion_form="SBDM", diffusion_norm=1.0, last_step="Mean",
                   last_step_size=0.04, num_steps=250):
        """-> sample_fn(x, model, **model_kwargs) returning the list of states (transport.py:286-341).
        SDE: dx = [v + w(t) s] dt + sqrt(2 w(t)) dW, followed by one deterministic last step of `last_step_size`:
        None | "Mean" (drift only) | "Tweedie" (posterior mean) | "Euler" (probability-flow step).
        Note: one model call per drift evaluation would do; like the reference this evaluates the model twice (velocity
        and score) per drift."""
        num_steps = num_steps if sampling_method == "Euler" else num_steps // 2
        if last_step is None:
            last_step_size = 0.0
        elif last_step_size == -1:
            last_step_size = 1.0 / num_steps
        ps = self.transport.path_sampler

        def diffusion_fn(x, t):
            return ps.compute_diffusion(x, t, form=diffusion_form, norm=diffusion_norm)

        def sde_drift(x, t, model, **kw):
            return self.drift(x, t, model, **kw) + diffusion_fn(x, t) * self.score(x, t, model, **kw)

        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, diffusion_form=diffusion_form,
                                               sde=True, eval=True, reverse=False, last_step_size=last_step_size)
        solver = sde(sde_drift, diffusion_fn, t0=t0, t1=t1, num_steps=num_steps, sampler_type=sampling_method)

        if last_step is None:
            last = lambda x, t, model, **kw: x  # noqa: E731
        elif last_step == "Mean":
            last = lambda x, t, model, **kw: x + sde_drift(x, t, model, **kw) * last_step_size  # noqa: E731
        elif last_step == "Tweedie":
            def last(x, t, model, **kw):
                a, sg = ps.compute_alpha_t(t)[0][0], ps.compute_sigma_t(t)[0][0]
                return x / a + (sg ** 2) / a * self.score(x, t, model, **kw)
        elif last_step == "Euler":
            last = lambda x, t, model, **kw: x + self.drift(x, t, model, **kw) * last_step_size  # noqa: E731
        else:
            raise NotImplementedError(f"last_step={last_step!r}")

        def _sample(init, model, **model_kwargs):
            xs = solver.sample(init, model, **model_kwargs)
            ts = th.ones(init.size(0), device=init.device, dtype=init.dtype) * t1
            with th.no_grad():
                xs.append(last(xs[-1], ts, model, **model_kwargs))
            assert len(xs) == num_steps, "Samples does not match the number of steps"
            return xs

        return _sample
