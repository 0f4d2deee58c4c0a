"""Interpolation plans x_t = alpha_t x1 + sigma_t x0 of the flow-matching transport (dimsum/transport/path.py:21-246):
Linear (ICPlan), GVP (sin/cos) and VP. t runs from noise (0) to data (1)."""
import math

import torch as th


def expand_t_like_x(t, x):
    """(B,) -> (B, 1, ..., 1)"""
    return t.view(t.size(0), *([1] * (x.dim() - 1)))


class ICPlan:
    """alpha = t, sigma = 1 - t"""

    def __init__(self, sigma=0.0, **_):
        self.sigma = sigma

    def compute_alpha_t(self, t):
        return t, 1

    def compute_sigma_t(self, t):
        return 1 - t, -1

    def compute_d_alpha_alpha_ratio_t(self, t):
        return 1 / t

    def compute_drift(self, x, t):
        """SDE drift / diffusion in score parametrisation (path.py:43-51)"""
        t = expand_t_like_x(t, x)
        ratio = self.compute_d_alpha_alpha_ratio_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        return -(ratio * x), ratio * (sigma_t ** 2) - sigma_t * d_sigma_t

    def compute_diffusion(self, x, t, form="constant", norm=1.0):
        """diffusion coefficient w(t) of the sampling SDE (path.py:52-76); only the requested form is evaluated"""
        t = expand_t_like_x(t, x)
        if form == "none":
            return th.zeros((1,), device=t.device)
        if form == "constant":
            return th.full((1,), norm, device=t.device)
        if form == "SBDM":
            return norm * 2.0 * self.compute_drift(x, t.reshape(t.shape[0]))[1]
        if form == "sigma":
            return norm * self.compute_sigma_t(t)[0]
        if form == "linear":
            return norm * (1 - t)
        if form == "decreasing":
            return 0.25 * (norm * th.cos(math.pi * t) + 1) ** 2
        if form == "increasing-decreasing":
            return norm * th.sin(math.pi * t) ** 2
        if form == "log":
            return norm * th.log(t - t ** 2 + 1)
        raise NotImplementedError(f"Diffusion form {form} not implemented")

    def get_score_from_velocity(self, velocity, x, t):
        t = expand_t_like_x(t, x)
        alpha_t, d_alpha_t = self.compute_alpha_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        reverse_alpha_ratio = alpha_t / d_alpha_t
        var = sigma_t ** 2 - reverse_alpha_ratio * d_sigma_t * sigma_t
        return (reverse_alpha_ratio * velocity - x) / var

    def compute_mu_t(self, t, x0, x1):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[0] * x1 + self.compute_sigma_t(t)[0] * x0

    compute_xt = compute_mu_t

    def compute_ut(self, t, x0, x1, xt):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[1] * x1 + self.compute_sigma_t(t)[1] * x0

    def plan(self, t, x0, x1):
        xt = self.compute_xt(t, x0, x1)
        return t, xt, self.compute_ut(t, x0, x1, xt)


class GVPCPlan(ICPlan):
    """alpha = sin(pi t / 2), sigma = cos(pi t / 2)   (path.py:228-246)"""

    def compute_alpha_t(self, t):
        return th.sin(t * math.pi / 2), math.pi / 2 * th.cos(t * math.pi / 2)

    def compute_sigma_t(self, t):
        return th.cos(t * math.pi / 2), -math.pi / 2 * th.sin(t * math.pi / 2)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return math.pi / (2 * th.tan(t * math.pi / 2))


class VPCPlan(ICPlan):
    """variance-preserving path (path.py:191-225)"""

    def __init__(self, sigma_min=0.1, sigma_max=20.0, **kw):
        super().__init__(**kw)
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def _log_mean(self, t):
        return -0.25 * ((1 - t) ** 2) * (self.sigma_max - self.sigma_min) - 0.5 * (1 - t) * self.sigma_min

    def _d_log_mean(self, t):
        return 0.5 * (1 - t) * (self.sigma_max - self.sigma_min) + 0.5 * self.sigma_min

    def compute_alpha_t(self, t):
        a = th.exp(self._log_mean(t))
        return a, a * self._d_log_mean(t)

    def compute_sigma_t(self, t):
        p = 2 * self._log_mean(t)
        s = th.sqrt(1 - th.exp(p))
        return s, th.exp(p) * (2 * self._d_log_mean(t)) / (-2 * s)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return self._d_log_mean(t)

    def compute_drift(self, x, t):
        t = expand_t_like_x(t, x)
        beta = self.sigma_min + (1 - t) * (self.sigma_max - self.sigma_min)
        return -0.5 * beta * x, beta / 2
