"""SDE definitions of the sampling path (mirror of reference ``score_sde_pytorch/sde_lib.py``).

Scalars and (B,)-shaped schedule values are computed with torch exactly as the reference does
(they are a few floats per step); everything that touches the (B, C, L, L) state runs in the HIP
kernels of libt2p_hip.so (``sampling.py`` here).  Names, constructor arguments and attributes
follow the reference so that callers can switch imports:

  VESDE   sde_lib.py:199-245     VPSDE   sde_lib.py:106-157     subVPSDE   sde_lib.py:160-196
"""
from __future__ import annotations

import abc

import numpy as np
import torch


class SDE(abc.ABC):
    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        ...

    @abc.abstractmethod
    def marginal_prob_std(self, t):
        """std of p_t(x | x_0) for a (B,) tensor of times."""

    @abc.abstractmethod
    def prior_scale(self):
        """x_T = randn * prior_scale()."""

    def prior_sampling(self, shape):
        return torch.randn(*shape) * self.prior_scale()


class VESDE(SDE):
    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000):
        super().__init__(N)
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        # float32, ascending (sde_lib.py:210)
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(self.sigma_min), np.log(self.sigma_max), N))

    @property
    def T(self):
        return 1

    def marginal_prob_std(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def prior_scale(self):
        return self.sigma_max

    def discretize_coeffs(self, t):
        """(drift scale a, G) such that f = a * x; VE: f = 0, G = sqrt(sigma_k^2 - sigma_{k-1}^2)
        with the truncating index of sde_lib.py:237-245."""
        t = t.detach().cpu()
        timestep = (t * (self.N - 1) / self.T).long()
        sigma = self.discrete_sigmas[timestep]
        adjacent = torch.where(timestep == 0, torch.zeros_like(t), self.discrete_sigmas[timestep - 1])
        return torch.zeros_like(t), torch.sqrt(sigma ** 2 - adjacent ** 2)

    def g_table(self, eps):
        """G of loop step i for the reference's ``timesteps = linspace(T, eps, N)`` (sampling.py:257)."""
        ts = torch.linspace(self.T, eps, self.N)
        return self.discretize_coeffs(ts)[1].float().contiguous()

    def label_table(self, eps):
        """Time label the score network receives at loop step i: ``round((T - t_i) (N - 1))`` in the
        reference's float32 arithmetic (models/utils.py:159-171 on ``linspace(T, eps, N)``).  It equals
        ``i`` only for tiny eps: with get_pc_sampler's own default eps = 1e-3 and N = 1000, 499 labels differ."""
        ts = torch.linspace(self.T, eps, self.N)
        return torch.round((self.T - ts) * (self.N - 1)).long().to(torch.int32).contiguous()


class VPSDE(SDE):
    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.discrete_betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    @property
    def T(self):
        return 1

    def marginal_prob_std(self, t):
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return torch.sqrt(1.0 - torch.exp(2.0 * log_mean_coeff))

    def prior_scale(self):
        return 1.0

    def discretize_coeffs(self, t):
        """DDPM discretisation (sde_lib.py:148-157): f = (sqrt(alpha) - 1) x, G = sqrt(beta)."""
        t = t.detach().cpu()
        timestep = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas[timestep]
        alpha = self.alphas[timestep]
        return torch.sqrt(alpha) - 1.0, torch.sqrt(beta)


    # ---- per-step tables of the fused sampler (loop step i of ``timesteps = linspace(T, eps, N)``, sampling.py:257) ----
    def _steps(self, eps):
        ts = torch.linspace(self.T, eps, self.N)
        return ts, (ts * (self.N - 1) / self.T).long()

    def g_table(self, eps):
        """G = sqrt(beta_k) of loop step i (sde_lib.py:148-157)."""
        ts, _ = self._steps(eps)
        return self.discretize_coeffs(ts)[1].float().contiguous()

    def label_table(self, eps):
        """Integer part of the time label of loop step i (index of the std / sigma tables, models/utils.py:150-153)."""
        ts, _ = self._steps(eps)
        return (ts * (self.N - 1)).long().to(torch.int32).contiguous()

    def vp_tables(self, eps):
        """(label_f, score_scale, x_coef, corr_alpha), float32[N] each: the fractional label ``t (N - 1)`` the network embeds,
        ``-1 / sqrt_1m_alphas_cumprod[label]`` (models/utils.py:150-157), ``2 - sqrt(alpha_k)`` (x - f, sde_lib.py:148-157) and
        ``alpha_k`` (sampling.py:184-186), all in the reference's float32 arithmetic."""
        ts, k = self._steps(eps)
        label_f = ts * (self.N - 1)
        std = self.sqrt_1m_alphas_cumprod[label_f.long()]
        a, _ = self.discretize_coeffs(ts)
        return tuple(t.float().contiguous() for t in (label_f, -1.0 / std, 1.0 - a, self.alphas[k]))


class subVPSDE(SDE):
    """Present for the config surface only: the reference's ``subVPSDE.sde`` does not accept the
    ``context`` argument its callers pass (sde_lib.py:177 vs :89), so it cannot be sampled there
    either, and the driver never selects it (sampling_6d.py:76-82)."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000):
        super().__init__(N)
        self.beta_0 = beta_min
        self.beta_1 = beta_max

    @property
    def T(self):
        return 1

    def marginal_prob_std(self, t):
        log_mean_coeff = -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0
        return 1 - torch.exp(2.0 * log_mean_coeff)

    def prior_scale(self):
        return 1.0
