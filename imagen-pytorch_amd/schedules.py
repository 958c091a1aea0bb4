"""Continuous-time Gaussian diffusion schedules (host side).

Mirror of the reference's `GaussianDiffusionContinuousTimes` (ip.py:212-318) for the sampling path: the
log-SNR schedules and the per-step scalar coefficients of the DDPM posterior.  These are O(T) scalars per
stage — they are evaluated once on the host (fp32 torch on CPU, exactly the reference's formulas) and
uploaded as a `[T, 8]` table that the sampler kernels index with the device step counter
("precompute per-step scalar table on host").
"""
from __future__ import annotations

import math

import torch
from torch import nn


def beta_linear_log_snr(t: torch.Tensor) -> torch.Tensor:
    """ip.py:212-214."""
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def alpha_cosine_log_snr(t: torch.Tensor, s: float = 0.008) -> torch.Tensor:
    """ip.py:216-218 (its log() clamps at 1e-5)."""
    return -torch.log(((torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** -2) - 1).clamp(min=1e-5))


def log_snr_to_alpha_sigma(log_snr: torch.Tensor):
    """ip.py:220-221."""
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


COEF_COLS = 8  # [alpha, sigma, alpha_next, sigma_next, c, nonzero, log_snr, 0] — layout shared with csrc/sampler.hip


class GaussianDiffusionContinuousTimes(nn.Module):
    def __init__(self, *, noise_schedule, timesteps=1000):
        super().__init__()
        if noise_schedule == "linear":
            self.log_snr = beta_linear_log_snr
        elif noise_schedule == "cosine":
            self.log_snr = alpha_cosine_log_snr
        else:
            raise ValueError(f'invalid noise schedule {noise_schedule}')
        self.noise_schedule = noise_schedule
        self.num_timesteps = timesteps

    def get_times(self, batch_size, noise_level, *, device=None):
        return torch.full((batch_size,), noise_level, device=device, dtype=torch.float32)

    def get_condition(self, times):
        return None if times is None else self.log_snr(times)

    def sample_random_times(self, batch_size, *, device=None):
        """ip.py:239-240."""
        return torch.rand((batch_size,), device=device, dtype=torch.float32)

    def get_sampling_timesteps(self, batch=None, *, device=None):
        """ip.py:245-250: T consecutive (t, t_next) pairs of linspace(1, 0, T+1).  With `batch` (the reference's call) every
        element is a (2, batch) tensor that unpacks into the per-sample `times, times_next`; without it (the coefficient tables
        below) the pairs are fp32 scalars."""
        times = torch.linspace(1., 0., self.num_timesteps + 1, device=device)
        if batch is None:
            return [(times[i], times[i + 1]) for i in range(self.num_timesteps)]
        pairs = torch.stack((times[:-1], times[1:]))                       # (2, T)
        return tuple(pairs[:, i, None].expand(2, batch) for i in range(self.num_timesteps))

    # ---- tensor forms of the reference's scheduler API (ip.py:252-318), for callers that drive single steps themselves
    # (Imagen.p_mean_variance / p_sample below); the sampling loop proper uses the coefficient tables instead
    @staticmethod
    def _per_sample(v: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        return v.reshape(v.shape + (1,) * (like.ndim - v.ndim)) if v.ndim < like.ndim else v

    def _times(self, t, like: torch.Tensor) -> torch.Tensor:
        if isinstance(t, (int, float)):
            t = torch.full((like.shape[0],), float(t), device=like.device, dtype=like.dtype)
        return t

    def q_posterior(self, x_start, x_t, t, *, t_next=None):
        """Posterior q(x_{t_next} | x_t, x_0): mean, variance and log(variance clamped at 1e-20) (ip.py:252-270)."""
        if t_next is None:
            t_next = (t - 1. / self.num_timesteps).clamp(min=0.)
        l, ln = self._per_sample(self.log_snr(t), x_t), self._per_sample(self.log_snr(t_next), x_t)
        alpha, _ = log_snr_to_alpha_sigma(l)
        alpha_n, sigma_n = log_snr_to_alpha_sigma(ln)
        c = -torch.special.expm1(l - ln)
        mean = alpha_n * (x_t * (1 - c) / alpha + c * x_start)
        var = (sigma_n ** 2) * c
        return mean, var, torch.log(var.clamp(min=1e-20))

    def q_sample(self, x_start, t, noise=None):
        """x_t = alpha_t x_0 + sigma_t eps; also returns log_snr (unpadded), alpha, sigma (ip.py:272-284)."""
        t = self._times(t, x_start)
        if noise is None:
            noise = torch.randn_like(x_start)
        l = self.log_snr(t).type(x_start.dtype)
        alpha, sigma = log_snr_to_alpha_sigma(self._per_sample(l, x_start))
        return alpha * x_start + sigma * noise, l, alpha, sigma

    def q_sample_from_to(self, x_from, from_t, to_t, noise=None):
        """Move a sample between noise levels (the inpainting re-noising, ip.py:286-307)."""
        from_t, to_t = self._times(from_t, x_from), self._times(to_t, x_from)
        if noise is None:
            noise = torch.randn_like(x_from)
        alpha, sigma = log_snr_to_alpha_sigma(self._per_sample(self.log_snr(from_t), x_from))
        alpha_to, sigma_to = log_snr_to_alpha_sigma(self._per_sample(self.log_snr(to_t), x_from))
        return x_from * (alpha_to / alpha) + noise * (sigma_to * alpha - sigma * alpha_to) / alpha

    def predict_start_from_v(self, x_t, t, v):
        """ip.py:309-313."""
        alpha, sigma = log_snr_to_alpha_sigma(self._per_sample(self.log_snr(t), x_t))
        return alpha * x_t - sigma * v

    def predict_start_from_noise(self, x_t, t, noise):
        """ip.py:315-318."""
        alpha, sigma = log_snr_to_alpha_sigma(self._per_sample(self.log_snr(t), x_t))
        return (x_t - sigma * noise) / alpha.clamp(min=1e-8)

    def step_coefficients(self) -> torch.Tensor:
        """[T, 8] fp32 table for the sampler kernels: ip.py:256-268 (posterior), 315-318 (x0), 2162-2163 (nonzero mask)."""
        rows = []
        for t, t_next in self.get_sampling_timesteps():
            l, ln = self.log_snr(t), self.log_snr(t_next)
            alpha, sigma = log_snr_to_alpha_sigma(l)
            alpha_n, sigma_n = log_snr_to_alpha_sigma(ln)
            c = -torch.special.expm1(l - ln)
            nonzero = 0.0 if float(t_next) == 0.0 else 1.0
            rows.append(torch.stack([alpha, sigma, alpha_n, sigma_n, c, torch.tensor(nonzero), l, torch.tensor(0.0)]))
        return torch.stack(rows).float().contiguous()

    def inpaint_coefficients(self, resample_times: int, philox: bool):
        """Tables for the inpainting resample loop (ip.py:2237-2275), one row per INNER iteration (timestep i, resample r = R-1..0):
        the step table with every row repeated R times, the blend weights of `img*~m + q_sample(known, t)*m` (w0 = alpha_t on the
        known image, sigma_t on the noise) and the re-noising weights of q_sample_from_to(x, t_next -> t) (ip.py:286-307), which are
        the identity on the last resample of a timestep and on the whole last timestep.  The noise weight sits in column 4 (the
        kernel's own Philox draw) when `philox`, else in column 1 (an injected noise image passed as t1)."""
        R = resample_times
        pairs = self.get_sampling_timesteps()
        blend, renoise = [], []
        nz = 4 if philox else 1
        for t, t_next in pairs:
            l, ln = self.log_snr(t), self.log_snr(t_next)
            alpha_t, sigma_t = log_snr_to_alpha_sigma(l)
            alpha_n, sigma_n = log_snr_to_alpha_sigma(ln)
            last_t = float(t_next) == 0.0
            for r in reversed(range(R)):
                b = torch.zeros(8)
                b[0], b[nz] = alpha_t, sigma_t
                blend.append(b)
                q = torch.zeros(8)
                if r == 0 or last_t:
                    q[0] = 1.0
                else:
                    q[0], q[nz] = alpha_t / alpha_n, (sigma_t * alpha_n - sigma_n * alpha_t) / alpha_n
                renoise.append(q)
        step = self.step_coefficients().repeat_interleave(R, dim=0).contiguous()
        return step, torch.stack(blend).float().contiguous(), torch.stack(renoise).float().contiguous()

    def q_sample_coefficients(self, t: float):
        """alpha, sigma of q(x_t | x_0) at noise level t (ip.py:272-284) as python floats (fp32-rounded)."""
        l = self.log_snr(torch.tensor(t, dtype=torch.float32))
        a, s = log_snr_to_alpha_sigma(l)
        return float(a), float(s), float(l)
