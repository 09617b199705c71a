"""Noise schedules, forward noising and training losses (reference: diffusion/scheduler.py).

Schedule tables are built once on the host exactly as the reference builds them (fp64 beta / alpha_sq for the
variance-preserving schedules, fp32 sigmas for rectified flow) and kept as the same public attributes
(`n_steps`, `beta`, `alpha_sq`, `sigmas`). The per-step work - noising and the MSE loss with its gradient -
runs in fused HIP kernels that index a device copy of the table (uploaded once per device, not per call).
"""
import math
from dataclasses import dataclass

import torch

from dm_hip import ops


def get_scheduler(name, **kwargs):
    if name == "beta":
        return BetaScheduler(**kwargs)
    if name == "rectified_flow":
        return RectifiedFlowScheduler(**kwargs)
    raise ValueError(f"unknown scheduler type {name}")


class BaseScheduler:
    pred_type = "noise"

    def get_loss(self, x: torch.Tensor, model, gen=None, **model_call_kwargs):
        raise NotImplementedError()

    def diffuse(self, x: torch.Tensor, eps: torch.Tensor, t: torch.Tensor):
        raise NotImplementedError()

    # device copies of host tables, keyed by (attribute, device)
    def _on_device(self, attr: str, device):
        cache = self.__dict__.setdefault("_device_tables", {})
        host = getattr(self, attr)
        key = (attr, str(device))
        hit = cache.get(key)
        if hit is None or hit[0] is not host:
            cache[key] = (host, host.to(device).contiguous())
        return cache[key][1]

    def _sample_eps_t(self, x, gen):
        # RNG draw order of the reference: eps first, then t (diffusion/scheduler.py:62-67, 136-141)
        eps = torch.randn_like(x, generator=gen)
        t = torch.randint(0, self.n_steps, size=(x.size(0),), device=x.device, generator=gen)
        return eps, t


@dataclass
class VPSchedule:
    alpha_sq: torch.Tensor
    sigma_sq: torch.Tensor
    log_snr: torch.Tensor

    @staticmethod
    def from_alpha_sq(alpha_sq):
        sigma_sq = 1 - alpha_sq
        return VPSchedule(alpha_sq=alpha_sq, sigma_sq=sigma_sq, log_snr=torch.log(alpha_sq / sigma_sq))

    @staticmethod
    def from_log_snr(log_snr):
        alpha_sq = torch.sigmoid(log_snr)
        return VPSchedule(alpha_sq=alpha_sq, sigma_sq=1 - alpha_sq, log_snr=log_snr)


class VariancePreservingScheduler(BaseScheduler):
    def __init__(self, n_steps):
        self.n_steps = n_steps

    def get_schedule(self, t: torch.LongTensor) -> VPSchedule:
        raise NotImplementedError()

    def get_loss(self, x: torch.Tensor, model, gen=None, **model_call_kwargs):
        eps, t = self._sample_eps_t(x, gen)
        x_t = self.diffuse(x, t, eps)
        eps_pred = model(x_t, t, **model_call_kwargs)
        return ops.mse_loss(eps_pred, eps, mode=0)

    def diffuse(self, x, t, eps):
        # x_t = sqrt(alpha_sq[t]) x + sqrt(1 - alpha_sq[t]) eps, square roots in fp64 then rounded to fp32
        return ops.diffuse(x, eps, t, tab64=self._alpha_sq_table(x.device))

    def _alpha_sq_table(self, device):
        raise NotImplementedError()


class BetaScheduler(VariancePreservingScheduler):
    def __init__(self, schedule="linear", beta_start=1e-4, beta_end=2e-2, cos_s=0.008, cos_max_beta=0.999,
                 n_steps=1000):
        if schedule == "linear":
            beta = torch.linspace(beta_start, beta_end, steps=n_steps, dtype=torch.float64)
        elif schedule == "cosine":
            beta = self.alpha_cosine(cos_s, cos_max_beta, n_steps)
        else:
            raise ValueError(f"unknown schedule type {schedule}")
        self.n_steps = n_steps
        self.beta = beta
        self.alpha_sq = torch.cumprod(1 - beta, dim=0)

    def get_schedule(self, t: torch.LongTensor) -> VPSchedule:
        return VPSchedule.from_alpha_sq(self._on_device("alpha_sq", t.device)[t])

    def _alpha_sq_table(self, device):
        return self._on_device("alpha_sq", device)

    @staticmethod
    def alpha_cosine(s=0.008, max_beta=0.999, n_steps=1000):
        def alpha_bar(u):
            return math.cos((u + s) / (1 + s) * math.pi / 2) ** 2

        betas = [min(1 - alpha_bar((i + 1) / n_steps) / alpha_bar(i / n_steps), max_beta) for i in range(n_steps)]
        return torch.tensor(betas, dtype=torch.float64)


class RectifiedFlowScheduler(BaseScheduler):
    pred_type = "velocity"

    def __init__(self, n_steps=1000):
        self.n_steps = n_steps
        self.sigmas = torch.linspace(1, n_steps, steps=n_steps) / n_steps

    def get_loss(self, x, model, gen=None, **model_call_kwargs):
        eps, t = self._sample_eps_t(x, gen)
        x_t = self.diffuse(x, t, eps)
        v_pred = model(x_t, t, **model_call_kwargs)
        return ops.mse_loss(v_pred, eps, x=x, mode=1)  # target x - eps

    def diffuse(self, x, t, eps):
        # x_t = (1 - sigma[t]) x + sigma[t] eps
        return ops.diffuse(x, eps, t, tab32=self._on_device("sigmas", x.device))
