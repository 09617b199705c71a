"""Sampling loops (reference: diffusion/sampler.py).

Every per-step scalar (DDPM coefficient tables, DDIM alpha/sigma pairs, Euler dt) is computed ONCE per
`sample()` call on the host with the reference's own arithmetic and dtypes, so the loop body is: one UNet
forward (pred_fn) + one fused in-place update kernel, with no table upload and no device->host sync per step.
A pred_fn that offers `run_loop` (harness.generate.GraphedPredictor) takes the whole loop over: forward, noise draw
and update are then ONE captured HIP graph replayed n_steps times, the step's scalars coming from a device table.
"""
from typing import Callable, Optional

import torch
import tqdm.auto as tqdm

from dm_hip import ops
from diffusion.scheduler import BetaScheduler, RectifiedFlowScheduler, VariancePreservingScheduler


def get_sampler(name, **kwargs):
    if name == "ddpm":
        return DDPMSampler(**kwargs)
    if name == "ddim":
        return DDIMSampler(**kwargs)
    if name == "euler":
        return RectifiedFlowEulerSampler(**kwargs)
    raise ValueError(f"unknown sampler {name}")


def timestep_grid(n_sched: int, n_steps: int) -> torch.Tensor:
    """The reference's step grid: int64 truncation of linspace(n_sched-1, 0, n_steps+1), last point dropped."""
    return torch.linspace(n_sched - 1, 0, steps=n_steps + 1, dtype=torch.int64)[:-1]


class BaseSampler:
    pred_type = "noise"

    def __init__(self, n_steps, pbar=False, pbar_kwargs=None):
        self.n_steps = n_steps
        self.pbar = pbar
        self.pbar_kwargs = {"leave": False}
        if pbar_kwargs is not None:
            self.pbar_kwargs.update(pbar_kwargs)

    def set_steps(self, n_steps):
        self.n_steps = n_steps

    def prepare_iterator(self, iterator):
        if self.pbar:
            iterator = tqdm.tqdm(iterator, **self.pbar_kwargs)
        return iterator

    def sample(self, z: torch.Tensor, scheduler, pred_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
               **kwargs):
        raise NotImplementedError()

    # shared loop plumbing -------------------------------------------------------------------------------
    def _loop(self, state, t_steps, t_rows, rows, mode, clip, pred_fn, gen=None):
        """rows[i] = (c0, c1, c2, c3) of step i (see dm_sampler_step); gen: the noise generator of DDPM, else None."""
        run_loop = getattr(pred_fn, "run_loop", None)
        if run_loop is not None and not self.pbar:
            return run_loop(state, t_steps, rows, mode, clip, gen)
        n = self.n_steps
        for i in self.prepare_iterator(range(n)):
            pred = pred_fn(state, t_rows[i])
            last = i == n - 1
            noise = torch.randn_like(state, generator=gen) if (gen is not None and not last) else None
            c0, c1, c2, c3 = rows[i]
            ops.sampler_step(state, pred, noise, c0, c1, c2, c3, mode, clip, last)
        return state

    def _begin(self, z, scheduler):
        t_steps = timestep_grid(scheduler.n_steps, self.n_steps)
        # one [n_steps, B] device tensor: row i is the t argument of step i (no per-step host->device traffic)
        t_rows = t_steps.to(z.device)[:, None].expand(self.n_steps, z.size(0)).contiguous()
        state = z.detach().to(torch.float32).contiguous().clone()
        return t_steps, t_rows, state


class DDPMSampler(BaseSampler):
    @torch.no_grad()
    def sample(self, z: torch.Tensor, scheduler: BetaScheduler, pred_fn, seed: Optional[int] = None):
        gen = torch.Generator(device=z.device)
        if seed is not None:
            gen.manual_seed(seed)
        t_steps, t_rows, state = self._begin(z, scheduler)

        # coefficient tables in the sample dtype, as the reference derives them (diffusion/sampler.py:71-74)
        beta = scheduler.beta.to(dtype=state.dtype)
        coef = (beta / torch.sqrt(1 - scheduler.alpha_sq).to(dtype=state.dtype)).tolist()
        inv_sqrt_alpha = (1 / torch.sqrt(1 - beta)).tolist()
        sigma = torch.sqrt(beta).tolist()

        rows = [(inv_sqrt_alpha[t], coef[t], sigma[t], 0.0) for t in t_steps.tolist()]
        return self._loop(state, t_steps, t_rows, rows, 0, False, pred_fn, gen).to(z.dtype)


class DDIMSampler(BaseSampler):
    def __init__(self, n_steps, eta=0.0, clip_latent=True, pbar=False, pbar_kwargs=None):
        super().__init__(n_steps, pbar, pbar_kwargs)
        self.eta = eta  # kept for config compatibility; the reference never uses it (always deterministic)
        self.clip_latent = clip_latent

    @torch.no_grad()
    def sample(self, z: torch.Tensor, scheduler: VariancePreservingScheduler, pred_fn):
        t_steps, t_rows, state = self._begin(z, scheduler)
        sch = scheduler.get_schedule(t_steps)  # host, fp64
        alpha = torch.sqrt(sch.alpha_sq).to(state.dtype).tolist()
        sigma = torch.sqrt(sch.sigma_sq).to(state.dtype).tolist()

        n = self.n_steps
        rows = [(alpha[i], sigma[i]) + ((alpha[i + 1], sigma[i + 1]) if i < n - 1 else (0.0, 0.0)) for i in range(n)]
        return self._loop(state, t_steps, t_rows, rows, 1, self.clip_latent, pred_fn).to(z.dtype)


class RectifiedFlowEulerSampler(BaseSampler):
    pred_type = "velocity"

    @torch.no_grad()
    def sample(self, z: torch.Tensor, scheduler: RectifiedFlowScheduler, pred_fn):
        t_steps, t_rows, state = self._begin(z, scheduler)
        sig = scheduler.sigmas[t_steps]  # fp32, host
        nxt = torch.cat([sig[1:], torch.zeros(1, dtype=sig.dtype)])
        dt = (nxt - sig).tolist()        # fp32 subtraction as in the reference; negative: backward Euler steps

        rows = [(d, 0.0, 0.0, 0.0) for d in dt]
        return self._loop(state, t_steps, t_rows, rows, 2, False, pred_fn).to(z.dtype)
