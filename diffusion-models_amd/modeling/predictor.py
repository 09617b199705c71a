"""Prediction closures with classifier-free guidance (reference: modeling/predictor.py:6-32)."""
from typing import Callable

import torch

from dm_hip import ops


class BasePredictor:
    def get_pred_fn(self, args, **kwargs) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
        raise NotImplementedError()


class GuidedPredictor(BasePredictor):
    def pred_conditional(self, z, t, cond=None, uncond_mask=None):
        raise NotImplementedError()

    def get_pred_fn(self, cond=None, guidance_scale=1.0):
        unconditional = cond is None or guidance_scale == 0.0
        plain = guidance_scale == 1.0

        def pred_fn(z, t):
            if unconditional:
                return self.pred_conditional(z, t)
            if plain:
                return self.pred_conditional(z, t, cond=cond)
            # Doubled batch exactly as the reference builds it: rows are interleaved (z0, z0, z1, z1, ...) while
            # the mask marks the FIRST half unconditional and the halves are split back by position.
            mask = torch.cat([torch.ones_like(cond), torch.zeros_like(cond)], dim=0)
            both = self.pred_conditional(
                torch.repeat_interleave(z, 2, dim=0),
                torch.repeat_interleave(t, 2, dim=0),
                cond=torch.repeat_interleave(cond, 2, dim=0),
                uncond_mask=mask,
            )
            pred_uncond, pred_cond = torch.chunk(both, 2, dim=0)
            return ops.cfg_combine(pred_uncond, pred_cond, guidance_scale)

        return pred_fn
