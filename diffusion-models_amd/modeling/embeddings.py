"""Timestep embedding (reference: modeling/embeddings.py:10-47)."""
import math

import torch
from torch import nn

from dm_hip import ops
from modeling.activation import get_activation, is_silu
from modeling.layers import EmbLinear


class PeriodicEncoding(nn.Module):
    """t -> [sin(t f_k) | cos(t f_k)], f_k = exp(-ln(max_period) * k / (half - 1)) (end-point inclusive)."""

    def __init__(self, n_dim: int, max_period: int = 10000):
        super().__init__()
        assert n_dim % 2 == 0
        ramp = torch.linspace(0, 1, steps=n_dim // 2, dtype=torch.float32)
        self.register_buffer("freq", torch.exp(-math.log(max_period) * ramp))

    def forward(self, x: torch.Tensor):
        return ops.timestep_embed(x, self.freq)


class MLPEmbedding(nn.Module):
    def __init__(self, time_emb_dim: int, out_dim: int, activation: str = "swish"):
        super().__init__()
        self.linear1 = EmbLinear(time_emb_dim, out_dim)
        self.nonlinearity = get_activation(activation)
        self.linear2 = EmbLinear(out_dim, out_dim)

    def forward(self, x: torch.Tensor):
        h = self.linear1(x)
        if is_silu(self.nonlinearity):
            return self.linear2(h, act_in=True)
        return self.linear2(self.nonlinearity(h))
