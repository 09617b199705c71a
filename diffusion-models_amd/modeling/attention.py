"""Spatial self-attention layer (reference: modeling/attention.py:35-84)."""
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from dm_hip import ops
from modeling.layers import GroupNorm, Linear


class Attention2D(nn.Module):
    def __init__(self, n_dim: int, n_heads: Optional[int] = None, n_dim_head: Optional[int] = None,
                 norm_groups: int = 32, dropout: float = 0.0):
        super().__init__()
        assert n_dim_head is not None or n_heads is not None
        if n_dim_head is None:
            assert n_dim % n_heads == 0
            n_dim_head = n_dim // n_heads
        elif n_heads is None:
            assert n_dim % n_dim_head == 0
            n_heads = n_dim // n_dim_head
        self.n_heads = n_heads
        self.n_dim_head = n_dim_head
        self.n_dim_inner = n_heads * n_dim_head
        self._dropout = float(dropout)
        self.register_buffer("dropout_p", torch.tensor(dropout))

        self.norm = GroupNorm(norm_groups, n_dim)
        self.qkv_proj = Linear(n_dim, 3 * self.n_dim_inner)
        self.out_proj = Linear(self.n_dim_inner, n_dim)

    def _attention_with_dropout(self, tokens):
        """Attention with dropout on the attention weights (modeling/attention.py:75-78, 92-96: softmax, dropout,
        times v). No configuration of the reference trains with it, so it is not a kernel of the library: the weights
        are materialised ([B, heads, L, L], fp32 softmax) and the two products are library batched GEMMs. The mask
        comes from torch's generator - the reference's comes from inside its fused attention kernel, a stream no
        other implementation can reproduce; only the distribution is the same."""
        B, L, _ = tokens.shape
        q, k, v = tokens.view(B, L, 3, self.n_heads, self.n_dim_head).permute(2, 0, 3, 1, 4)   # each [B, heads, L, d]
        w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)).float() * self.n_dim_head ** -0.5, dim=-1)
        w = F.dropout(w, p=self._dropout, training=True).to(tokens.dtype)
        return torch.matmul(w, v).transpose(1, 2).reshape(B, L, self.n_dim_inner)

    def forward(self, x: torch.Tensor):
        B, C, H, W = x.shape
        L = H * W
        x = ops.as_nhwc(x)
        # the residual branch reads x through the norm's fork, so that its gradient is added inside the norm's backward
        # kernel instead of by a separate accumulation pass
        h, x, _ = self.norm(x, fork=True)                 # [B, C, H, W] pixel-major == tokens [B, L, C]
        qkv = self.qkv_proj(h)                            # [B, 3*inner, H, W] == tokens [B, L, (3, heads, d)]
        tokens = qkv.permute(0, 2, 3, 1).reshape(B, L, 3 * self.n_dim_inner)
        if self.training and self._dropout > 0:
            att = self._attention_with_dropout(tokens)
        else:
            att = ops.attention(tokens, self.n_heads, self.n_dim_head)  # [B, L, inner]
        att = att.view(B, H, W, self.n_dim_inner).permute(0, 3, 1, 2)   # pixel-major view, no copy
        return self.out_proj(att, res=x)
