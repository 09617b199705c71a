"""Activation lookup (reference: modeling/activation.py:5-13). SiLU is fused into the GroupNorm and linear
kernels; ReLU/GELU remain constructible and run as plain torch modules on the bf16 activations."""
from torch import nn

_TABLE = {"relu": nn.ReLU, "gelu": nn.GELU, "swish": nn.SiLU, "silu": nn.SiLU}


def get_activation(name: str):
    if name not in _TABLE:
        raise ValueError(f'unknown activation "{name}"')
    return _TABLE[name]()


def is_silu(module) -> bool:
    return isinstance(module, nn.SiLU)
