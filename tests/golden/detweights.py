"""Deterministic weights shared by the golden-vector generator and the tests.

The reference initialises with torch's default initialisers (modeling/unet.py:44-126 uses
plain nn.Conv2d / nn.Linear / nn.GroupNorm / nn.Embedding).  Golden vectors must not depend on torch's
RNG stream, and multi-megabyte state dicts cannot be committed, so both sides rebuild the SAME weights
from numpy's frozen legacy generator, walking the state dict in registration order.
"""
import numpy as np

_BUFFER_SUFFIXES = ("uncond_class", "freq", "dropout_p")


def det_tensor(name, shape, rs):
    shape = tuple(int(s) for s in shape)
    if name.endswith(_BUFFER_SUFFIXES):
        return None  # buffers keep their constructed value
    if "norm" in name and name.endswith("weight"):
        return (1.0 + 0.2 * rs.uniform(-1, 1, size=shape)).astype(np.float32)
    if "norm" in name and name.endswith("bias"):
        return (0.2 * rs.uniform(-1, 1, size=shape)).astype(np.float32)
    if name.endswith("class_embedding.weight"):
        w = rs.normal(0, 1, size=shape).astype(np.float32)
        w[-1] = 0.0  # padding_idx row
        return w
    if name.endswith("weight"):
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / np.sqrt(fan_in)
        return rs.uniform(-b, b, size=shape).astype(np.float32)
    if name.endswith("bias"):
        return (0.1 * rs.uniform(-1, 1, size=shape)).astype(np.float32)
    raise ValueError(f"no rule for {name}")


def det_state_dict(names_shapes, seed):
    """names_shapes: iterable of (name, shape) in state-dict order -> {name: np.ndarray or None}."""
    rs = np.random.RandomState(seed)
    return {n: det_tensor(n, s, rs) for n, s in names_shapes}


def load_det_weights(module, seed):
    """Overwrite every parameter of a torch module (reference or build) in place."""
    import torch
    sd = module.state_dict()
    vals = det_state_dict([(k, v.shape) for k, v in sd.items()], seed)
    with torch.no_grad():
        for k, v in sd.items():
            if vals[k] is not None:
                v.copy_(torch.from_numpy(vals[k]))
    return module
