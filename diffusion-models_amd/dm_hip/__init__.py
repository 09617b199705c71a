"""ctypes binding of libdm_hip.so (the C ABI declared in include/dm_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950). There is no fallback: importing
this package without the built library raises, and every operator needs a HIP device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DM_HIP_LIBRARY: another build of the same library, e.g. to compare two builds on one machine)
LIB_PATH = os.environ.get("DM_HIP_LIBRARY") or os.path.join(_HERE, "libdm_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950). There is no CPU or stock-PyTorch fallback for these operators."
    )

lib = ctypes.CDLL(LIB_PATH)

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int) for n in
                ("B", "Hs", "Ws", "C1", "C2", "Cout", "KH", "KW", "stride", "pad", "ups", "transposed", "Ho", "Wo")]


class GnDesc(ctypes.Structure):
    _fields_ = [("B", c_int), ("HW", c_int), ("C1", c_int), ("C2", c_int), ("G", c_int), ("eps", c_float),
                ("act", c_int), ("has_ss", c_int), ("has_pre", c_int)]


class ConvProj(ctypes.Structure):
    """dm_conv_proj: the 1x1 projection of a second input that rides in a 3x3 convolution's launch."""
    _fields_ = [("x1", c_void_p), ("x2", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("C1", c_int), ("C2", c_int)]


class LinearItem(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("y", c_void_p), ("dy", c_void_p), ("dw", c_void_p),
                ("db", c_void_p), ("N", c_int)]


class SumItem(ctypes.Structure):
    """dm_sum_item: one ordered sum of partial vectors into up to three accumulating outputs."""
    _fields_ = [("part", c_void_p), ("stride", c_long), ("nparts", c_int), ("n", c_int), ("s1", c_int), ("s2", c_int),
                ("out0", c_void_p), ("out1", c_void_p), ("out2", c_void_p), ("dual0", c_void_p)]


class WgradItem(ctypes.Structure):
    """dm_wgrad_item: one layer of a batched weight-gradient call."""
    _fields_ = [("desc", ConvDesc), ("x1", c_void_p), ("x2", c_void_p), ("dy", c_void_p), ("dw", c_void_p)]


class WtItem(ctypes.Structure):
    """dm_wt_item: one weight of a batched forward->data-gradient re-layout (element offset, dims, first tile)."""
    _fields_ = [("offset", c_long), ("Cout", c_int), ("T", c_int), ("Cin", c_int), ("tile_begin", c_int)]


P = c_void_p
_SIGNATURES = {
    "dm_abi_version": (c_int, []),
    "dm_config_set": (c_int, [ctypes.c_char_p, c_int]),
    "dm_config_get": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int)]),
    "dm_config_reset": (None, []),
    "dm_conv2d_igemm": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, P]),
    "dm_conv2d_gn_fusable": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(GnDesc)]),
    "dm_conv2d_igemm_gn": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, P, ctypes.POINTER(GnDesc)] + [P] * 8),
    "dm_conv2d_proj_fusable": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvProj)]),
    "dm_conv2d_igemm_proj": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, ctypes.POINTER(ConvProj), P, P]),
    "dm_conv2d_wgrad_ws_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "dm_conv2d_wgrad": (c_int, [ctypes.POINTER(ConvDesc), P, P, P, P, P, ctypes.c_size_t, P]),
    "dm_conv2d_wgrad_group_ws_bytes": (ctypes.c_size_t, [P, c_int]),
    "dm_conv2d_wgrad_group": (c_int, [P, c_int, P, ctypes.c_size_t, P]),
    "dm_pack_conv_weight": (c_int, [P, c_long, c_long, c_long, c_long, c_int, c_int, c_int, c_int, P, P, P]),
    "dm_transpose_conv_weights": (c_int, [P, P, P, c_int, c_int, P]),
    "dm_colsum_ws_floats": (c_long, [c_long, c_int]),
    "dm_ordered_sum_group": (c_int, [P, c_int, P]),
    "dm_colsum_bf16": (c_int, [P, P, c_long, c_int, P, P, P]),
    "dm_colsum_bf16_dual": (c_int, [P, P, P, c_long, c_int, P, P, P]),
    "dm_colsum_f32": (c_int, [P, P, c_int, c_int, P]),
    "dm_in_conv_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "dm_out_conv_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "dm_out_conv_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "dm_edge_conv_wgrad_ws_floats": (c_long, [c_int] * 6),
    "dm_edge_conv_wgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "dm_groupnorm_ws_floats": (c_long, [ctypes.POINTER(GnDesc)]),
    "dm_groupnorm_act_fwd": (c_int, [ctypes.POINTER(GnDesc)] + [P] * 11),
    "dm_noalloc_init": (c_int, []),
    "dm_groupnorm_noalloc_max_ws_floats": (c_long, []),
    "dm_groupnorm_act_fwd_noalloc": (c_int, [ctypes.POINTER(GnDesc)] + [P] * 10),
    "dm_groupnorm_act_bwd": (c_int, [ctypes.POINTER(GnDesc)] + [P] * 16 + [c_int, P, P]),
    "dm_groupnorm_param_grads": (c_int, [ctypes.POINTER(GnDesc)] + [P] * 6),
    "dm_attention_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "dm_attention_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "dm_timestep_embed": (c_int, [P, P, P, c_int, c_int, P]),
    "dm_linear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dm_linear_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dm_linear_group_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "dm_linear_group_bwd_ws_floats": (c_long, [P, c_int, c_int, c_int]),
    "dm_linear_group_bwd": (c_int, [P, P, c_int, P, P, c_int, c_int, c_int, P]),
    "dm_class_embed_add": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "dm_class_embed_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dm_upsample2x_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "dm_upsample2x_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "dm_avgpool2x_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "dm_avgpool2x_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "dm_diffuse": (c_int, [P, P, P, P, P, P, c_int, c_long, c_int, P]),
    "dm_mse_partials": (c_int, []),
    "dm_mse_loss_fwd": (c_int, [P, P, P, P, P, c_long, c_int, P]),
    "dm_mse_loss_bwd": (c_int, [P, P, P, P, P, c_long, c_int, P]),
    "dm_sampler_step": (c_int, [P, P, P, c_float, c_float, c_float, c_float, c_int, c_int, c_int, c_long, P]),
    "dm_sampler_step_dev": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_long, P]),
    "dm_cfg_combine": (c_int, [P, P, P, c_float, c_long, P]),
    "dm_u8_batch": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_long, c_int, P]),
    "dm_nn2_ws_floats": (c_long, [c_int, c_int]),
    "dm_nn2_distances": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P]),
    "dm_sumsq_partials": (c_int, []),
    "dm_grad_norm": (c_int, [P, c_long, c_float, P, P, P]),
    "dm_adamw_ema_step": (c_int, [P, P, P, P, P, P, P, c_float, c_long] + [c_float] * 8 + [P, c_float, c_int, P]),
    "dm_optimizer_tick": (c_int, [P, P] + [c_float] * 6 + [P]),
    "dm_optimizer_tick_lr": (c_int, [P, P, c_float, c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_double,
                                     ctypes.c_double] + [c_float] * 5 + [P]),
    "dm_lr_multiplier": (ctypes.c_double, [c_int] + [ctypes.c_longlong] * 3 + [ctypes.c_double] * 2),
}

EXPORTS = tuple(_SIGNATURES) + ("dm_last_error",)

for _name, (_res, _args) in _SIGNATURES.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args
lib.dm_last_error.restype = ctypes.c_char_p
lib.dm_last_error.argtypes = []


class DmHipError(RuntimeError):
    pass


def check(rc):
    """Map a non-zero dm_status to an exception carrying the library's message."""
    if rc != 0:
        raise DmHipError(f"libdm_hip status {rc}: {lib.dm_last_error().decode()}")


def reset_config():
    """Every run-time switch of the library back to its built-in value, then DM_HIP_CONFIG="key=value,key=value" from
    the environment applied on top (tuning runs; unknown keys are errors). Runs at import; tests call it after
    every case so that no case leaves a switch behind for the next."""
    lib.dm_config_reset()
    for kv in filter(None, os.environ.get("DM_HIP_CONFIG", "").split(",")):
        k, _, v = kv.partition("=")
        check(lib.dm_config_set(k.strip().encode(), int(v)))


reset_config()
