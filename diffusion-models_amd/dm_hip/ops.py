"""torch.autograd bindings of the libdm_hip operators.

Activations travel between operators as bf16 tensors of logical shape [B, C, H, W] in channels_last memory
format, i.e. physically pixel-major [B][H][W][C] as the kernels expect; token tensors [B, L, C] are the same
memory. Parameters stay fp32 in the reference's shapes. Every function here needs a HIP device tensor.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import ConvDesc, ConvProj, GnDesc, LinearItem, SumItem, WgradItem, check, lib

BF16 = torch.bfloat16
CL = torch.channels_last

# When True, packed bf16 weight copies are rebuilt on every forward (needed while a whole training step is
# captured into a HIP graph: the weights change between replays but the Python-side version check runs once).
FORCE_REPACK = False
# Bumped by whoever rewrites parameter memory behind torch's back (the fused optimizer kernel); part of the
# pack cache key, since such writes do not advance tensor._version.
WEIGHTS_EPOCH = 0
# When True, parameter gradients are accumulated by the kernels straight into an existing fp32 `param.grad`
# (e.g. a view of a flat gradient arena) instead of being returned to autograd as fresh tensors: no zero-fill, no
# extra add pass. GRAD_READY(param), if set, is told when such a gradient has been enqueued.
DIRECT_GRADS = False
GRAD_READY = None
# When True (and DIRECT_GRADS), weight/bias gradient kernels of the convolutions are enqueued on a second stream:
# they depend only on tensors that exist when a layer's backward starts and nothing but the optimizer reads their
# results, so they overlap the data-gradient / GroupNorm chain. Whoever sets it must make the consumer of the
# gradients wait for side_stream(device).
SIDE_WGRAD = 0   # number of side streams to rotate over (0: off)
_SIDE = {}
_SIDE_NEXT = 0
# Side streams that have been given work since whoever joins them last cleared this set. Under HIP-graph capture only
# these may be waited for: a stream that took no part in the capture holds no captured work, and waiting for it from
# the capturing stream is an error.
SIDE_TOUCHED = set()
# Tensors that kernels on a side stream read and that autograd may otherwise rewrite IN PLACE behind them: a gradient
# handed back unchanged as the gradient of another input (the residual branch of a convolution) becomes the
# accumulation buffer when a second gradient arrives for the same tensor, and autograd adds into it on the main
# stream if it holds the last reference. A reference kept here until the side streams are joined forces that sum out
# of place. (record_stream only guards the memory against reuse after the tensor dies.)
SIDE_KEEP = []


# HIP graphs are never destroyed. ROCm 7.2: destroying an instantiated graph whose capture forked onto other streams
# frees stream objects that the runtime's list of graph-launch streams still names, and the next launch of ANY graph
# with forks reads them (a segmentation fault inside hipGraphLaunch once that memory has been reused;
# `DM_DESTROY_GRAPHS=1 python tools/diag_seg.py ddim ddpm euler h3` reproduces it and prints the native call stack).
# Every graph this package captures is created through new_graph(), which keeps it alive until the process ends;
# callers that capture graphs of their own around these operators should do the same.
_GRAPHS = []


def new_graph():
    g = torch.cuda.CUDAGraph()
    if os.environ.get("DM_DESTROY_GRAPHS") != "1":   # (set only by the reproducer of the fault)
        _GRAPHS.append(g)
    return g


class SumCollector:
    """Deferred ordered sums (dm_sum_item): the producers of parameter-gradient partials fill in one item each
    instead of launching their own sum, and flush() runs all of them in ONE launch. The partial buffers are kept
    alive until then. flush() must be called on a stream that is ordered after every producer."""

    CHUNK = 256

    def __init__(self):
        # items live in fixed-size ctypes arrays (a producer writes through the reference slot() hands out, so an
        # array is never moved); a model that files more sums than one array holds simply opens the next one
        self.chunks = [(SumItem * self.CHUNK)()]
        self.count = 0
        self.keep = []

    def slot(self, *tensors):
        """ctypes reference to the next free item; `tensors` are the scratch buffers the item will point into."""
        c, i = divmod(self.count, self.CHUNK)
        if c == len(self.chunks):
            self.chunks.append((SumItem * self.CHUNK)())
        ref = ctypes.byref(self.chunks[c][i])
        self.count += 1
        self.keep.extend(tensors)
        return ref

    def flush(self):
        if self.count:
            cur = torch.cuda.current_stream()
            for t in self.keep:
                t.record_stream(cur)
            for c in range((self.count + self.CHUNK - 1) // self.CHUNK):
                n = min(self.CHUNK, self.count - c * self.CHUNK)
                check(lib.dm_ordered_sum_group(self.chunks[c], n, cur.cuda_stream))
        self.count = 0
        self.keep = []


# When set (a SumCollector) and DIRECT_GRADS is on, the ordered sums behind bias / GroupNorm-affine / edge-conv
# gradients are collected there instead of being launched one by one; whoever sets it calls flush() after backward
# (after joining the side streams) and before anything reads those gradients.
PENDING_SUMS = None


class WgradCollector:
    """Deferred convolution weight gradients (dm_wgrad_item): with DIRECT_GRADS on, a convolution's backward files its
    weight gradient here instead of launching it, and every `group` layers (and at flush()) they run as ONE call of
    dm_conv2d_wgrad_group on the side stream - fewer, longer pixel slices per layer, fewer partial tiles. The operands
    (dy and the saved inputs) are kept alive until the launch; GRAD_READY is told about the parameters at the launch.
    flush() must be called before the side streams are joined."""

    def __init__(self, group=6, tail=2):
        self.group = group
        self.tail = tail          # group size once fewer than `group` layers of the pass are still to come
        self.expected = 0         # layers one backward pass files here (0: unknown); set by begin_pass()
        self.seen = 0
        self.items = (WgradItem * 64)()
        self.count = 0
        self.keep = []      # operand tensors of the pending items
        self.params = []
        self.pending_ids = set()   # id() of the parameters whose gradient is filed here and not launched yet

    def add(self, desc, x1, x2, dy, dw, param):
        it = self.items[self.count]
        it.desc, it.x1, it.x2, it.dy, it.dw = desc, x1.data_ptr(), _p(x2), dy.data_ptr(), dw.data_ptr()
        self.count += 1
        self.keep += [t for t in (x1, x2, dy) if t is not None]
        self.params.append(param)
        self.pending_ids.add(id(param))
        self.seen += 1

    def begin_pass(self, expected=0):
        """Before a backward pass; expected: how many layers it will file (lets the last launches be small: whatever
        is launched after the last data-gradient kernel has nothing left to run beside)."""
        self.expected, self.seen = expected, 0

    def flush_if_full(self):
        """Called by an operator after its own last launch (flush() may end a HIP-graph capture through GRAD_READY,
        and a side stream forked by the operator belongs to the capture that was open then)."""
        limit = self.group
        if self.expected and self.expected - self.seen < self.group:
            limit = self.tail
        if self.count >= min(limit, len(self.items)):
            self.flush()

    def flush(self):
        if not self.count:
            return
        n, self.count = self.count, 0
        keep, self.keep = self.keep, []
        params, self.params = self.params, []
        self.pending_ids.clear()
        st, stream_w = _param_stream(keep, True)
        nbytes = lib.dm_conv2d_wgrad_group_ws_bytes(self.items, n)
        ws = _scratch(keep[0].device, (nbytes + 3) // 4 + 64, stream_w)
        base = (ws.data_ptr() + 255) // 256 * 256
        check(lib.dm_conv2d_wgrad_group(self.items, n, base, nbytes, st))
        for prm in params:
            _grad_done(prm)

    def drop(self):
        self.count, self.keep, self.params = 0, [], []
        self.pending_ids.clear()


# When set (a WgradCollector) and DIRECT_GRADS is on, convolution weight gradients are collected there.
PENDING_WGRADS = None


def _deferred(direct, *tensors):
    """The `deferred` argument of a producer: a collector slot when sums are being collected, else None (sum now)."""
    if direct and PENDING_SUMS is not None:
        return PENDING_SUMS.slot(*tensors)
    return None


def side_stream(device, index=0):
    s = _SIDE.get((device, index))
    if s is None:
        s = _SIDE[(device, index)] = torch.cuda.Stream(device=device)
    return s


def side_streams(device):
    """Every side stream created so far on `device` (the streams a consumer of the gradients must wait for)."""
    return [s for (d, _), s in _SIDE.items() if d == device]


# Named auxiliary streams (aux_stream): whole branches of the network that a module runs beside the main stream (the
# embedding path). Autograd runs such a branch's backward on the same stream, where it may write parameter gradients.
_AUX = {}         # raw stream handle -> torch stream
# The embedding path of the UNet (time MLP, class table, every block's projection: a dozen small fp32 kernels that
# nothing but the first block's second GroupNorm waits for) runs on an auxiliary stream beside the first convolutions,
# and its backward beside the last ones.
EMB_SIDE_STREAM = os.environ.get("DM_EMB_SIDE", "1") != "0"


def aux_stream(device, name):
    s = _SIDE.get((device, name))
    if s is None:
        s = _SIDE[(device, name)] = torch.cuda.Stream(device=device)
        _AUX[s.cuda_stream] = s
    return s


def aux_streams():
    return list(_AUX.values())


LAUNCHES = 0     # operator launches so far (every call takes its stream from _stream / _param_stream)


def _stream():
    """Raw handle of the current stream (what the kernels launch on); work on an auxiliary stream marks it as one
    that the consumer of the gradients has to wait for."""
    global LAUNCHES
    LAUNCHES += 1
    raw = torch.cuda.current_stream().cuda_stream
    if _AUX:
        s = _AUX.get(raw)
        if s is not None:
            SIDE_TOUCHED.add(s)
    return raw


def _p(t):
    return None if t is None else t.data_ptr()


def _require_device(t, who):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: libdm_hip operators run on the GPU only; got a {t.device} tensor "
                           "(there is no CPU fallback)")


def is_nhwc(x):
    """True if a [B,C,H,W] tensor is dense pixel-major in memory."""
    if x.dim() != 4:
        return False
    B, C, H, W = x.shape
    return x.stride() == (H * W * C, 1, W * C, C) or x.numel() == 0


def as_nhwc(x):
    """bf16 + pixel-major memory; no copy when the tensor already is."""
    _require_device(x, "as_nhwc")
    if x.dtype == BF16 and is_nhwc(x):
        return x
    B, C, H, W = x.shape
    out = torch.empty_strided((B, C, H, W), (H * W * C, 1, W * C, C), dtype=BF16, device=x.device)
    out.copy_(x)
    return out


def empty_nhwc(B, C, H, W, device):
    return torch.empty_strided((B, C, H, W), (H * W * C, 1, W * C, C), dtype=BF16, device=device)


def _grad_sink(param, ohwi=False):
    """The tensor a kernel may accumulate this parameter's gradient into directly, or None."""
    if not DIRECT_GRADS or param is None:
        return None
    g = param.grad
    if g is None or g.dtype != torch.float32 or not g.is_cuda:
        return None
    if ohwi and g.dim() == 4:
        co, ci, kh, kw = g.shape
        if g.stride() != (kh * kw * ci, 1, kw * ci, ci):
            return None
    elif not g.is_contiguous():
        return None
    return g


def _grad_done(param):
    if GRAD_READY is not None:
        GRAD_READY(param)


def f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _scratch(device, floats, stream=None):
    """Scratch for one kernel call, from torch's caching allocator: under HIP-graph capture it comes from the
    graph's private pool, so a captured pointer can never be handed to another tensor while the graph lives (a
    grow-on-demand buffer shared by all calls could). `stream`: the side stream that will run the kernel."""
    if stream is None:
        return torch.empty(max(int(floats), 1), dtype=torch.float32, device=device)
    with torch.cuda.stream(stream):
        return torch.empty(max(int(floats), 1), dtype=torch.float32, device=device)


def _param_stream(tensors, enable=True):
    """(raw stream, torch stream or None) for a kernel that only produces parameter gradients: the next side stream
    when the caller turned that mode on (it then waits for the current stream, and `tensors` are kept alive for
    it), else the current stream."""
    if not (SIDE_WGRAD and enable):
        return _stream(), None
    global _SIDE_NEXT, LAUNCHES
    LAUNCHES += 1
    side = side_stream(tensors[0].device, _SIDE_NEXT % SIDE_WGRAD)
    _SIDE_NEXT += 1
    side.wait_stream(torch.cuda.current_stream())
    SIDE_TOUCHED.add(side)
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    return side.cuda_stream, side


# ------------------------------------------------------------------------------------------------------------
class WeightPack:
    """bf16 copies of one conv/linear weight in the layouts the implicit-GEMM kernels read:
    forward [Cout][T][Cin] and data-gradient [Cin][T][Cout]."""

    def __init__(self):
        self.key = None
        self.w_fwd = None
        self.w_dgrad = None
        self.bound = None

    def bind(self, weight, w_fwd, w_dgrad):
        """Use externally maintained copies (e.g. slices of a bf16 mirror of a parameter arena that the optimizer
        kernel refreshes) for as long as `weight` is not rewritten through torch."""
        self.bound = (weight.data_ptr(), weight._version, w_fwd, w_dgrad)

    def get(self, weight):
        if self.bound is not None:
            ptr, version, w_fwd, w_dgrad = self.bound
            if weight.data_ptr() == ptr and weight._version == version:
                return w_fwd, w_dgrad
            self.bound = None  # the parameter was rewritten behind the mirror: pack it here from now on
        w4 = weight if weight.dim() == 4 else weight[:, :, None, None]
        key = (weight.data_ptr(), weight._version, weight.device, WEIGHTS_EPOCH)
        if FORCE_REPACK or key != self.key or self.w_fwd is None:
            Cout, Cin, KH, KW = w4.shape
            if self.w_fwd is None or self.w_fwd.device != weight.device or self.w_fwd.numel() != w4.numel():
                self.w_fwd = torch.empty(w4.numel(), dtype=BF16, device=weight.device)
                self.w_dgrad = torch.empty(w4.numel(), dtype=BF16, device=weight.device)
            s = w4.stride()
            src = w4.detach()
            if src.dtype != torch.float32:
                src = src.float()
                s = src.stride()
            check(lib.dm_pack_conv_weight(src.data_ptr(), s[0], s[1], s[2], s[3], Cout, Cin, KH, KW,
                                          self.w_fwd.data_ptr(), self.w_dgrad.data_ptr(), _stream()))
            self.key = key
        return self.w_fwd, self.w_dgrad


def refresh_packs(module):
    """Bring the packed bf16 copies of every layer's weight up to date on the current stream (what a layer's next eager
    call would do): for callers about to replay a HIP graph that was captured while the copies were current - the graph
    reads the copies' buffers and holds no packing launch of its own. Returns the buffers' addresses (they stay where
    they are unless a layer's weight changes size or a bound mirror is dropped: a graph is good for one such tuple)."""
    where = []
    for m in module.modules():
        pack, w = getattr(m, "_pack", None), getattr(m, "weight", None)
        if isinstance(pack, WeightPack) and w is not None and w.is_cuda:
            where.append(pack.get(w)[0].data_ptr())
    return tuple(where)


def _conv_desc(B, Hs, Ws, C1, C2, Cout, KH, KW, stride, pad, ups, transposed, Ho, Wo):
    return ConvDesc(B, Hs, Ws, C1, C2, Cout, KH, KW, stride, pad, ups, transposed, Ho, Wo)


class GradBox:
    """Carries the gradient of a skip connection from its late consumer's backward (the up path, which runs first in
    backward) to its early consumer's (the next layer of the down path), whose kernel adds it to the gradient it
    computes - instead of autograd adding the two with a pass of its own. The early consumer creates the box for its
    input tensor (grad_box), the late one routes the tensor through skip_tap."""
    __slots__ = ("grad", "tapped", "delivered")

    def __init__(self):
        self.grad, self.tapped, self.delivered = None, False, False

    def take(self, shape):
        """The gradient left by the tap (None if the tensor was never tapped); the early consumer's backward calls it."""
        if self.tapped and not self.delivered:
            raise RuntimeError("skip connection: the early consumer's backward ran before the late consumer's "
                               "(a gradient would be lost)")
        g, self.grad = self.grad, None
        self.tapped = self.delivered = False
        if g is None:
            return None
        g = as_nhwc(g)
        if tuple(g.shape) != tuple(shape) or g.dtype != BF16:
            raise RuntimeError(f"skip connection: gradient {tuple(g.shape)} {g.dtype} does not match the input {tuple(shape)}")
        return g


SKIP_BOXES = os.environ.get("DM_SKIP_BOXES", "1") != "0"
_BOX_KINDS = os.environ.get("DM_SKIP_BOX_KINDS", "gn,conv").split(",")


def grad_box(x, kind="gn"):
    """A box for the gradient that a later skip_tap of `x` (the same tensor object) will leave; None when gradients
    are off or x needs none."""
    if not (SKIP_BOXES and kind in _BOX_KINDS and torch.is_grad_enabled() and x.requires_grad):
        return None
    box = GradBox()
    x._dm_grad_box = box
    return box


class _SkipTapFn(Function):
    @staticmethod
    def forward(ctx, x, box):
        ctx.box = box
        ctx.set_materialize_grads(False)
        box.tapped, box.delivered = True, False
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.box.grad = g
        ctx.box.delivered = True
        return None, None


def skip_tap(x):
    """`x` for the late consumer of a skip connection: if an early consumer has left a box on the tensor, the gradient
    this use produces goes into the box (and not to autograd's accumulation); else `x` as it is."""
    box = getattr(x, "_dm_grad_box", None)
    if box is None or box.tapped or not torch.is_grad_enabled():
        return x
    return _SkipTapFn.apply(x, box)


class FusedNorm:
    """The GroupNorm (+ scale/shift or pre-add) (+ SiLU) that consumes a convolution's output, handed to conv2d so
    that layers whose output tiles hold whole samples and groups produce it in the convolution's epilogue."""

    def __init__(self, gamma, beta, groups, eps, act, scale_shift=None, pre_add=None):
        self.gamma, self.beta, self.groups, self.eps, self.act = gamma, beta, int(groups), float(eps), bool(act)
        self.scale_shift, self.pre_add = scale_shift, pre_add


class _ConvFn(Function):
    """y = conv2d(cat(x1, x2) [nearest-upsampled x2 if ups], weight) + bias + res
    Returns (y, y_norm, mean, rstd): the last three are the fused GroupNorm's results when `norm` is given and the
    layer qualifies, else None."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, res, stride, pad, ups, pack, bias_grad=True, bias_twin=None, norm=None,
                box=None, raw=True):
        _require_device(x1, "conv2d")
        x1 = as_nhwc(x1)
        x2 = as_nhwc(x2) if x2 is not None else None
        res = as_nhwc(res) if res is not None else None
        w4 = weight if weight.dim() == 4 else weight[:, :, None, None]
        Cout, Cin, KH, KW = w4.shape
        B, C1, Hs, Ws = x1.shape
        C2 = x2.shape[1] if x2 is not None else 0
        if C1 + C2 != Cin:
            raise RuntimeError(f"conv2d: input has {C1 + C2} channels, weight expects {Cin}")
        Hi, Wi = Hs << ups, Ws << ups
        Ho = (Hi + 2 * pad - KH) // stride + 1
        Wo = (Wi + 2 * pad - KW) // stride + 1
        w_fwd, _ = pack.get(weight)
        b32 = f32c(bias.detach()) if bias is not None else None
        d = _conv_desc(B, Hs, Ws, C1, C2, Cout, KH, KW, stride, pad, ups, 0, Ho, Wo)
        y = y_norm = mean = rstd = None
        if norm is not None:
            ss, pre = norm.scale_shift, norm.pre_add
            g = GnDesc(B, Ho * Wo, Cout, 0, norm.groups, norm.eps, int(norm.act), int(ss is not None),
                       int(pre is not None))
            if not lib.dm_conv2d_gn_fusable(ctypes.byref(d), ctypes.byref(g)):
                norm = None
        if norm is not None:
            ss32 = f32c(ss.detach()) if ss is not None else None
            pre32 = f32c(pre.detach()) if pre is not None else None
            if ss32 is not None and tuple(ss32.shape) != (B, 2 * Cout):
                raise RuntimeError(f"group_norm: scale_shift must be [{B}, {2 * Cout}], got {tuple(ss32.shape)}")
            if pre32 is not None and tuple(pre32.shape) != (B, Cout):
                raise RuntimeError(f"group_norm: pre_add must be [{B}, {Cout}], got {tuple(pre32.shape)}")
            y_norm = empty_nhwc(B, Cout, Ho, Wo, x1.device)
            if raw:   # the raw output and the statistics: what the backward pass will read
                y = empty_nhwc(B, Cout, Ho, Wo, x1.device)
                mean = torch.empty((B, norm.groups), dtype=torch.float32, device=x1.device)
                rstd = torch.empty((B, norm.groups), dtype=torch.float32, device=x1.device)
            check(lib.dm_conv2d_igemm_gn(ctypes.byref(d), x1.data_ptr(), _p(x2), w_fwd.data_ptr(), _p(b32), _p(res),
                                         _p(y), ctypes.byref(g), f32c(norm.gamma.detach()).data_ptr(),
                                         f32c(norm.beta.detach()).data_ptr(), _p(ss32), _p(pre32), y_norm.data_ptr(),
                                         _p(mean), _p(rstd), _stream()))
            ctx.mark_non_differentiable(*[t for t in (y_norm, mean, rstd) if t is not None])
            ctx.set_materialize_grads(False)    # (no zero tensors made for the gradients of those three)
        else:
            y = empty_nhwc(B, Cout, Ho, Wo, x1.device)
            check(lib.dm_conv2d_igemm(ctypes.byref(d), x1.data_ptr(), _p(x2), w_fwd.data_ptr(), _p(b32), _p(res),
                                      y.data_ptr(), _stream()))
        ctx.save_for_backward(x1, x2, weight)
        ctx.params = (weight, bias)
        ctx.bias_twin = bias_twin
        if box is not None and (ups or x2 is not None):
            raise RuntimeError("conv2d: a gradient box needs a single, un-upsampled input")
        ctx.box = box
        ctx.cfg = (stride, pad, ups, pack, bias is not None and bias_grad, res is not None,
                   (B, C1, C2, Hs, Ws, Hi, Wi, Ho, Wo))
        return y, y_norm, mean, rstd

    @staticmethod
    def backward(ctx, dy, *_unused):
        x1, x2, weight = ctx.saved_tensors
        dx1, dx2, dw, db, dres = _conv_backward(x1, x2, weight, ctx.params, ctx.bias_twin, ctx.box, ctx.cfg, dy,
                                                ctx.needs_input_grad[:5])
        return dx1, dx2, dw, db, dres, None, None, None, None, None, None, None, None, None


def _conv_backward(x1, x2, weight, params, bias_twin, box, cfg, dy, needs):
    """Backward of one convolution y = conv(x1 | x2) + bias (+ res): the data gradient on the current stream, the
    weight / bias gradients on a side stream when the caller set that mode up. needs: which of (x1, x2, weight, bias,
    res) want a gradient. Returns (dx1, dx2, dw, db, dres); dw / db are None when the gradient went straight into the
    parameter's gradient buffer."""
    stride, pad, ups, pack, has_bias, has_res, dims = cfg
    B, C1, C2, Hs, Ws, Hi, Wi, Ho, Wo = dims
    w4 = weight if weight.dim() == 4 else weight[:, :, None, None]
    Cout, Cin, KH, KW = w4.shape
    if dy is None:
        dy = torch.zeros((B, Cout, Ho, Wo), dtype=BF16, device=x1.device)
    dy = as_nhwc(dy)
    st = _stream()
    dx1 = dx2 = dw = db = None
    done = []
    wparam, bparam = params
    wsink = _grad_sink(wparam, ohwi=True) if needs[2] else None
    want_db = has_bias and needs[3]
    bsink = _grad_sink(bparam) if want_db else None
    on_side = (wsink is not None or not needs[2]) and (bsink is not None or not want_db)
    st_w, stream_w = _param_stream((dy, x1, x2), on_side)   # dy (and the saved inputs) are complete
    if needs[0] or needs[1]:
        _, w_dgrad = pack.get(weight)
        dxf = empty_nhwc(B, Cin, Hi, Wi, dy.device)
        d = _conv_desc(B, Ho, Wo, Cout, 0, Cin, KH, KW, stride, pad, 0, 1, Hi, Wi)
        # the gradient a skip connection's late consumer left for this input rides the epilogue's residual add
        late = box.take((B, Cin, Hi, Wi)) if box is not None else None
        check(lib.dm_conv2d_igemm(ctypes.byref(d), dy.data_ptr(), None, w_dgrad.data_ptr(), None, _p(late),
                                  dxf.data_ptr(), st))
        if ups:
            dxs = empty_nhwc(B, Cin, Hs, Ws, dy.device)
            check(lib.dm_upsample2x_bwd(dxf.data_ptr(), dxs.data_ptr(), B, Hs, Ws, Cin, st))
            dxf = dxs
        if x2 is None:
            dx1 = dxf
        else:
            dx1, dx2 = dxf[:, :C1], dxf[:, C1:]
    if needs[2]:
        dwf = wsink if wsink is not None else torch.zeros((Cout, KH, KW, Cin), dtype=torch.float32, device=dy.device)
        d = _conv_desc(B, Hs, Ws, C1, C2, Cout, KH, KW, stride, pad, ups, 0, Ho, Wo)
        # split-K partial tiles go through a scratch workspace (no float atomics, order-independent sums),
        # allocated on the stream that runs the kernel
        if wsink is not None and PENDING_WGRADS is not None:
            PENDING_WGRADS.add(d, x1, x2, dy, dwf, wparam)   # launched with its neighbours; reports the parameter then
        else:
            ws_bytes = lib.dm_conv2d_wgrad_ws_bytes(ctypes.byref(d))
            ws = _scratch(dy.device, (ws_bytes + 3) // 4, stream_w)
            check(lib.dm_conv2d_wgrad(ctypes.byref(d), x1.data_ptr(), _p(x2), dy.data_ptr(), dwf.data_ptr(),
                                      ws.data_ptr(), ws.numel() * 4, st_w))
            if wsink is not None:
                done.append(wparam)
        if wsink is None:
            dw = dwf.permute(0, 3, 1, 2)
            if weight.dim() == 2:
                dw = dw.reshape(Cout, Cin)
    if want_db:
        dbf = bsink if bsink is not None else torch.zeros(Cout, dtype=torch.float32, device=dy.device)
        twin = bias_twin
        tsink = _grad_sink(twin) if (twin is not None and bsink is not None) else None
        M = B * Ho * Wo
        cws = _scratch(dy.device, lib.dm_colsum_ws_floats(M, Cout), stream_w)
        if tsink is not None:
            # the twin's convolution produced the other term of the same sum (and ran with bias_grad=False)
            check(lib.dm_colsum_bf16_dual(dy.data_ptr(), dbf.data_ptr(), tsink.data_ptr(), M, Cout, cws.data_ptr(),
                                          _deferred(True, cws), st_w))
            done.append(twin)
        else:
            check(lib.dm_colsum_bf16(dy.data_ptr(), dbf.data_ptr(), M, Cout, cws.data_ptr(),
                                     _deferred(bsink is not None and twin is None, cws), st_w))
            if twin is not None:
                g = dbf.detach().to(twin.dtype)
                if twin.grad is None:
                    twin.grad = g.clone()
                else:
                    twin.grad.add_(g)   # in place: a gradient that is a view of a flat arena stays one
        if bsink is not None:
            done.append(bparam)
        else:
            db = dbf
    # reported only now, after this operator's last launch: a listener may close a HIP-graph capture at such a
    # report, and the side stream forked above belongs to the capture that was open then
    for prm in done:
        _grad_done(prm)
    if PENDING_WGRADS is not None:
        PENDING_WGRADS.flush_if_full()
    dres = dy if (has_res and needs[4]) else None
    if dres is not None and (stream_w is not None or SIDE_WGRAD):
        SIDE_KEEP.append(dy)
    return dx1, dx2, dw, db, dres


def conv2d(x1, weight, bias, pack, x2=None, res=None, stride=1, pad=0, ups=0, bias_grad=True, bias_twin=None,
           norm=None, box=None):
    """bias_twin: the bias parameter of another convolution whose output is added to this one's (run with
    bias_grad=False): it receives this bias's gradient, the same column sums, from the same launch.
    bias_grad=False: the bias is added but its gradient is left to the consumer of the output (group_norm_act's
    upstream_bias), which gets it for free.
    norm: a FusedNorm describing the GroupNorm that consumes the output. The call then returns (y, done) where
    done is (y_norm, mean, rstd) from the convolution's epilogue, to be passed to group_norm_act as `precomputed`,
    or None when the layer does not qualify and the GroupNorm has to run as a launch of its own.
    With gradients switched off (sampling) a fused launch writes neither y nor the statistics, which only a backward
    pass reads: the call returns (None, (y_norm, None, None)).
    box: a GradBox made for x1 (grad_box): the gradient it holds at backward time is added to x1's gradient in the
    data-gradient kernel's epilogue."""
    y, y_norm, mean, rstd = _ConvFn.apply(x1, x2, weight, bias, res, stride, pad, ups, pack, bias_grad, bias_twin, norm,
                                          box, torch.is_grad_enabled())
    if norm is None:
        return y
    return y, ((y_norm, mean, rstd) if y_norm is not None else None)


class _ConvProjFn(Function):
    """y = conv3x3(h, weight) + bias + conv1x1(r1 | r2, wr) + br in ONE launch (dm_conv2d_igemm_proj): a ConvBlock's
    exit when the block changes the channel count. Backward: the two convolutions' own backward passes."""

    @staticmethod
    def forward(ctx, h, weight, bias, pack, r1, r2, wr, br, packr):
        h, r1 = as_nhwc(h), as_nhwc(r1)
        r2 = as_nhwc(r2) if r2 is not None else None
        B, C, H, W = h.shape
        Cout = weight.shape[0]
        Cr1, Cr2 = r1.shape[1], (r2.shape[1] if r2 is not None else 0)
        w_fwd, _ = pack.get(weight)
        wr_fwd, _ = packr.get(wr)
        y = empty_nhwc(B, Cout, H, W, h.device)
        b32 = f32c(bias.detach()) if bias is not None else None
        br32 = f32c(br.detach()) if br is not None else None
        d = _conv_desc(B, H, W, C, 0, Cout, 3, 3, 1, 1, 0, 0, H, W)
        proj = ConvProj(r1.data_ptr(), _p(r2), wr_fwd.data_ptr(), _p(br32), Cr1, Cr2)
        check(lib.dm_conv2d_igemm_proj(ctypes.byref(d), h.data_ptr(), None, w_fwd.data_ptr(), _p(b32), ctypes.byref(proj),
                                       y.data_ptr(), _stream()))
        ctx.save_for_backward(h, r1, r2, weight, wr)
        ctx.params = (weight, bias, wr, br)
        ctx.packs = (pack, packr)
        ctx.dims = (B, C, Cout, Cr1, Cr2, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, r1, r2, weight, wr = ctx.saved_tensors
        _, bias, _, br = ctx.params
        pack, packr = ctx.packs
        B, C, Cout, Cr1, Cr2, H, W = ctx.dims
        need = ctx.needs_input_grad
        # one launch of column sums over dy serves both biases (the two convolutions' outputs are the terms of one sum)
        twin = br if (br is not None and bias is not None and need[7] and need[2]) else None
        dh, _, dw, db, _ = _conv_backward(h, None, weight, (weight, bias), twin, None,
                                          (1, 1, 0, pack, bias is not None, False, (B, C, 0, H, W, H, W, H, W)), dy,
                                          (need[0], False, need[1], need[2], False))
        dr1, dr2, dwr, dbr, _ = _conv_backward(r1, r2, wr, (wr, br), None, None,
                                               (1, 0, 0, packr, br is not None and twin is None, False,
                                                (B, Cr1, Cr2, H, W, H, W, H, W)), dy,
                                               (need[4], need[5] and r2 is not None, need[6], need[7], False))
        return dh, dw, db, None, dr1, dr2, dwr, dbr, None


def conv2d_proj(h, weight, bias, pack, r1, r2, wr, br, packr):
    """conv3x3(h) + bias + conv1x1(r1 | r2) + br in one launch where the layer allows it (dm_conv2d_proj_fusable);
    returns None where it does not, and the caller runs the two convolutions one after the other."""
    if weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) or not h.is_cuda:
        return None
    B, C, H, W = h.shape
    d = _conv_desc(B, H, W, C, 0, weight.shape[0], 3, 3, 1, 1, 0, 0, H, W)
    proj = ConvProj(None, None, None, None, r1.shape[1], r2.shape[1] if r2 is not None else 0)
    if tuple(r1.shape[2:]) != (H, W) or not lib.dm_conv2d_proj_fusable(ctypes.byref(d), ctypes.byref(proj)):
        return None
    return _ConvProjFn.apply(h, weight, bias, pack, r1, r2, wr, br, packr)


# ------------------------------------------------------------------------------------------------------------
class _GroupNormActFn(Function):
    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, ss, pre, G, eps, act, fork, upstream_bias, precomputed=None, box=None):
        _require_device(x1, "group_norm")
        if box is not None and (x2 is not None or upstream_bias is not None):
            raise RuntimeError("group_norm: a gradient box needs a single input and no upstream_bias")
        ctx.box = box
        x1 = as_nhwc(x1)
        x2 = as_nhwc(x2) if x2 is not None else None
        B, C1, H, W = x1.shape
        C2 = x2.shape[1] if x2 is not None else 0
        C = C1 + C2
        g32, b32 = f32c(gamma.detach()), f32c(beta.detach())
        ss32 = f32c(ss.detach()) if ss is not None else None
        pre32 = f32c(pre.detach()) if pre is not None else None
        if ss32 is not None and tuple(ss32.shape) != (B, 2 * C):
            raise RuntimeError(f"group_norm: scale_shift must be [{B}, {2 * C}], got {tuple(ss32.shape)}")
        if pre32 is not None and tuple(pre32.shape) != (B, C):
            raise RuntimeError(f"group_norm: pre_add must be [{B}, {C}], got {tuple(pre32.shape)}")
        d = GnDesc(B, H * W, C1, C2, G, eps, int(act), int(ss32 is not None), int(pre32 is not None))
        if precomputed is not None:
            # the convolution that produced x1 has normalised it in its epilogue (conv2d's `norm`)
            y, mean, rstd = precomputed
            if x2 is not None or tuple(y.shape) != (B, C, H, W) or tuple(mean.shape) != (B, G):
                raise RuntimeError("group_norm: precomputed results do not belong to this input")
            y = y.detach()
        else:
            y = empty_nhwc(B, C, H, W, x1.device)
            mean = torch.empty((B, G), dtype=torch.float32, device=x1.device)
            rstd = torch.empty((B, G), dtype=torch.float32, device=x1.device)
            ws = _scratch(x1.device, lib.dm_groupnorm_ws_floats(ctypes.byref(d)))
            check(lib.dm_groupnorm_act_fwd(ctypes.byref(d), x1.data_ptr(), _p(x2), g32.data_ptr(), b32.data_ptr(),
                                           _p(ss32), _p(pre32), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                           ws.data_ptr(), _stream()))
        ctx.save_for_backward(x1, x2, g32, b32, ss32, pre32, mean, rstd)
        ctx.params = (gamma, beta, upstream_bias)
        ctx.cfg = (d, B, C1, C2, H, W)
        if fork:
            # Hand the inputs out again for a second consumer (the residual branch): its gradient then arrives here
            # and is added inside the backward kernel instead of by a separate accumulation pass.
            return y, x1.view_as(x1), (x2.view_as(x2) if x2 is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy, da1=None, da2=None):
        x1, x2, g32, b32, ss32, pre32, mean, rstd = ctx.saved_tensors
        d, B, C1, C2, H, W = ctx.cfg
        C = C1 + C2
        dev = x1.device
        if dy is None:
            dy = torch.zeros((B, C, H, W), dtype=BF16, device=dev).contiguous(memory_format=CL)
        dy = as_nhwc(dy)
        dextra = None
        if da1 is not None or da2 is not None:
            if x2 is None:
                dextra = as_nhwc(da1)
            else:
                if da1 is None:
                    da1 = torch.zeros_like(x1)
                if da2 is None:
                    da2 = torch.zeros_like(x2)
                full = (H * W * C, 1, W * C, C)
                if (da1.dtype == BF16 and da2.dtype == BF16 and da1.stride() == full and da2.stride() == full
                        and da2.data_ptr() == da1.data_ptr() + 2 * C1):
                    dextra = da1  # the two gradients are the channel slices of one dense [B][HW][C] buffer
                else:
                    dextra = as_nhwc(torch.cat([da1, da2], dim=1))
        # the gradient a skip connection's late consumer left for x1: a second addend of the backward kernel
        dextra2 = ctx.box.take((B, C1, H, W)) if ctx.box is not None else None
        if dextra is None:
            dextra, dextra2 = dextra2, None
        dx1 = empty_nhwc(B, C1, H, W, dev)
        dx2 = empty_nhwc(B, C2, H, W, dev) if x2 is not None else None
        gs, bs = _grad_sink(ctx.params[0]), _grad_sink(ctx.params[1])
        direct = gs is not None and bs is not None
        if direct:
            dg, db_ = gs, bs
        else:
            dgb = torch.zeros((2, C), dtype=torch.float32, device=dev)
            dg, db_ = dgb[0], dgb[1]
        # bias gradient of the convolution that produced x1 (= sum of dx), when asked for
        ub = ctx.params[2]
        dub = ub_sink = None
        if ub is not None and ctx.needs_input_grad[10]:
            if dextra is not None or x2 is not None:
                raise RuntimeError("group_norm: upstream_bias needs a single, unforked input")
            ub_sink = _grad_sink(ub)
            dub = ub_sink if ub_sink is not None else torch.zeros(C, dtype=torch.float32, device=dev)
        dss = torch.empty((B, 2 * C), dtype=torch.float32, device=dev) if ss32 is not None else None
        dpre = torch.empty((B, C), dtype=torch.float32, device=dev) if pre32 is not None else None
        ws = _scratch(dev, lib.dm_groupnorm_ws_floats(ctypes.byref(d)))
        partials = torch.empty((B, 3, C), dtype=torch.float32, device=dev)
        check(lib.dm_groupnorm_act_bwd(ctypes.byref(d), x1.data_ptr(), _p(x2), g32.data_ptr(), b32.data_ptr(),
                                       _p(ss32), _p(pre32), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr(),
                                       _p(dextra), _p(dextra2), dx1.data_ptr(), _p(dx2), _p(dss), _p(dpre), partials.data_ptr(),
                                       int(dub is not None), ws.data_ptr(), _stream()))
        # the batch sums of the per-sample terms feed nothing but the optimizer: collected for one grouped launch
        # after backward when the gradients go straight to their buffers, else summed right here
        all_direct = direct and (dub is None or ub_sink is not None)
        check(lib.dm_groupnorm_param_grads(ctypes.byref(d), partials.data_ptr(), dg.data_ptr(), db_.data_ptr(),
                                           _p(dub), _deferred(all_direct, partials), _stream()))
        if ub_sink is not None:
            _grad_done(ub)
            dub = None
        if direct:
            _grad_done(ctx.params[0])
            _grad_done(ctx.params[1])
            return dx1, dx2, None, None, dss, dpre, None, None, None, None, dub, None, None
        return dx1, dx2, dg, db_, dss, dpre, None, None, None, None, dub, None, None


def group_norm_act(x1, gamma, beta, groups, eps, act, x2=None, scale_shift=None, pre_add=None, fork=False,
                   upstream_bias=None, precomputed=None, box=None):
    """With fork=True returns (y, x1', x2'): x1'/x2' are the inputs again, to be used by a second consumer whose
    gradient is then folded into this operator's backward kernel.
    upstream_bias: the bias parameter of the convolution whose output x1 is (and that was run with bias_grad=False):
    its gradient, the sum of dx over samples and pixels, falls out of this operator's backward sums.
    precomputed: (y, mean, rstd) that conv2d(..., norm=FusedNorm(same arguments)) returned with x1: the forward
    launch is skipped, the backward pass is the same.
    box: a GradBox made for x1 (grad_box): the gradient it holds at backward time is added to dx1 inside the backward
    kernel."""
    return _GroupNormActFn.apply(x1, x2, gamma, beta, scale_shift, pre_add, groups, eps, act, fork, upstream_bias,
                                 precomputed, box)


# ------------------------------------------------------------------------------------------------------------
class _AttentionFn(Function):
    @staticmethod
    def forward(ctx, qkv, H, D, scale):
        _require_device(qkv, "attention")
        if qkv.dtype != BF16 or not qkv.is_contiguous():
            qkv = qkv.to(BF16).contiguous()
        B, L, F = qkv.shape
        if F != 3 * H * D:
            raise RuntimeError(f"attention: packed qkv has {F} features, expected 3*{H}*{D}")
        out = torch.empty((B, L, H * D), dtype=BF16, device=qkv.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
        check(lib.dm_attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, L, H, D, scale, _stream()))
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (B, L, H, D, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, L, H, D, scale = ctx.cfg
        if dout.dtype != BF16 or not dout.is_contiguous():
            dout = dout.to(BF16).contiguous()
        dqkv = torch.empty_like(qkv)
        check(lib.dm_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(),
                                   B, L, H, D, scale, _stream()))
        return dqkv, None, None, None


def attention(qkv, n_heads, head_dim, scale=None):
    return _AttentionFn.apply(qkv, n_heads, head_dim, float(head_dim ** -0.5 if scale is None else scale))


# ------------------------------------------------------------------------------------------------------------
class _LinearFn(Function):
    """fp32 embedding-path linear: y = f(x) @ w.T + b, f = SiLU if act_in"""

    @staticmethod
    def forward(ctx, x, weight, bias, act_in):
        _require_device(x, "linear")
        x32, w32 = f32c(x), f32c(weight.detach())
        b32 = f32c(bias.detach()) if bias is not None else None
        M, K = x32.shape
        N = w32.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        check(lib.dm_linear_fwd(x32.data_ptr(), w32.data_ptr(), _p(b32), y.data_ptr(), M, K, N, int(act_in), _stream()))
        ctx.save_for_backward(x32, w32)
        ctx.params = (weight, bias)
        ctx.cfg = (M, K, N, int(act_in), bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x32, w32 = ctx.saved_tensors
        M, K, N, act_in, has_bias = ctx.cfg
        dy = f32c(dy)
        dev = dy.device
        dx = torch.empty((M, K), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        wparam, bparam = ctx.params
        dw = db = ws = bs = None
        if ctx.needs_input_grad[1]:
            ws = _grad_sink(wparam)
            dw = ws if ws is not None else torch.zeros((N, K), dtype=torch.float32, device=dev)
        if has_bias and ctx.needs_input_grad[2]:
            bs = _grad_sink(bparam)
            db = bs if bs is not None else torch.zeros((N,), dtype=torch.float32, device=dev)
        check(lib.dm_linear_bwd(x32.data_ptr(), w32.data_ptr(), dy.data_ptr(), _p(dx), _p(dw), _p(db), M, K, N, act_in,
                                _stream()))
        if ws is not None:
            _grad_done(wparam)
            dw = None
        if bs is not None:
            _grad_done(bparam)
            db = None
        return dx, dw, db, None


def linear(x, weight, bias, act_in=False):
    return _LinearFn.apply(x, weight, bias, act_in)


class _LinearGroupFn(Function):
    """P fp32 linears sharing one input, one launch each way: y_p = f(x) @ w_p.T + b_p.
    apply(x, act_in, w_0, b_0, w_1, b_1, ...) -> (y_0, y_1, ...)"""

    @staticmethod
    def forward(ctx, x, act_in, *wb):
        _require_device(x, "linear_group")
        x32 = f32c(x)
        M, K = x32.shape
        ws, bs = wb[0::2], wb[1::2]
        P = len(ws)
        w32 = [f32c(w.detach()) for w in ws]
        b32 = [f32c(b.detach()) if b is not None else None for b in bs]
        ys = [torch.empty((M, w.shape[0]), dtype=torch.float32, device=x.device) for w in w32]
        items = (LinearItem * P)()
        for i in range(P):
            items[i].w, items[i].b, items[i].y, items[i].N = w32[i].data_ptr(), _p(b32[i]), ys[i].data_ptr(), w32[i].shape[0]
        check(lib.dm_linear_group_fwd(x32.data_ptr(), items, P, M, K, int(act_in), _stream()))
        ctx.save_for_backward(x32, *w32)
        ctx.params = (ws, bs)
        ctx.cfg = (M, K, int(act_in))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x32, *w32 = ctx.saved_tensors
        ws, bs = ctx.params
        M, K, act_in = ctx.cfg
        P = len(w32)
        dev = x32.device
        dx = torch.empty((M, K), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        items = (LinearItem * P)()
        keep, grads, sinks = [], [None] * (2 * P), []
        for i in range(P):
            N = w32[i].shape[0]
            dy = f32c(dys[i]) if dys[i] is not None else torch.zeros((M, N), dtype=torch.float32, device=dev)
            keep.append(dy)
            wsink = _grad_sink(ws[i]) if ctx.needs_input_grad[2 + 2 * i] else None
            dw = wsink if wsink is not None else (torch.zeros((N, K), dtype=torch.float32, device=dev)
                                                  if ctx.needs_input_grad[2 + 2 * i] else None)
            bsink = db = None
            if bs[i] is not None and ctx.needs_input_grad[3 + 2 * i]:
                bsink = _grad_sink(bs[i])
                db = bsink if bsink is not None else torch.zeros((N,), dtype=torch.float32, device=dev)
            items[i].w, items[i].dy, items[i].dw, items[i].db, items[i].N = w32[i].data_ptr(), dy.data_ptr(), _p(dw), _p(db), N
            if wsink is None:
                grads[2 * i] = dw
            else:
                sinks.append(ws[i])
            if bsink is None:
                grads[2 * i + 1] = db
            else:
                sinks.append(bs[i])
        scratch = None
        if dx is not None:
            scratch = torch.empty(lib.dm_linear_group_bwd_ws_floats(items, P, M, K), dtype=torch.float32, device=dev)
        check(lib.dm_linear_group_bwd(x32.data_ptr(), items, P, _p(dx), _p(scratch), M, K, act_in, _stream()))
        for prm in sinks:
            _grad_done(prm)
        return (dx, None) + tuple(grads)


def linear_group(x, layers, act_in=False):
    """layers: sequence of (weight, bias) with a common input width; returns the list of outputs."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    return list(_LinearGroupFn.apply(x, act_in, *flat))


def timestep_embed(t, freq):
    """[sin(t*freq) | cos(t*freq)], fp32 [B, 2*half]; no gradient (t is an integer step index)."""
    _require_device(freq, "timestep_embed")
    t64 = t.to(device=freq.device, dtype=torch.int64).contiguous()
    f32 = f32c(freq)
    B, half = t64.shape[0], f32.shape[0]
    out = torch.empty((B, 2 * half), dtype=torch.float32, device=freq.device)
    check(lib.dm_timestep_embed(t64.data_ptr(), f32.data_ptr(), out.data_ptr(), B, half, _stream()))
    return out


class _ClassEmbedAddFn(Function):
    @staticmethod
    def forward(ctx, emb, table, cls, mask, uncond_idx):
        _require_device(emb, "class_embed_add")
        out = f32c(emb).clone()
        t32 = f32c(table.detach())
        B, E = out.shape
        cls64 = cls.to(torch.int64).contiguous() if cls is not None else None
        m32 = mask.to(torch.float32).contiguous() if mask is not None else None
        check(lib.dm_class_embed_add(out.data_ptr(), t32.data_ptr(), _p(cls64), _p(m32), uncond_idx, B, E, _stream()))
        ctx.save_for_backward(cls64, m32)
        ctx.cfg = (uncond_idx, B, E, tuple(table.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        cls64, m32 = ctx.saved_tensors
        uncond_idx, B, E, tshape = ctx.cfg
        dout = f32c(dout)
        dtable = None
        if ctx.needs_input_grad[1]:
            dtable = torch.zeros(tshape, dtype=torch.float32, device=dout.device)
            check(lib.dm_class_embed_bwd(dout.data_ptr(), dtable.data_ptr(), _p(cls64), _p(m32), uncond_idx, tshape[0],
                                         B, E, _stream()))
        return dout, dtable, None, None, None


def class_embed_add(emb, table, cls, mask, uncond_idx):
    return _ClassEmbedAddFn.apply(emb, table, cls, mask, uncond_idx)


# ------------------------------------------------------------------------------------------------------------
class _InConvFn(Function):
    """3x3 'same' conv from an NCHW fp32 image to pixel-major bf16 features."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_device(x, "in_conv")
        x32, w32 = f32c(x), f32c(weight.detach())
        b32 = f32c(bias.detach()) if bias is not None else None
        B, Cs, H, W = x32.shape
        Cb = w32.shape[0]
        y = empty_nhwc(B, Cb, H, W, x.device)
        check(lib.dm_in_conv_fwd(x32.data_ptr(), w32.data_ptr(), _p(b32), y.data_ptr(), B, Cs, H, W, Cb, _stream()))
        ctx.save_for_backward(x32)
        ctx.params = (weight, bias)
        ctx.cfg = (B, Cs, H, W, Cb, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x32,) = ctx.saved_tensors
        B, Cs, H, W, Cb, has_bias = ctx.cfg
        dy = as_nhwc(dy)
        weight, bias = ctx.params
        dx = None
        if ctx.needs_input_grad[0]:
            # (training never asks for it; guidance or inversion through the network does.) The data gradient of a 3x3
            # 'same' convolution is the convolution of dy with the kernels flipped and the channel roles swapped: the
            # features -> image kernel of the last layer
            w_t = f32c(weight.detach().permute(1, 0, 2, 3).flip(2, 3))
            dx = torch.empty(B, Cs, H, W, dtype=torch.float32, device=dy.device)
            check(lib.dm_out_conv_fwd(dy.data_ptr(), w_t.data_ptr(), None, dx.data_ptr(), B, Cs, H, W, Cb, _stream()))
        dw, db = _edge_wgrad(x32, dy, weight, bias, (B, Cs, H, W, Cb), True, (x32, dy))
        return dx, dw, db


def _edge_wgrad(small, big, weight, bias, shape, big_is_output, wait_for):
    """Weight/bias gradient of an edge convolution: straight into the parameters' gradient buffers and on the side
    stream when the caller set those modes up (as the other convolutions do), else as fresh tensors for autograd."""
    B, Cs, H, W, Cb = shape
    dev = big.device
    wsink = _grad_sink(weight)
    bsink = _grad_sink(bias) if bias is not None else None
    direct = wsink is not None and (bias is None or bsink is not None)
    wshape = (Cb, Cs, 3, 3) if big_is_output else (Cs, Cb, 3, 3)
    dw = wsink if direct else torch.zeros(wshape, dtype=torch.float32, device=dev)
    db = None
    if bias is not None:
        db = bsink if direct else torch.zeros((Cb if big_is_output else Cs,), dtype=torch.float32, device=dev)
    st, stream_w = _param_stream(wait_for, direct)
    ws = _scratch(dev, lib.dm_edge_conv_wgrad_ws_floats(B, Cs, H, W, Cb, 1 if big_is_output else 0), stream_w)
    check(lib.dm_edge_conv_wgrad(small.data_ptr(), big.data_ptr(), dw.data_ptr(), _p(db), B, Cs, H, W, Cb,
                                 1 if big_is_output else 0, ws.data_ptr(), _deferred(direct, ws), st))
    if direct:
        _grad_done(weight)
        if bias is not None:
            _grad_done(bias)
        return None, None
    return dw, db


def in_conv(x, weight, bias):
    return _InConvFn.apply(x, weight, bias)


class _OutConvFn(Function):
    """3x3 'same' conv from pixel-major bf16 features to an NCHW fp32 image."""

    @staticmethod
    def forward(ctx, h, weight, bias):
        _require_device(h, "out_conv")
        h = as_nhwc(h)
        w32 = f32c(weight.detach())
        b32 = f32c(bias.detach()) if bias is not None else None
        B, Cb, H, W = h.shape
        Cs = w32.shape[0]
        y = torch.empty((B, Cs, H, W), dtype=torch.float32, device=h.device)
        check(lib.dm_out_conv_fwd(h.data_ptr(), w32.data_ptr(), _p(b32), y.data_ptr(), B, Cs, H, W, Cb, _stream()))
        ctx.save_for_backward(h, w32)
        ctx.params = (weight, bias)
        ctx.cfg = (B, Cs, H, W, Cb, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w32 = ctx.saved_tensors
        B, Cs, H, W, Cb, has_bias = ctx.cfg
        dy = f32c(dy)
        st = _stream()
        dh = dw = db = None
        if ctx.needs_input_grad[0]:
            dh = empty_nhwc(B, Cb, H, W, dy.device)
            check(lib.dm_out_conv_dgrad(dy.data_ptr(), w32.data_ptr(), dh.data_ptr(), B, Cs, H, W, Cb, st))
        if ctx.needs_input_grad[1]:
            weight, bias = ctx.params
            dw, db = _edge_wgrad(dy, h, weight, bias, (B, Cs, H, W, Cb), False, (dy, h))
        return dh, dw, db


def out_conv(h, weight, bias):
    return _OutConvFn.apply(h, weight, bias)


# ------------------------------------------------------------------------------------------------------------
class _Resample2xFn(Function):
    """mode 'up': nearest x2 upsampling; mode 'avg': 2x2 average pooling. bf16 pixel-major."""

    @staticmethod
    def forward(ctx, x, mode):
        _require_device(x, "resample2x")
        x = as_nhwc(x)
        B, C, H, W = x.shape
        st = _stream()
        if mode == "up":
            y = empty_nhwc(B, C, 2 * H, 2 * W, x.device)
            check(lib.dm_upsample2x_fwd(x.data_ptr(), y.data_ptr(), B, H, W, C, st))
        else:
            y = empty_nhwc(B, C, H // 2, W // 2, x.device)
            check(lib.dm_avgpool2x_fwd(x.data_ptr(), y.data_ptr(), B, H, W, C, st))
        ctx.cfg = (mode, B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        mode, B, C, H, W = ctx.cfg
        dy = as_nhwc(dy)
        dx = empty_nhwc(B, C, H, W, dy.device)
        st = _stream()
        if mode == "up":
            check(lib.dm_upsample2x_bwd(dy.data_ptr(), dx.data_ptr(), B, H, W, C, st))
        else:
            check(lib.dm_avgpool2x_bwd(dy.data_ptr(), dx.data_ptr(), B, H, W, C, st))
        return dx, None


def upsample2x(x):
    return _Resample2xFn.apply(x, "up")


def avgpool2x(x):
    return _Resample2xFn.apply(x, "avg")


# ------------------------------------------------------------------------------------------------------------
def diffuse(x, eps, t, tab64=None, tab32=None):
    """x_t = a[t] x + s[t] eps with (a, s) from the variance-preserving table (fp64 alpha_sq) or the
    rectified-flow sigmas (fp32)."""
    _require_device(x, "diffuse")
    x32, e32 = f32c(x), f32c(eps)
    t64 = t.to(device=x.device, dtype=torch.int64).contiguous()
    B = x32.shape[0]
    n = x32.numel() // B
    out = torch.empty_like(x32)
    mode = 0 if tab64 is not None else 1
    check(lib.dm_diffuse(x32.data_ptr(), e32.data_ptr(), t64.data_ptr(), _p(tab64), _p(tab32), out.data_ptr(), B, n,
                         mode, _stream()))
    return out


class _MseLossFn(Function):
    """mean((pred - target)^2); target = eps (mode 0) or x - eps (mode 1)"""

    @staticmethod
    def forward(ctx, pred, x, eps, mode):
        _require_device(pred, "mse_loss")
        p32, e32 = f32c(pred), f32c(eps)
        x32 = f32c(x) if x is not None else None
        partials = torch.empty(lib.dm_mse_partials(), dtype=torch.float32, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        check(lib.dm_mse_loss_fwd(p32.data_ptr(), _p(x32), e32.data_ptr(), partials.data_ptr(), loss.data_ptr(),
                                  p32.numel(), mode, _stream()))
        ctx.save_for_backward(p32, x32, e32)
        ctx.mode = mode
        return loss

    @staticmethod
    def backward(ctx, gout):
        p32, x32, e32 = ctx.saved_tensors
        g = f32c(gout).reshape(1)
        dpred = torch.empty_like(p32)
        check(lib.dm_mse_loss_bwd(p32.data_ptr(), _p(x32), e32.data_ptr(), g.data_ptr(), dpred.data_ptr(), p32.numel(),
                                  ctx.mode, _stream()))
        return dpred, None, None, None


def mse_loss(pred, eps, x=None, mode=0):
    return _MseLossFn.apply(pred, x, eps, mode)


def sampler_step(z, pred, noise, c0, c1, c2, c3, mode, clip, last):
    """In-place sampler update on z (fp32, contiguous)."""
    _require_device(z, "sampler_step")
    if z.dtype != torch.float32 or not z.is_contiguous():
        raise RuntimeError("sampler_step: z must be a contiguous fp32 tensor")
    p32 = f32c(pred)
    n32 = f32c(noise) if noise is not None else None
    check(lib.dm_sampler_step(z.data_ptr(), p32.data_ptr(), _p(n32), c0, c1, c2, c3, mode, int(clip), int(last),
                              z.numel(), _stream()))
    return z


def sampler_step_dev(z, pred, noise, coefs, step, t_table, t_out, n_steps, mode, clip):
    """The sampler update of loop step step[0] with its scalars from the device table coefs [n_steps, 4], followed by
    step[0] += 1 and t_out[:] = t_table[step[0]] (see dm_sampler_step_dev): graph-capturable, no host scalars."""
    _require_device(z, "sampler_step_dev")
    if z.dtype != torch.float32 or not z.is_contiguous():
        raise RuntimeError("sampler_step_dev: z must be a contiguous fp32 tensor")
    if (coefs.dtype != torch.float32 or tuple(coefs.shape) != (n_steps, 4) or not coefs.is_contiguous()
            or t_table.dtype != torch.int64 or t_table.numel() != n_steps or step.dtype != torch.int64
            or t_out.dtype != torch.int64 or t_out.numel() != z.shape[0]):
        raise RuntimeError("sampler_step_dev: coefs must be fp32 [n_steps, 4], t_table int64 [n_steps], step int64 [1], "
                           "t_out int64 [batch]")
    p32 = f32c(pred)
    n32 = f32c(noise) if noise is not None else None
    check(lib.dm_sampler_step_dev(z.data_ptr(), p32.data_ptr(), _p(n32), coefs.data_ptr(), step.data_ptr(),
                                  t_table.data_ptr(), t_out.data_ptr(), n_steps, z.shape[0], mode, int(clip), z.numel(),
                                  _stream()))
    return z


def cfg_combine(pred_uncond, pred_cond, scale):
    _require_device(pred_uncond, "cfg_combine")
    u, c = f32c(pred_uncond), f32c(pred_cond)
    out = torch.empty_like(u)
    check(lib.dm_cfg_combine(u.data_ptr(), c.data_ptr(), out.data_ptr(), float(scale), u.numel(), _stream()))
    return out
