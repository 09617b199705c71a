"""GPU: each libdm_hip operator (called through the C ABI via dm_hip.ops) against a plain PyTorch fp32 CPU
reference of the same op on the same seeded inputs.

Tolerances (relative L2 error against the fp32 reference, written per test): activations are stored in bf16
(8 significant bits, unit round-off 2^-9 = 2e-3), products accumulate in fp32. fp32 kernels are held to ~1e-5;
the diffusion elementwise kernels are bit-exact.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.parity import T, golden, max_abs, rel_err

pytestmark = pytest.mark.gpu

BF16_FWD = 6e-3     # one bf16 rounding of inputs + one of the output
BF16_BWD = 1.2e-2   # gradients pass through one more bf16 tensor


def dev():
    return torch.device("cuda:0")


def q(x):
    """Round to bf16 and back: the reference then sees exactly the values the kernels see."""
    return x.to(torch.bfloat16).to(torch.float32)


def rnd(rs, *shape, scale=1.0):
    return T((rs.standard_normal(shape) * scale).astype(np.float32))


# ------------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, C1, C2, Cout, H, W, k, stride, ups
    (2, 64, 0, 64, 8, 8, 3, 1, 0),
    (3, 128, 0, 128, 16, 16, 3, 1, 0),      # M = 768: six full tiles
    (2, 64, 0, 192, 7, 7, 3, 1, 0),         # ragged M (98), Cout not a multiple of 128
    (2, 128, 64, 128, 8, 8, 1, 1, 0),       # two-source 1x1 (res_proj over a skip concat)
    (2, 64, 128, 64, 5, 9, 3, 1, 0),        # two-source 3x3, non-square
    (2, 64, 0, 64, 16, 16, 3, 2, 0),        # stride 2 pad 1
    (2, 64, 0, 128, 7, 7, 3, 2, 0),         # stride 2 on an odd size (7 -> 4)
    (2, 64, 0, 128, 4, 4, 3, 1, 1),         # fused nearest x2 upsample
    (1, 256, 0, 512, 8, 8, 3, 1, 0),        # deep reduction (K = 2304)
    (5, 64, 0, 64, 1, 1, 1, 1, 0),          # degenerate: a linear layer
]


CONV_CASES += [
    (2, 64, 0, 128, 32, 32, 3, 1, 0),       # tiles of 8 whole rows inside one image
    (3, 128, 0, 128, 16, 16, 3, 1, 0),      # one image per tile
    (9, 64, 64, 64, 8, 8, 3, 1, 0),         # four whole images per tile, ragged last tile, two sources
    (3, 64, 0, 64, 28, 28, 3, 1, 0),        # tiles straddle image boundaries at arbitrary rows
    (11, 128, 0, 64, 7, 7, 3, 1, 0),
    (2, 64, 0, 64, 8, 8, 3, 1, 1),          # fused upsample through the patch path
    (5, 128, 0, 64, 32, 32, 3, 2, 0),       # stride 2 on an even size: the data gradient runs by pixel parity class
    (3, 64, 0, 128, 12, 20, 3, 2, 0),       # ... classes of 90 pixels: a ragged tile per class
]


@pytest.mark.parametrize("patch", [1, 2, 0])
@pytest.mark.parametrize("B,C1,C2,Cout,H,W,k,stride,ups", CONV_CASES)
def test_conv2d_fwd_bwd(B, C1, C2, Cout, H, W, k, stride, ups, patch):
    """patch=1: the kernels a training step takes (patch convolution, nine-tap weight gradient where the shape allows);
    2: the three-tap weight-gradient kernel instead of the nine-tap one; 0: the gathered convolution, the single-tap
    weight gradient and the stride-2 data gradient over all nine taps."""
    from dm_hip import ops, lib
    lib.dm_config_set(b"wgrad_row9", 2 if patch == 1 else 0)   # 2: on every shape that qualifies
    patch = int(patch > 0)
    lib.dm_config_set(b"conv_patch", patch)
    lib.dm_config_set(b"wgrad_row3", patch)   # patch=0 also selects the single-tap weight-gradient kernel
    lib.dm_config_set(b"dgrad_subpixel", patch)   # ... and the stride-2 data gradient over all nine taps
    try:
        _conv2d_case(B, C1, C2, Cout, H, W, k, stride, ups)
    finally:
        lib.dm_config_set(b"conv_patch", 1)
        lib.dm_config_set(b"wgrad_row3", 1)
        lib.dm_config_set(b"wgrad_row9", 3)
        lib.dm_config_set(b"dgrad_subpixel", 1)


@pytest.mark.parametrize("B,C1,C2,Cout,H,W,mode", [
    (2, 128, 0, 256, 16, 16, 2),       # one chunk of input channels, two output tiles
    (3, 128, 128, 128, 9, 11, 2),      # two sources (a chunk each), 297 pixels: a ragged last pixel tile
    (1, 384, 128, 384, 8, 8, 2),       # four chunks through the two weight buffers, three output tiles
    (32, 128, 0, 256, 32, 32, 1),      # the shape class the default takes (>= 32768 pixels, <= 256 input channels)
    (2, 256, 0, 128, 16, 16, 1),       # ... and one it leaves to the gathered kernel (few pixels)
])
def test_conv1x1_streaming(B, C1, C2, Cout, H, W, mode):
    """The streaming 1x1 kernel (activations from global memory straight into MFMA operands, transposed product,
    output from registers) against the gathered kernel on the same operands - bias, residual, two sources - and
    against torch fp32; forward and, through autograd, the data gradient (the same kernel on the transposed weights)."""
    from dm_hip import lib, ops
    rs = np.random.RandomState(B * 7 + Cout + H)
    x1 = q(rnd(rs, B, C1, H, W))
    x2 = q(rnd(rs, B, C2, H, W)) if C2 else None
    w = q(rnd(rs, Cout, C1 + C2, 1, 1, scale=(C1 + C2) ** -0.5))
    bias = 0.1 * rnd(rs, Cout)
    res = q(rnd(rs, B, Cout, H, W))
    gy = q(rnd(rs, B, Cout, H, W))
    d = dev()

    def run(stream):
        lib.dm_config_set(b"conv1x1_stream", stream)
        t1 = x1.to(d).to(torch.bfloat16).requires_grad_(True)
        t2 = x2.to(d).to(torch.bfloat16).requires_grad_(True) if x2 is not None else None
        wd = w.to(d).requires_grad_(True)
        y = ops.conv2d(t1, wd, bias.to(d), ops.WeightPack(), x2=t2, res=res.to(d).to(torch.bfloat16))
        y.backward(gy.to(d).to(torch.bfloat16))
        torch.cuda.synchronize()
        return y.detach().float(), t1.grad.float(), (t2.grad.float() if t2 is not None else None), wd.grad.clone()

    ya, g1a, g2a, gwa = run(0)
    yb, g1b, g2b, gwb = run(mode)
    # same products, fp32 sums in another order: a few outputs fall to the other bf16 neighbour
    for a_, b_ in ((ya, yb), (g1a, g1b)) + (((g2a, g2b),) if g2a is not None else ()):
        diff = (a_ - b_).abs()
        assert float(diff.max()) <= 2.0 ** -7 * max(1.0, float(a_.abs().max()))
        assert float((diff > 0).float().mean()) < 2e-2
    assert torch.equal(gwa, gwb)      # (the weight gradient does not go through this kernel)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(xin, w, bias) + res
    assert rel_err(yb.cpu(), ref) < BF16_FWD
    gref = F.conv_transpose2d(gy, w)
    gx = g1b if g2b is None else torch.cat([g1b, g2b], 1)
    assert rel_err(gx.cpu(), gref) < BF16_FWD


def _conv2d_case(B, C1, C2, Cout, H, W, k, stride, ups):
    from dm_hip import ops
    rs = np.random.RandomState(hash((B, C1, C2, Cout, H, W, k, stride, ups)) % (2 ** 31))
    pad = (k - 1) // 2
    x1 = q(rnd(rs, B, C1, H, W))
    x2 = q(rnd(rs, B, C2, H, W)) if C2 else None
    w = rnd(rs, Cout, C1 + C2, k, k, scale=(1.0 / np.sqrt((C1 + C2) * k * k)))
    b = rnd(rs, Cout, scale=0.1)
    # reference (fp32, CPU) with bf16-rounded weights, as the kernel packs them
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    xin = xin.clone().requires_grad_(True)
    wq = q(w).clone().requires_grad_(True)
    bb = b.clone().requires_grad_(True)
    xu = F.interpolate(xin, scale_factor=2, mode="nearest") if ups else xin
    y_ref = F.conv2d(xu, wq, bb, stride=stride, padding=pad)
    res = q(rnd(rs, *y_ref.shape))
    resr = res.clone().requires_grad_(True)
    y_ref = y_ref + resr
    gy = q(rnd(rs, *y_ref.shape))
    y_ref.backward(gy)

    d = dev()
    x1d = x1.to(d).requires_grad_(True)
    x2d = x2.to(d).requires_grad_(True) if x2 is not None else None
    wd = w.to(d).requires_grad_(True)
    bd = b.to(d).requires_grad_(True)
    resd = res.to(d).requires_grad_(True)
    pack = ops.WeightPack()
    y = ops.conv2d(x1d, wd, bd, pack, x2=x2d, res=resd, stride=stride, pad=pad, ups=ups)
    assert tuple(y.shape) == tuple(y_ref.shape) and y.dtype == torch.bfloat16
    assert rel_err(y.float(), y_ref) < BF16_FWD
    y.backward(gy.to(d))
    gx = x1d.grad.float() if x2d is None else torch.cat([x1d.grad.float(), x2d.grad.float()], 1)
    assert rel_err(gx, xin.grad) < BF16_FWD
    assert rel_err(wd.grad, wq.grad) < 2e-3           # fp32 accumulation of exact bf16 products
    assert rel_err(bd.grad, bb.grad) < 1e-4
    assert rel_err(resd.grad.float(), resr.grad) < 1e-6


@pytest.mark.parametrize("B,C1,C2,Cout,H,W,ups,k,stride", [
    (64, 128, 0, 128, 32, 32, 0, 3, 1),     # many pixel slices per tile
    (32, 64, 64, 192, 16, 16, 0, 3, 1),     # two sources, Cout not a multiple of the 128-channel tile
    (16, 128, 0, 64, 8, 8, 1, 3, 1),        # fused upsample, half-empty output-channel tile
    (2, 64, 0, 64, 8, 8, 0, 3, 1),          # a single slice
    (64, 256, 128, 128, 32, 32, 0, 1, 1),   # 1x1 over two sources: single-tap kernel, 128 x 128 tiles
    (8, 128, 0, 256, 16, 16, 0, 1, 1),
    (16, 64, 0, 64, 16, 16, 0, 1, 1),       # 64 x 64 tiles
    (8, 64, 64, 192, 8, 8, 0, 1, 1),        # 64-wide tiles on both axes, ragged output-channel tile
    (8, 64, 0, 128, 16, 16, 0, 3, 2),       # stride 2: 128 x 64 tiles
    (8, 128, 0, 72, 14, 14, 0, 3, 2),       # 64 x 128 tiles, Cout = 72, 7x7 output
    (4, 64, 0, 64, 28, 28, 0, 3, 1),        # width 28 does not divide 64: single-tap kernel on a 3x3 layer
])
def test_conv2d_wgrad_workspace(B, C1, C2, Cout, H, W, ups, k, stride):
    """The weight gradient (split-K partial tiles to a workspace, summed in slice order by a second kernel) against
    torch's fp32 conv weight gradient on the same bf16-rounded operands; it accumulates into dw, leaves memory
    behind dw alone, does not depend on the workspace's contents and gives bit-identical results from run to run."""
    import ctypes
    from dm_hip import ConvDesc, check, lib
    d_ = dev()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + Cout)
    bf = torch.bfloat16
    pad = (k - 1) // 2
    Hi, Wi = H << ups, W << ups
    Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    Cin = C1 + C2
    x1c = torch.randn(B, H, W, C1, generator=g).to(bf)
    x2c = torch.randn(B, H, W, C2, generator=g).to(bf) if C2 else None
    dyc = torch.randn(B, Ho, Wo, Cout, generator=g).to(bf)
    x1, x2, dy = x1c.to(d_), (x2c.to(d_) if C2 else None), dyc.to(d_)
    desc = ConvDesc(B, H, W, C1, C2, Cout, k, k, stride, pad, ups, 0, Ho, Wo)
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    nbytes = lib.dm_conv2d_wgrad_ws_bytes(ctypes.byref(desc))
    assert nbytes > 0 and nbytes % (64 * 64 * 4) == 0
    # fp32 reference on the CPU: dw[co][kh][kw][ci]
    xin = (x1c if x2c is None else torch.cat([x1c, x2c], -1)).float().permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    wz = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    F.conv2d(xin, wz, None, stride=stride, padding=pad).backward(dyc.float().permute(0, 3, 1, 2))
    want = wz.grad.permute(0, 2, 3, 1).reshape(-1)
    base = torch.randn(Cout * k * k * Cin + 64, generator=g).to(d_)       # dw accumulates; 64 guard floats behind it
    outs = []
    for fill in (0.0, float("nan"), 1.0):                                  # the workspace's contents do not matter
        ws = torch.full((nbytes // 4,), fill, device=d_)
        out = base.clone()
        check(lib.dm_conv2d_wgrad(ctypes.byref(desc), p(x1), p(x2), p(dy), out.data_ptr(), ws.data_ptr(), nbytes, st))
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0][-64:], base[-64:])
    assert rel_err(outs[0][:-64] - base[:-64], want) < 2e-5    # exact bf16 products, fp32 sums in another order
    # too small a workspace is an error, not a silent fallback
    small = torch.empty(16, device=d_)
    assert lib.dm_conv2d_wgrad(ctypes.byref(desc), p(x1), p(x2), p(dy), out.data_ptr(), small.data_ptr(), 64, st) != 0


def test_conv2d_wgrad_group():
    """Several layers' weight gradients in one call: runs of 3x3 / stride-1 layers share one launch (other pixel
    splits than the one-layer calls choose), other layers in the list go out one by one. Every layer against torch's
    fp32 conv weight gradient on the same bf16-rounded operands and against the one-layer entry point; accumulating,
    independent of the workspace's contents, bit-identical from run to run."""
    import ctypes
    from dm_hip import ConvDesc, WgradItem, check, lib
    d_ = dev()
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(99)
    bf = torch.bfloat16
    #          B, C1, C2, Cout, H, W, ups, k, stride
    layers = [(32, 128, 0, 128, 8, 8, 0, 3, 1),
              (32, 64, 64, 192, 8, 8, 0, 3, 1),      # two sources, ragged output-channel tile
              (8, 128, 0, 64, 8, 8, 1, 3, 1),        # fused upsample: 16x16 output
              (8, 64, 0, 128, 16, 16, 0, 1, 1),      # 1x1: not a three-tap layer, splits the run
              (16, 64, 0, 64, 32, 32, 0, 3, 1),
              (4, 64, 0, 64, 28, 28, 0, 3, 1),       # width 28: single-tap kernel on a 3x3 layer
              (8, 256, 0, 128, 16, 16, 0, 3, 1)] + [(4, 64, 0, 64, 8, 8, 0, 3, 1)] * 9   # a run longer than 8
    items = (WgradItem * len(layers))()
    keep, wants, bases, singles = [], [], [], []
    for i, (B, C1, C2, Cout, H, W, ups, k, stride) in enumerate(layers):
        pad = (k - 1) // 2
        Hi, Wi = H << ups, W << ups
        Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
        Cin = C1 + C2
        x1c = torch.randn(B, H, W, C1, generator=g).to(bf)
        x2c = torch.randn(B, H, W, C2, generator=g).to(bf) if C2 else None
        dyc = torch.randn(B, Ho, Wo, Cout, generator=g).to(bf)
        x1, x2, dy = x1c.to(d_), (x2c.to(d_) if C2 else None), dyc.to(d_)
        xin = (x1c if x2c is None else torch.cat([x1c, x2c], -1)).float().permute(0, 3, 1, 2)
        if ups:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        wz = torch.zeros(Cout, Cin, k, k, requires_grad=True)
        F.conv2d(xin, wz, None, stride=stride, padding=pad).backward(dyc.float().permute(0, 3, 1, 2))
        wants.append(wz.grad.permute(0, 2, 3, 1).reshape(-1))
        base = torch.randn(Cout * k * k * Cin + 64, generator=g).to(d_)
        bases.append(base)
        desc = ConvDesc(B, H, W, C1, C2, Cout, k, k, stride, pad, ups, 0, Ho, Wo)
        it = items[i]
        it.desc, it.x1, it.x2, it.dy = desc, x1.data_ptr(), (x2.data_ptr() if C2 else None), dy.data_ptr()
        keep += [x1, x2, dy]
        nb = lib.dm_conv2d_wgrad_ws_bytes(ctypes.byref(desc))
        ws1 = torch.empty(nb // 4, device=d_)
        one = base.clone()
        check(lib.dm_conv2d_wgrad(ctypes.byref(desc), x1.data_ptr(), it.x2, dy.data_ptr(), one.data_ptr(), ws1.data_ptr(), nb, st))
        singles.append(one)
    nbytes = lib.dm_conv2d_wgrad_group_ws_bytes(items, len(layers))
    assert nbytes > 0 and nbytes % 256 == 0
    runs = []
    for fill in (0.0, float("nan")):
        ws = torch.full((nbytes // 4 + 64,), fill, device=d_)
        outs = [b.clone() for b in bases]
        for i, o in enumerate(outs):
            items[i].dw = o.data_ptr()
        check(lib.dm_conv2d_wgrad_group(items, len(layers), (ws.data_ptr() + 255) // 256 * 256, nbytes, st))
        runs.append(outs)
    torch.cuda.synchronize()
    for i in range(len(layers)):
        a, b = runs[0][i], runs[1][i]
        assert torch.equal(a, b), i
        assert torch.equal(a[-64:], bases[i][-64:]), i
        assert rel_err(a[:-64] - bases[i][:-64], wants[i]) < 2e-5, i
        assert rel_err(a[:-64] - bases[i][:-64], singles[i][:-64] - bases[i][:-64]) < 2e-5, i
    assert torch.equal(runs[0][3], singles[3]) and torch.equal(runs[0][5], singles[5])   # launched as the one-layer call does
    assert lib.dm_conv2d_wgrad_group(items, len(layers), (ws.data_ptr() + 255) // 256 * 256, 1024, st) != 0


def test_parameter_gradient_kernels_are_bitwise_reproducible():
    """Every parameter-gradient reduction (GroupNorm affine and upstream bias, conv bias column sums, edge-conv
    weights, class table) sums in a fixed order: two runs on the same inputs must agree bit for bit."""
    from dm_hip import ops
    d = dev()

    def run():
        outs = []
        # GroupNorm, both kernel families (256 and 1024 pixels), with the upstream-bias output
        for H in (16, 32):
            x = q(rnd(rs2, 24, 128, H, H)).to(d).requires_grad_(True)
            ga, be, ub = (rnd(rs2, 128).to(d).requires_grad_(True) for _ in range(3))
            y = ops.group_norm_act(x, ga, be, 32, 1e-5, True, upstream_bias=ub)
            y.backward(q(rnd(rs2, *y.shape)).to(d))
            outs += [ga.grad, be.grad, ub.grad]
        # conv bias (column sums) and weight
        x = q(rnd(rs2, 16, 64, 16, 16)).to(d)
        w, b = rnd(rs2, 64, 64, 3, 3, scale=0.05).to(d).requires_grad_(True), rnd(rs2, 64).to(d).requires_grad_(True)
        y = ops.conv2d(x, w, b, ops.WeightPack(), pad=1)
        y.backward(q(rnd(rs2, *y.shape)).to(d))
        outs += [w.grad, b.grad]
        # edge convolutions (matrix-core kernel at width 16, VALU kernel at width 7)
        for Wd in (16, 7):
            img = rnd(rs2, 8, 3, Wd, Wd).to(d)
            w1, b1 = rnd(rs2, 64, 3, 3, 3).to(d).requires_grad_(True), rnd(rs2, 64).to(d).requires_grad_(True)
            h = ops.in_conv(img, w1, b1)
            w2, b2 = rnd(rs2, 3, 64, 3, 3).to(d).requires_grad_(True), rnd(rs2, 3).to(d).requires_grad_(True)
            o = ops.out_conv(h, w2, b2)
            o.backward(rnd(rs2, *o.shape).to(d))
            outs += [w1.grad, b1.grad, w2.grad, b2.grad]
        # class table
        emb, table = rnd(rs2, 64, 32).to(d).requires_grad_(True), rnd(rs2, 11, 32).to(d).requires_grad_(True)
        cls = T(rs2.randint(0, 10, size=(64,)).astype(np.int64)).to(d)
        ops.class_embed_add(emb, table, cls, None, 10).backward(rnd(rs2, 64, 32).to(d))
        outs.append(table.grad)
        torch.cuda.synchronize()
        return [o.clone() for o in outs]

    rs2 = np.random.RandomState(78)
    a = run()
    rs2 = np.random.RandomState(78)
    b = run()
    assert all(torch.equal(u, v) for u, v in zip(a, b)), [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]


GN_CASES = [
    # B, C1, C2, H, W, act, ss, pre
    (2, 64, 0, 8, 8, True, False, False),
    (2, 128, 0, 4, 4, True, True, False),
    (3, 128, 64, 4, 4, True, False, False),     # group size 6: vectors straddle groups, groups straddle sources
    (2, 256, 128, 8, 8, True, True, False),     # group size 12
    (2, 64, 0, 7, 7, False, False, False),      # plain GroupNorm (attention)
    (2, 128, 0, 5, 3, True, False, True),       # 'addition' embedding mixing
    (2, 512, 512, 2, 2, True, True, False),     # widest concat of the CIFAR model
    (2, 128, 0, 32, 32, True, True, False),     # 1024 pixels: 16 pixels per thread in the single-pass forward
    (3, 256, 0, 16, 16, True, True, False),     # 256 pixels: single-pass forward and backward
    (2, 64, 0, 28, 28, True, False, False),     # 784 pixels: ragged pixel lanes
    (2, 128, 0, 14, 14, False, False, True),    # 196 pixels, pre-add
    (2, 72, 56, 8, 8, True, True, False),       # sources meet inside a 64-channel slab
]


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("B,C1,C2,H,W,act,ss,pre", GN_CASES)
def test_groupnorm_act_fwd_bwd(B, C1, C2, H, W, act, ss, pre, fused):
    from dm_hip import lib
    lib.dm_config_set(b"gn_fused", fused)   # single-pass kernels where they apply / always the two-sweep kernels
    try:
        _groupnorm_case(B, C1, C2, H, W, act, ss, pre)
    finally:
        lib.dm_config_set(b"gn_fused", 1)


def _groupnorm_case(B, C1, C2, H, W, act, ss, pre):
    from dm_hip import ops
    rs = np.random.RandomState(B * 1000 + C1 + C2 + H)
    C, G = C1 + C2, 32
    x1 = q(rnd(rs, B, C1, H, W) + 0.5)
    x2 = q(rnd(rs, B, C2, H, W, scale=2.0)) if C2 else None
    gamma, beta = 1 + 0.2 * rnd(rs, C), 0.2 * rnd(rs, C)
    sst = 0.5 * rnd(rs, B, 2 * C) if ss else None
    pret = rnd(rs, B, C) if pre else None

    leaves = [t.clone().requires_grad_(True) for t in (x1, gamma, beta)]
    x1r, gr, br = leaves
    x2r = x2.clone().requires_grad_(True) if x2 is not None else None
    ssr = sst.clone().requires_grad_(True) if ss else None
    prer = pret.clone().requires_grad_(True) if pre else None
    xin = x1r if x2r is None else torch.cat([x1r, x2r], 1)
    if pre:
        xin = xin + prer[:, :, None, None]
    v = F.group_norm(xin, G, gr, br, eps=1e-5)
    if ss:
        sc, sh = torch.chunk(ssr[:, :, None, None], 2, dim=1)
        v = (1 + sc) * v + sh
    y_ref = F.silu(v) if act else v
    gy = q(rnd(rs, *y_ref.shape))
    y_ref.backward(gy)

    d = dev()
    mk = lambda t: t.to(d).requires_grad_(True) if t is not None else None  # noqa: E731
    x1d, x2d, gd, bd, ssd, pred_ = mk(x1), mk(x2), mk(gamma), mk(beta), mk(sst), mk(pret)
    y = ops.group_norm_act(x1d, gd, bd, G, 1e-5, act, x2=x2d, scale_shift=ssd, pre_add=pred_)
    assert rel_err(y.float(), y_ref) < BF16_FWD
    y.backward(gy.to(d))
    assert rel_err(x1d.grad.float(), x1r.grad) < BF16_FWD
    if x2d is not None:
        assert rel_err(x2d.grad.float(), x2r.grad) < BF16_FWD
    assert rel_err(gd.grad, gr.grad) < 1e-4 and rel_err(bd.grad, br.grad) < 1e-4
    if ss:
        assert rel_err(ssd.grad, ssr.grad) < 1e-4
    if pre:
        assert rel_err(pred_.grad, prer.grad) < 2e-4


CONV_PROJ_CASES = [
    # B, C (3x3 input = output channels of conv1), Cout, Cr1, Cr2, H, W
    (2, 128, 128, 256, 128, 32, 32),      # up.2 block 0 of the CIFAR model: tiles of 8 rows inside an image
    (3, 256, 256, 128, 0, 16, 16),        # down.1 block 0: one image per tile, single-source projection
    (9, 512, 512, 256, 0, 8, 8),          # four images per tile, ragged last tile, 64-wide tiles
    (5, 64, 192, 64, 64, 8, 8),           # Cout not a multiple of 128
    (2, 128, 128, 64, 192, 28, 28),       # tiles straddle image boundaries at arbitrary rows
]


@pytest.mark.parametrize("B,C,Cout,Cr1,Cr2,H,W", CONV_PROJ_CASES)
def test_conv3x3_with_fused_1x1_projection(B, C, Cout, Cr1, Cr2, H, W):
    """modeling/blocks.py:73-75, `conv2(h) + res_proj(x)` as ONE launch (dm_conv2d_igemm_proj) against torch's two
    convolutions in fp32 on the bf16-rounded operands, forward and every gradient; and bit for bit against the two
    launches it replaces where no intermediate rounding separates them (the gradients)."""
    from dm_hip import ops, lib
    rs = np.random.RandomState(B * 100 + C + Cr1 + H)
    h = q(rnd(rs, B, C, H, W))
    r1 = q(rnd(rs, B, Cr1, H, W))
    r2 = q(rnd(rs, B, Cr2, H, W)) if Cr2 else None
    w = rnd(rs, Cout, C, 3, 3, scale=1.0 / np.sqrt(9 * C))
    wr = rnd(rs, Cout, Cr1 + Cr2, 1, 1, scale=1.0 / np.sqrt(Cr1 + Cr2))
    b, br = rnd(rs, Cout, scale=0.1), rnd(rs, Cout, scale=0.1)
    leaves = [t.clone().requires_grad_(True) for t in (h, r1, q(w), b, q(wr), br)]
    hr, r1r, wq, bq, wrq, brq = leaves
    r2r = r2.clone().requires_grad_(True) if r2 is not None else None
    rin = r1r if r2r is None else torch.cat([r1r, r2r], 1)
    y_ref = F.conv2d(hr, wq, bq, padding=1) + F.conv2d(rin, wrq, brq)
    gy = q(rnd(rs, *y_ref.shape))
    y_ref.backward(gy)

    d = dev()
    mk = lambda t: t.to(d).requires_grad_(True) if t is not None else None  # noqa: E731

    def run(fused):
        lib.dm_config_set(b"conv_proj_fuse", int(fused))
        hd, r1d, r2d, wd, bd, wrd, brd = mk(h), mk(r1), mk(r2), mk(w), mk(b), mk(wr), mk(br)
        pack, packr = ops.WeightPack(), ops.WeightPack()
        y = ops.conv2d_proj(hd, wd, bd, pack, r1d, r2d, wrd, brd, packr)
        assert (y is not None) == fused
        if y is None:
            y = ops.conv2d(hd, wd, bd, pack, res=ops.conv2d(r1d, wrd, brd, packr, x2=r2d), pad=1)
        y.backward(gy.to(d))
        return y, [t.grad if t is not None else None for t in (hd, r1d, r2d, wd, bd, wrd, brd)]

    y, grads = run(True)
    y0, grads0 = run(False)
    assert y.dtype == torch.bfloat16 and rel_err(y.float(), y_ref) < BF16_FWD
    # (the separate form rounds the projected residual to bf16 before the sum: the fused result is the closer one)
    assert rel_err(y0.float(), y_ref) < 1.5 * BF16_FWD and rel_err(y.float(), y_ref) <= rel_err(y0.float(), y_ref) * 1.05
    refs = [hr.grad, r1r.grad, r2r.grad if r2r is not None else None, wq.grad, bq.grad, wrq.grad, brq.grad]
    tols = [BF16_FWD, BF16_FWD, BF16_FWD, 2e-3, 1e-4, 2e-3, 1e-4]
    for g1, g0, ref, tol in zip(grads, grads0, refs, tols):
        if ref is None:
            assert g1 is None
            continue
        assert rel_err(g1.float(), ref) < tol
        assert torch.equal(g1, g0)                    # same kernels on the same dy either way
    # shapes the kernel does not take fall back (and say so by returning None)
    assert ops.conv2d_proj(mk(h), mk(w), mk(b), ops.WeightPack(), mk(q(rnd(rs, B, 72, H, W))), None,
                           mk(rnd(rs, Cout, 72, 1, 1)), None, ops.WeightPack()) is None


def test_parameter_used_twice_collects_both_gradient_terms():
    """One GroupNorm affine pair applied twice in a forward pass files two deferred sums for the same gradient buffers:
    the grouped ordered sum must add both (it runs items with overlapping outputs one after the other)."""
    from dm_hip import ops
    rs = np.random.RandomState(77)
    B, C, H, W, G = 3, 64, 8, 8, 32
    x = q(rnd(rs, B, C, H, W))
    gamma, beta = 1 + 0.2 * rnd(rs, C), 0.2 * rnd(rs, C)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    h = F.silu(F.group_norm(xr, G, gr, br, eps=1e-5))
    y_ref = F.silu(F.group_norm(q(h.detach()) + (h - h.detach()), G, gr, br, eps=1e-5))   # (the kernel's bf16 hand-over)
    gy = q(rnd(rs, *y_ref.shape))
    y_ref.backward(gy)
    d = dev()
    xd = x.to(d).requires_grad_(True)
    gd, bd = gamma.to(d).requires_grad_(True), beta.to(d).requires_grad_(True)
    gd.grad, bd.grad = torch.zeros_like(gd), torch.zeros_like(bd)
    sums = ops.SumCollector()
    y = ops.group_norm_act(ops.group_norm_act(xd, gd, bd, G, 1e-5, True), gd, bd, G, 1e-5, True)
    ops.DIRECT_GRADS, ops.PENDING_SUMS = True, sums
    try:
        y.backward(gy.to(d))
        assert sums.count == 2
        sums.flush()
    finally:
        ops.DIRECT_GRADS, ops.PENDING_SUMS = False, None
    assert rel_err(gd.grad, gr.grad) < 5e-3 and rel_err(bd.grad, br.grad) < 5e-3
    # more deferred sums than one block of the collector holds: it grows instead of failing in the middle of backward
    many = ops.SumCollector()
    part = torch.ones(2, 64, device=d)
    outs = [torch.zeros(64, device=d) for _ in range(many.CHUNK + 3)]
    for o in outs:
        ref_ = many.slot(part)
        it = ref_._obj
        it.part, it.stride, it.nparts, it.n, it.s1, it.s2 = part.data_ptr(), 64, 2, 64, 64, 64
        it.out0, it.out1, it.out2, it.dual0 = o.data_ptr(), None, None, None
    many.flush()
    assert all(float(o.sum()) == 128.0 for o in outs)


CONV_GN_CASES = [
    # B, C1, C2, Cout, H, W, ups, act, mix, fusable
    (3, 128, 0, 256, 16, 16, 0, True, "pre", True),     # one sample per tile, 128-wide tiles, four threads per sum
    (9, 256, 0, 256, 8, 8, 0, True, "pre", True),       # four samples per tile, ragged last tile
    (8, 128, 128, 256, 8, 8, 0, True, "ss", True),      # skip concat into the conv, scale/shift mixing
    (5, 64, 0, 64, 16, 16, 0, False, None, True),       # 64-wide tiles, no activation, no embedding
    (4, 128, 0, 128, 8, 8, 1, True, "pre", True),       # through the fused upsampling gather (8x8 -> 16x16)
    (12, 512, 0, 512, 8, 8, 0, True, "pre", True),      # groups of 16 channels, four per 64-wide tile (the 8x8 stage)
    (2, 64, 0, 64, 32, 32, 0, True, "pre", False),      # a sample spans four tiles: separate launches
    (4, 128, 0, 128, 4, 4, 0, True, "ss", False),       # sixteen samples per tile: more sums than threads
    (2, 64, 0, 192, 16, 16, 0, True, None, False),      # 192 channels: groups of 6 do not tile 64 columns evenly
]


@pytest.mark.parametrize("B,C1,C2,Cout,H,W,ups,act,mix,fusable", CONV_GN_CASES)
def test_conv_with_groupnorm_in_its_epilogue(B, C1, C2, Cout, H, W, ups, act, mix, fusable):
    """conv2d(..., norm=FusedNorm) + group_norm_act(precomputed=...) against the same two operators as separate
    launches (conv_gn_fuse=0) and against torch fp32. The raw convolution output is bit-identical; statistics agree
    to fp32 summation order (1e-5), the normalised tensor to one bf16 rounding of values that agree to ~1e-6, the
    gradients (same backward kernels, fed those statistics) to 1e-4 of their norm (1e-3 where they are bf16)."""
    from dm_hip import lib, ops
    rs = np.random.RandomState(B * 100 + Cout + H)
    G = 32
    Cin = C1 + C2
    x1 = q(rnd(rs, B, C1, H, W))
    x2 = q(rnd(rs, B, C2, H, W)) if C2 else None
    w = q(rnd(rs, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5))
    bias = 0.1 * rnd(rs, Cout)
    gamma, beta = 1 + 0.2 * rnd(rs, Cout), 0.2 * rnd(rs, Cout)
    e = {"pre": rnd(rs, B, Cout), "ss": 0.5 * rnd(rs, B, 2 * Cout), None: None}[mix]
    Ho, Wo = H << ups, W << ups
    gy = q(rnd(rs, B, Cout, Ho, Wo))
    d = dev()

    def run(fuse):
        lib.dm_config_set(b"conv_gn_fuse", fuse)
        try:
            mk = lambda t: t.to(d).requires_grad_(True) if t is not None else None  # noqa: E731
            t = dict(x1=mk(x1), x2=mk(x2), w=mk(w), bias=mk(bias), gamma=mk(gamma), beta=mk(beta), e=mk(e))
            kw = {} if mix is None else {"pre_add" if mix == "pre" else "scale_shift": t["e"]}
            pack = ops.WeightPack()
            y, done = ops.conv2d(t["x1"].to(torch.bfloat16), t["w"], t["bias"], pack, x2=None if x2 is None else
                                 t["x2"].to(torch.bfloat16), pad=1, ups=ups, bias_grad=False,
                                 norm=ops.FusedNorm(t["gamma"], t["beta"], G, 1e-5, act, **kw))
            assert (done is not None) == bool(fuse and fusable)
            z = ops.group_norm_act(y, t["gamma"], t["beta"], G, 1e-5, act, upstream_bias=t["bias"],
                                   precomputed=done, **kw)
            z.backward(gy.to(d))
            torch.cuda.synchronize()
            stats = None if done is None else (done[1].clone(), done[2].clone())
            return y.detach(), z.detach(), stats, {k: v.grad for k, v in t.items() if v is not None}
        finally:
            lib.dm_config_set(b"conv_gn_fuse", 1)

    y0, z0, _, g0 = run(0)
    y1, z1, stats, g1 = run(1)
    assert torch.equal(y0, y1)
    if fusable:
        yf = y1.float() + (0 if mix != "pre" else e.to(d)[:, :, None, None])
        grp = yf.reshape(B, G, -1)
        assert rel_err(stats[0], grp.mean(2)) < 1e-5
        assert rel_err(stats[1], (grp.var(2, unbiased=False) + 1e-5).rsqrt()) < 1e-5
        diff = (z0.float() - z1.float()).abs()
        assert float(diff.max()) <= 2.0 ** -7 * max(1.0, float(z0.float().abs().max()))   # at most one bf16 step
        assert float((diff > 0).float().mean()) < 2e-2
        for k in g0:
            # (bf16 gradients: statistics that differ in the last bits flip a rounding here and there)
            assert rel_err(g1[k].float(), g0[k].float()) < (1e-3 if k in ("x1", "x2") else 1e-4), k
    else:
        assert torch.equal(z0, z1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k

    # torch fp32 on the same bf16-rounded operands
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    v = q(F.conv2d(xin, w, bias, padding=1))
    assert rel_err(y1.float().cpu(), v) < BF16_FWD
    if mix == "pre":
        v = v + e[:, :, None, None]
    v = F.group_norm(v, G, gamma, beta, eps=1e-5)
    if mix == "ss":
        sc, sh = torch.chunk(e[:, :, None, None], 2, dim=1)
        v = (1 + sc) * v + sh
    v = F.silu(v) if act else v
    assert rel_err(z1.float().cpu(), v) < 1.5 * BF16_FWD


@pytest.mark.parametrize("H,Cout", [(16, 256), (8, 256)])
def test_conv_groupnorm_epilogue_is_batch_independent(H, Cout):
    """A batch of 128 takes 128-wide output tiles (16x16) or is steered to 64-wide ones to qualify (8x8), a batch of
    3 takes 64-wide tiles: the normalised output and the statistics of a sample are the same bits in both, because
    the sums over a sample's rows follow one order whatever the tile width."""
    from dm_hip import ops
    rs = np.random.RandomState(H)
    B, Cin = 128 if H == 16 else 512, 64
    d = dev()
    x = q(rnd(rs, B, Cin, H, H)).to(d).to(torch.bfloat16)
    w = q(rnd(rs, Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5)).to(d)
    bias, gamma, beta = (0.1 * rnd(rs, Cout)).to(d), (1 + 0.2 * rnd(rs, Cout)).to(d), (0.2 * rnd(rs, Cout)).to(d)
    e = rnd(rs, B, Cout).to(d)
    pack = ops.WeightPack()
    out = []
    for n in (B, 3):
        y, done = ops.conv2d(x[:n].contiguous(memory_format=torch.channels_last), w, bias, pack, pad=1,
                             norm=ops.FusedNorm(gamma, beta, 32, 1e-5, True, pre_add=e[:n].contiguous()))
        assert done is not None
        out.append((y, *done))
    for big, small in zip(*out):
        assert torch.equal(big[:3], small)
    # with gradients off the launch writes the normalised tensor only (nothing reads the raw output or the statistics
    # without a backward pass): same bits
    with torch.no_grad():
        y, done = ops.conv2d(x[:3].contiguous(memory_format=torch.channels_last), w, bias, pack, pad=1,
                             norm=ops.FusedNorm(gamma, beta, 32, 1e-5, True, pre_add=e[:3].contiguous()))
    assert y is None and done[1] is None and done[2] is None
    assert torch.equal(done[0], out[1][1])


@pytest.mark.parametrize("early", ["norm_fork", "norm", "conv_s2"])
def test_skip_gradient_box(early):
    """A tensor with two consumers: an early one (GroupNorm with / without the residual fork, or a stride-2 conv) and
    a late one reached through skip_tap. With the box the late consumer's gradient is added inside the early
    consumer's backward kernel; without it (boxes off) autograd adds the two. Both against torch fp32."""
    from dm_hip import ops
    rs = np.random.RandomState(5)
    B, C, H = 3, 64, 16
    x0 = q(rnd(rs, B, C, H, H))
    gamma, beta = 1 + 0.2 * rnd(rs, C), 0.2 * rnd(rs, C)
    w = q(rnd(rs, C, C, 3, 3, scale=(9 * C) ** -0.5))
    late_w = q(rnd(rs, B, C, H, H))
    go = q(rnd(rs, B, C, H // 2 if early == "conv_s2" else H, H // 2 if early == "conv_s2" else H))
    d = dev()

    def ours(boxes):
        ops.SKIP_BOXES = boxes
        try:
            x = x0.to(d).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            xin = x * 1.0                          # a non-leaf, as in the network
            g_, b_ = gamma.to(d).requires_grad_(True), beta.to(d).requires_grad_(True)
            box = ops.grad_box(xin)
            assert (box is not None) == boxes
            if early == "conv_s2":
                y = ops.conv2d(xin, w.to(d), None, ops.WeightPack(), stride=2, pad=1, box=box)
            elif early == "norm_fork":
                h, xa, _ = ops.group_norm_act(xin, g_, b_, 32, 1e-5, True, fork=True, box=box)
                y = h + 0.5 * xa
            else:
                y = ops.group_norm_act(xin, g_, b_, 32, 1e-5, True, box=box)
            tapped = ops.skip_tap(xin)
            assert (tapped is not xin) == boxes
            z = (tapped * late_w.to(d).to(torch.bfloat16)).float().sum()
            (y.float() * go.to(d)).sum().add(z).backward()
            return x.grad.float().cpu()
        finally:
            ops.SKIP_BOXES = True

    xr = x0.clone().requires_grad_(True)
    if early == "conv_s2":
        yr = F.conv2d(xr, w, None, stride=2, padding=1)
    else:
        hr = F.silu(F.group_norm(xr, 32, gamma, beta, eps=1e-5))
        yr = hr + 0.5 * xr if early == "norm_fork" else hr
    ((yr * go).sum() + (xr * late_w).sum()).backward()
    on, off = ours(True), ours(False)
    assert rel_err(on, xr.grad) < BF16_BWD and rel_err(off, xr.grad) < BF16_BWD
    assert rel_err(on, off) < BF16_FWD


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("B,L,H,D", [(2, 64, 8, 8), (2, 49, 16, 8), (1, 256, 4, 16), (3, 16, 2, 32), (1, 100, 1, 64),
                                     (3, 70, 5, 16), (2, 1, 3, 8), (128, 64, 64, 8)])
def test_attention_fwd_bwd(B, L, H, D, mfma):
    """Head dims 8 and 16 run on the matrix cores (mfma=1) or, switched off, on the VALU kernels that also serve
    head dims 32 and 64; both against torch's fp32 scaled_dot_product_attention. The last case is the cos_cifar
    mid-block attention at training batch size."""
    from dm_hip import lib
    if not mfma and D > 16:
        pytest.skip("head dims above 16 have one implementation")
    lib.dm_config_set(b"attn_mfma", mfma)
    try:
        _attention_case(B, L, H, D)
    finally:
        lib.dm_config_set(b"attn_mfma", 1)


def _attention_case(B, L, H, D):
    from dm_hip import ops
    rs = np.random.RandomState(B + L + H + D)
    qkv = q(rnd(rs, B, L, 3 * H * D))
    qr = qkv.clone().requires_grad_(True)
    t = qr.view(B, L, 3, H, D).permute(2, 0, 3, 1, 4)
    o_ref = F.scaled_dot_product_attention(t[0], t[1], t[2]).transpose(1, 2).reshape(B, L, H * D)
    go = q(rnd(rs, B, L, H * D))
    o_ref.backward(go)
    qd = qkv.to(dev()).to(torch.bfloat16).requires_grad_(True)
    o = ops.attention(qd, H, D)
    assert rel_err(o.float(), o_ref) < BF16_FWD
    o.backward(go.to(dev()).to(torch.bfloat16))
    assert rel_err(qd.grad.float(), qr.grad) < BF16_BWD


@pytest.mark.parametrize("M,K,N,act", [(8, 64, 128, False), (128, 128, 1024, True), (5, 64, 64, True), (33, 128, 192, False),
                                       (9, 30, 70, True), (20, 256, 192, True), (7, 512, 130, False), (17, 203, 64, True),
                                       (6, 260, 64, False)])
def test_linear_fwd_bwd(M, K, N, act):
    """(the cases from K = 203 up walk the input features through LDS in more than one chunk; K = 30, 203: widths that
    are not multiples of 4 take the scalar staging and inner loops)"""
    from dm_hip import ops
    rs = np.random.RandomState(M + K + N)
    x, w, b = rnd(rs, M, K), rnd(rs, N, K, scale=K ** -0.5), rnd(rs, N, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y_ref = F.linear(F.silu(xr) if act else xr, wr, br)
    gy = rnd(rs, M, N)
    y_ref.backward(gy)
    d = dev()
    xd, wd, bd = (t.to(d).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd, act_in=act)
    assert rel_err(y, y_ref) < 1e-5
    y.backward(gy.to(d))
    assert rel_err(xd.grad, xr.grad) < 1e-5 and rel_err(wd.grad, wr.grad) < 1e-5 and rel_err(bd.grad, br.grad) < 1e-5


@pytest.mark.parametrize("M,K,widths", [(8, 64, (128, 256, 64)), (128, 128, (128, 256, 512, 512, 1024)), (5, 30, (70, 64)),
                                        (16, 256, (128, 384)), (3, 516, (64, 200))])
def test_linear_group_fwd_bwd(M, K, widths):
    """All projections of one input in one launch each way (what a UNet forward does with its emb_proj layers), against
    torch fp32 layer by layer; the input's gradient is the sum over the layers. The last two cases have more input
    features than one LDS chunk of the forward kernel holds."""
    from dm_hip import ops
    rs = np.random.RandomState(M + K + len(widths))
    x = rnd(rs, M, K)
    ws = [rnd(rs, n, K, scale=K ** -0.5) for n in widths]
    bs = [rnd(rs, n, scale=0.1) for n in widths]
    gys = [rnd(rs, M, n) for n in widths]
    xr = x.clone().requires_grad_(True)
    wr, br = [w.clone().requires_grad_(True) for w in ws], [b.clone().requires_grad_(True) for b in bs]
    refs = [F.linear(F.silu(xr), w, b) for w, b in zip(wr, br)]
    sum((y * gy).sum() for y, gy in zip(refs, gys)).backward()
    d = dev()
    xd = x.to(d).requires_grad_(True)
    wd, bd = [w.to(d).requires_grad_(True) for w in ws], [b.to(d).requires_grad_(True) for b in bs]
    ys = ops.linear_group(xd, list(zip(wd, bd)), act_in=True)
    for y, ref in zip(ys, refs):
        assert rel_err(y, ref) < 1e-5
    sum((y * gy.to(d)).sum() for y, gy in zip(ys, gys)).backward()
    assert rel_err(xd.grad, xr.grad) < 1e-5
    for a, b in zip(wd + bd, wr + br):
        assert rel_err(a.grad, b.grad) < 1e-5


def test_timestep_embedding_matches_reference_vectors():
    from dm_hip import ops
    g = golden("embeddings.npz")
    for dim in (64, 128):
        out = ops.timestep_embed(T(g["t"]).to(dev()), T(g[f"freq_{dim}"]).to(dev()))
        # sin/cos of arguments up to ~1e3 rad: device and host libm agree to a few ulp of the argument
        assert max_abs(out, T(g[f"enc_{dim}"])) < 2e-6


def test_class_embed():
    from dm_hip import ops
    rs = np.random.RandomState(5)
    B, E = 6, 64
    emb, table = rnd(rs, B, E), rnd(rs, 11, E)
    cls = T(np.array([0, 3, 9, 9, 1, 10], dtype=np.int64))
    mask = T(np.array([0, 1, 0, 0, 1, 0], dtype=np.float32))
    d = dev()
    for c, m in [(cls, mask), (cls, None), (None, None)]:
        er, tr = emb.clone().requires_grad_(True), table.clone().requires_grad_(True)
        idx = torch.full((B,), 10) if c is None else (c if m is None else torch.where(m.bool(), torch.full_like(c, 10), c))
        ref = er + F.embedding(idx, tr, padding_idx=10)
        ref.backward(torch.ones_like(ref) * 0.5)
        ed, td = emb.to(d).requires_grad_(True), table.to(d).requires_grad_(True)
        out = ops.class_embed_add(ed, td, None if c is None else c.to(d), None if m is None else m.to(d), 10)
        assert max_abs(out, ref) == 0.0
        out.backward(torch.ones_like(out) * 0.5)
        assert max_abs(ed.grad, er.grad) == 0.0 and max_abs(td.grad, tr.grad) < 1e-6


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("B,Cs,H,W,Cb", [(2, 3, 8, 8, 64), (3, 1, 7, 7, 64), (2, 3, 16, 16, 128), (1, 3, 5, 9, 128),
                                         (5, 4, 9, 6, 128), (2, 2, 32, 32, 32), (130, 3, 32, 32, 128)])
def test_edge_convs(B, Cs, H, W, Cb, mfma):
    """in_conv / out_conv forward and backward against torch fp32. mfma=1: the forward and data-gradient kernels on
    the matrix cores (fp32 operands as hi + lo bf16 pairs), mfma=0: the VALU kernels (which also serve feature widths
    that are not multiples of 16 / 32). The last case has a ragged last workgroup."""
    from dm_hip import lib
    lib.dm_config_set(b"edge_mfma", mfma)
    try:
        _edge_convs_case(B, Cs, H, W, Cb)
    finally:
        lib.dm_config_set(b"edge_mfma", 1)


def _edge_convs_case(B, Cs, H, W, Cb):
    from dm_hip import ops
    rs = np.random.RandomState(B + Cs + H + Cb)
    d = dev()
    # image -> features
    x, w, b = rnd(rs, B, Cs, H, W), rnd(rs, Cb, Cs, 3, 3, scale=0.2), rnd(rs, Cb, scale=0.1)
    wr, br, xr = w.clone().requires_grad_(True), b.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, padding=1)
    gy = q(rnd(rs, *y_ref.shape))
    y_ref.backward(gy)
    wd, bd, xd = w.to(d).requires_grad_(True), b.to(d).requires_grad_(True), x.to(d).requires_grad_(True)
    y = ops.in_conv(xd, wd, bd)
    assert rel_err(y.float(), y_ref) < 4e-3          # one bf16 rounding of the output
    y.backward(gy.to(d))
    # the gradient w.r.t. the image (guidance / inversion through the network; training does not ask for it)
    assert xd.grad.shape == x.shape and rel_err(xd.grad, xr.grad) < 1e-5
    # the weight gradient takes the image in bf16, as the reference's autocast conv does
    wq = w.clone().requires_grad_(True)
    F.conv2d(q(x), wq, None, padding=1).backward(gy)
    # (matrix-core kernel; the VALU kernel for odd widths keeps the image in fp32)
    assert min(rel_err(wd.grad, wq.grad), rel_err(wd.grad, wr.grad)) < 1e-4 and rel_err(bd.grad, br.grad) < 1e-4
    assert rel_err(wd.grad, wr.grad) < 4e-3
    # features -> image
    h, w2, b2 = q(rnd(rs, B, Cb, H, W)), rnd(rs, Cs, Cb, 3, 3, scale=Cb ** -0.5), rnd(rs, Cs, scale=0.1)
    hr, w2r, b2r = (t.clone().requires_grad_(True) for t in (h, w2, b2))
    o_ref = F.conv2d(hr, w2r, b2r, padding=1)
    go = rnd(rs, *o_ref.shape)
    o_ref.backward(go)
    hd, w2d, b2d = (t.to(d).requires_grad_(True) for t in (h, w2, b2))
    o = ops.out_conv(hd, w2d, b2d)
    assert o.dtype == torch.float32 and rel_err(o, o_ref) < 1e-5
    o.backward(go.to(d))
    assert rel_err(hd.grad.float(), hr.grad) < 4e-3
    # the weight gradient takes the output gradient in bf16 (the dtype it has under the reference's autocast)
    w2q = w2.clone().requires_grad_(True)
    F.conv2d(h, w2q, None, padding=1).backward(q(go))
    assert min(rel_err(w2d.grad, w2q.grad), rel_err(w2d.grad, w2r.grad)) < 1e-4 and rel_err(b2d.grad, b2r.grad) < 1e-4
    assert rel_err(w2d.grad, w2r.grad) < 4e-3


def test_resample():
    from dm_hip import ops
    rs = np.random.RandomState(3)
    x = q(rnd(rs, 2, 64, 4, 6))
    d = dev()
    for fn, ref in [(ops.upsample2x, lambda t: F.interpolate(t, scale_factor=2, mode="nearest")),
                    (ops.avgpool2x, lambda t: F.avg_pool2d(t, 2, 2))]:
        xr = x.clone().requires_grad_(True)
        y_ref = ref(xr)
        gy = q(rnd(rs, *y_ref.shape))
        y_ref.backward(gy)
        xd = x.to(d).requires_grad_(True)
        y = fn(xd)
        assert rel_err(y.float(), y_ref) < 4e-3
        y.backward(gy.to(d))
        assert rel_err(xd.grad.float(), xr.grad) < 4e-3


def test_diffuse_and_loss_bitexact_vs_reference_vectors():
    from diffusion.scheduler import get_scheduler
    from dm_hip import ops
    g = golden("diffusion.npz")
    d = dev()
    x = T(g["x"]).to(d)
    scheds = {"lin": get_scheduler("beta", schedule="linear"), "cos": get_scheduler("beta", schedule="cosine"),
              "rf": get_scheduler("rectified_flow")}
    for k, s in scheds.items():
        eps, t = T(g[f"eps_{k}"]).to(d), T(g[f"t_{k}"]).to(d)
        xt = s.diffuse(x, t, eps)
        assert np.array_equal(xt.cpu().numpy(), g[f"xt_{k}"]), k       # bit-exact
        pred = 0.5 * xt
        loss = ops.mse_loss(pred, eps, x=x, mode=1) if k == "rf" else ops.mse_loss(pred, eps, mode=0)
        assert abs(float(loss) - float(g[f"loss_{k}"])) < 2e-7 * max(1.0, float(g[f"loss_{k}"]))
    # gradient of the loss
    p = (0.5 * x).clone().requires_grad_(True)
    eps = T(g["eps_rf"]).to(d)
    ops.mse_loss(p, eps, x=x, mode=1).backward()
    ref = 2 * (p.detach() - (x - eps)) / p.numel()
    assert rel_err(p.grad, ref) < 1e-6


def test_sampler_step_and_cfg_bitexact():
    from dm_hip import ops
    rs = np.random.RandomState(11)
    z, p, n = rnd(rs, 4, 3, 8, 8), rnd(rs, 4, 3, 8, 8), rnd(rs, 4, 3, 8, 8)
    d = dev()
    c0, c1, c2, c3 = (float(np.float32(v)) for v in (1.0101, 0.0703, 0.1414, 0.9))
    f = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
    out = ops.sampler_step(z.to(d).clone(), p.to(d), n.to(d), c0, c1, c2, 0.0, 0, False, False)
    assert torch.equal(out.cpu(), f(c0) * (z - f(c1) * p) + f(c2) * n)
    out = ops.sampler_step(z.to(d).clone(), p.to(d), None, c0, c1, c2, 0.0, 0, False, False)
    assert torch.equal(out.cpu(), f(c0) * (z - f(c1) * p))
    x0 = torch.clip((z - f(c1) * p) / f(c0), -1, 1)
    out = ops.sampler_step(z.to(d).clone(), p.to(d), None, c0, c1, c2, c3, 1, True, False)
    assert torch.equal(out.cpu(), f(c2) * x0 + f(c3) * p)
    out = ops.sampler_step(z.to(d).clone(), p.to(d), None, c0, c1, c2, c3, 1, True, True)
    assert torch.equal(out.cpu(), x0)
    out = ops.sampler_step(z.to(d).clone(), p.to(d), None, -0.02, 0, 0, 0, 2, False, False)
    assert torch.equal(out.cpu(), z - f(-0.02) * p)
    assert torch.equal(ops.cfg_combine(z.to(d), p.to(d), 2.5).cpu(), z + 2.5 * (p - z))


def test_adamw_ema_step_matches_torch():
    import ctypes
    from dm_hip import lib, check
    rs = np.random.RandomState(21)
    n = 100003
    p0, g = rnd(rs, n), rnd(rs, n, scale=3.0)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=2e-4, betas=(0.9, 0.99))
    d = dev()
    p, m, v, ema = p0.to(d), torch.zeros(n, device=d), torch.zeros(n, device=d), p0.to(d).clone()
    wb = torch.empty(n, dtype=torch.bfloat16, device=d)
    partials = torch.empty(lib.dm_sumsq_partials(), device=d)
    norm = torch.empty(1, device=d)
    ema_ref = p0.clone()
    st = torch.cuda.current_stream().cuda_stream
    for step in range(1, 4):
        gs = g * (0.5 ** step)
        ref.grad = gs.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        decay = 0.0 if step == 1 else min(1 - (1 + (step - 1)) ** -0.75, 0.9999)
        ema_ref.sub_((1 - decay) * (ema_ref - ref.detach()))
        # the buffer holds the SUM over `world` ranks, grad_scale = 1 / world turns it into the mean on the fly
        # (powers of two: exact); on the last step the kernel also zeroes the gradient behind its read
        world = (1, 2, 4)[step - 1]
        gd = (gs * world).to(d)
        check(lib.dm_grad_norm(gd.data_ptr(), n, 1.0 / world, partials.data_ptr(), norm.data_ptr(), st))
        assert abs(float(norm) - float(tn)) < 1e-5 * float(tn)
        check(lib.dm_adamw_ema_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), ema.data_ptr(),
                                    wb.data_ptr(), norm.data_ptr(), 1.0, n, 2e-4, 0.9, 0.99, 1e-8, 1e-2,
                                    1 - 0.9 ** step, 1 - 0.99 ** step, decay, None, 1.0 / world, int(step == 3), st))
        assert torch.equal(gd.cpu(), torch.zeros(n) if step == 3 else gs * world)
        # same formulas as torch's AdamW; the global norm is reduced in a different order (rel 5e-7) and FMA
        # contraction moves results by an ulp: parameters of magnitude <= 8 agree to 2 ulps
        assert max_abs(p, ref.detach()) < 2e-6
        assert max_abs(ema, ema_ref) < 2e-6
        assert torch.equal(wb.float().cpu(), p.cpu().to(torch.bfloat16).float())


def test_transpose_conv_weights_batched():
    """Forward-layout [Cout][T][Cin] bf16 arena -> data-gradient layout [Cin][T][Cout] at the same offsets; exact."""
    import ctypes

    from dm_hip import WtItem, check, lib
    shapes = [(128, 9, 64), (64, 1, 192), (72, 9, 8), (8, 4, 136)]  # (Cout, T, Cin), ragged tiles included
    offs, items, o, tiles = [], [], 0, 0
    for co, t, ci in shapes:
        offs.append(o)
        items.append((o, co, t, ci, tiles))
        tiles += t * ((co + 63) // 64) * ((ci + 63) // 64)
        o += (co * t * ci + 63) // 64 * 64
    src = torch.randn(o, device=dev()).to(torch.bfloat16)
    dst = torch.zeros_like(src)
    rec = np.zeros(len(items), dtype=np.dtype([("offset", np.int64), ("Cout", np.int32), ("T", np.int32),
                                               ("Cin", np.int32), ("tile_begin", np.int32)]))
    assert rec.dtype.itemsize == ctypes.sizeof(WtItem)
    for i, it in enumerate(items):
        rec[i] = it
    table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev())
    check(lib.dm_transpose_conv_weights(src.data_ptr(), dst.data_ptr(), table.data_ptr(), len(items), tiles,
                                        torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for (co, t, ci), off in zip(shapes, offs):
        n = co * t * ci
        want = src[off:off + n].view(co, t, ci).permute(2, 1, 0).contiguous().view(-1)
        assert torch.equal(dst[off:off + n], want), (co, t, ci)
