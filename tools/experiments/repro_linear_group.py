"""Does the grouped embedding projection (dm_linear_group_fwd) give the same bits on every launch? A HIP graph holds
N launches of it into N separate outputs on one branch and, optionally, a second branch of unrelated small kernels; the
graph is replayed R times without host synchronisation and every output is compared with a synchronised eager launch.

  DM_HIP_LIBRARY=tools/bin/libdm_hip_fast.so python tools/experiments/repro_linear_group.py [other: 0 none, 1 in_conv, 2 bf16 GEMM, 3 bf16 GEMM before each launch on the same branch, 4 HBM-bound elementwise, 5 fp32 GEMM, 6 this library's 3x3 forward, 7 its weight gradient, 8 fp16 GEMM, 9 full-size first convolution, 10 matrix-core attention, 11 small bf16 GEMM] [N] [R]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]

import torch  # noqa: E402

from dm_hip import ops  # noqa: E402

other = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
R = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
M, K = 8, 64
widths = [128, 256, 256, 256, 256, 256, 128, 128]
x = torch.randn(M, K, generator=g).to(dev)
layers = [(torch.randn(n, K, generator=g).mul(0.1).to(dev), torch.randn(n, generator=g).mul(0.1).to(dev)) for n in widths]
img = torch.randn(8, 3, 16, 16, generator=g).to(dev)
w_in = torch.randn(64, 3, 3, 3, generator=g).mul(0.1).to(dev)
b_in = torch.zeros(64, device=dev)
A = torch.randn(2048, 2048, generator=g).to(dev).bfloat16()
A32 = torch.randn(2048, 2048, generator=g).to(dev)
big = torch.randn(64 * 1024 * 1024, generator=g).to(dev)     # 256 MB: an HBM-bound neighbour without matrix cores
from modeling.layers import Conv2d  # noqa: E402
conv = Conv2d(256, 256, kernel_size=3, padding="same").to(dev)
feat = ops.as_nhwc(torch.randn(128, 256, 16, 16, generator=g).to(dev))
gfeat = ops.as_nhwc(torch.randn(128, 256, 16, 16, generator=g).to(dev))


A16 = A.half()
img_full = torch.randn(128, 3, 32, 32, generator=g).to(dev)
w_full = torch.randn(128, 3, 3, 3, generator=g).mul(0.1).to(dev)
b_full = torch.zeros(128, device=dev)
qkv = torch.randn(128, 64, 3 * 64 * 8, generator=g).to(dev).bfloat16()


def neighbour(kind):
    if kind == 1:
        return ops.in_conv(img, w_in, b_in)
    if kind == 2:
        return A @ A
    if kind == 4:
        return big * 1.5
    if kind == 5:
        return A32 @ A32
    if kind == 6:                      # this library's 3x3 forward kernel (direct-to-LDS loads, 16x16x32 bf16 MFMA)
        with torch.no_grad():
            return conv(feat)
    if kind == 7:                      # its weight-gradient kernel (transposing LDS reads) and the slab sum
        with torch.enable_grad():
            conv.weight.grad = None
            conv(feat).backward(gfeat)
        return conv.weight.grad
    if kind == 8:                      # fp16 GEMM
        return A16 @ A16
    if kind == 9:                      # the first convolution at full size (image -> 128 features, MFMA on hi/lo bf16 pairs)
        return ops.in_conv(img_full, w_full, b_full)
    if kind == 10:                     # matrix-core attention, 64 heads of 8 over 64 positions
        return ops.attention(qkv, 64, 8)
    if kind == 11:                     # a small bf16 GEMM
        return A[:512, :512] @ A[:512, :512]
    raise ValueError(kind)

with torch.no_grad():
    ref = torch.cat([y.flatten() for y in ops.linear_group(x, layers, act_in=True)])
    torch.cuda.synchronize()
    side, aux = torch.cuda.Stream(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for _ in range(2):   # warm-up, then capture
        outs, keep = [], []
        graph = ops.new_graph() if _ == 1 else None
        ctx = torch.cuda.graph(graph, stream=side) if graph is not None else torch.cuda.stream(side)
        with ctx:
            if other not in (0, 3):
                aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(aux):
                    for i in range(N):
                        keep.append(neighbour(other))
            for i in range(N):
                if other == 3:
                    keep.append(A @ A)    # a heavy matrix-core kernel right before, on the same branch
                outs.append(ops.linear_group(x, layers, act_in=True))
            if other not in (0, 3):
                torch.cuda.current_stream().wait_stream(aux)
        torch.cuda.synchronize()
    bad = 0
    for r in range(R):
        graph.replay()
        graph.replay()
        torch.cuda.synchronize()
        for ys in outs:
            got = torch.cat([y.flatten() for y in ys])
            if not torch.equal(got, ref):
                bad += 1
                if bad <= 3:
                    idx = (got != ref).nonzero().flatten()
                    print(f"replay {r}: {idx.numel()} elements differ, first at {int(idx[0])}, max |d| {float((got - ref).abs().max()):.3e}")
    print(f"other-branch {other}: {bad} of {R * N} outputs (second launch of each pair) differ")
