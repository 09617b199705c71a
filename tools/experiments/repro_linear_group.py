"""Does the grouped embedding projection (dm_linear_group_fwd) give the same bits on every launch? A HIP graph holds
N launches of it into N separate outputs on one branch and, optionally, a second branch of unrelated small kernels; the
graph is replayed R times without host synchronisation and every output is compared with a synchronised eager launch.

  DM_HIP_LIBRARY=tools/bin/libdm_hip_fast.so python tools/experiments/repro_linear_group.py [other=1] [N] [R]
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
            if other in (1, 2):
                aux.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(aux):
                    for i in range(N):
                        keep.append(ops.in_conv(img, w_in, b_in) if other == 1 else A @ A)
            for i in range(N):
                if other == 3:
                    keep.append(A @ A)    # a heavy matrix-core kernel right before, on the same branch
                outs.append(ops.linear_group(x, layers, act_in=True))
            if other in (1, 2):
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
