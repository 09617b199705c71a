"""As repro_linear_group.py, for the BACKWARD kernels of the embedding path (grouped linear, plain linear, class table)
and the edge convolution's weight gradient: gradients computed beside bf16 GEMMs of another graph branch must equal,
bit for bit, the gradients of a quiet launch.   python tools/experiments/repro_backward.py [N] [R]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]

import torch  # noqa: E402

from dm_hip import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
M, K = 128, 512          # the full-size model's embedding width and batch
widths = [128, 256, 256, 512, 512, 256, 128]
x = torch.randn(M, K, generator=g).to(dev).requires_grad_()
layers = [(torch.randn(n, K, generator=g).mul(0.05).to(dev).requires_grad_(),
           torch.randn(n, generator=g).mul(0.1).to(dev).requires_grad_()) for n in widths]
gys = [torch.randn(M, n, generator=g).to(dev) for n in widths]
w1 = torch.randn(K, 128, generator=g).mul(0.05).to(dev).requires_grad_()
b1 = torch.randn(K, generator=g).mul(0.1).to(dev).requires_grad_()
x1 = torch.randn(M, 128, generator=g).to(dev).requires_grad_()
gy1 = torch.randn(M, K, generator=g).to(dev)
table = torch.randn(11, K, generator=g).to(dev).requires_grad_()
cls = torch.randint(0, 10, (M,), generator=g).to(dev)
mask = (torch.rand(M, generator=g) < 0.2).float().to(dev)
img = torch.randn(M, 3, 32, 32, generator=g).to(dev)
w_in = torch.randn(128, 3, 3, 3, generator=g).mul(0.1).to(dev).requires_grad_()
b_in = torch.zeros(128, device=dev).requires_grad_()
A = torch.randn(2048, 2048, generator=g).to(dev).bfloat16()
params = [x, x1, w1, b1, table, w_in, b_in] + [t for wb in layers for t in wb]


def work():
    ys = ops.linear_group(x, layers, act_in=True)
    y1 = ops.linear(x1, w1, b1, act_in=True)
    e = ops.class_embed_add(y1, table, cls, mask, 10)
    h = ops.in_conv(img, w_in, b_in)
    loss = sum((y * gy).sum() for y, gy in zip(ys, gys)) + (e * gy1).sum() + h.float().square().mean()
    return torch.autograd.grad(loss, params)


want = torch.cat([t.flatten() for t in work()])
torch.cuda.synchronize()
side, aux = torch.cuda.Stream(), torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
for phase in ("warm-up", "capture"):
    outs, keep = [], []
    graph = ops.new_graph() if phase == "capture" else None
    with (torch.cuda.graph(graph, stream=side) if graph is not None else torch.cuda.stream(side)):
        aux.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(aux), torch.no_grad():
            for _ in range(4 * N):
                keep.append(A @ A)
        for _ in range(N):
            outs.append(work())
        torch.cuda.current_stream().wait_stream(aux)
    torch.cuda.synchronize()
bad = 0
names = ["x", "x1", "w1", "b1", "table", "w_in", "b_in"] + [f"{k}{i}" for i in range(len(layers)) for k in "wb"]
for r in range(R):
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    for grads in outs:
        got = torch.cat([t.flatten() for t in grads])
        if not torch.equal(got, want):
            bad += 1
            if bad <= 3:
                print("differ:", [n for n, a, b in zip(names, grads, work()) if not torch.equal(a, b)])
print(f"{bad} of {R * N} gradient sets differ from the quiet launch")
