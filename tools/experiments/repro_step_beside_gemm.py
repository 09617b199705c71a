"""Forward + backward of a whole UNet step beside bf16 GEMMs of another graph branch: loss and every gradient must equal,
bit for bit, those of a quiet launch (a generic detector for results that depend on what else runs on the chip).
  python tools/experiments/repro_step_beside_gemm.py [tiny|full] [replays]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]

import torch  # noqa: E402

from dm_hip import ops  # noqa: E402
from diffusion.scheduler import get_scheduler  # noqa: E402
from harness.train_step import TrainStep  # noqa: E402
from modeling import get_model  # noqa: E402
from tests.parity import TINY_CFG, det_sd_for, load_cfg  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
if which == "tiny":
    cfg, B, side_px = TINY_CFG, 8, 16
else:
    cfg = load_cfg("cos_cifar")["model_cfg"]
    B, side_px = 32, 32
m = get_model("unet", **cfg)
m.load_state_dict(det_sd_for(m, 11))
step = TrainStep(m.to(dev).train(), get_scheduler("beta", n_steps=1000), lr=2e-4, clip_grad_norm=1.0, use_ema=True)
x = (torch.rand(B, 3, side_px, side_px, generator=g) * 2 - 1).to(dev)
cls = torch.randint(0, 10, (B,), generator=g).to(dev)
mask = (torch.rand(B, generator=g) < 0.25).float().to(dev)
eps = torch.randn(B, 3, side_px, side_px, generator=g).to(dev)
t = torch.randint(0, 1000, (B,), generator=g).to(dev)
args = (x, cls, mask, eps, t)
A = torch.randn(2048, 2048, generator=g).to(dev).bfloat16()

step.arena.grad.zero_()
loss = step.loss_given_noise(*args)
step._backward(loss, overlap=False)
torch.cuda.synchronize()
want_loss, want_grad = float(loss.detach()), step.arena.grad.clone()

side, aux = torch.cuda.Stream(), torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
ops.FORCE_REPACK = True
n_gemm = 40 if which == "tiny" else 400
for phase in ("warm-up", "capture"):
    graph = ops.new_graph() if phase == "capture" else None
    with (torch.cuda.graph(graph, stream=side) if graph is not None else torch.cuda.stream(side)):
        keep = []
        aux.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(aux), torch.no_grad():
            for _ in range(n_gemm):
                keep.append(A @ A)
        step.arena.grad.zero_()
        lc = step.loss_given_noise(*args)
        step._backward(lc, overlap=False)
        torch.cuda.current_stream().wait_stream(aux)
    torch.cuda.synchronize()
ops.FORCE_REPACK = False
lc = lc.detach()
bad = 0
for r in range(R):
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    if float(lc) != want_loss or not torch.equal(step.arena.grad, want_grad):
        bad += 1
        if bad <= 3:
            names = [n for (n, p), o in zip(step.arena.params, step.arena.offsets)
                     if not torch.equal(step.arena.grad[o:o + p.numel()], want_grad[o:o + p.numel()])]
            print(f"replay {r}: loss {float(lc)!r} vs {want_loss!r}; {len(names)} gradients differ, first {names[:4]}")
print(f"{which}: {bad} of {R} replays beside GEMMs differ from the quiet step")
