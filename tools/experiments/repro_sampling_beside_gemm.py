"""Sampling (generate(): graph-replayed DDIM steps in two interleaved chains) while bf16 GEMMs run on another stream:
the images must equal those of a quiet run bit for bit.   python tools/experiments/repro_sampling_beside_gemm.py [runs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]

import torch  # noqa: E402

from diffusion.sampler import get_sampler  # noqa: E402
from diffusion.scheduler import get_scheduler  # noqa: E402
from harness.generate import generate  # noqa: E402
from modeling import get_model  # noqa: E402
from tests.parity import det_sd_for, load_cfg  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = load_cfg("cos_cifar")
model = get_model("unet", **cfg["model_cfg"])
model.load_state_dict(det_sd_for(model, 11))
model.to(dev).eval()
sched = get_scheduler("beta", schedule="cosine")
sampler = get_sampler("ddim", n_steps=10, pbar=False)
g = torch.Generator().manual_seed(2)
z = torch.randn(64, 3, 32, 32, generator=g)
cls = torch.randint(0, 10, (64,), generator=g)
A = torch.randn(2048, 2048, generator=g).to(dev).bfloat16()
want = generate(model, sampler, sched, z, cls, guidance_scale=1.0, batch_size=64)
torch.cuda.synchronize()
aux = torch.cuda.Stream()
bad = 0
for r in range(runs):
    with torch.cuda.stream(aux):
        for _ in range(600):
            A @ A
    got = generate(model, sampler, sched, z, cls, guidance_scale=1.0, batch_size=64)
    torch.cuda.synchronize()
    if not torch.equal(got, want):
        bad += 1
        print(f"run {r}: {int((got != want).sum())} values differ, max |d| {float((got.float() - want.float()).abs().max()):.3e}")
print(f"{bad} of {runs} sampling runs beside GEMMs differ from the quiet run")
