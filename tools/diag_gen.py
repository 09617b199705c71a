"""Diagnostic: generate() with DDPM on the tiny UNet, eager against graph replay with 1, 2 and 3 interleaved chains;
prints where and by how much the images differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-models_amd"))
import torch
from tests.parity import TINY_CFG, det_sd_for
from diffusion.sampler import get_sampler
from diffusion.scheduler import get_scheduler
from harness.generate import generate
from modeling import get_model

dev = torch.device("cuda:0")
model = get_model("unet", **TINY_CFG)
model.load_state_dict(det_sd_for(model, 5))
model = model.to(dev)
for name, kw in (("ddpm", {"n_steps": 5}), ("ddim", {"n_steps": 4})):
    sched = get_scheduler("beta", n_steps=1000)
    sampler = get_sampler(name, **kw)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(5, TINY_CFG["in_channels"], 16, 16, generator=g)
    cls = torch.randint(0, 10, (5,), generator=g)
    eager = generate(model, sampler, sched, z, cls, batch_size=2, use_graph=False, chains=1)
    for chains, use_graph in ((1, True), (2, True), (2, True), (2, False), (3, True), (2, True)):
        again = generate(model, sampler, sched, z, cls, batch_size=2, use_graph=use_graph, chains=chains)
        d = (again - eager).abs().flatten(1).max(1).values
        print(name, "chains", chains, "graph", use_graph, "max diff per image", [f"{v:.2e}" for v in d.tolist()], flush=True)
