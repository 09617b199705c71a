import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
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
sched = get_scheduler("beta", n_steps=1000)
sampler = get_sampler("ddim", n_steps=4)
g = torch.Generator().manual_seed(3)
z = torch.randn(5, 3, 16, 16, generator=g)
cls = torch.randint(0, 10, (5,), generator=g)
for bs in (2, 5, 1):
    e = generate(model, sampler, sched, z, cls, batch_size=bs, use_graph=False)
    gr = generate(model, sampler, sched, z, cls, batch_size=bs, use_graph=True)
    print(bs, [(float((e[i] - gr[i]).abs().max())) for i in range(5)])
e2 = generate(model, sampler, sched, z, cls, batch_size=2, use_graph=False)
e5 = generate(model, sampler, sched, z, cls, batch_size=5, use_graph=False)
print("eager bs2 vs bs5", [(float((e2[i] - e5[i]).abs().max())) for i in range(5)])
a = generate(model, sampler, sched, z, cls, batch_size=5, use_graph=False)
b = generate(model, sampler, sched, z, cls, batch_size=5, use_graph=False)
print("eager twice bs5", [(float((a[i] - b[i]).abs().max())) for i in range(5)])
x = torch.randn(5, 3, 16, 16, generator=g).to(dev); t = torch.full((5,), 999, device=dev); c = cls.to(dev)
with torch.no_grad():
    o1 = model(x, t, cls=c); o2 = model(x, t, cls=c); o3 = model(x[:2].contiguous(), t[:2], cls=c[:2])
print("fwd twice", float((o1 - o2).abs().max()), "fwd batch-subset", float((o1[:2] - o3).abs().max()), float(o1.abs().max()))
