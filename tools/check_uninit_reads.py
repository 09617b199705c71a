"""Poison the caching allocator's free blocks with NaN bit patterns between two forwards and report the first
module whose output changes: that module (or its kernel) reads memory nobody wrote."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch
from tests.parity import TINY_CFG, det_sd_for
from modeling import get_model
dev = torch.device("cuda:0")
model = get_model("unet", **TINY_CFG)
model.load_state_dict(det_sd_for(model, 5))
model = model.to(dev).eval()
g = torch.Generator().manual_seed(3)
outs = {}
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        if torch.is_tensor(o):
            outs.setdefault(name, []).append(o.detach().float().clone())
    return f
for n, m in model.named_modules():
    if n and not list(m.children()):
        m.register_forward_hook(hook(n))
def poison():
    junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 3000, 70000, 500000)]
    del junk
for B, hw in ((1, 16), (2, 16), (1, 32), (5, 32)):
    x = torch.randn(B, 3, hw, hw, generator=g).to(dev); t = torch.full((B,), 500, device=dev)
    c = torch.randint(0, 10, (B,), generator=g).to(dev)
    outs.clear()
    with torch.no_grad():
        model(x, t, cls=c)
        poison()
        model(x, t, cls=c)
    print("B", B, "hw", hw)
    for n, (a, b) in ((k, v) for k, v in outs.items() if len(v) == 2):
        d = float((a - b).abs().max())
        if d > 0 or torch.isnan(b).any():
            print("   first differing module:", n, tuple(a.shape), "max diff", d)
            break
