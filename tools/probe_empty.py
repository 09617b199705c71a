"""Edge-case probe: an empty batch and a batch of one through the UNet (forward, backward) and the samplers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
from modeling import get_model  # noqa: E402
from tests.parity import TINY_CFG  # noqa: E402

dev = torch.device("cuda:0")
m = get_model("unet", **TINY_CFG).to(dev).train()
for B in (1, 0):
    x = torch.randn(B, 3, 16, 16, device=dev, requires_grad=True)
    t = torch.randint(0, 1000, (B,), device=dev)
    cls = torch.randint(0, 10, (B,), device=dev)
    try:
        y = m(x, t, cls, torch.zeros(B, device=dev))
        torch.cuda.synchronize()
        print("forward B=%d ok" % B, tuple(y.shape))
        if B:
            y.float().sum().backward()
            torch.cuda.synchronize()
            print("backward B=%d ok" % B, tuple(x.grad.shape))
    except Exception as e:  # noqa: BLE001
        print("B=%d failed: %r" % (B, e))
        break
