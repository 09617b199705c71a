"""Full-size stability run: N eager training steps of a shipped configuration on structured synthetic images (smooth
class-dependent patterns, so that the denoiser has something to learn); prints the loss every 50 steps and fails on a
non-finite value. usage: python tools/soak.py [config] [steps]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cos_cifar"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    dev = torch.device("cuda:0")
    cfg = bench.load_cfg(name)
    model, sched, step = bench.build_step(cfg, dev)
    B, c = cfg["batch_size"], cfg["model_cfg"]["in_channels"]
    hw = 28 if cfg["dataset"] == "mnist" else 32
    g = torch.Generator().manual_seed(1)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, hw), torch.linspace(-1, 1, hw), indexing="ij")
    acc, window = 0.0, []
    for s in range(1, n + 1):
        cls = torch.randint(0, 10, (B,), generator=g)
        ph = torch.rand(B, 1, 1, 1, generator=g) * 2 * math.pi
        k = (1 + cls.float()).view(B, 1, 1, 1)
        x = torch.sin(k * xx + ph) * torch.cos(0.5 * k * yy) * torch.ones(1, c, 1, 1)
        loss = float(step.step(x.to(dev), cls.to(dev)))
        if not math.isfinite(loss):
            print(f"step {s}: loss {loss}")
            sys.exit(1)
        acc += loss
        if s % 50 == 0:
            window.append(acc / 50)
            print(f"step {s:5d}: mean loss of the last 50 steps {acc / 50:.4f}", flush=True)
            acc = 0.0
    pn = float(step.arena.data.norm())
    print(f"parameter norm {pn:.3f}, ema norm {float(step.ema.norm()):.3f}")
    assert window[-1] < 0.5 * window[0], "the loss did not fall"


if __name__ == "__main__":
    main()
