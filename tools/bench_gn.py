"""Timing of the GroupNorm(+SiLU) kernels on the shapes of the CIFAR UNet at batch 128: achieved GB/s against the
ALGORITHMIC bytes (every tensor once: forward 4 B/element, backward 6, +2 with a residual gradient)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
from dm_hip import GnDesc, check, lib  # noqa: E402

# (B, HW, C1, C2, count per training step)
SHAPES = [(128, 1024, 128, 0, 8), (128, 1024, 256, 0, 2), (128, 1024, 256, 128, 2), (128, 256, 256, 0, 6),
          (128, 256, 512, 0, 1), (128, 256, 512, 256, 2), (128, 256, 256, 128, 1), (128, 64, 512, 0, 11),
          (128, 64, 512, 512, 2), (128, 64, 512, 256, 1)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    tf = tb = 0.0
    for B, HW, C1, C2, cnt in SHAPES:
        C = C1 + C2
        bf = torch.bfloat16
        x1 = torch.randn(B, HW, C1, device=dev).to(bf)
        x2 = torch.randn(B, HW, C2, device=dev).to(bf) if C2 else None
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        ss = torch.randn(B, 2 * C, device=dev) * 0.1
        y = torch.empty(B, HW, C, device=dev, dtype=bf)
        dy = torch.randn(B, HW, C, device=dev).to(bf)
        dx1 = torch.empty_like(x1)
        dx2 = torch.empty_like(x2) if C2 else None
        mean, rstd = torch.empty(B, 32, device=dev), torch.empty(B, 32, device=dev)
        dss = torch.empty(B, 2 * C, device=dev)
        pp = torch.empty(B, 3, C, device=dev)
        dex = torch.randn(B, HW, C, device=dev).to(bf)
        d = GnDesc(B, HW, C1, C2, 32, 1e-5, 1, 1, 0)
        ws = torch.zeros(lib.dm_groupnorm_ws_floats(ctypes.byref(d)), device=dev)
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        t_f = timeit(lambda: check(lib.dm_groupnorm_act_fwd(ctypes.byref(d), p(x1), p(x2), p(g), p(b), p(ss), None, p(y),
                                                            p(mean), p(rstd), p(ws), st)))
        def bwd(extra):
            return timeit(lambda: check(lib.dm_groupnorm_act_bwd(
                ctypes.byref(d), p(x1), p(x2), p(g), p(b), p(ss), None, p(mean), p(rstd), p(dy), p(dex) if extra else None, None,
                p(dx1), p(dx2), p(dss), None, p(pp), 0, p(ws), st)))
        t_b, t_e = bwd(False), bwd(True)
        n = B * HW * C
        tf += t_f * cnt
        tb += 0.5 * (t_b + t_e) * cnt
        print(f"B={B} HW={HW:5d} C={C1}+{C2:<4d} {n*2/1e6:7.1f} MB x{cnt:2d} | fwd {t_f*1e3:6.1f}us {4*n/t_f/1e9:5.2f} TB/s"
              f" | bwd {t_b*1e3:6.1f}us {6*n/t_b/1e9:5.2f} TB/s | bwd+res {t_e*1e3:6.1f}us {8*n/t_e/1e9:5.2f} TB/s", flush=True)
    print(f"per training step (counts above, half the backward calls with a residual gradient): fwd {tf:.3f} ms, bwd {tb:.3f} ms")


if __name__ == "__main__":
    main()
