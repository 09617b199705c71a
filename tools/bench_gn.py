"""Timing of the GroupNorm(+SiLU) kernels on the shapes of the CIFAR UNet at batch 128 (achieved GB/s against the
algorithmic bytes: forward 6 B/element, backward 10 B/element)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
from dm_hip import GnDesc, check, lib  # noqa: E402

SHAPES = [(128, 1024, 128, 0), (128, 1024, 256, 0), (128, 1024, 256, 128), (128, 256, 256, 0), (128, 256, 512, 256),
          (128, 64, 512, 0), (128, 64, 512, 512)]


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
    for B, HW, C1, C2 in SHAPES:
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
        dg, db, dss = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.empty(B, 2 * C, device=dev)
        d = GnDesc(B, HW, C1, C2, 32, 1e-5, 1, 1, 0)
        ws = torch.zeros(lib.dm_groupnorm_ws_floats(ctypes.byref(d)), device=dev)
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        t_f = timeit(lambda: check(lib.dm_groupnorm_act_fwd(ctypes.byref(d), p(x1), p(x2), p(g), p(b), p(ss), None, p(y),
                                                            p(mean), p(rstd), p(ws), st)))
        t_b = timeit(lambda: check(lib.dm_groupnorm_act_bwd(ctypes.byref(d), p(x1), p(x2), p(g), p(b), p(ss), None,
                                                            p(mean), p(rstd), p(dy), None, p(dx1), p(dx2), p(dg), p(db),
                                                            p(dss), None, None, p(ws), st)))
        n = B * HW * C
        tf += t_f
        tb += t_b
        print(f"B={B} HW={HW:5d} C={C1}+{C2:<4d} {n*2/1e6:7.1f} MB | fwd {t_f*1e3:6.1f}us {6*n/t_f/1e9:6.0f} GB/s"
              f" | bwd {t_b*1e3:6.1f}us {10*n/t_b/1e9:6.0f} GB/s", flush=True)
    print(f"total fwd {tf:.3f} ms, bwd {tb:.3f} ms")


if __name__ == "__main__":
    main()
