"""How the small-image single-pass GroupNorm kernels scale with the number of workgroups (HW = 64, 256; batch sweep):
time per launch back to back (operands cache-warm), to separate per-workgroup cost from memory traffic."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
from dm_hip import GnDesc, check, lib  # noqa: E402
from bench_gn import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    bf = torch.bfloat16
    for HW, C in ((64, 512), (256, 256)):
        for B in (16, 32, 64, 128, 256, 512):
            x1 = torch.randn(B, HW, C, device=dev).to(bf)
            g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
            ss = torch.randn(B, 2 * C, device=dev) * 0.1
            y = torch.empty(B, HW, C, device=dev, dtype=bf)
            dy = torch.randn(B, HW, C, device=dev).to(bf)
            dx1 = torch.empty_like(x1)
            mean, rstd = torch.empty(B, 32, device=dev), torch.empty(B, 32, device=dev)
            dss = torch.empty(B, 2 * C, device=dev)
            pp = torch.empty(B, 3, C, device=dev)
            d = GnDesc(B, HW, C, 0, 32, 1e-5, 1, 1, 0)
            ws = torch.zeros(lib.dm_groupnorm_ws_floats(ctypes.byref(d)), device=dev)
            p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
            t_f = timeit(lambda: check(lib.dm_groupnorm_act_fwd(ctypes.byref(d), p(x1), None, p(g), p(b), p(ss), None, p(y),
                                                                p(mean), p(rstd), p(ws), st)), iters=50)
            t_b = timeit(lambda: check(lib.dm_groupnorm_act_bwd(
                ctypes.byref(d), p(x1), None, p(g), p(b), p(ss), None, p(mean), p(rstd), p(dy), None, None,
                p(dx1), None, p(dss), None, p(pp), 0, p(ws), st)), iters=50)
            n = B * HW * C
            print(f"HW={HW:4d} C={C} B={B:4d} workgroups={B * C // 64:5d} | fwd {t_f*1e3:6.1f} us {4*n/t_f/1e9:5.2f} TB/s | "
                  f"bwd {t_b*1e3:6.1f} us {6*n/t_b/1e9:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
