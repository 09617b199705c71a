"""Timing of the UNet's edge convolutions (image <-> features) at batch 128, 3x32x32, 128 features: the library in
the tree against an older build (tools/bin/libdm_hip_old.so, if present), back to back in one process."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
from bench_gn import timeit  # noqa: E402

P, I = ctypes.c_void_p, ctypes.c_int


def load(path):
    lib = ctypes.CDLL(path)
    for name, n_ptr in (("dm_in_conv_fwd", 4), ("dm_out_conv_fwd", 4), ("dm_out_conv_dgrad", 3)):
        f = getattr(lib, name)
        f.restype, f.argtypes = I, [P] * n_ptr + [I] * 5 + [P]
    return lib


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    B, Cs, H, W, Cb = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (128, 3, 32, 32, 128)
    x = torch.randn(B, Cs, H, W, device=dev)
    h = torch.randn(B, H, W, Cb, device=dev).to(torch.bfloat16)
    w_in, b_in = torch.randn(Cb, Cs, 3, 3, device=dev) * 0.1, torch.randn(Cb, device=dev)
    w_out, b_out = torch.randn(Cs, Cb, 3, 3, device=dev) * 0.1, torch.randn(Cs, device=dev)
    y_big = torch.empty_like(h)
    y_small = torch.empty_like(x)
    libs = [("tree", os.path.join(ROOT, "diffusion-models_amd", "dm_hip", "libdm_hip.so"))]
    old = os.path.join(ROOT, "tools", "bin", "libdm_hip_old.so")
    if os.path.exists(old):
        libs.append(("old", old))
    p = lambda t: t.data_ptr()  # noqa: E731
    mb = B * H * W * Cb * 2 / 1e6
    for tag, path in libs * 2:
        lib = load(path)
        t1 = timeit(lambda: lib.dm_in_conv_fwd(p(x), p(w_in), p(b_in), p(y_big), B, Cs, H, W, Cb, st), iters=50)
        t2 = timeit(lambda: lib.dm_out_conv_fwd(p(h), p(w_out), p(b_out), p(y_small), B, Cs, H, W, Cb, st), iters=50)
        t3 = timeit(lambda: lib.dm_out_conv_dgrad(p(x), p(w_out), p(y_big), B, Cs, H, W, Cb, st), iters=50)
        print(f"{tag:5s} in_conv fwd {t1*1e3:6.1f} us ({mb/t1/1e3:4.2f} TB/s) | out_conv fwd {t2*1e3:6.1f} us ({mb/t2/1e3:4.2f} TB/s)"
              f" | out_conv dgrad {t3*1e3:6.1f} us ({mb/t3/1e3:4.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
