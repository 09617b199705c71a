"""Per-layer timing of the implicit-GEMM conv kernels (forward, data gradient, weight gradient) on the shapes of
the CIFAR UNet at batch 128. Prints TFLOP/s per shape; used to steer kernel work."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
from dm_hip import ConvDesc, check, lib  # noqa: E402

SHAPES = [
    # name, B, H, W, C1, C2, Cout, k, stride, ups
    ("L32 128->128", 128, 32, 32, 128, 0, 128, 3, 1, 0),
    ("L32 256->128", 128, 32, 32, 256, 0, 128, 3, 1, 0),
    ("L32 384->128", 128, 32, 32, 384, 0, 128, 3, 1, 0),
    ("L32 up 256->256", 128, 16, 16, 256, 0, 256, 3, 1, 1),
    ("L16 128->256", 128, 16, 16, 128, 0, 256, 3, 1, 0),
    ("L16 256->256", 128, 16, 16, 256, 0, 256, 3, 1, 0),
    ("L16 768->256", 128, 16, 16, 768, 0, 256, 3, 1, 0),
    ("L16 up 512->512", 128, 8, 8, 512, 0, 512, 3, 1, 1),
    ("L8 256->512", 128, 8, 8, 256, 0, 512, 3, 1, 0),
    ("L8 512->512", 128, 8, 8, 512, 0, 512, 3, 1, 0),
    ("L8 1024->512", 128, 8, 8, 1024, 0, 512, 3, 1, 0),
    ("L32 s2 128->128", 128, 32, 32, 128, 0, 128, 3, 2, 0),
    ("L32 1x1 256+128->128", 128, 32, 32, 256, 128, 128, 1, 1, 0),
    ("L8 1x1 512+512->512", 128, 8, 8, 512, 512, 512, 1, 1, 0),
    ("attn qkv 512->1536", 128, 8, 8, 512, 0, 1536, 1, 1, 0),
]


COLD = os.environ.get("DM_BENCH_COLD") == "1"
_flush = None


def timeit(fn, iters=20, warm=()):
    """Average time of fn(). With DM_BENCH_COLD=1 every call is preceded (outside the timed span) by a sweep over a
    768 MB buffer that evicts the caches and then by a touch of the tensors in `warm` - the state a layer finds inside
    a training step: activations just written by the previous kernel, weights not seen since the last step."""
    global _flush
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if not COLD:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    if _flush is None:
        _flush = torch.empty(768 << 20, dtype=torch.uint8, device="cuda:0")
    total = 0.0
    for _ in range(iters):
        _flush.zero_()
        for t in warm:
            t.add_(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
    return total / iters


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    only = sys.argv[1] if len(sys.argv) > 1 else None
    which = sys.argv[2] if len(sys.argv) > 2 else "fdw"   # f(orward) d(ata gradient) w(eight gradient)
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
    if os.environ.get("DM_SLAB_OVERHEAD"):
        check(lib.dm_config_set(b"wgrad_slab_overhead", int(os.environ["DM_SLAB_OVERHEAD"])))
    for name, B, H, W, C1, C2, Cout, k, stride, ups in SHAPES:
        if only and only not in name:
            continue
        pad = (k - 1) // 2
        Hi, Wi = H << ups, W << ups
        Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
        Cin = C1 + C2
        bf = torch.bfloat16
        x1 = torch.randn(B, H, W, C1, device=dev).to(bf)
        x2 = torch.randn(B, H, W, C2, device=dev).to(bf) if C2 else None
        w = (torch.randn(Cout, k * k, Cin, device=dev) * 0.02).to(bf)
        wt = (torch.randn(Cin, k * k, Cout, device=dev) * 0.02).to(bf)
        y = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=bf)
        dy = torch.randn(B, Ho, Wo, Cout, device=dev).to(bf)
        dx = torch.empty(B, Hi, Wi, Cin, device=dev, dtype=bf)
        dw = torch.zeros(Cout, k * k, Cin, device=dev)
        if os.environ.get("DM_BENCH_ZERO") == "1":   # all-zero operands: the clock the chip holds without data toggling
            for t_ in (x1, x2, w, wt, dy):
                if t_ is not None:
                    t_.zero_()
        flops = 2.0 * B * Ho * Wo * Cout * Cin * k * k
        d = ConvDesc(B, H, W, C1, C2, Cout, k, k, stride, pad, ups, 0, Ho, Wo)
        dt = ConvDesc(B, Ho, Wo, Cout, 0, Cin, k, k, stride, pad, 0, 1, Hi, Wi)
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        t_f0 = t_f = t_d = t_w = float("inf")
        if "f" in which:
            lib.dm_config_set(b"conv_patch", 0)
            t_f0 = timeit(lambda: check(lib.dm_conv2d_igemm(ctypes.byref(d), p(x1), p(x2), p(w), None, None, p(y), st)), warm=[t for t in (x1, x2) if t is not None])
            lib.dm_config_set(b"conv_patch", 1)
            t_f = timeit(lambda: check(lib.dm_conv2d_igemm(ctypes.byref(d), p(x1), p(x2), p(w), None, None, p(y), st)),
                         warm=[t for t in (x1, x2) if t is not None])
        if "d" in which:
            t_d = timeit(lambda: check(lib.dm_conv2d_igemm(ctypes.byref(dt), p(dy), None, p(wt), None, None, p(dx), st)), warm=[dy])
        if "w" in which:
            wsb = lib.dm_conv2d_wgrad_ws_bytes(ctypes.byref(d))
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            t_w = timeit(lambda: check(lib.dm_conv2d_wgrad(ctypes.byref(d), p(x1), p(x2), p(dy), p(dw), p(ws), wsb, st)), warm=[dy])
        for key, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[key][0] += flops
            tot[key][1] += t
        print(f"{name:24s} M={B*Ho*Wo:6d} N={Cout:4d} K={Cin*k*k:5d} GF={flops/1e9:7.1f} | gather-fwd {flops/t_f0/1e9:5.0f} TF | fwd {t_f*1e3:7.1f}us {flops/t_f/1e9:7.0f} TF"
              f" | dgrad {t_d*1e3:7.1f}us {flops/t_d/1e9:7.0f} TF | wgrad {t_w*1e3:7.1f}us {flops/t_w/1e9:7.0f} TF", flush=True)
    for key, (f, t) in tot.items():
        if t:
            print(f"total {key}: {t:.3f} ms, {f/t/1e9:.0f} TF")


if __name__ == "__main__":
    main()
