"""conv1 + norm2 of a ConvBlock as two launches against one launch with the GroupNorm in the convolution's epilogue,
on the CIFAR UNet's small-image layers at batch 128. Prints microseconds per variant (warm caches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
from dm_hip import lib, ops  # noqa: E402

SHAPES = [
    # name, B, H, W, Cin, Cout, ups
    ("L16 128->256", 128, 16, 16, 128, 256, 0),
    ("L16 256->256", 128, 16, 16, 256, 256, 0),
    ("L16 512->256", 128, 16, 16, 512, 256, 0),
    ("L8 256->256", 128, 8, 8, 256, 256, 0),
    ("L8 512->256", 128, 8, 8, 512, 256, 0),
    ("L16 64ch", 128, 16, 16, 64, 64, 0),
]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, B, H, W, Cin, Cout, ups in SHAPES:
        x = torch.randn(B, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * (9 * Cin) ** -0.5
        bias = torch.randn(Cout, device=dev) * 0.1
        gamma, beta = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        e = torch.randn(B, Cout, device=dev)
        pack = ops.WeightPack()
        norm = ops.FusedNorm(gamma, beta, 32, 1e-5, True, pre_add=e)

        def conv_only():
            return ops.conv2d(x, w, bias, pack, pad=1, ups=ups)

        def gn_only(y):
            return ops.group_norm_act(y, gamma, beta, 32, 1e-5, True, pre_add=e)

        def both():
            y, done = ops.conv2d(x, w, bias, pack, pad=1, ups=ups, norm=norm)
            return ops.group_norm_act(y, gamma, beta, 32, 1e-5, True, pre_add=e, precomputed=done)

        with torch.no_grad():
            y = conv_only()
            t_conv = timeit(conv_only)
            t_gn = timeit(lambda: gn_only(y))
            lib.dm_config_set(b"conv_gn_fuse", 0)
            t_sep = timeit(both)
            lib.dm_config_set(b"conv_gn_fuse", 1)
            t_fused = timeit(both)
        print(f"{name:16s} conv {t_conv:7.1f} us  gn {t_gn:6.1f} us  two launches {t_sep:7.1f} us  fused {t_fused:7.1f} us",
              flush=True)


if __name__ == "__main__":
    main()
