"""How much of the optimizer tail (clip + AdamW + EMA + mirrors + gradient zeroing: HBM-bound) hides under the UNet's
forward pass (MFMA-bound) when the two run on different streams of one HIP graph: times the forward alone, the update
alone, and both forked from one point and joined at the end."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import torch  # noqa: E402
import bench  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    from dm_hip import ops
    device = torch.device("cuda:0")
    cfg = bench.load_cfg("cos_cifar")
    model, sched, step = bench.build_step(cfg, device)
    x, cls = bench.synthetic_batch(cfg, 0, device, cfg["batch_size"])
    for _ in range(2):
        step.step(x, cls)
    torch.cuda.synchronize()
    mask = torch.zeros(x.shape[0], device=device)

    def forward():
        with torch.no_grad():
            return sched.get_loss(x, model, cls=cls, uncond_mask=mask)

    side = torch.cuda.Stream()
    other = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    ops.FORCE_REPACK = True
    graphs = {}
    with torch.cuda.stream(side):
        forward(); step.update()
        torch.cuda.synchronize()
        for name in ("forward", "update", "both"):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                if name == "forward":
                    forward()
                elif name == "update":
                    step.update()
                else:
                    other.wait_stream(side)
                    with torch.cuda.stream(other):
                        step.update()
                    forward()
                    side.wait_stream(other)
            graphs[name] = g
    ops.FORCE_REPACK = False
    torch.cuda.current_stream().wait_stream(side)
    t = {k: timed(g.replay) for k, g in graphs.items()}
    print(f"forward {t['forward']:.3f} ms  update {t['update']:.3f} ms  forked {t['both']:.3f} ms  "
          f"(hidden: {t['forward'] + t['update'] - t['both']:.3f} ms)")


if __name__ == "__main__":
    main()
