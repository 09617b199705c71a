"""Benchmark: CIFAR-10 UNet training step (arg_json/cos_cifar.json, bf16 MFMA compute, batch 128 per GPU) on N
MI355X GPUs of one node, synthetic data, through the libdm_hip kernels.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = bernoulli/eps/t draws + noising + UNet forward + loss + backward + (all-reduce) + global-norm clip +
AdamW + EMA, exactly the work of train.py:429-452. Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel family (the implicit-GEMM convolution: forward
and data-gradient launches) from per-launch HIP-event timings taken in one extra instrumented step after the
timed region; `cpu_baseline` times the CPU oracle's training step on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# Hardware queues per process (HIP's default is 4), read when the runtime library loads. A step uses a main stream,
# one or two side streams, an auxiliary one, the graph's own and the collective library's: with more streams than
# hardware queues, unrelated streams share a queue and wait behind each other's barriers. (No effect on the one-GPU
# numbers: 9.00 ms/step with 4 and with 8.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "diffusion-models_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_GFLOP_PER_IMG = {"cos_cifar": 16.383, "beta_cifar": 16.383, "rf_cifar_long": 16.383, "rf_cifar": 16.2,
                     "rf_mnist": 3.108, "fid_test": 2.881}
MFMA_BF16_PEAK_TFLOPS = 2500.0


def measured_conv_traffic():
    """Average HBM-side bytes per forward/data-gradient conv launch, from the newest profiles/r*_conv_traffic.json
    (rocprofv3 --pmc passes of this benchmark; counters cannot be read from inside the run) and that file's name.
    (None, None) if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_conv_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return round(json.load(f)["conv_fwd_dgrad_bytes_per_launch"]), "profiles/" + os.path.basename(files[-1])


def load_cfg(name):
    with open(os.path.join(ROOT, "tests", "golden", "arg_json", name + ".json")) as f:
        return json.load(f)


def build_step(cfg, device, use_ema=True, bucket_mb=25):
    from modeling import get_model
    from diffusion.scheduler import get_scheduler
    from harness.train_step import TrainStep
    torch.manual_seed(cfg.get("seed", 42))
    model = get_model(cfg["model_type"], **cfg["model_cfg"]).to(device).train()
    sched = get_scheduler(cfg["scheduler_type"], **cfg["scheduler_cfg"])
    step = TrainStep(model, sched, lr=cfg["lr"], betas=(0.9, 0.99), clip_grad_norm=cfg.get("clip_grad_norm", 1.0),
                     use_ema=cfg.get("use_ema", True) and use_ema, ema_inv_gamma=cfg.get("ema_inv_gamma", 1.0),
                     ema_power=cfg.get("ema_power", 0.75), p_uncond=cfg.get("p_uncond", 0.2), bucket_mb=bucket_mb)
    return model, sched, step


def synthetic_batch(cfg, rank, device, batch):
    g = torch.Generator().manual_seed(42 + rank)
    c = cfg["model_cfg"]["in_channels"]
    hw = 28 if cfg["dataset"] == "mnist" else 32
    x = torch.rand(batch, c, hw, hw, generator=g) * 2 - 1
    cls = torch.randint(0, 10, (batch,), generator=g)
    return x.to(device), cls.to(device)


HBM_PEAK_TBS = 8.0


def _gn_elems(d):
    return d.B * d.HW * (d.C1 + d.C2)


# entry point -> (kernel family label, algorithmic bytes of one call from its ctypes arguments): every tensor the
# operator has to touch, once (bf16 activations 2 B, fp32 4 B); descriptors arrive as ctypes byref objects
HBM_OPS = {
    "dm_groupnorm_act_fwd": ("gn_fwd (GroupNorm+scale/shift+SiLU forward)", lambda a: 4.0 * _gn_elems(a[0]._obj)),
    "dm_groupnorm_act_bwd": ("gn_bwd (GroupNorm+SiLU backward)",
                             lambda a: (6.0 + 2.0 * bool(a[10]) + 2.0 * bool(a[11])) * _gn_elems(a[0]._obj)),
    "dm_colsum_bf16": ("colsum (conv bias gradient)", lambda a: 2.0 * a[2] * a[3]),
    "dm_colsum_bf16_dual": ("colsum (conv bias gradient)", lambda a: 2.0 * a[3] * a[4]),
    "dm_grad_norm": ("sumsq (global gradient norm)", lambda a: 4.0 * a[1]),
    "dm_adamw_ema_step": ("adamw_ema (clip+AdamW+EMA+bf16 mirror)", lambda a: 38.0 * a[8]),
    "dm_diffuse": ("diffuse (noising)", lambda a: 12.0 * a[6] * a[7]),
    "dm_mse_loss_fwd": ("mse_fwd (loss)", lambda a: (12.0 if a[1] else 8.0) * a[5]),
    "dm_mse_loss_bwd": ("mse_bwd (loss gradient)", lambda a: (16.0 if a[1] else 12.0) * a[5]),
    "dm_sampler_step": ("sampler_step (DDPM/DDIM/Euler update)", lambda a: (16.0 if a[2] else 12.0) * a[10]),
}
CONV_OPS = ("dm_conv2d_igemm", "dm_conv2d_igemm_gn", "dm_conv2d_igemm_proj", "dm_conv2d_wgrad", "dm_conv2d_wgrad_group")
# Kernels of a few microseconds: an event bracket around ONE eager call times the host's launch path, not the kernel.
# These are timed as BURST launches of the same call replayed from a HIP graph (idempotent or throw-away outputs).
BURST_OPS = ("dm_diffuse", "dm_mse_loss_fwd", "dm_mse_loss_bwd", "dm_sampler_step", "dm_grad_norm")
BURST = 20


class AbiTimer:
    """Wraps entry points of the C ABI with HIP events on the launching stream (torch's current stream: the timed
    passes run with the side streams off). Convolutions are recorded with their flops, the HBM-bound operators with
    their algorithmic bytes."""

    def __init__(self):
        from dm_hip import lib
        self.lib = lib
        self.records = []

    def __enter__(self):
        self._orig = {n: getattr(self.lib, n) for n in CONV_OPS + tuple(HBM_OPS)}
        for name, fn in self._orig.items():
            setattr(self.lib, name, self._wrap(name, fn))
        return self

    def _wrap(self, name, fn):
        def call(*args):
            if name == "dm_conv2d_wgrad_group":     # (items, n, ws, ws_bytes, stream): the layers of one batched call
                work = nbytes = 0.0
                for i in range(args[1]):
                    d = args[0][i].desc
                    m = d.B * d.Ho * d.Wo
                    work += 2.0 * m * d.Cout * (d.C1 + d.C2) * d.KH * d.KW
                    nbytes += 2.0 * (d.B * d.Hs * d.Ws * (d.C1 + d.C2) + m * d.Cout) + 4.0 * d.Cout * d.KH * d.KW * (d.C1 + d.C2)
                shape = None
            elif name in CONV_OPS:
                d = args[0]._obj
                m = d.B * d.Ho * d.Wo
                work = 2.0 * m * d.Cout * (d.C1 + d.C2) * d.KH * d.KW
                # algorithmic bytes: every operand once (input, weights, output, residual of the fused epilogue)
                nbytes = 2.0 * (d.B * d.Hs * d.Ws * (d.C1 + d.C2) + d.Cout * d.KH * d.KW * (d.C1 + d.C2) + m * d.Cout)
                if name == "dm_conv2d_igemm_proj":   # + the 1x1 projection of the second input, in the same launch
                    pr = args[5]._obj
                    work += 2.0 * m * d.Cout * (pr.C1 + pr.C2)
                    nbytes += 2.0 * (m * (pr.C1 + pr.C2) + d.Cout * (pr.C1 + pr.C2))
                elif name.startswith("dm_conv2d_igemm") and args[5]:
                    nbytes += 2.0 * m * d.Cout
                if name == "dm_conv2d_igemm_gn":    # the normalised tensor as well; flops: the convolution's only
                    nbytes += 2.0 * m * d.Cout
                shape = (m, d.Cout, d.C1 + d.C2, d.KH, d.stride, d.ups, d.transposed)
            else:
                work, nbytes, shape = 0.0, float(HBM_OPS[name][1](args)), None
                if name.startswith("dm_groupnorm_act"):
                    g = args[0]._obj
                    shape = ("gn", name[17:], g.B, g.HW, g.C1, g.C2, bool(len(args) > 10 and name.endswith("bwd") and args[10]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            ms = self._burst_ms(fn, args) if name in BURST_OPS else None
            self.records.append((name, work, e0, e1, nbytes, shape, ms))
            return rc
        return call

    @staticmethod
    def _burst_ms(fn, args):
        """Average duration of one launch of fn(*args) (its last argument is the stream) inside a graph of BURST of
        them, replayed once warm; None if the capture fails (the single-call bracket is used then)."""
        try:
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            from dm_hip import ops
            g = ops.new_graph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(g, stream=st):
                    for _ in range(BURST):
                        fn(*args[:-1], st.cuda_stream)
                g.replay()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
            torch.cuda.current_stream().wait_stream(st)
            e1.synchronize()
            return e0.elapsed_time(e1) / BURST
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench] burst timing failed ({e!r}); single-call bracket used\n")
            return None

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(self.lib, name, fn)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        dump = os.environ.get("DM_BENCH_DUMP_CONVS")
        for name, flops, e0, e1, nbytes, shape, burst_ms in self.records:
            ms = burst_ms if burst_ms is not None else e0.elapsed_time(e1)
            if dump and shape and shape[0] == "gn":
                sys.stderr.write(f"[gn] {shape[1]} B={shape[2]} HW={shape[3]:5d} C={shape[4]}+{shape[5]} extra={int(shape[6])} "
                                 f"{nbytes / 1e6:7.1f} MB {ms * 1e3:7.1f} us {nbytes / ms / 1e9:6.2f} TB/s\n")
            elif dump and shape:
                sys.stderr.write(f"[conv] {name[11:]:6s} M={shape[0]:6d} N={shape[1]:4d} C={shape[2]:4d} k={shape[3]} s={shape[4]} "
                                 f"ups={shape[5]} T={shape[6]} {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF\n")
            s = out.setdefault(name, {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0})
            s["launches"] += 1
            s["flops"] += flops
            s["bytes"] += nbytes
            s["ms"] += ms
        return out


def hbm_rows(summary):
    """One row per HBM-bound operator family: algorithmic bytes and HIP-event time per launch -> TB/s against 8 TB/s."""
    fam = {}
    for name, s in summary.items():
        if name not in HBM_OPS:
            continue
        f = fam.setdefault(HBM_OPS[name][0], {"launches": 0, "bytes": 0.0, "ms": 0.0})
        for k in f:
            f[k] += s[k]
    rows = []
    for label, f in fam.items():
        if f["ms"] <= 0:
            continue
        tbs = f["bytes"] / (f["ms"] * 1e-3) / 1e12
        rows.append({"kernel": label, "launches": f["launches"], "bytes_per_launch": round(f["bytes"] / f["launches"]),
                     "us": round(f["ms"] * 1e3 / f["launches"], 2), "tb_s": round(tbs, 3),
                     "frac": round(tbs / HBM_PEAK_TBS, 4)})
    return rows


def cpu_baseline(cfg, budget_s=20.0, batch=16):
    """The CPU oracle's training step (oracle/unet_oracle.py) on the host cores: a bounded sample."""
    from modeling import get_model
    from oracle import unet_oracle as O
    torch.manual_seed(42)
    m = get_model(cfg["model_type"], **cfg["model_cfg"])
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = O.param_names(sd)
    opt = O.AdamWState({k: sd[k] for k in names}, lr=cfg["lr"])
    if cfg["scheduler_type"] == "beta":
        sc = cfg["scheduler_cfg"]
        beta = O.beta_cosine(sc.get("cos_s", 0.008), sc.get("cos_max_beta", 0.999), sc["n_steps"]) \
            if sc.get("schedule") == "cosine" else O.beta_linear(n_steps=sc["n_steps"])
        kind, table = "vp", O.alpha_sq_from_beta(beta)
    else:
        kind, table = "rf", O.rf_sigmas(cfg["scheduler_cfg"]["n_steps"])
    g = torch.Generator().manual_seed(42)
    c = cfg["model_cfg"]["in_channels"]
    hw = 28 if cfg["dataset"] == "mnist" else 32
    x = torch.rand(batch, c, hw, hw, generator=g) * 2 - 1
    cls = torch.randint(0, 10, (batch,), generator=g)
    steps, t_total = 0, 0.0
    for i in range(1000):
        mask = torch.bernoulli(0.2 * torch.ones_like(cls, dtype=torch.float32), generator=g)
        eps, t = O.draw_eps_t(x, 1000, g)
        t0 = time.perf_counter()
        O.train_step(sd, cfg["model_cfg"], opt, (x, cls, mask, eps, t), kind, table, clip=1.0)
        dt = time.perf_counter() - t0
        if i > 0:  # first step is warm-up
            steps += 1
            t_total += dt
            if t_total > budget_s:
                break
    return {"value": round(batch * steps / t_total, 3), "unit": "img/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{steps} oracle train steps of {cfg['output_dir'].split('/')[-1]} at batch "
                                      f"{batch} (fp32, torch CPU primitives) after 1 warm-up step"}


def sampling_row(model, cfg, device, rank, world, config_name, sampler_type=None, n_steps=None, batch=256,
                 repeats=1, warmup=1, use_graph=True, chains=1):
    """Sampling throughput of one (sampler, n_steps) setting: every rank runs the whole n-step loop on `batch` images
    (train.py:476-504: get_pred_fn + sampler.sample + clip), UNet forward replayed from a HIP graph. Returns a dict
    (on every rank; times are the max over ranks)."""
    from diffusion.sampler import get_sampler
    from diffusion.scheduler import get_scheduler
    from harness.generate import GraphedPredictor, interleaved_pred_fn
    sched = get_scheduler(cfg["scheduler_type"], **cfg["scheduler_cfg"])
    stype = sampler_type or cfg["sampler_type"]
    scfg = dict(cfg["sampler_cfg"]) if stype == cfg["sampler_type"] else {}
    scfg["n_steps"] = n_steps or scfg.get("n_steps", 50)
    scfg["pbar"] = False
    sampler = get_sampler(stype, **scfg)
    g = torch.Generator().manual_seed(42 + rank)
    c = cfg["model_cfg"]["in_channels"]
    hw = 28 if cfg["dataset"] == "mnist" else 32
    z = torch.randn(batch, c, hw, hw, generator=g).to(device)
    cls = torch.randint(0, 10, (batch,), generator=g).to(device)
    was_training = model.training
    model.eval()
    pred = interleaved_pred_fn(model, cls, cfg.get("guidance_scale", 1.0), chains)
    if use_graph:
        pred = GraphedPredictor(pred)

    def run():
        with torch.no_grad():
            return torch.clip(sampler.sample(z.clone(), sched, pred), -1, 1)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if use_graph:   # capture the loop's graph outside the timed region (two replays of it)
        pred.replay_limit = 2
        run()
        pred.replay_limit = None
    for _ in range(warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(repeats):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    if was_training:
        model.train()
    per = elapsed / repeats
    n = scfg["n_steps"]
    fwd_tflop = FWD_GFLOP_PER_IMG.get(config_name, 0.0) * batch * n / 1e3
    tf = fwd_tflop / per if fwd_tflop else None
    return {"sampler": stype, "n_steps": n, "batch_per_gpu": batch, "img_s": round(batch * world / per, 2),
            "s_per_batch": round(per, 4), "unet_forward_ms": round(per * 1e3 / n, 3),
            "tflops": round(tf, 1) if tf else None,
            "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4) if tf else None, "hip_graph": bool(use_graph),
            "interleaved_sub_batches": chains,
            # the reference samples in fp32; bound of the difference measured on reference-trained networks against
            # the reference's own images (tests/test_gpu_model.py): relative L2 <= 6.2e-3 over DDIM-50 / Euler-50,
            # 8.9e-5 over DDPM ancestral sampling with all 1000 steps
            "activations": "bf16 (the reference samples in fp32)",
            "image_error_bound_vs_fp32_reference": 1e-3 if (stype == "ddpm" and n >= 1000) else 1.5e-2}


def bench_sampling(args, cfg, device, rank, world):
    """--mode sample: the sampling loop of the config as the headline line (one "step" = one whole n-step loop)."""
    from modeling import get_model
    torch.manual_seed(42)
    model = get_model(cfg["model_type"], **cfg["model_cfg"]).to(device).eval()
    row = sampling_row(model, cfg, device, rank, world, args.config, sampler_type=args.sampler,
                       n_steps=args.sample_steps, batch=args.sample_batch, repeats=args.steps, warmup=args.warmup,
                       use_graph=not args.no_graph, chains=args.sample_chains)
    if rank == 0:
        emit({
            "metric": f"sample imgs/sec @{row['n_steps']} steps ({row['sampler']})", "value": row["img_s"],
            "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(row["s_per_batch"] * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"arg_json/{args.config}.json sampling loop, {row['sampler']} {row['n_steps']} steps, "
                                   f"batch {args.sample_batch}/GPU", "parallelism": f"shard{world}",
                       "hip_graph": not args.no_graph},
            "unet_forward_ms": row["unet_forward_ms"], "step_tflops_achieved": row["tflops"],
            "roofline": {"bound": "mfma", "kernel": "UNet forward (all kernels of one sampler step)",
                         "achieved": row["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": row["frac"],
                         "traffic": None},
            "cpu_baseline": None})
    if world > 1:
        dist.destroy_process_group()


_REAL_STDOUT = None


def emit(line: dict):
    """The one JSON line, on the process's original stdout."""
    text = json.dumps(line) + "\n"
    if _REAL_STDOUT is None:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, text.encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cos_cifar")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's batch_size)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step into a HIP graph")
    ap.add_argument("--split-graph", action="store_true",
                    help="capture forward+backward and the optimizer tail as two graphs with the gradient all-reduce "
                         "between them (what runs on more than one GPU; on one GPU for testing)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="one GPU: create an RCCL process group of one rank and run the data-parallel collectives "
                         "(implies --split-graph)")
    ap.add_argument("--pg-port", type=int, default=29731)
    ap.add_argument("--bucket-mb", type=int, default=25, help="gradient all-reduce bucket size (MiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--mode", choices=["train", "sample"], default="train",
                    help="train: the training step (the headline metric); sample: the sampling loop of the config")
    ap.add_argument("--sample-batch", type=int, default=256, help="images per GPU per sampling batch (generation_batch_size)")
    ap.add_argument("--sample-steps", type=int, default=None, help="sampler steps (default: the config's)")
    ap.add_argument("--sample-chains", type=int, default=2,
                    help="run every UNet forward of the sampling loop as this many interleaved sub-batches on streams of their own")
    ap.add_argument("--sampler", default=None, help="sampler type override for --mode sample (ddpm, ddim, euler)")
    ap.add_argument("--no-sampling", action="store_true",
                    help="train mode: skip the sampling rows (config sampler, and DDPM ancestral 1000 steps on VP configs)")
    ap.add_argument("--side-streams", type=int, default=None,
                    help="streams the weight-gradient kernels rotate over beside the main one (0: none; default: the harness's)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # A rank that is stuck (a collective nobody joins, a hung launch) says where and ENDS: Python stacks of every thread
    # to stderr and exit status 1 after DM_BENCH_WATCHDOG seconds (default: 15 minutes on more than one GPU, where the
    # launcher then takes the other ranks down; off on one GPU unless set). The whole default run takes 2-3 minutes.
    watchdog = int(os.environ.get("DM_BENCH_WATCHDOG", "900" if world > 1 else "0"))
    if watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(watchdog, exit=True)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.force_collectives:
        args.split_graph = True
    if os.environ.get("DM_BENCH_SHARE_GPU") == "1":
        local_rank = 0      # rehearsal on a one-GPU box: every rank on the one device (with DM_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # Everything runs on a stream of the benchmark's own, not the legacy default stream: that one synchronises
    # implicitly with other streams, which does not mix with a collective library's streams working between the
    # launches (two ranks over gloo replaying the graph chain on it hung; never on a created stream).
    torch.cuda.set_stream(torch.cuda.Stream(device))
    if world > 1 or args.force_collectives:
        # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the one JSON
        # line by pointing file descriptor 1 at stderr for the run; emit() writes to the saved descriptor
        sys.stdout.flush()
        global _REAL_STDOUT
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        # (a collective that does not complete within five minutes is aborted by the process group's own watchdog,
        # which ends the process with an error)
        dist.init_process_group(os.environ.get("DM_BENCH_BACKEND", "nccl"), device_id=device,
                                timeout=datetime.timedelta(minutes=5))
    elif args.force_collectives:
        # a process group of one rank over RCCL: every collective of the multi-GPU path really runs (one-GPU rehearsal)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{args.pg_port}", rank=0, world_size=1,
                                device_id=device)

    cfg = load_cfg(args.config)
    if args.mode == "sample":
        return bench_sampling(args, cfg, device, rank, world)
    batch = args.batch or cfg["batch_size"]
    model, sched, step = build_step(cfg, device, bucket_mb=args.bucket_mb)
    if args.side_streams is not None:
        step.side_wgrad = args.side_streams
    x, cls = synthetic_batch(cfg, rank, device, batch)

    from dm_hip import ops
    graph = None
    run = lambda: step.step(x, cls)  # noqa: E731
    # eager warm-up (also primes allocator pools and packed weights)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    split = world > 1 or args.split_graph
    n_segments = None
    if not args.no_graph:
        try:
            ops.FORCE_REPACK = True
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            if not split:
                with torch.cuda.stream(side):
                    graph = ops.new_graph()
                    run()  # warm-up on the capture stream
                    torch.cuda.synchronize()
                    with torch.cuda.graph(graph, stream=side):
                        loss_static = step.step(x, cls)
                    run = graph.replay
            else:
                # More than one GPU: forward + backward is a chain of graphs cut wherever a gradient bucket becomes
                # complete; at replay each bucket's RCCL all-reduce is issued behind the graph that completes it and
                # runs beside the rest of backward (collectives stay outside graph capture); the optimizer tail is
                # one more graph behind the last collective. --split-graph: the same form on one GPU (the cuts are
                # made, the collectives run only with --force-collectives).
                step.reducer.force = world == 1
                with torch.cuda.stream(side):
                    step.forward_backward(x, cls); step.reduce(); step.update()  # warm-up on the capture stream
                torch.cuda.current_stream().wait_stream(side)
                loss_static = step.capture_segmented(x, cls, stream=side)
                torch.cuda.current_stream().wait_stream(side)
                g_upd = ops.new_graph()
                with torch.cuda.graph(g_upd, stream=side):
                    step.update()
                graph = step.segments[0][0]
                n_segments = len(step.segments[0])

                def run():
                    step.replay_segmented()
                    g_upd.replay()
            torch.cuda.current_stream().wait_stream(side)
        except Exception as e:  # capture is an optimisation; the eager path is the same kernels
            sys.stderr.write(f"[bench] graph capture failed, running eagerly: {e!r}\n")
            graph = None
            step.reducer.force = False
            run = lambda: step.step(x, cls)  # noqa: E731
        finally:
            ops.FORCE_REPACK = False

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    ms_per_step = elapsed / args.steps * 1e3
    value = batch * world * args.steps / elapsed

    # dominant-kernel roofline and per-kernel HBM rates from one instrumented eager step (every launch alone on one
    # stream, HIP events around each entry point)
    roof, hbm, summ = None, None, {}
    if rank == 0:
        step.side_wgrad = 0
        step.reducer.enabled = False     # this rank's own step: the other ranks run no collectives beside it
        with AbiTimer() as ct:
            step.step(x, cls)
            # the sampler update kernels: three eager steps of each family on a generation-sized batch
            from diffusion.sampler import get_sampler
            zz = torch.randn(256, *x.shape[1:], device=device)
            fake = lambda z_, t_: z_  # noqa: E731  (the update kernel is what is timed here, not the network)
            if sched.pred_type == "noise":
                get_sampler("ddim", n_steps=3, pbar=False).sample(zz, sched, fake)
                get_sampler("ddpm", n_steps=3, pbar=False).sample(zz, sched, fake)
            else:
                get_sampler("euler", n_steps=3, pbar=False).sample(zz, sched, fake)
        summ = ct.summary()
        ig = {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0}
        for key in ("dm_conv2d_igemm", "dm_conv2d_igemm_gn", "dm_conv2d_igemm_proj"):
            for k, v in summ.get(key, {}).items():
                ig[k] += v
        fused_gn = summ.get("dm_conv2d_igemm_gn", {}).get("launches", 0)
        fused_proj = summ.get("dm_conv2d_igemm_proj", {}).get("launches", 0)
        wg = {"launches": 0, "flops": 0.0, "ms": 0.0, "bytes": 0.0}
        for key in ("dm_conv2d_wgrad", "dm_conv2d_wgrad_group"):
            for k, v in summ.get(key, {}).items():
                wg[k] += v
        if ig and ig["ms"] > 0:
            achieved = ig["flops"] / (ig["ms"] * 1e-3) / 1e12
            traffic, traffic_file = measured_conv_traffic()
            roof = {"bound": "mfma",
                    "kernel": "conv_patch_kernel<128,6> / conv_patch_kernel<64,7> / conv_igemm_kernel<128>: every "
                              "dm_conv2d_igemm / _gn / _proj launch of a step (forward + data gradient; "
                              f"{fused_gn} of them carry the following GroupNorm in their epilogue, priced at the "
                              f"convolution's flops alone; {fused_proj} carry a block's 1x1 residual projection, priced "
                              "with its flops)",
                    "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": f"{traffic_file} (rocprofv3 --pmc passes of this benchmark: "
                                      "2 x FETCH_SIZE + WRITE_SIZE; not measured by this run)" if traffic else None,
                    "algorithmic_bytes_per_launch": round(ig["bytes"] / ig["launches"]),
                    "launches_per_step": ig["launches"], "ms_per_step": round(ig["ms"], 3),
                    "wgrad": {"kernel": "conv_wgrad9_kernel<4> / conv_wgrad3_kernel<4> / conv_wgrad_kernel<..> + their slab-sum kernels",
                              "achieved": round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12, 2),
                              "frac": round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                              "launches_per_step": wg["launches"],
                              "ms_per_step": round(wg["ms"], 3)} if wg and wg["ms"] > 0 else None}
        hbm = hbm_rows(summ)

    # sampling rows (BASELINE metric: "sample imgs/sec @N steps"): the config's own sampler, and DDPM ancestral
    # sampling over all 1000 steps where the scheduler is variance preserving
    sampling = None
    if not args.no_sampling:
        sampling = [sampling_row(model, cfg, device, rank, world, args.config, batch=args.sample_batch, chains=args.sample_chains)]
        if sched.pred_type == "noise":
            sampling.append(sampling_row(model, cfg, device, rank, world, args.config, sampler_type="ddpm",
                                         n_steps=cfg["scheduler_cfg"]["n_steps"], batch=args.sample_batch, warmup=0,
                                         chains=args.sample_chains))

    if rank == 0:
        step_tflop = 3 * FWD_GFLOP_PER_IMG.get(args.config, 0.0) * batch / 1e3
        line = {
            "metric": "train imgs/sec (node) CIFAR-10 UNet",
            "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"arg_json/{args.config}.json train step (UNet {cfg['model_cfg']['block_out_channels']}, "
                                   f"{cfg['scheduler_cfg'].get('schedule', cfg['scheduler_type'])} schedule, AdamW+clip+EMA)",
                       "per_gpu_batch": batch, "global_batch": batch * world, "image": "3x32x32",
                       "parallelism": f"dp{world}", "hip_graph": graph is not None,
                       **({"graph_segments": n_segments} if n_segments else {})},
            "step_tflops_achieved": round(step_tflop / (ms_per_step * 1e-3), 2) if step_tflop else None,
            "step_mfma_frac": round(step_tflop / (ms_per_step * 1e-3) / MFMA_BF16_PEAK_TFLOPS / world, 4) if step_tflop else None,
            "roofline": roof,
            "hbm": hbm,
            "sampling": sampling,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, budget_s=args.cpu_budget)
        else:
            line["cpu_baseline"] = None
        emit(line)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
