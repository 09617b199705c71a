"""Batched, rank-sharded sample generation (reference: train.py:141-160 seeds, train.py:476-504 and
save_fid_eval.py:126-172 generation loops).

Each rank takes one contiguous shard of the (z, cls) seeds, runs the whole n-step sampler on batches of
`batch_size` independently (no collective inside the loop), clips to [-1, 1] as the reference does, and the
shards are gathered once at the end.

The UNet forward of a sampler step is ~150 kernel launches whose arguments do not change from step to step
except for the contents of z and t, so it is captured ONCE per (batch size, guidance mode) into a HIP graph
together with the sampler's update kernel (its per-step scalars come from a device table indexed by a device step
counter, which also writes the next step's t) and, for DDPM, the noise draw: a sampler step is ONE graph launch.
"""
import weakref
from typing import Callable, Dict, Optional, Tuple

import os

import torch
import torch.distributed as dist


def make_generation_seed(dataset: str, n_examples: int, seed: Optional[int] = None, sample_labels: bool = False):
    """train.py:141-160: z ~ N(0, I) and labels from (separate, identically seeded) CPU generators."""
    def get_generator():
        return torch.Generator().manual_seed(seed) if seed is not None else None

    if dataset == "mnist":
        image_shape, n_classes = (1, 28, 28), 10
    elif dataset == "cifar10":
        image_shape, n_classes = (3, 32, 32), 10
    else:
        raise ValueError(f"unknown dataset {dataset}")
    z = torch.randn((n_examples, *image_shape), generator=get_generator())
    if sample_labels:
        cls = torch.randint(0, n_classes, (n_examples,), generator=get_generator())
    else:
        cls = torch.arange(n_examples) % n_classes
    return {"z": z, "cls": cls}


# model -> {(batch, labels?, guidance, chains, addresses of everything a graph reads): (label buffer, GraphedPredictor)}
_CAPTURED = weakref.WeakKeyDictionary()


class GraphedPredictor:
    """pred_fn(z, t) replayed from a HIP graph.

    The graph is captured at the first call, against the z buffer of that call (samplers update their state in
    place, so every later step passes the same buffer and no copy of z is needed; a different buffer of the
    same shape is copied into the captured one). t is copied into a captured int64 buffer. The returned
    prediction is the graph's output buffer: it is overwritten by the next call, which is what a sampler loop
    wants (the update kernel consumes it before the next forward).
    """

    def __init__(self, pred_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], warmup: int = 2):
        self.pred_fn = pred_fn
        self.warmup = warmup
        self.graph = None
        self.z = self.t = self.out = None
        self.loops = {}
        self.replay_limit = None   # when set, run_loop stops after this many steps (capture / warm-up runs)

    def _capture(self, z, t):
        self.z = z
        self.t = t.clone()
        side = torch.cuda.Stream(device=z.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # packs weights, sizes workspaces
                self.pred_fn(self.z, self.t)
        torch.cuda.current_stream().wait_stream(side)
        from dm_hip import ops
        self.graph = ops.new_graph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = self.pred_fn(self.z, self.t)

    # ---- a whole sampling loop as replays of ONE graph (offered to the samplers as pred_fn.run_loop) -----------
    def _capture_loop(self, key, state, n_steps, mode, clip, noisy):
        """Capture [UNet forward -> (noise draw) -> sampler update from the device table -> step counter / next t] once
        per (shape, n_steps, sampler family); nothing in it depends on the step index."""
        from dm_hip import ops
        dev, B = state.device, state.shape[0]
        L = {"z": torch.empty_like(state), "t": torch.zeros(B, dtype=torch.int64, device=dev),
             "step": torch.zeros(1, dtype=torch.int64, device=dev),
             "coefs": torch.zeros((n_steps, 4), dtype=torch.float32, device=dev),
             "t_table": torch.zeros(n_steps, dtype=torch.int64, device=dev),
             "gen": torch.Generator(device=dev) if noisy else None}
        L["z"].copy_(state)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # packs weights, primes the allocator
                self.pred_fn(L["z"], L["t"])
        torch.cuda.current_stream().wait_stream(side)
        graph = ops.new_graph()
        if noisy:
            graph.register_generator_state(L["gen"])
        with torch.cuda.graph(graph), torch.no_grad():
            pred = self.pred_fn(L["z"], L["t"])
            noise = torch.randn_like(L["z"], generator=L["gen"]) if noisy else None
            ops.sampler_step_dev(L["z"], pred, noise, L["coefs"], L["step"], L["t_table"], L["t"], n_steps, mode, clip)
        L["graph"] = graph
        self.loops[key] = L
        return L

    def run_loop(self, state, t_steps, rows, mode, clip, gen=None):
        """The sampler loop `for i: pred = pred_fn(state, t_i); update(state, pred, rows[i])` (diffusion/sampler.py) as
        n_steps replays of one HIP graph. gen: DDPM's noise generator (its seed and offset are taken over, so the noise
        is the sequence eager randn_like calls would draw). Returns the final state (a new tensor)."""
        n_steps = len(rows)
        key = (tuple(state.shape), n_steps, mode, bool(clip), gen is not None)
        L = self.loops.get(key) or self._capture_loop(key, state, n_steps, mode, clip, gen is not None)
        L["z"].copy_(state)
        L["coefs"].copy_(torch.tensor(rows, dtype=torch.float32))
        L["t_table"].copy_(t_steps.to(torch.int64))
        L["t"].copy_(L["t_table"][:1].expand(state.shape[0]))
        L["step"].zero_()
        if gen is not None:
            L["gen"].set_state(gen.get_state())
        for _ in range(n_steps if self.replay_limit is None else min(n_steps, self.replay_limit)):
            L["graph"].replay()
        return L["z"].clone()

    def __call__(self, z, t):
        if self.graph is None or z.shape != self.z.shape or z.dtype != self.z.dtype:
            self._capture(z, t)
        if z.data_ptr() != self.z.data_ptr():
            self.z.copy_(z)
        self.t.copy_(t)
        self.graph.replay()
        return self.out


def interleaved_pred_fn(model, cond: Optional[torch.Tensor], guidance_scale: float, chains: int):
    """pred_fn(z, t) that runs the batch as `chains` independent sub-batches, each on a stream of its own. The rows of
    a batch do not interact anywhere in the network (GroupNorm is per sample), so the result is the same bit for bit;
    what changes is the schedule: a forward pass is a strict chain of MFMA-bound convolutions and HBM-bound GroupNorm
    kernels, and two chains half a layer out of phase keep both the matrix cores and the memory system busy.
    Works eagerly and under HIP-graph capture (the forks and joins are captured)."""
    if chains <= 1:
        return model.get_pred_fn(cond=cond, guidance_scale=guidance_scale)
    device = next(model.parameters()).device
    streams = [torch.cuda.Stream(device=device) for _ in range(chains)]
    fns = {}

    def pred(z, t):
        B = z.shape[0]
        bounds = [shard_bounds(B, chains, c) for c in range(chains)]
        if B not in fns:   # per sub-batch closures over views of the (persistent) label buffer
            fns[B] = [model.get_pred_fn(cond=None if cond is None else cond[lo:hi], guidance_scale=guidance_scale)
                      for lo, hi in bounds]
        from dm_hip import ops
        cur = torch.cuda.current_stream()
        out = torch.empty_like(z, dtype=torch.float32)
        # the network's own side stream for the embedding path stays off inside the chains: a fork inside a forked
        # stream, once per chain in one capture, ends in a segmentation fault of the process (ROCm 7.2;
        # DM_CHAIN_EMB_SIDE=1 switches it on to reproduce that), and the other chain already fills the gaps it would fill
        emb_side = ops.EMB_SIDE_STREAM
        if os.environ.get("DM_CHAIN_EMB_SIDE", "0") != "1":
            ops.EMB_SIDE_STREAM = False
        try:
            for s, (lo, hi), f in zip(streams, bounds, fns[B]):
                if hi == lo:
                    continue
                s.wait_stream(cur)
                for v in (z, t, out):
                    v.record_stream(s)
                with torch.cuda.stream(s):
                    out[lo:hi].copy_(f(z[lo:hi], t[lo:hi]))
        finally:
            ops.EMB_SIDE_STREAM = emb_side
        for s in streams:
            cur.wait_stream(s)
        return out

    return pred


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n items for `rank`; the first n % world ranks hold one more item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@torch.no_grad()
def generate(model, sampler, scheduler, z: torch.Tensor, cls: Optional[torch.Tensor] = None,
             guidance_scale: float = 1.0, batch_size: int = 256, use_graph: bool = True,
             process_group=None, gather: bool = True, sampler_kwargs: Optional[Dict] = None,
             chains: int = 2) -> torch.Tensor:
    """Generate one image per row of z (train.py:476-504), sharded over the ranks of `process_group`.

    z / cls are the full seed tensors (host or device), identical on every rank. Returns the clipped fp32
    samples on the host: all of them on every rank when `gather`, else this rank's shard.
    Every batch starts its own `sampler.sample()` call, so a DDPM sampler re-creates its default-seeded noise
    generator per batch exactly as the reference does.
    chains: run every UNet forward as this many interleaved sub-batches (interleaved_pred_fn); same images, about
    15 % more of them per second at batch 256 with 2 (measured: UNet forward 5.9 -> 5.1 ms).
    """
    device = next(model.parameters()).device
    world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(process_group) if world > 1 else 0
    lo, hi = shard_bounds(z.size(0), world, rank)
    was_training = model.training
    model.eval()
    sampler_kwargs = sampler_kwargs or {}
    # Captured predictors (and the label buffers their graphs read) live with the model and serve every later call:
    # graphs are never destroyed (dm_hip.ops.new_graph), so one per call would pile up. A graph reads the parameters
    # through the pointers it was captured with: the key names them all, and the packed bf16 copies that a graph reads
    # in place of a weight are brought up to date here, outside the graph.
    cache = _CAPTURED.setdefault(model, {})
    where = None
    if use_graph:
        from dm_hip import ops
        where = tuple(t.data_ptr() for t in list(model.parameters()) + list(model.buffers())) + ops.refresh_packs(model)
    cond_buf: Dict[int, torch.Tensor] = {}
    outs = []
    for b0 in range(lo, hi, batch_size):
        b1 = min(hi, b0 + batch_size)
        zb = z[b0:b1].to(device)
        cond = None
        if use_graph:
            key = (b1 - b0, cls is not None, float(guidance_scale) if guidance_scale is not None else None, chains, where)
            if key not in cache:
                # one persistent label buffer per captured predictor: its graph keeps reading it
                cond = torch.empty(b1 - b0, dtype=torch.int64, device=device) if cls is not None else None
                cache[key] = (cond, GraphedPredictor(interleaved_pred_fn(model, cond, guidance_scale, chains)))
            cond, pred_fn = cache[key]
            if cond is not None:
                cond.copy_(cls[b0:b1])
        else:
            if cls is not None:
                cond = cond_buf.setdefault(b1 - b0, torch.empty(b1 - b0, dtype=torch.int64, device=device))
                cond.copy_(cls[b0:b1])
            pred_fn = interleaved_pred_fn(model, cond, guidance_scale, chains)
        samples = sampler.sample(zb, scheduler, pred_fn, **sampler_kwargs)
        outs.append(torch.clip(samples, -1, 1).cpu())
    if was_training:
        model.train()
    shape = tuple(z.shape[1:])
    mine = torch.cat(outs, dim=0) if outs else torch.empty((0, *shape), dtype=torch.float32)
    if world == 1 or not gather:
        return mine
    return gather_shards(mine, z.size(0), world, process_group, device)


def gather_shards(mine: torch.Tensor, n_total: int, world: int, process_group=None, device=None) -> torch.Tensor:
    """The one collective of generation: all_gather of the per-rank shards (padded to the largest shard)."""
    backend = dist.get_backend(process_group)
    dev = device if backend == "nccl" else torch.device("cpu")
    longest = max(shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world))
    padded = torch.zeros((longest, *mine.shape[1:]), dtype=mine.dtype, device=dev)
    padded[:mine.size(0)] = mine.to(dev)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=process_group)
    sizes = [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]
    return torch.cat([p[:s].cpu() for p, s in zip(parts, sizes)], dim=0)
