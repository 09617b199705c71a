"""The training step of the reference's Trainer.train_on_batch (train.py:429-452), as a reusable harness.

    uncond_mask = bernoulli(p_uncond)            RNG draw 1
    loss = scheduler.get_loss(x, model, ...)     RNG draws 2 (eps) and 3 (t)
    loss.backward()
    clip_grad_norm_(params, max_norm)            global L2 norm over all parameters
    AdamW(lr, betas, eps=1e-8, weight_decay=1e-2).step(); zero_grad(); EMA.step()

MI355X-first layout: all parameters live in ONE flat fp32 arena (the nn.Parameters are views into it), and so
do gradients, Adam moments and the EMA shadow. The optimizer tail is then two kernels over contiguous memory
(norm, fused clip+AdamW+EMA) instead of ~660 per-tensor launches, and data-parallel gradient averaging is an
all-reduce over contiguous slices of the gradient arena, launched bucket by bucket (reverse registration order,
the order gradients become ready) while backward is still running.
"""
import contextlib
import ctypes
import sys
import math
import os
from typing import Optional

import torch
import torch.distributed as dist

import numpy as np

from dm_hip import WtItem, check, lib, ops
from harness.ddp import BucketedAllReduce
from modeling.layers import Conv2d

_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary


def arena_layout(names, numels, late=()):
    """Element offset of every parameter in the flat arenas, and their total size. `names` / `numels` are in
    registration order and so is the returned list (offsets[i] belongs to parameter i), but the PHYSICAL order differs in
    one respect: the parameters listed in `late` come first, in front of parameter 0. The data-parallel reducer cuts
    the arena into buckets from its end to its start - the order in which backward produces gradients - so the front
    of the arena is the bucket that leaves last. `late` names the parameters whose gradients exist only when backward
    is over wherever they are registered (the embedding path of the UNet: one grouped launch behind the last block's
    backward produces the gradients of every block's emb_proj). Left in registration order they would sit in every
    bucket and hold every all-reduce back to the end of backward."""
    late = set(late)
    order = [i for i, n in enumerate(names) if n in late] + [i for i, n in enumerate(names) if n not in late]
    offsets, off = [0] * len(names), 0
    for i in order:
        offsets[i] = off
        off += (numels[i] + _ALIGN - 1) // _ALIGN * _ALIGN
    return offsets, off


def late_region(names, offsets, late=()):
    """Where the `late` parameters of arena_layout end: [0, that) holds them and nothing else (0: there are none, or
    there is nothing behind them)."""
    late = set(late)
    rest = [o for n, o in zip(names, offsets) if n not in late]
    return min(rest) if (rest and any(n in late for n in names)) else 0


class FlatArena:
    """Parameters of a module re-homed into one contiguous fp32 buffer. self.params / names / shapes / offsets are in
    registration order (the order of state_dict() and of the optimizer state); the physical order is arena_layout's."""

    def __init__(self, module: torch.nn.Module):
        self.params = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        device = self.params[0][1].device
        conv_weights = {id(m.weight) for m in module.modules() if isinstance(m, Conv2d)}
        late_fn = getattr(module, "late_gradient_parameter_names", None)
        self.late = set(late_fn()) if late_fn is not None else set()
        self.offsets, off = arena_layout([n for n, _ in self.params], [p.numel() for _, p in self.params], self.late)
        self.front = late_region([n for n, _ in self.params], self.offsets, self.late)   # end of the late region
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.names = [n for n, _ in self.params]
        self.shapes = [tuple(p.shape) for _, p in self.params]
        # conv weights are stored physically [Cout][KH][KW][Cin]: the layout the weight-gradient kernel produces
        self.ohwi = [id(p) in conv_weights and p.dim() == 4 for _, p in self.params]
        for i, (n, p) in enumerate(self.params):
            view = self.view_of(self.data, i)
            view.copy_(p.data)
            p.data = view
            p.grad = self.view_of(self.grad, i)

    def view_of(self, buf: torch.Tensor, i: int) -> torch.Tensor:
        """Parameter i's slice of an arena-shaped buffer (data, grad, Adam moments, EMA shadow) as a view of the
        parameter's logical shape (for a conv weight: the reference's [Cout, Cin, KH, KW], strided)."""
        o, shape = self.offsets[i], self.shapes[i]
        n = 1
        for d in shape:
            n *= d
        flat = buf[o:o + n]
        if self.ohwi[i]:
            co, ci, kh, kw = shape
            return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat.view(shape)


LR_KINDS = {"constant": 0, "constant_with_warmup": 1, "linear": 2, "cosine": 3, "cosine_with_restarts": 4,
            "polynomial": 5, "inverse_sqrt": 6}


def lr_schedule_args(name, lr, warmup_steps=0, max_steps=None, cfg=None):
    """(kind, warmup, total, p0, p1) of dm_optimizer_tick_lr for the schedule that
    transformers.get_scheduler(name, optimizer, num_warmup_steps, num_training_steps, **cfg) builds in the reference
    (train.py:281-291), with its defaults and its argument errors."""
    if name is None:
        return (0, 0, 0, 0.0, 0.0)
    if name not in LR_KINDS:
        raise ValueError(f'unknown lr scheduler "{name}" (known: {sorted(LR_KINDS)})')
    # cfg is what the reference splats into get_scheduler (TrainArgs.lr_schduler_cfg): its one legal key is
    # scheduler_specific_kwargs, the dict that reaches the schedule function
    top = dict(cfg or {})
    cfg = dict(top.pop("scheduler_specific_kwargs", None) or {})
    if top:
        raise TypeError(f"get_scheduler() got an unexpected keyword argument {sorted(top)[0]!r}")
    kind = LR_KINDS[name]
    if kind <= 1:
        cfg = {}   # (get_scheduler does not pass them on to the constant rules)
    if kind >= 2 and kind <= 5 and max_steps is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")
    p0 = p1 = 0.0
    if kind == 3:
        p0 = float(cfg.pop("num_cycles", 0.5))
    elif kind == 4:
        p0 = float(int(cfg.pop("num_cycles", 1)))
    elif kind == 5:
        lr_end, p0 = float(cfg.pop("lr_end", 1e-7)), float(cfg.pop("power", 1.0))
        if not lr > lr_end:
            raise ValueError(f"lr_end ({lr_end}) must be smaller than initial lr ({lr})")
        p1 = lr_end / lr
    elif kind == 6:
        timescale = cfg.pop("timescale", None)
        p0 = float(timescale if timescale is not None else (warmup_steps or 10_000))
    if cfg:
        raise TypeError(f"unexpected arguments for the {name} lr scheduler: {sorted(cfg)}")
    return (kind, 0 if kind == 0 else int(warmup_steps), int(max_steps or 0), p0, p1)


class TrainStep:
    def __init__(self, model, scheduler, lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=None,
                 clip_grad_norm: Optional[float] = 1.0, use_ema=True, ema_inv_gamma=1.0, ema_power=0.75,
                 ema_max_decay=0.9999, p_uncond=0.2, bucket_mb=25, process_group=None, side_wgrad=None, wgrad_group=None,
                 optimizer="adamw", lr_scheduler=None, lr_warmup_steps=0, max_steps=None, lr_scheduler_cfg=None):
        self.model, self.scheduler = model, scheduler
        # train.py:265-279: torch.optim.AdamW (decoupled decay, default 1e-2) or torch.optim.Adam (default: none; with
        # none the two are the same update)
        if optimizer not in ("adamw", "adam"):
            raise ValueError(f"unknown optimizer {optimizer}")
        if weight_decay is None:
            weight_decay = 1e-2 if optimizer == "adamw" else 0.0
        elif optimizer == "adam" and weight_decay:
            raise NotImplementedError("Adam with a coupled L2 penalty is not built; the reference never sets one")
        self.optimizer_name = optimizer
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        # lr is the BASE rate; the rate of a step comes from the schedule, evaluated on the device (train.py:446-447)
        self.lr_scheduler_name = lr_scheduler
        self.lr_sched = lr_schedule_args(lr_scheduler, lr, lr_warmup_steps, max_steps, lr_scheduler_cfg)
        self.clip = float(clip_grad_norm) if clip_grad_norm else 0.0
        self.p_uncond = p_uncond
        self.side_wgrad = side_wgrad
        self.ema_cfg = (ema_inv_gamma, ema_power, ema_max_decay)
        self.arena = FlatArena(model)
        dev = self.arena.data.device
        n = self.arena.numel
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ema = self.arena.data.clone() if use_ema else None
        self.partials = torch.empty(lib.dm_sumsq_partials(), dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        self.steps = 0
        self._init_weight_mirrors()
        self.reducer = BucketedAllReduce(self.arena.grad, self.arena.offsets, bucket_mb << 20, process_group,
                                         front=self.arena.front)
        self.world = self.reducer.world
        if self.side_wgrad is None:
            self.side_wgrad = 1
        if self.world > 1 and self.side_wgrad and dev.type == "cuda":
            # every side stream that may carry parameter-gradient kernels (side_wgrad can be raised after
            # construction; streams that were never created carry nothing)
            self.reducer.extra_streams = lambda: [ops.side_stream(dev, 0)] + [
                s for s in ops.side_streams(dev) if s is not ops.side_stream(dev, 0)]
        self._in_backward = False
        self._report = False   # backward in progress reports finished gradients to the reducer
        self._seg = None   # state of a segmented capture in progress (capture_segmented)
        self._update_captured = False   # update() has run under HIP-graph capture: that graph holds lr, betas, eps, wd by value
        self._replay_stream = None
        self._sums = ops.SumCollector()
        if wgrad_group is None:
            # Layers per weight-gradient launch (<= 1: every layer launches its own, the default). Measured on the
            # CIFAR UNet at batch 128: groups of 4-8 cut the weight-gradient kernels' own time by 12 % (fewer partial
            # tiles, fuller rounds of workgroups) but the step got 5 % SLOWER - bunched into a few long launches the
            # MFMA-bound weight gradients no longer interleave with the HBM-bound GroupNorm kernels of the main
            # stream, which is where the second stream earns its keep.
            wgrad_group = int(os.environ.get("DM_WGRAD_GROUP", "0"))
        self._wgrads = None
        if wgrad_group > 1:
            self._wgrads = ops.WgradCollector(group=wgrad_group, tail=int(os.environ.get("DM_WGRAD_TAIL", "2")))
            from modeling.layers import Linear
            # every Conv2d / token Linear runs once per forward, so files one weight gradient per backward
            self._n_wgrads = sum(isinstance(m, (Conv2d, Linear)) for m in model.modules())
        self._param_index = {id(p): i for i, (_, p) in enumerate(self.arena.params)}
        # gradients that autograd accumulates itself report through this hook; gradients the kernels write in
        # place report through ops.GRAD_READY (set around backward); both are no-ops while the reducer is inactive
        for _, p in self.arena.params:
            p.register_post_accumulate_grad_hook(self._on_accumulated)

    # ---- bf16 weight mirrors ------------------------------------------------------------------------------
    def _init_weight_mirrors(self):
        """bf16 copies of the whole parameter arena in the two layouts the conv kernels read. The arena stores a
        conv weight as [Cout][KH][KW][Cin] and a linear weight as [out][in], which IS the forward operand layout,
        so the forward mirror is written by the optimizer kernel itself; the data-gradient mirror ([Cin][T][Cout]
        at the same offsets) is one batched transposition per step instead of two packing launches per layer."""
        a = self.arena
        dev = a.data.device
        self.w16 = torch.empty(a.numel, dtype=torch.bfloat16, device=dev)
        self.w16_t = torch.zeros(a.numel, dtype=torch.bfloat16, device=dev)
        offset_of = {id(p): o for (_, p), o in zip(a.params, a.offsets)}
        items, tiles, self._bound = [], 0, []
        for mod in self.model.modules():
            pack, w = getattr(mod, "_pack", None), getattr(mod, "weight", None)
            if not isinstance(pack, ops.WeightPack) or w is None or id(w) not in offset_of:
                continue
            if w.dim() == 4:
                co, ci, kh, kw = w.shape
                if w.stride() != (kh * kw * ci, 1, kw * ci, ci):
                    continue
            elif w.dim() == 2:
                (co, ci), kh, kw = w.shape, 1, 1
            else:
                continue
            if co % 8 or ci % 8:
                continue
            o = offset_of[id(w)]
            items.append((o, co, kh * kw, ci, tiles))
            tiles += kh * kw * ((co + 63) // 64) * ((ci + 63) // 64)
            self._bound.append((pack, w, o, w.numel()))
        self._wt_tiles = tiles
        self._wt_items = None
        if items:
            rec = np.zeros(len(items), dtype=np.dtype([("offset", np.int64), ("Cout", np.int32), ("T", np.int32),
                                                       ("Cin", np.int32), ("tile_begin", np.int32)]))
            assert rec.dtype.itemsize == ctypes.sizeof(WtItem)
            for i, it in enumerate(items):
                rec[i] = it
            self._wt_items = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
            self._wt_n = len(items)
        self.refresh_packed()

    def _transpose_mirror(self):
        if self._wt_items is not None:
            check(lib.dm_transpose_conv_weights(self.w16.data_ptr(), self.w16_t.data_ptr(), self._wt_items.data_ptr(),
                                                self._wt_n, self._wt_tiles, torch.cuda.current_stream().cuda_stream))

    def refresh_packed(self):
        """Rebuild both mirrors from the fp32 arena and (re)bind them to the layers. Call after writing parameters
        through torch (load_state_dict, copy_); the optimizer step keeps them current by itself."""
        self.w16.copy_(self.arena.data)
        self._transpose_mirror()
        for pack, w, o, n in self._bound:
            pack.bind(w, self.w16[o:o + n], self.w16_t[o:o + n])

    def _on_accumulated(self, p):
        """Autograd's post-accumulate hook. It also fires for a parameter whose operator returned no gradient because
        a kernel writes it in place - harmless once that kernel has been enqueued, but a convolution weight filed with
        the weight-gradient collector has no kernel yet: the collector reports it when it launches."""
        if self._wgrads is not None and id(p) in self._wgrads.pending_ids:
            return
        self._on_grad_ready(p)

    def _on_grad_ready(self, p):
        if self._in_backward and self._report:
            i = self._param_index.get(id(p))
            if i is not None:
                self.reducer.on_ready(i)

    # ---- one step ----------------------------------------------------------------------------------------
    def _backward(self, loss, overlap=True):
        """Backward into the gradient arena. overlap=True: launch each bucket's all-reduce as soon as it is complete
        (eager multi-GPU steps); False: the collectives are left to _reduce(), so that this phase contains no
        communication and can be captured into one HIP graph on every rank; "cut": the phase is being captured
        (capture_segmented) and the capture is cut into a new graph wherever a bucket becomes complete."""
        self.reducer.begin()
        self._in_backward = True
        cut = overlap == "cut"
        overlapped = self.reducer.active and bool(overlap)
        self.reducer.on_bucket = self._cut if cut else None
        self._report = overlapped
        ops.DIRECT_GRADS, ops.GRAD_READY = True, (self._on_grad_ready if overlapped else None)
        ops.SIDE_WGRAD = int(self.side_wgrad)
        # The ordered sums behind bias / GroupNorm-affine / edge-conv gradients run as ONE launch after backward -
        # except when buckets are all-reduced while backward runs: those gradients are spread over every bucket, so
        # each is then summed where it is produced and its bucket can leave early. A cut capture collects them and
        # sums what has been collected at every cut.
        ops.PENDING_SUMS = None if (overlapped and not cut) else self._sums
        # convolution weight gradients run several layers per launch
        ops.PENDING_WGRADS = self._wgrads
        if self._wgrads is not None:
            self._wgrads.begin_pass(self._n_wgrads)
        ops.SIDE_TOUCHED.clear()
        ops.SIDE_KEEP.clear()
        try:
            loss.backward()
            if self._wgrads is not None:
                self._wgrads.flush()    # the layers still waiting for neighbours (may complete buckets / cut)
        except BaseException:
            self._sums.count, self._sums.keep = 0, []   # a failed backward leaves nothing to sum
            if self._wgrads is not None:
                self._wgrads.drop()
            raise
        finally:
            ops.DIRECT_GRADS, ops.GRAD_READY, ops.SIDE_WGRAD, ops.PENDING_SUMS = False, None, 0, None
            ops.PENDING_WGRADS = None
            self._in_backward = self._report = False
            self.reducer.on_bucket = None
        self._join_sides()
        self._sums.flush()

    def _join_sides(self):
        """The weight/bias gradient kernels ran on side streams, the embedding path's backward on an auxiliary one:
        the current stream waits for every one that was given work since the last join, before anything reads the
        gradient arena. While capturing, also for every auxiliary stream that autograd pulled into the capture (a
        wait for a gradient is enough); never for a stream outside the capture. Returns the streams waited for."""
        cur = torch.cuda.current_stream()
        joined = list(ops.SIDE_TOUCHED)
        capturing = torch.cuda.is_current_stream_capturing()
        for aux in ops.aux_streams():
            if aux in joined:
                continue
            if capturing:
                with torch.cuda.stream(aux):
                    inside = torch.cuda.is_current_stream_capturing()
                if not inside:
                    continue
            joined.append(aux)
        for side in joined:
            cur.wait_stream(side)
        ops.SIDE_TOUCHED.clear()
        ops.SIDE_KEEP.clear()
        return joined

    # ---- forward + backward captured as a chain of HIP graphs cut at bucket boundaries ---------------------------
    def _cut(self, b):
        """Bucket b is complete (called from inside backward, between two operators, while capturing): close the
        current graph behind everything that produced the bucket and open the next one. At replay the bucket's
        all-reduce is issued between the two graphs and runs beside the rest of backward."""
        seg = self._seg
        if len(self.reducer.launched) == len(self.reducer.buckets):
            return   # the last bucket: nothing of backward is left to run beside it, its collective follows the chain
        seg["cut"].add(b)
        seg["launches"].append(ops.LAUNCHES - seg["mark"])
        seg["mark"] = ops.LAUNCHES
        # on the capture's own stream: the report may come from an operator running on an auxiliary stream
        with torch.cuda.stream(seg["stream"]):
            joined = self._join_sides()
            self._sums.flush()
            seg["graphs"][-1].capture_end()
            seg["after"].append([b])
            g = ops.new_graph()
            g.capture_begin(pool=seg["pool"], capture_error_mode="relaxed")
            seg["graphs"].append(g)
            # auxiliary streams go on into the new capture: operators queued on them later (autograd keeps a branch
            # on its stream without any new wait) must land in a graph, not run eagerly during capture
            for aux in ops.aux_streams():
                if aux in joined:
                    aux.wait_stream(seg["stream"])
                    ops.SIDE_TOUCHED.add(aux)

    def capture_segmented(self, x=None, label=None, stream=None, loss_fn=None):
        """Capture forward + backward of one step (train.py:429-440) on (x, label) as a chain of HIP graphs, cut at
        every point of backward where a gradient bucket of the data-parallel reducer becomes complete; returns the
        (static) loss tensor. replay_segmented() then replays the chain, issuing each bucket's all-reduce behind the
        graph that completes it, so the collectives overlap the remaining backward although every kernel launch
        comes from a graph. With the reducer inactive (one rank, not forced) there is a single graph.
        loss_fn: a callable returning the loss instead of (x, label) (e.g. loss_given_noise with fixed draws).
        The caller runs update() (or a graph of it) afterwards."""
        if self._seg is not None:
            raise RuntimeError("capture_segmented: a capture is already in progress")
        stream = stream or torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        seg = self._seg = {"graphs": [], "after": [], "pool": torch.cuda.graph_pool_handle(), "stream": stream,
                           "cut": set(), "launches": [], "mark": ops.LAUNCHES}
        with torch.cuda.stream(stream):
            # (relaxed mode: autograd runs backward on its own thread, so a capture begun on this one is ended
            # on that one at the first cut, and the last one the other way round)
            g = ops.new_graph()
            g.capture_begin(pool=seg["pool"], capture_error_mode="relaxed")
            seg["graphs"].append(g)
            try:
                if loss_fn is not None:
                    loss = loss_fn()
                else:
                    uncond_mask = None
                    if label is not None:
                        uncond_mask = torch.bernoulli(self.p_uncond * torch.ones_like(label))
                    loss = self.scheduler.get_loss(x, self.model, cls=label, uncond_mask=uncond_mask)
                self._backward(loss, overlap="cut")
                loss = loss.detach()
            finally:
                self._seg = None
                seg["graphs"][-1].capture_end()
            seg["after"].append([b for b in range(len(self.reducer.buckets)) if b not in seg["cut"]])
            seg["launches"].append(ops.LAUNCHES - seg["mark"])
        torch.cuda.current_stream().wait_stream(stream)
        self.segments = (seg["graphs"], seg["after"])
        self.segment_launches = seg["launches"]    # operator launches captured into each graph of the chain
        if os.environ.get("DM_DDP_DEBUG", "0") not in ("", "0"):
            sys.stderr.write(f"[segments] buckets after each graph {seg['after']}, launches per graph {seg['launches']}\n")
        return loss

    def replay_segmented(self):
        """Replay the chain captured by capture_segmented(), each bucket's all-reduce issued right behind the graph
        that completes it; returns when every collective has been enqueued and the current stream waits for all of
        them (gradients averaged). The chain runs on a stream of its own, bracketed by waits against the current
        stream, when that is the legacy default stream (whose implicit synchronisation with other streams does not
        mix with graph launches interleaved with a collective library's copy streams)."""
        graphs, after = self.segments
        cur = torch.cuda.current_stream()
        own = None
        if cur == torch.cuda.default_stream(cur.device) and os.environ.get("DM_REPLAY_OWN_STREAM", "1") != "0":
            if self._replay_stream is None:
                self._replay_stream = torch.cuda.Stream(cur.device)
            own = self._replay_stream
            own.wait_stream(cur)
        with torch.cuda.stream(own) if own is not None else contextlib.nullcontext():
            self.reducer.begin()
            debug = os.environ.get("DM_DDP_DEBUG") == "2"   # (serialises: every graph is waited for before the next)
            for i, (g, buckets) in enumerate(zip(graphs, after)):
                if debug:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                g.replay()
                if debug:
                    e1.record()
                    torch.cuda.synchronize()
                    sys.stderr.write(f"[replay] graph {i}: {e0.elapsed_time(e1):.3f} ms, then buckets {buckets}\n")
                for b in buckets:
                    self.reducer.launch(b)
            self.reducer.finish(plain=True)
        if own is not None:
            cur.wait_stream(own)

    def _reduce(self):
        """Finish the data-parallel mean of the gradient arena (launches whatever buckets are still outstanding)."""
        self.reducer.finish()

    def _update(self):
        """Clip, AdamW, EMA, bf16 weight mirrors, zero the gradients (train.py:441-450)."""
        a = self.arena
        st = torch.cuda.current_stream().cuda_stream
        if torch.cuda.is_current_stream_capturing():
            self._update_captured = True
        # the arena holds the SUM of the ranks' gradients: the mean is taken on the fly by both kernels
        gscale = 1.0 / self.world if self.reducer.active and self.reducer.have_group else 1.0
        if self.clip > 0:
            check(lib.dm_grad_norm(a.grad.data_ptr(), a.numel, gscale, self.partials.data_ptr(), self.norm.data_ptr(), st))
        ig, pw, mx = self.ema_cfg
        check(lib.dm_optimizer_tick_lr(self.step_dev.data_ptr(), self.hyper.data_ptr(), self.lr, *self.lr_sched,
                                       self.betas[0], self.betas[1], ig, pw, mx, st))
        check(lib.dm_adamw_ema_step(a.data.data_ptr(), a.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                    self.ema.data_ptr() if self.ema is not None else None, self.w16.data_ptr(),
                                    self.norm.data_ptr() if self.clip > 0 else None, self.clip, a.numel, self.lr,
                                    self.betas[0], self.betas[1], self.eps, self.wd, 1.0, 1.0, 0.0,
                                    self.hyper.data_ptr(), gscale, 1, st))   # (zeroes the gradients behind its read)
        self._transpose_mirror()
        self.steps += 1
        ops.WEIGHTS_EPOCH += 1  # packed bf16 copies of the weights are stale now

    def last_lr(self):
        """lr_scheduler.get_last_lr()[0] of the reference's log line (train.py:639-640): the rate the NEXT step runs at."""
        return self.lr * lib.dm_lr_multiplier(self.lr_sched[0], self.steps, *self.lr_sched[1:])

    def _backward_and_update(self, loss):
        self._backward(loss)
        self._reduce()
        self._update()

    # ---- the step in three phases, for callers that capture the local phases into HIP graphs and keep the
    # collectives outside them (bench.py on more than one GPU): forward_backward -> reduce -> update
    def forward_backward(self, x, label=None):
        uncond_mask = None
        if label is not None:
            uncond_mask = torch.bernoulli(self.p_uncond * torch.ones_like(label))
        loss = self.scheduler.get_loss(x, self.model, cls=label, uncond_mask=uncond_mask)
        self._backward(loss, overlap=False)
        return loss.detach()

    def reduce(self):
        self.reducer.begin()
        self._reduce()

    def update(self):
        self._update()

    def step(self, x, label=None):
        """train.py:429-452. Returns the loss tensor (device scalar; call .item() to mirror the reference's sync)."""
        uncond_mask = None
        if label is not None:
            uncond_mask = torch.bernoulli(self.p_uncond * torch.ones_like(label))
        loss = self.scheduler.get_loss(x, self.model, cls=label, uncond_mask=uncond_mask)
        self._backward_and_update(loss)
        return loss.detach()

    def loss_given_noise(self, x, label, uncond_mask, eps, t):
        """The loss of scheduler.get_loss with the random draws supplied by the caller."""
        x_t = self.scheduler.diffuse(x, t, eps)
        pred = self.model(x_t, t, cls=label, uncond_mask=uncond_mask)
        if self.scheduler.pred_type == "velocity":
            return ops.mse_loss(pred, eps, x=x, mode=1)
        return ops.mse_loss(pred, eps, mode=0)

    def step_given_noise(self, x, label, uncond_mask, eps, t):
        """Same step with the random draws supplied by the caller (parity tests)."""
        loss = self.loss_given_noise(x, label, uncond_mask, eps, t)
        self._backward_and_update(loss)
        return loss.detach(), self.norm.clone()

    # ---- weights, optimizer and EMA state in / out ------------------------------------------------------------
    def model_state_dict(self):
        """The model's state dict in the reference's layout (contiguous tensors): what train.py:366 saves as model.pt."""
        return {k: v.detach().contiguous().clone() for k, v in self.model.state_dict().items()}

    def ema_state_dict(self):
        """The EMA weights as a model state dict: what train.py:367-371 saves as ema_model.pt."""
        if self.ema is None:
            raise RuntimeError("ema_state_dict: this TrainStep was built with use_ema=False")
        sd = self.model_state_dict()
        for i, n in enumerate(self.arena.names):
            sd[n] = self.arena.view_of(self.ema, i).contiguous().clone()
        return sd

    def state_dict(self):
        """Everything train.py:375-392 (save_latest_ckpt) keeps to resume a run, in the formats of the objects the
        reference holds: torch.optim.AdamW.state_dict() (per-parameter step / exp_avg / exp_avg_sq in parameter
        order) and diffusers' EMAModel.state_dict() (shadow_params list + optimization_step + warm-up settings)."""
        a = self.arena
        n = len(a.params)
        opt_state = {}
        if self.steps > 0:
            for i in range(n):
                opt_state[i] = {"step": torch.tensor(float(self.steps)),
                                "exp_avg": a.view_of(self.m, i).contiguous().clone(),
                                "exp_avg_sq": a.view_of(self.v, i).contiguous().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(n))}
        sd = {"global_steps": self.steps, "model_state_dict": self.model_state_dict(),
              "optimizer_state_dict": {"state": opt_state, "param_groups": [group]}}
        if self.lr_scheduler_name is not None:
            # as torch's LambdaLR leaves them (train.py:390-391): the group holds the current rate and the base rate
            # as `initial_lr`; the scheduler's own state is its position
            group["lr"], group["initial_lr"] = self.last_lr(), self.lr
            sd["lr_scheduler_state_dict"] = {"base_lrs": [self.lr], "last_epoch": self.steps, "_step_count": self.steps + 1,
                                             "_get_lr_called_within_step": False, "_last_lr": [self.last_lr()],
                                             "lr_lambdas": [None]}
        if self.ema is not None:
            ig, pw, mx = self.ema_cfg
            sd["ema_model_state_dict"] = {
                "decay": mx, "min_decay": 0.0, "optimization_step": self.steps, "update_after_step": 0,
                "use_ema_warmup": True, "inv_gamma": ig, "power": pw,
                "shadow_params": [a.view_of(self.ema, i).contiguous().clone() for i in range(n)]}
        return sd

    def load_state_dict(self, sd):
        """Inverse of state_dict() (train.py:394-415 load_latest_ckpt): a run resumed from it continues exactly as the
        uninterrupted run would (same Adam bias corrections, same EMA warm-up position)."""
        a = self.arena
        n = len(a.params)
        self.model.load_state_dict(sd["model_state_dict"])
        opt = sd["optimizer_state_dict"]["state"]
        self.m.zero_()
        self.v.zero_()
        steps = int(sd.get("global_steps", 0))
        for i in range(n):
            st = opt.get(i, opt.get(str(i)))
            if st is None:
                continue
            a.view_of(self.m, i).copy_(st["exp_avg"])
            a.view_of(self.v, i).copy_(st["exp_avg_sq"])
            steps = max(steps, int(float(st["step"])))
        g = sd["optimizer_state_dict"]["param_groups"][0]
        base_lr = g.get("initial_lr", g["lr"]) if self.lr_scheduler_name is not None else g["lr"]
        loaded = (base_lr, tuple(g["betas"]), g["eps"], g["weight_decay"])
        if self._update_captured and loaded != (self.lr, tuple(self.betas), self.eps, self.wd):
            # those scalars are kernel arguments: a graph captured before this call would go on using the old ones
            raise RuntimeError("load_state_dict: the checkpoint's optimizer hyper-parameters differ from the ones a captured "
                               "update() graph holds by value; load the state before capturing, or build a new TrainStep")
        self.lr, self.betas, self.eps, self.wd = loaded
        if self.ema is not None:
            e = sd.get("ema_model_state_dict")
            if e is None:
                self.ema.copy_(a.data)
            else:
                for i, t in enumerate(e["shadow_params"]):
                    a.view_of(self.ema, i).copy_(t)
                self.ema_cfg = (e.get("inv_gamma", self.ema_cfg[0]), e.get("power", self.ema_cfg[1]),
                                e.get("decay", self.ema_cfg[2]))
                steps = max(steps, int(e.get("optimization_step", 0)))
        self.steps = steps
        self.step_dev.fill_(steps)
        a.grad.zero_()
        self.refresh_packed()
        ops.WEIGHTS_EPOCH += 1
