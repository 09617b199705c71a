"""Bucketed gradient all-reduce over a flat gradient arena, overlapped with backward.

The gradient arena is one contiguous fp32 buffer holding every parameter's gradient at a fixed offset (registration
order, except that the parameters whose gradients appear last whatever their place - the embedding path - sit at its
front: harness.train_step.arena_layout). It is cut into contiguous buckets from its END to its start - the order in
which backward produces gradients - and a bucket's all-reduce is launched the moment every parameter inside it has
reported its gradient, while backward is still running on the remaining layers.
On MI355X nodes the backend is RCCL over xGMI ("nccl" in torch.distributed); the same code runs over gloo on
CPU tensors, which is how the multi-process tests exercise it.
"""
from typing import List, Sequence, Tuple

import os
import sys
import time

import torch
import torch.distributed as dist


def cut_buckets(offsets: Sequence[int], total: int, bucket_elems: int, front: int = 0) -> List[Tuple[int, int]]:
    """[lo, hi) element ranges covering [0, total), boundaries on parameter offsets, each >= bucket_elems
    (except possibly the last one or two), listed from the END of the arena to its start. `offsets` may come in any
    order (a parameter's index need not follow its position in the arena).
    front: [0, front) - the parameters whose gradients exist only when backward is over - is a bucket of its own, the
    last one, whatever the bucket size: merged into the bucket behind it, it would hold that bucket's all-reduce back to
    the end of backward as well."""
    offsets = sorted(offsets)
    if front and front not in offsets:
        raise ValueError(f"cut_buckets: front={front} is not a parameter boundary")
    buckets, hi = [], total
    for i in range(len(offsets) - 1, -1, -1):
        lo = offsets[i]
        if (hi - lo >= bucket_elems and lo >= front) or lo == front or i == 0:
            if hi > lo:
                if lo == front and front and hi - lo < bucket_elems and buckets:
                    buckets[-1] = (lo, buckets[-1][1])     # a short run in front of the late region joins its neighbour
                else:
                    buckets.append((lo, hi))
            hi = lo
    return buckets


class BucketedAllReduce:
    def __init__(self, grad_arena: torch.Tensor, offsets: Sequence[int], bucket_bytes: int = 25 << 20, group=None,
                 front: int = 0):
        self.grad = grad_arena
        self.group = group
        self.have_group = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.have_group else 1
        # Buckets are tracked (and on_bucket / collectives issued) on more than one rank; `force` turns the tracking on
        # for a single rank too: with a process group of one rank the collectives then really run (a way to exercise
        # the RCCL path on a one-GPU box), without a process group only on_bucket is called.
        self.force = False
        self.enabled = True    # False: no collectives at all (a rank that runs a step of its own, e.g. to time kernels)
        # Optional callable(bucket index): called INSTEAD of launching the bucket's all-reduce when the bucket becomes
        # complete during backward (used to cut a HIP-graph capture there; the collective is issued at replay).
        self.on_bucket = None
        self.buckets = cut_buckets(offsets, grad_arena.numel(), max(1, bucket_bytes // grad_arena.element_size()), front)
        self.bucket_of, self.size = {}, [0] * len(self.buckets)
        for i, o in enumerate(offsets):
            for b, (lo, hi) in enumerate(self.buckets):
                if lo <= o < hi:
                    self.bucket_of[i] = b
                    self.size[b] += 1
                    break
        # Optional: a callable returning the HIP streams the collectives must be ordered after IN ADDITION to the
        # current one (every stream parameter-gradient kernels run on); None: the current stream only.
        self.extra_streams = None
        self.early = 0   # buckets launched before finish() in the last step (i.e. while backward was running)
        # A collective that has not completed this many seconds after finish() started waiting for it is reported
        # (rank, bucket, state of every stream) and raised as an error, so that a stuck rank ends its process with a
        # non-zero status instead of waiting for ever. Applies where waiting blocks the host (gloo) and, with
        # DM_DDP_DEBUG set, to every backend (RCCL's own watchdog covers the device-side wait otherwise).
        self.timeout_s = float(os.environ.get("DM_DDP_TIMEOUT", "120"))
        self.order = []  # bucket indices in the order their collectives were issued in the last step
        # Device tensors over a backend without device collectives (gloo: the two-process tests and rehearsals that
        # share one GPU): the reducer stages a bucket through ONE pinned host arena itself - device-to-host copy on the
        # issuing stream when the bucket is complete, the host all-reduce and the copy back in finish() - instead of
        # handing device tensors to the backend, whose own staging allocates pinned memory per collective and drives
        # the HIP runtime from its worker threads. Every HIP call then comes from the caller's threads, in the order
        # the RCCL path would issue its collectives, and the stream ordering under test is the same.
        self.staged = self.have_group and grad_arena.is_cuda and dist.get_backend(group) != "nccl"
        self.host = torch.empty(grad_arena.numel(), dtype=grad_arena.dtype, pin_memory=True) if self.staged else None
        self.begin()

    def begin(self):
        """Call before backward."""
        self.ready, self.launched, self.handles = set(), set(), []
        self.count = [0] * len(self.buckets)
        self.order = []

    def on_ready(self, param_index: int):
        """Parameter `param_index`'s gradient has been enqueued. Launches its bucket when the bucket is complete."""
        if not self.active or param_index in self.ready:
            return
        self.ready.add(param_index)
        b = self.bucket_of[param_index]
        self.count[b] += 1
        if self.count[b] == self.size[b]:
            if self.on_bucket is not None:
                self.launched.add(b)
                self.on_bucket(b)
            else:
                self._launch(b)

    @property
    def active(self):
        return self.enabled and (self.world > 1 or self.force)

    def launch(self, b):
        """All-reduce bucket b now, ordered after the current stream only (replay of captured graph segments: the
        producers of the bucket ran inside graphs launched on the current stream)."""
        if self.active and b not in self.launched:
            self._launch(b, plain=True)

    def _launch(self, b, plain=False):
        lo, hi = self.buckets[b]
        self.launched.add(b)
        self.order.append(b)
        if not self.have_group:
            return
        if _debug():
            cur = torch.cuda.current_stream() if self.grad.is_cuda else None
            _say(self, f"issue #{len(self.handles)} bucket {b} [{lo}, {hi}) from stream "
                       f"{cur.cuda_stream if cur is not None else '-'}"
                       f"{' via side streams' if (self.extra_streams is not None and not plain) else ''}")
        sides = list(self.extra_streams()) if (self.extra_streams is not None and not plain) else []
        if sides:
            # issue from the first side stream after making it wait for the main one and for the other side streams:
            # the collective is then ordered after the gradients written on any of them, and the main stream is not
            # held up
            side = sides[0]
            side.wait_stream(torch.cuda.current_stream())
            for other in sides[1:]:
                side.wait_stream(other)
            with torch.cuda.stream(side):
                h = self._issue(lo, hi)
        else:
            h = self._issue(lo, hi)
        self.handles.append(h)

    def _issue(self, lo, hi):
        """The all-reduce of arena[lo:hi) on the current stream: a handle of the backend, or (staged form) the event
        behind the bucket's copy to the host arena."""
        if not self.staged:
            return dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.host[lo:hi].copy_(self.grad[lo:hi], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return _Staged(self, lo, hi, ev)

    def finish(self, plain=False):
        """Call after backward: reduces buckets whose parameters never all reported (unused parameters) and makes
        the current stream wait for every collective. The arena then holds the SUM of the ranks' gradients."""
        if not self.active:
            return
        self.early = len(self.launched)
        for b in range(len(self.buckets)):
            if b not in self.launched:
                self._launch(b, plain)
        for i, h in enumerate(self.handles):
            self._wait(i, h)
        self.handles = []
        # (the arena now holds the SUM over ranks; the optimizer kernels apply 1 / world on the fly)

    def _wait(self, i, h):
        """Make the current stream (and, where the backend blocks, the host) wait for collective number i."""
        blocking = self.have_group and dist.get_backend(self.group) != "nccl"
        if not (blocking or _debug()):
            h.wait()    # RCCL: a device-side wait enqueued on the current stream, returns at once
            return
        t0, told = time.monotonic(), 0.0
        while not h.is_completed():
            waited = time.monotonic() - t0
            if waited > self.timeout_s:
                _say(self, f"collective #{i} (bucket {self.order[i] if i < len(self.order) else '?'}) not complete after "
                           f"{waited:.0f} s; issue order {self.order}; {self._stream_state()}", force=True)
                raise RuntimeError(f"data-parallel all-reduce #{i} did not complete within {self.timeout_s:.0f} s")
            if _debug() and waited - told > 5.0:
                told = waited
                _say(self, f"waiting for #{i} since {waited:.0f} s; {self._stream_state()}")
            time.sleep(0.0005)
        h.wait()
        if _debug():
            _say(self, f"done #{i} (bucket {self.order[i] if i < len(self.order) else '?'})")

    def _stream_state(self):
        """Which of the streams involved still hold unfinished work (debug output of a stuck wait)."""
        if not self.grad.is_cuda:
            return "cpu tensors"
        streams = {"current": torch.cuda.current_stream()}
        if self.extra_streams is not None:
            for k, s_ in enumerate(self.extra_streams()):
                streams[f"side{k}"] = s_
        return "streams " + ", ".join(f"{k}:{'idle' if s_.query() else 'busy'}" for k, s_ in streams.items())


class _Staged:
    """Handle of a bucket on its way through the pinned host arena (see BucketedAllReduce.staged)."""

    def __init__(self, red, lo, hi, event):
        self.red, self.lo, self.hi, self.event = red, lo, hi, event

    def is_completed(self):
        return self.event.query()    # the copy to the host has landed; wait() does the rest

    def wait(self):
        red = self.red
        self.event.synchronize()
        dist.all_reduce(red.host[self.lo:self.hi], op=dist.ReduceOp.SUM, group=red.group)
        red.grad[self.lo:self.hi].copy_(red.host[self.lo:self.hi], non_blocking=True)


def _debug():
    return os.environ.get("DM_DDP_DEBUG", "0") not in ("", "0")


def _say(red, text, force=False):
    if force or _debug():
        rank = dist.get_rank(red.group) if red.have_group else 0
        sys.stderr.write(f"[ddp rank {rank} t={time.monotonic():.3f}] {text}\n")
        sys.stderr.flush()
