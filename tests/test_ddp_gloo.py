"""CPU, world_size 2 over gloo: the bucketed, backward-overlapped gradient all-reduce of the training harness."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.parity import PKG, ROOT


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from harness.ddp import BucketedAllReduce, cut_buckets
        sizes = [7, 1000, 64, 3000, 5, 128, 4096, 9, 640]
        offsets, off = [], 0
        for n in sizes:
            offsets.append(off)
            off += (n + 63) // 64 * 64
        total = off
        g = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(total, generator=g)
        # (the reducer leaves the SUM over ranks; the optimizer kernels apply 1 / world on the fly)
        expect = sum(torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
        red = BucketedAllReduce(grads, offsets, bucket_bytes=4 * 2000)
        assert red.world == world and len(red.buckets) > 2
        # gradients become ready in a scrambled, rank-identical order; one parameter never reports
        order = [8, 6, 7, 5, 3, 4, 1, 2]
        red.begin()
        for i in order:
            red.on_ready(i)
            red.on_ready(i)  # duplicate notifications are ignored
        launched_early = len(red.launched)
        red.finish()
        ok = torch.allclose(grads, expect, atol=1e-6) and red.order[:launched_early] == sorted(red.order[:launched_early])
        q.put((rank, ok, launched_early, len(red.buckets)))
    finally:
        dist.destroy_process_group()


def test_cut_buckets_partition():
    sys.path.insert(0, PKG)
    from harness.ddp import cut_buckets
    offsets = [0, 64, 1088, 1152, 4160]
    total = 4224
    b = cut_buckets(offsets, total, 1000)
    assert b[0][1] == total and b[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(b, b[1:]):
        assert hi2 == lo            # contiguous, walking towards the start of the arena
    assert all(lo in offsets for lo, _ in b)
    assert all(hi - lo >= 1000 for lo, hi in b[:-1])
    assert cut_buckets([0], 10, 1 << 20) == [(0, 10)]
    assert cut_buckets(offsets[::-1], total, 1000) == b      # offsets may come in any order


def test_no_early_bucket_holds_an_embedding_path_parameter():
    """The benchmark model at the default bucket size: the gradients of the embedding path (time MLP, class table, every
    block's emb_proj) exist only when backward is over, so the arena keeps them in front of everything else, where
    the LAST bucket is cut; every other bucket is complete - and its all-reduce can start - while backward is still
    running."""
    import json
    sys.path.insert(0, PKG)
    from harness.ddp import cut_buckets
    from harness.train_step import arena_layout, late_region
    from modeling import get_model
    with open(os.path.join(ROOT, "tests", "golden", "arg_json", "cos_cifar.json")) as f:
        cfg = json.load(f)
    with torch.device("meta"):
        model = get_model(cfg["model_type"], **cfg["model_cfg"])
    named = [(n, p.numel()) for n, p in model.named_parameters()]
    names, numels = [n for n, _ in named], [k for _, k in named]
    late = set(model.late_gradient_parameter_names())
    assert len(late) == 4 + 1 + 2 * 17 and late <= set(names)
    offsets, total = arena_layout(names, numels, late)
    assert sorted(offsets)[0] == 0 and len(set(offsets)) == len(offsets) and total >= sum(numels)
    front = late_region(names, offsets, late)
    assert 0 < front < (8 << 20) // 4
    for mb in (1, 25, 50, 100, 1000):      # whatever the bucket size, the embedding path is the last bucket, alone
        buckets = cut_buckets(offsets, total, (mb << 20) // 4, front)
        assert buckets[-1] == (0, front) and (len(buckets) >= 8 or mb > 25)
        assert buckets[0][1] == total and all(b[0] == a[1] for a, b in zip(buckets[1:], buckets))
        for n, o in zip(names, offsets):
            b = next(i for i, (lo, hi) in enumerate(buckets) if lo <= o < hi)
            assert (b == len(buckets) - 1) == (n in late), (mb, n, b)
    # apart from the embedding path the physical order is the registration order: a bucket is a run of consecutive layers
    rest = [o for n, o in zip(names, offsets) if n not in late]
    assert rest == sorted(rest) and min(rest) > max(o for n, o in zip(names, offsets) if n in late)
    # registration order (no late set): every bucket would hold such a parameter - what the layout is there to avoid
    offsets0, total0 = arena_layout(names, numels)
    buckets0 = cut_buckets(offsets0, total0, (25 << 20) // 4)
    held = {next(i for i, (lo, hi) in enumerate(buckets0) if lo <= o < hi) for n, o in zip(names, offsets0) if n in late}
    assert held == set(range(len(buckets0)))


def test_bucketed_allreduce_world2():
    world, port = 2, 29611
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    for rank, ok, early, nb in res:
        assert ok, f"rank {rank}: summed gradients wrong (or buckets issued out of order)"
        assert 0 < early < nb, "some (not all) buckets must have been launched before finish()"
