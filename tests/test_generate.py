"""The generation harness (harness/generate.py): sharding and gathering on CPU (gloo, world size 2), and on the GPU
the graph-replayed predictor against eager execution."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.parity import PKG, ROOT, TINY_CFG, det_sd_for

for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def test_shard_bounds_partition_contiguously():
    from harness.generate import shard_bounds
    for n in (0, 1, 7, 256, 10000, 50001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_generation_seed_follows_the_reference_recipe():
    """train.py:141-160: z and the labels come from two generators seeded alike; without sample_labels the labels
    cycle through the classes."""
    from harness.generate import make_generation_seed
    s = make_generation_seed("cifar10", 6, seed=42, sample_labels=True)
    assert torch.equal(s["z"], torch.randn((6, 3, 32, 32), generator=torch.Generator().manual_seed(42)))
    assert torch.equal(s["cls"], torch.randint(0, 10, (6,), generator=torch.Generator().manual_seed(42)))
    s = make_generation_seed("mnist", 12, seed=1)
    assert s["z"].shape == (12, 1, 28, 28) and s["cls"].tolist() == [i % 10 for i in range(12)]
    with pytest.raises(ValueError):
        make_generation_seed("imagenet", 1)


def _gather_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from harness.generate import gather_shards, shard_bounds
        n = 7  # uneven: 4 + 3
        full = torch.arange(n * 2 * 3, dtype=torch.float32).view(n, 2, 3)
        lo, hi = shard_bounds(n, world, rank)
        out = gather_shards(full[lo:hi].clone(), n, world)
        q.put((rank, torch.equal(out, full)))
    finally:
        dist.destroy_process_group()


def test_gather_shards_world2_uneven():
    world, port = 2, 29613
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ok for _, ok in (q.get(timeout=5) for _ in range(world)))


@pytest.mark.gpu
@pytest.mark.parametrize("sampler_name,kwargs", [("ddim", {"n_steps": 4}), ("ddpm", {"n_steps": 5}), ("euler", {"n_steps": 3})])
def test_generate_graph_replay_equals_eager(sampler_name, kwargs):
    """Same seeds, same weights: the HIP-graph replayed UNet gives bit-identical images to eager launches, including
    a ragged last batch (5 images in batches of 2) and clipping to [-1, 1]."""
    from diffusion.sampler import get_sampler
    from diffusion.scheduler import get_scheduler
    from harness.generate import generate
    from modeling import get_model
    dev = torch.device("cuda:0")
    model = get_model("unet", **TINY_CFG)
    model.load_state_dict(det_sd_for(model, 5))
    model = model.to(dev)
    sched = get_scheduler("rectified_flow", n_steps=1000) if sampler_name == "euler" else get_scheduler("beta", n_steps=1000)
    sampler = get_sampler(sampler_name, **kwargs)
    g = torch.Generator().manual_seed(3)
    hw = 16
    z = torch.randn(5, TINY_CFG["in_channels"], hw, hw, generator=g)
    cls = torch.randint(0, 10, (5,), generator=g) if TINY_CFG.get("n_class_embeddings") else None
    eager = generate(model, sampler, sched, z, cls, batch_size=2, use_graph=False, chains=1)
    graphed = generate(model, sampler, sched, z, cls, batch_size=2, use_graph=True, chains=1)
    assert eager.shape == z.shape and eager.dtype == torch.float32
    assert float(eager.abs().max()) <= 1.0
    assert torch.equal(eager, graphed)
    # every UNet forward as interleaved sub-batches on streams of their own (2: the default; 3: ragged sub-batches,
    # one of them empty on the last batch of one image): the same images, eagerly and replayed
    for chains, use_graph in ((2, True), (2, False), (3, True)):
        again = generate(model, sampler, sched, z, cls, batch_size=2, use_graph=use_graph, chains=chains)
        assert torch.equal(eager, again), (chains, use_graph)


@pytest.mark.gpu
def test_generate_reuses_its_graphs_and_follows_the_weights():
    """generate() keeps the predictors it captured with the model (graphs are never destroyed, so they must not pile
    up): a second call replays the same graphs, and after the parameters have been rewritten in place (a training step,
    load_state_dict) the replayed graphs read the new weights - the packed bf16 copies are brought up to date outside
    the graph - and give the images of the eager loop bit for bit."""
    from diffusion.sampler import get_sampler
    from diffusion.scheduler import get_scheduler
    from dm_hip import ops
    from harness import generate as G
    from modeling import get_model
    dev = torch.device("cuda:0")
    model = get_model("unet", **TINY_CFG)
    model.load_state_dict(det_sd_for(model, 5))
    model = model.to(dev)
    sched = get_scheduler("beta", n_steps=1000)
    sampler = get_sampler("ddim", n_steps=3)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(4, TINY_CFG["in_channels"], 16, 16, generator=g)
    cls = torch.randint(0, 10, (4,), generator=g)
    first = G.generate(model, sampler, sched, z, cls, batch_size=4)
    n_graphs, n_cached = len(ops._GRAPHS), len(G._CAPTURED[model])
    again = G.generate(model, sampler, sched, z, cls.flip(0), batch_size=4)        # other labels, same graphs
    assert len(ops._GRAPHS) == n_graphs and len(G._CAPTURED[model]) == n_cached
    assert torch.equal(again, G.generate(model, sampler, sched, z, cls.flip(0), batch_size=4, use_graph=False))
    assert not torch.equal(again, first)
    with torch.no_grad():
        model.load_state_dict(det_sd_for(model, 6))                                 # in place: same addresses
    third = G.generate(model, sampler, sched, z, cls, batch_size=4)
    assert len(ops._GRAPHS) == n_graphs
    assert torch.equal(third, G.generate(model, sampler, sched, z, cls, batch_size=4, use_graph=False))
    assert not torch.equal(third, first)
