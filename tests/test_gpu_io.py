"""GPU: the rows either side of the hot path - the HBM-resident uint8 data path, the run directory layout
(checkpoints, resume), and the memorisation metric - against torch CPU evaluations of the reference's formulas."""
import json
import os

import numpy as np
import pytest
import torch

from tests.parity import TINY_CFG, det_sd_for

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape,augs", [((50, 32, 32, 3), None), ((50, 32, 32, 3), ["RandomHorizontalFlip"]),
                                        ((40, 28, 28), ["RandomHorizontalFlip", "RandomVerticalFlip"])])
def test_u8_batches_bitexact(shape, augs):
    """train.py:111-125: ToTensor + Normalize(0.5, 0.5) (the DataLoader's transform) and v / 127.5 - 1 (data_tensor),
    evaluated by torch on the CPU in fp32, against the device kernel - bit for bit, flips included."""
    from harness.data import DeviceDataset
    g = torch.Generator().manual_seed(1)
    data = torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)
    data.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)          # every byte value occurs
    labels = torch.randint(0, 10, (shape[0],), generator=g)
    ds = DeviceDataset(data, labels, dev(), augmentations=augs)
    idx = torch.randperm(shape[0], generator=g)[:17]
    flip = torch.randint(0, 4 if augs and len(augs) == 2 else (2 if augs else 1), (17,), generator=g).to(torch.uint8)
    x, y = ds.batch(idx, flip=flip)
    hwc = data if data.dim() == 4 else data[..., None]
    ref = hwc[idx].permute(0, 3, 1, 2).to(torch.float32)
    for b in range(17):
        if int(flip[b]) & 1:
            ref[b] = ref[b].flip(-1)
        if int(flip[b]) & 2:
            ref[b] = ref[b].flip(-2)
    want = ref.div(255).sub_(0.5).div_(0.5)     # ToTensor's div(255), Normalize's sub_(mean).div_(std)
    assert x.dtype == torch.float32 and tuple(x.shape) == (17, hwc.shape[3], shape[1], shape[2])
    assert torch.equal(x.cpu(), want) and torch.equal(y.cpu(), labels[idx])
    full = ds.data_tensor(chunk=16)
    assert torch.equal(full.cpu(), hwc.permute(0, 3, 1, 2).to(torch.float32) / 127.5 - 1)
    # random flips: both orientations occur, every image is the original or its mirror, labels follow
    if augs:
        xs, _ = ds.batch(torch.arange(shape[0]))
        plain = ds.batch(torch.arange(shape[0]), flip=torch.zeros(shape[0], dtype=torch.uint8))[0]
        same = (xs == plain).flatten(1).all(1)
        assert 0 < int(same.sum()) < shape[0]
    with pytest.raises(IndexError):
        ds.batch(torch.tensor([shape[0]]))


def test_memorization_metric_vs_cdist():
    """utils/validate_memo.py: nearest and second-nearest training image per generated image. Reference values by
    torch.cdist in fp64 on the CPU; the set includes exact and near copies of training rows (the memorised case,
    where the distance is a small difference of large squared norms)."""
    from harness.evaluation import calc_memorization_metric
    g = torch.Generator().manual_seed(2)
    n_ref, n_gen, D = 1500, 333, 3 * 32 * 32
    ref = torch.randint(0, 256, (n_ref, D), generator=g).float() / 127.5 - 1
    gen = torch.rand(n_gen, D, generator=g) * 2 - 1
    gen[:20] = ref[100:120]                                              # exact copies: d1 = 0
    gen[20:60] = ref[200:240] + 0.01 * torch.randn(40, D, generator=g)   # near copies
    gen[60:80] = 0.5 * (ref[300:320] + ref[320:340])                     # between two rows: ratio near 1
    dm = torch.cdist(gen.double(), ref.double())
    top2, _ = torch.topk(dm, 2, dim=1, largest=False)
    mem, d1, d2 = calc_memorization_metric(gen.view(n_gen, 3, 32, 32), ref.view(n_ref, 3, 32, 32), device=dev(), batch_size=128)
    assert d1.shape == (n_gen,) and d1.device.type == "cpu"
    # d^2 = |g|^2 + |r|^2 - 2 g.r in fp32 with |g|^2, |r|^2 ~ 1000 (3072 terms): good to ~1e-5 of the norms, i.e.
    # 2e-2 absolute on the squared distance (torch.cdist's own fp32 matrix-multiply path has the same cancellation)
    assert float((d1.double() ** 2 - top2[:, 0] ** 2).abs().max()) < 2e-2
    assert float((d2.double() ** 2 - top2[:, 1] ** 2).abs().max()) < 2e-2
    assert float(d1[:20].max()) < 0.15 and float((d1[20:60] - top2[20:60, 0]).abs().max()) < 2e-2
    assert float(((d2 - top2[:, 1]) / top2[:, 1]).abs().max()) < 1e-4
    want = float(torch.sum(top2[:, 0] / top2[:, 1] < 1 / 3)) / n_gen
    assert abs(float(mem) - want) < 1e-6 and want >= 60 / n_gen


def _step(seed=11, **kw):
    from diffusion.scheduler import get_scheduler
    from harness.train_step import TrainStep
    from modeling import get_model
    m = get_model("unet", **TINY_CFG)
    m.load_state_dict(det_sd_for(m, seed))
    return TrainStep(m.to(dev()).train(), get_scheduler("beta", n_steps=1000), lr=2e-4, clip_grad_norm=1.0, **kw)


def test_run_directory_roundtrip_and_resume(tmp_path):
    """ckpts/ckpt-%06d/{model,ema_model}.pt + train_args.json + latest.pt: a run resumed from latest.pt continues bit
    for bit like the uninterrupted run (Adam moments, bias-correction step, EMA warm-up position), the optimizer
    state has torch.optim.AdamW's format, and the checkpoint loads for evaluation as save_fid_eval.py does."""
    from harness import checkpoint as C
    from harness.generate import generate
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(8, 3, 16, 16, generator=g) * 2 - 1).to(dev())
    cls = torch.randint(0, 10, (8,), generator=g).to(dev())
    a = _step(use_ema=True)
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    for _ in range(3):
        a.step(x, cls)
    out = str(tmp_path / "run")
    cfg = {"model_type": "unet", "model_cfg": TINY_CFG, "scheduler_type": "beta",
           "scheduler_cfg": {"schedule": "linear", "n_steps": 1000}, "sampler_type": "ddim",
           "sampler_cfg": {"n_steps": 4, "eta": 0.0, "clip_latent": True, "pbar": False}, "dataset": "cifar10"}
    C.write_train_args(out, cfg)
    d = C.save_ckpt(a, out)
    assert d.endswith(os.path.join("ckpts", "ckpt-000003")) and sorted(os.listdir(d)) == ["ema_model.pt", "model.pt"]
    C.save_latest_ckpt(a, out)
    # the uninterrupted run goes on ...
    torch.manual_seed(6); torch.cuda.manual_seed(6)
    la = [float(a.step(x, cls)) for _ in range(2)]
    # ... and so does a fresh process-like object restored from latest.pt
    b = _step(seed=99, use_ema=True)
    ck = C.load_latest_ckpt(b, out)
    assert ck["global_steps"] == 3 and b.steps == 3
    torch.manual_seed(6); torch.cuda.manual_seed(6)
    lb = [float(b.step(x, cls)) for _ in range(2)]
    assert la == lb
    assert torch.equal(a.arena.data, b.arena.data) and torch.equal(a.ema, b.ema) and torch.equal(a.m, b.m)
    assert torch.equal(a.w16, b.w16) and torch.equal(a.w16_t, b.w16_t)
    # the reference's load_latest_ckpt (train.py:394-415) finds every key it indexes, in a form its calls accept
    import random as rd
    ref = torch.load(os.path.join(out, "latest.pt"), weights_only=False)     # (as the reference loads it; our own file)
    assert ref["global_steps"] == 3 and ref["epochs"] == 0 and ref["steps_in_epoch"] == 0
    np_before, rd_before = np.random.get_state(), rd.getstate()
    try:
        torch.set_rng_state(ref["rng_state"])
        torch.cuda.set_rng_state(ref["cuda_rng_state"])
        np.random.set_state(ref["np_rng_state"])
        rd.setstate(ref["rd_rng_state"])
    finally:
        np.random.set_state(np_before)
        rd.setstate(rd_before)
    assert set(ref["ema_model_state_dict"]) >= {"decay", "min_decay", "optimization_step", "update_after_step",
                                                "use_ema_warmup", "inv_gamma", "power", "shadow_params"}
    # torch.optim.AdamW accepts the optimizer state as its own
    sd = torch.load(os.path.join(out, "latest.pt"), weights_only=True)
    ref_params = [torch.nn.Parameter(torch.zeros(s)) for s in a.arena.shapes]
    opt = torch.optim.AdamW(ref_params, lr=1.0)
    opt.load_state_dict(sd["optimizer_state_dict"])
    assert opt.param_groups[0]["lr"] == 2e-4 and float(opt.state[ref_params[0]]["step"]) == 3.0
    assert all(t.is_contiguous() for t in sd["model_state_dict"].values())
    assert len(sd["ema_model_state_dict"]["shadow_params"]) == len(ref_params)
    # evaluation entry: EMA weights of the checkpoint, the config's scheduler and sampler
    model, sched, sampler, conf = C.load_for_eval(out, "ckpt-000003", use_ema=True, sampling_steps=3, device=dev())
    assert sampler.n_steps == 3 and conf["dataset"] == "cifar10"
    ema_sd = torch.load(os.path.join(d, "ema_model.pt"), weights_only=True)
    assert all(torch.equal(v.cpu(), ema_sd[k]) for k, v in model.state_dict().items())
    imgs = generate(model, sampler, sched, torch.randn(3, 3, 16, 16, generator=g), torch.tensor([1, 2, 3]), batch_size=2)
    assert imgs.shape == (3, 3, 16, 16) and bool(torch.isfinite(imgs).all())
    # the next checkpoint (here: the raw weights of the same one) loaded INTO that model: same object, same parameter
    # addresses, so generate() goes on replaying the graphs it captured for it
    from dm_hip import ops
    n_graphs = len(ops._GRAPHS)
    ptrs = [p.data_ptr() for p in model.parameters()]
    model2, _, _, _ = C.load_for_eval(out, "ckpt-000003", use_ema=False, sampling_steps=3, device=dev(), model=model)
    assert model2 is model and ptrs == [p.data_ptr() for p in model.parameters()]
    raw_sd = torch.load(os.path.join(d, "model.pt"), weights_only=True)
    assert all(torch.equal(v.cpu(), raw_sd[k]) for k, v in model.state_dict().items())
    z3 = torch.randn(3, 3, 16, 16, generator=g)
    imgs2 = generate(model, sampler, sched, z3, torch.tensor([1, 2, 3]), batch_size=2)
    assert len(ops._GRAPHS) == n_graphs
    assert torch.equal(imgs2, generate(model, sampler, sched, z3, torch.tensor([1, 2, 3]), batch_size=2, use_graph=False))
    with pytest.raises(RuntimeError):
        _step(use_ema=False).ema_state_dict()


def test_lr_schedule_on_device_resume_and_graph_replay(tmp_path):
    """lr_scheduler / optimizer options of the reference's trainer (train.py:265-291, 446-447): the rate of every step
    comes from the schedule evaluated on the device, so a replayed HIP graph follows it; a resumed run continues bit
    for bit; the saved scheduler and optimizer state load into torch's LambdaLR / AdamW at the same position."""
    transformers = pytest.importorskip("transformers")
    from dm_hip import lib, ops
    from harness import checkpoint as C
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(4, 3, 16, 16, generator=g) * 2 - 1).to(dev())
    cls = torch.randint(0, 10, (4,), generator=g).to(dev())
    kw = dict(use_ema=True, lr_scheduler="cosine", lr_warmup_steps=2, max_steps=8)
    a = _step(**kw)
    want = [2e-4 * lib.dm_lr_multiplier(a.lr_sched[0], k, *a.lr_sched[1:]) for k in range(9)]

    def on_device(step_obj, k):   # the fp32 rate the device derived for optimizer step k (its cos may differ in the last bit)
        return abs(float(step_obj.hyper[0]) - want[k]) <= 2e-7 * want[k]
    assert want[0] == 0.0 and want[2] == 2e-4 and 0 < want[7] < want[3] < 2e-4 and abs(want[8]) < 1e-20
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    p0 = a.arena.data.clone()
    a.step(x, cls)
    assert float(a.hyper[0]) == 0.0 and torch.equal(a.arena.data, p0)      # a warm-up from zero: step 0 moves nothing
    for k in (1, 2):
        a.step(x, cls)
        assert on_device(a, k) and a.last_lr() == want[k + 1]
    assert not torch.equal(a.arena.data, p0)
    out = str(tmp_path / "run")
    C.save_latest_ckpt(a, out)
    torch.manual_seed(6); torch.cuda.manual_seed(6)
    la = [float(a.step(x, cls)) for _ in range(2)]
    b = _step(seed=99, **kw)
    C.load_latest_ckpt(b, out)
    assert b.steps == 3 and b.lr == 2e-4
    torch.manual_seed(6); torch.cuda.manual_seed(6)
    lb = [float(b.step(x, cls)) for _ in range(2)]
    assert la == lb and torch.equal(a.arena.data, b.arena.data) and on_device(b, 4)
    # torch's objects take the saved state as their own and stand at the same position
    sd = torch.load(os.path.join(out, "latest.pt"), weights_only=True)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in a.arena.shapes]
    opt = torch.optim.AdamW(params, lr=2e-4, betas=(0.9, 0.99))
    sch = transformers.get_scheduler("cosine", optimizer=opt, num_warmup_steps=2, num_training_steps=8)
    opt.load_state_dict(sd["optimizer_state_dict"])
    sch.load_state_dict(sd["lr_scheduler_state_dict"])
    assert sch.last_epoch == 3 and abs(sch.get_last_lr()[0] - want[3]) < 1e-18 and opt.param_groups[0]["lr"] == want[3]
    opt.step(); sch.step()
    assert abs(opt.param_groups[0]["lr"] - want[4]) < 1e-18
    # one captured update follows the schedule on replay
    graph = ops.new_graph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            b.update()
        for k in (6, 7):     # (capture runs nothing: the step counter is still 5)
            graph.replay()
            side.synchronize()
            assert on_device(b, k - 1) and int(b.step_dev) == k
    torch.cuda.current_stream().wait_stream(side)
    # plain Adam (train.py:266-267): no decay; an unknown optimizer or a coupled penalty is refused
    assert _step(optimizer="adam").wd == 0.0 and _step().wd == 1e-2
    with pytest.raises(ValueError):
        _step(optimizer="sgd")
    with pytest.raises(NotImplementedError):
        _step(optimizer="adam", weight_decay=0.1)


def test_data_order_and_flips_resume_mid_epoch(tmp_path):
    """A run resumed from latest.pt sees the batches the uninterrupted run would have seen next: same rows (the
    shuffle order taken up at epochs / steps_in_epoch, train.py:598-605) and the same flips (the flip generator's state
    travels in latest.pt)."""
    from harness import checkpoint as C
    from harness.data import DeviceDataset
    g = torch.Generator().manual_seed(9)
    data = torch.randint(0, 256, (50, 8, 8, 3), generator=g, dtype=torch.uint8)
    labels = torch.randint(0, 10, (50,), generator=g)
    augs = ["RandomHorizontalFlip"]
    a = DeviceDataset(data, labels, dev(), augmentations=augs, seed=7)
    it = a.epochs(batch_size=8, seed=42)
    per_epoch = 7                                  # ceil(50 / 8)
    for _ in range(per_epoch + 3):                 # one whole epoch and three batches of the next
        next(it)
    step = _step(use_ema=False)
    out = str(tmp_path / "run")
    C.save_latest_ckpt(step, out, epochs=1, steps_in_epoch=3, dataset=a)
    want = [next(it) for _ in range(per_epoch)]    # runs over the end of the epoch
    b = DeviceDataset(data, labels, dev(), augmentations=augs, seed=12345)
    ck = C.load_latest_ckpt(_step(seed=5, use_ema=False), out, dataset=b)
    it2 = b.epochs(batch_size=8, seed=42, start_epoch=ck["epochs"], start_step=ck["steps_in_epoch"])
    for (x1, y1) in want:
        x2, y2 = next(it2)
        assert torch.equal(x1, x2) and torch.equal(y1, y2)


def test_fid_evaluation_log(tmp_path):
    """train.py:506-541 without the Inception network: memorisation ratio always, FID numbers when features and
    reference statistics are supplied; one JSON line per evaluation in fid_evaluations.jsonl."""
    from harness.evaluation import log_fid_evaluation
    g = torch.Generator().manual_seed(4)
    train = torch.rand(64, 3, 8, 8, generator=g) * 2 - 1
    gen = torch.cat([train[:8], torch.rand(24, 3, 8, 8, generator=g) * 2 - 1])
    feats = torch.randn(600, 16, generator=g)
    ref = (np.zeros(16), np.eye(16))
    out = str(tmp_path / "run")
    r = log_fid_evaluation(out, 1000, gen, train, feats, ref, fid_adjust_subsets=[300, 400, 600], device=dev())
    assert abs(r["memorization_ratio"] - 8 / 32) < 1e-6 and {"FID", "FID@inf", "FID@50k"} <= set(r)
    line = json.loads(open(os.path.join(out, "fid_evaluations.jsonl")).read().strip())
    assert line["steps"] == 1000 and line["memorization_ratio"] == r["memorization_ratio"]
