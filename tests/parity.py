"""Helpers shared by the parity tests, smoke() and bench.py: golden loading, deterministic weights, error
metrics, and construction of (oracle state dict, HIP model) pairs."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG = os.path.join(ROOT, "diffusion-models_amd")
for p in (ROOT, PKG, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)

from detweights import det_state_dict  # noqa: E402

TINY_CFG = dict(
    in_channels=3, out_channels=3,
    down_block_types=["UNetDownBlock", "UNetDownBlock"], up_block_types=["UNetUpBlock", "UNetUpBlock"],
    block_out_channels=[64, 128], n_blocks_per_layer=1, mid_attention=True, n_dim_attention_head=8,
    norm_groups=32, activation="silu", time_encoding_dim=64, embedding_dim=64,
    embedding_mixing="scale_shift", n_class_embeddings=10, dropout=0.0,
    downsample_use_conv=True, upsample_use_conv=True,
)
TINY_ATTN_CFG = dict(TINY_CFG, down_block_types=["UNetDownBlock", "UNetAttnDownBlock"],
                     up_block_types=["UNetAttnUpBlock", "UNetUpBlock"], embedding_mixing="addition")
# constructor switches no other fixture exercises (tests/golden/make_golden.py: VARIANT_CFG)
VARIANT_CFG = dict(
    in_channels=1, out_channels=2,
    down_block_types=["UNetDownBlock", "UNetAttnDownBlock", "UNetDownBlock"],
    up_block_types=["UNetUpBlock", "UNetAttnUpBlock", "UNetUpBlock"],
    block_out_channels=[64, 64, 128], n_blocks_per_layer=2, n_attention_heads=4, mid_attention=False,
    norm_groups=16, activation="silu", time_encoding_dim=32, embedding_dim=96, embedding_mixing="addition",
    n_class_embeddings=None, dropout=0.0, downsample_use_conv=False, upsample_use_conv=False,
)


# the networks of tests/golden/samples.npz (trained by the reference, weights stored rounded to bf16)
MINI_CFG = dict(TINY_CFG, block_out_channels=[64, 64])


def trained_mini_state(g, kind):
    """{parameter name: fp32 tensor} of the 'vp' or 'rf' network stored in samples.npz (bf16 bit patterns)."""
    prefix = f"w_{kind}_"
    return {k[len(prefix):]: torch.from_numpy(np.ascontiguousarray(g[k])).view(torch.bfloat16).float()
            for k in g.files if k.startswith(prefix)}


def ddpm_noise_stream(z, seed, n_steps):
    """The noise DDPMSampler.sample(z, ..., seed=seed) draws (diffusion/sampler.py:60-61, 83): a fresh generator on z's
    device seeded with `seed`, one randn_like(z) per step except the last. Regenerated on the CPU, where the
    reference's fixtures were made."""
    g = torch.Generator()
    g.manual_seed(seed)
    zc = z.cpu()
    return torch.stack([torch.randn_like(zc, generator=g) for _ in range(n_steps - 1)])


# ---- loss-curve recipe (tests/golden/train_curve_tiny.npz holds the reference's losses for it) -------------
CURVE = dict(weights_seed=900, data_seed=901, steps=1000, batch=16, n_images=64, lr=5e-4)


def curve_batches():
    """The 1000 (x, cls, mask, eps, t) batches of the loss-curve fixture: 64 structured 3x16x16 images (a 4x4 random
    pattern per image, upsampled, plus a class-dependent brightness), sampled with replacement; everything from one
    numpy RandomState so the reference and the device run see the same numbers."""
    rs = np.random.RandomState(CURVE["data_seed"])
    n, B = CURVE["n_images"], CURVE["batch"]
    labels = rs.randint(0, 10, size=(n,)).astype(np.int64)
    coarse = rs.uniform(-1, 1, size=(n, 3, 4, 4)).astype(np.float32)
    images = np.clip(0.7 * np.kron(coarse, np.ones((1, 1, 4, 4), np.float32)) + (labels[:, None, None, None] - 4.5) / 15.0, -1, 1)
    for _ in range(CURVE["steps"]):
        idx = rs.randint(0, n, size=(B,))
        mask = (rs.uniform(size=(B,)) < 0.2).astype(np.float32)
        eps = rs.standard_normal(size=(B, 3, 16, 16)).astype(np.float32)
        t = rs.randint(0, 1000, size=(B,)).astype(np.int64)
        yield (torch.from_numpy(images[idx].astype(np.float32)), torch.from_numpy(labels[idx]), torch.from_numpy(mask),
               torch.from_numpy(eps), torch.from_numpy(t))


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_cfg(name):
    with open(os.path.join(GOLDEN, "arg_json", name + ".json")) as f:
        return json.load(f)


def contract():
    with open(os.path.join(GOLDEN, "state_dict_contract.json")) as f:
        return json.load(f)


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def det_sd_for(module, seed):
    """Deterministic state dict (torch CPU tensors) for a module's state-dict layout; buffers keep their value."""
    sd = module.state_dict()
    vals = det_state_dict([(k, v.shape) for k, v in sd.items()], seed)
    return {k: (T(vals[k]) if vals[k] is not None else v.detach().cpu().clone()) for k, v in sd.items()}


def rel_err(a, b):
    """||a - b|| / ||b|| in fp64."""
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_abs(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def smoke_check(cfg=None, seed=900, B=4, tol_loss=2e-2, tol_grad=8e-2, hw=(16, 16)):
    """One tiny UNet training step on cuda:0 through the HIP kernels, checked against the CPU oracle: the loss and the
    gradient of EVERY parameter (relative L2 error per parameter). Returns (loss error, {name: gradient error})."""
    from modeling import get_model
    from diffusion.scheduler import get_scheduler
    from oracle import unet_oracle as O

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda:0")
    cfg = cfg or TINY_CFG
    model = get_model("unet", **cfg)
    sd = det_sd_for(model, seed)
    model.load_state_dict(sd)
    model.to(dev).train()
    rs = np.random.RandomState(seed + 1)
    x = T(rs.uniform(-1, 1, size=(B, 3) + tuple(hw)).astype(np.float32))
    cls = T(rs.randint(0, 10, size=(B,)).astype(np.int64))
    mask = T((rs.uniform(size=(B,)) < 0.3).astype(np.float32))
    eps = T(rs.standard_normal((B, 3) + tuple(hw)).astype(np.float32))
    t = T(rs.randint(0, 1000, size=(B,)).astype(np.int64))
    sched = get_scheduler("beta", schedule="cosine")

    from dm_hip import ops
    x_t = sched.diffuse(x.to(dev), t.to(dev), eps.to(dev))
    pred = model(x_t, t.to(dev), cls=cls.to(dev), uncond_mask=mask.to(dev))
    loss = ops.mse_loss(pred, eps.to(dev), mode=0)
    loss.backward()
    torch.cuda.synchronize()

    osd = {k: v.clone() for k, v in sd.items()}
    names = O.param_names(osd)
    params = {k: osd[k].requires_grad_(True) for k in names}
    live = dict(osd)
    live.update(params)
    ol = O.loss_vp(lambda a, b, **kw: O.unet_forward(live, cfg, a, b, **kw), x, t, eps,
                   O.alpha_sq_from_beta(O.beta_cosine()), cls=cls, uncond_mask=mask)
    og = torch.autograd.grad(ol, [params[k] for k in names], allow_unused=True)
    e_loss = abs(float(loss.detach()) - float(ol.detach())) / abs(float(ol.detach()))
    errs = {}
    got = dict(model.named_parameters())
    for k, g in zip(names, og):
        if g is None or float(g.norm()) < 1e-8:
            continue
        errs[k] = rel_err(got[k].grad, g)
    worst = max(errs.values())
    print(f"smoke: loss hip={float(loss.detach()):.6f} oracle={float(ol.detach()):.6f} rel={e_loss:.2e}; worst grad rel-err={worst:.3e} "
          f"over {len(errs)} parameters")
    assert e_loss < tol_loss, "loss mismatch against the CPU oracle"
    assert worst < tol_grad, ("gradient mismatch against the CPU oracle", sorted(errs.items(), key=lambda kv: -kv[1])[:5])
    return e_loss, errs
