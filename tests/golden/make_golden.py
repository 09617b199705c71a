"""Golden-vector generator.  Runs ONLY in the build container, where /root/reference is mounted.

It imports the unmodified reference (`modeling/`, `diffusion/`) on CPU in fp32, feeds it deterministic
inputs and deterministic weights (tests/golden/detweights.py), and stores inputs + expected outputs as small
.npz / .json fixtures next to this file.  Nothing from the reference's source text is stored: fixtures are
numbers only.  Re-run with:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = os.environ.get("DM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from detweights import load_det_weights  # noqa: E402

from modeling import get_model  # noqa: E402  (reference)
from modeling.blocks import ConvBlock, DownsampleBlock, UpsampleBlock  # noqa: E402
from modeling.attention import Attention2D  # noqa: E402
from modeling.embeddings import PeriodicEncoding, MLPEmbedding  # noqa: E402
from diffusion.scheduler import get_scheduler  # noqa: E402
from diffusion.sampler import get_sampler  # noqa: E402

torch.set_num_threads(8)
torch.manual_seed(0)

CONFIGS = ["beta_cifar", "cos_cifar", "fid_test", "rf_cifar", "rf_cifar_long", "rf_mnist"]


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, len(out), "arrays")


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


# ----------------------------------------------------------------------------------------------
# 0. configs: copy of the six arg_json files (pure data: the input format the build must load unchanged)
#    plus the state-dict contract (names, shapes) and parameter counts.
def dump_configs():
    os.makedirs(os.path.join(HERE, "arg_json"), exist_ok=True)
    contract = {}
    for c in CONFIGS:
        with open(os.path.join(REF, "arg_json", c + ".json")) as f:
            cfg = json.load(f)
        with open(os.path.join(HERE, "arg_json", c + ".json"), "w") as f:
            json.dump(cfg, f, indent=2)
        m = get_model(cfg["model_type"], **cfg["model_cfg"])
        sd = m.state_dict()
        contract[c] = {
            "keys": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
            "n_params": sum(p.numel() for p in m.parameters()),
            "param_names": [k for k, _ in m.named_parameters()],
        }
    with open(os.path.join(HERE, "state_dict_contract.json"), "w") as f:
        json.dump(contract, f)
    print("wrote state_dict_contract.json", {c: contract[c]["n_params"] for c in CONFIGS})


# ----------------------------------------------------------------------------------------------
# 1. schedules, t_steps grids, timestep embedding
def dump_schedules():
    lin = get_scheduler("beta", schedule="linear", n_steps=1000)
    cos = get_scheduler("beta", schedule="cosine", cos_s=0.008, cos_max_beta=0.999, n_steps=1000)
    lin_small = get_scheduler("beta", schedule="linear", beta_start=1e-3, beta_end=5e-2, n_steps=37)
    cos_small = get_scheduler("beta", schedule="cosine", cos_s=0.01, cos_max_beta=0.9, n_steps=50)
    rf = get_scheduler("rectified_flow", n_steps=1000)
    rf_small = get_scheduler("rectified_flow", n_steps=13)
    t = torch.tensor([0, 1, 500, 999])
    sch = lin.get_schedule(t)
    grids = {}
    for n_sched, n in [(1000, 10), (1000, 50), (1000, 1000), (1000, 7), (37, 5), (13, 13), (1000, 1)]:
        grids[f"tsteps_{n_sched}_{n}"] = torch.linspace(n_sched - 1, 0, steps=n + 1, dtype=torch.int64)[:-1]
    npz("schedules.npz",
        lin_beta=lin.beta, lin_alpha_sq=lin.alpha_sq, cos_beta=cos.beta, cos_alpha_sq=cos.alpha_sq,
        lin_small_beta=lin_small.beta, lin_small_alpha_sq=lin_small.alpha_sq,
        cos_small_beta=cos_small.beta, cos_small_alpha_sq=cos_small.alpha_sq,
        rf_sigmas=rf.sigmas, rf_small_sigmas=rf_small.sigmas,
        sched_t=t, sched_alpha_sq=sch.alpha_sq, sched_sigma_sq=sch.sigma_sq, sched_log_snr=sch.log_snr,
        **grids)

    ts = torch.tensor([0, 1, 2, 17, 500, 998, 999], dtype=torch.int64)
    out = {"t": ts}
    for dim in (64, 128):
        pe = PeriodicEncoding(dim)
        out[f"freq_{dim}"] = pe.freq
        out[f"enc_{dim}"] = pe(ts)
    mlp = MLPEmbedding(64, 64)
    load_det_weights(mlp, 11)
    out["mlp_out_64"] = mlp(out["enc_64"])
    npz("embeddings.npz", **out)


# ----------------------------------------------------------------------------------------------
# 2. per-layer forward + backward fixtures (weights from detweights seeds)
def run_fwd_bwd(module, inputs, seed_gy):
    """inputs: dict name -> tensor (float ones require grad). Returns out, grads of inputs, grads of params."""
    ins = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in inputs.items()}
    out = module(**ins)
    rs = np.random.RandomState(seed_gy)
    gy = rnd(rs, *out.shape)
    out.backward(gy)
    res = {"out": out, "gy": gy}
    for k, v in ins.items():
        res["in_" + k] = inputs[k]
        if v.is_floating_point():
            res["gin_" + k] = v.grad
    for k, p in module.named_parameters():
        res["gp_" + k] = p.grad
    return res


def dump_layers():
    rs = np.random.RandomState(1234)
    cases = {}
    # ConvBlock variants: (cin, cout, emb_dim, mixing, B, H, W)
    convblock_cases = {
        "cb_same_ss": (64, 64, 64, "scale_shift", 2, 8, 8),
        "cb_widen_ss": (64, 128, 64, "scale_shift", 2, 8, 8),
        "cb_concat_ss": (192, 128, 128, "scale_shift", 2, 4, 4),   # group size 6: vector straddles groups
        "cb_add": (64, 128, 64, "addition", 2, 8, 8),
        "cb_noemb": (128, 64, None, "addition", 2, 7, 7),
        "cb_mnist": (64, 64, 64, "scale_shift", 3, 14, 14),
    }
    for i, (name, (cin, cout, emb, mixing, B, H, W)) in enumerate(convblock_cases.items()):
        m = ConvBlock(cin, cout, norm_groups=32, activation="silu", embedding_dim=emb, embedding_mixing=mixing)
        load_det_weights(m, 100 + i)
        ins = {"x": rnd(rs, B, cin, H, W)}
        if emb is not None:
            ins["embedding"] = rnd(rs, B, emb)
        r = run_fwd_bwd(m, ins, 200 + i)
        r["meta"] = np.array([cin, cout, emb or 0, {"scale_shift": 1, "addition": 0}[mixing], B, H, W, 100 + i])
        cases[name] = r
        npz(f"layer_{name}.npz", **r)

    # down / up sample blocks
    for i, (name, mod, shape) in enumerate([
        ("down_conv", DownsampleBlock(64, 64, use_conv=True), (2, 64, 16, 16)),
        ("down_conv_odd", DownsampleBlock(64, 64, use_conv=True), (2, 64, 7, 7)),
        ("down_pool", DownsampleBlock(64, 64, use_conv=False), (2, 64, 8, 8)),
        ("up_conv", UpsampleBlock(64, 128, use_conv=True), (2, 64, 4, 4)),
        ("up_conv_odd", UpsampleBlock(64, 64, use_conv=True), (2, 64, 7, 7)),
        ("up_noconv", UpsampleBlock(64, 64, use_conv=False), (2, 64, 4, 4)),
    ]):
        if len(list(mod.parameters())):
            load_det_weights(mod, 300 + i)
        r = run_fwd_bwd(mod, {"x": rnd(rs, *shape)}, 400 + i)
        r["meta"] = np.array(list(shape) + [300 + i])
        npz(f"layer_{name}.npz", **r)

    # attention: (n_dim, head_dim, B, H, W)
    for i, (name, (c, d, B, H, W)) in enumerate({
        "attn_8x8": (64, 8, 2, 8, 8),
        "attn_7x7": (128, 8, 2, 7, 7),
        "attn_16x16_d16": (64, 16, 1, 16, 16),
    }.items()):
        m = Attention2D(c, n_dim_head=d, norm_groups=32)
        load_det_weights(m, 500 + i)
        m.train()
        r = run_fwd_bwd(m, {"x": rnd(rs, B, c, H, W)}, 600 + i)
        r["meta"] = np.array([c, d, B, H, W, 500 + i])
        npz(f"layer_{name}.npz", **r)


# ----------------------------------------------------------------------------------------------
# 3. whole-UNet fixtures
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


def unet_case(name, cfg, seed, B, hw, with_grads=True, store_all_grads=True):
    m = get_model("unet", **cfg)
    load_det_weights(m, seed)
    m.train()
    rs = np.random.RandomState(seed + 1)
    cin = cfg["in_channels"]
    x = rnd(rs, B, cin, hw, hw)
    t = torch.from_numpy(rs.randint(0, 1000, size=(B,)).astype(np.int64))
    cls = torch.from_numpy(rs.randint(0, 10, size=(B,)).astype(np.int64))
    mask = torch.from_numpy((rs.uniform(size=(B,)) < 0.4).astype(np.float32))
    xg = x.clone().requires_grad_(True)
    out = m(xg, t, cls=cls, uncond_mask=mask)
    res = {"x": x, "t": t, "cls": cls, "mask": mask, "out": out, "seed": np.array(seed)}
    with torch.no_grad():
        res["out_nocls"] = m(x, t)
        res["out_nomask"] = m(x, t, cls=cls)
    if with_grads:
        gy = rnd(rs, *out.shape)
        out.backward(gy)
        res["gy"] = gy
        res["gx"] = xg.grad
        names, norms, samples = [], [], []
        for k, p in m.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            names.append(k)
            norms.append(float(g.double().norm()))
            flat = g.flatten()
            samples.append(flat[:: max(1, flat.numel() // 16)][:16].numpy().copy())
            if store_all_grads:
                res["gp_" + k] = g
        res["grad_norms"] = np.array(norms)
        res["grad_names"] = np.array(names)
        res["grad_samples"] = np.stack([np.pad(s, (0, 16 - len(s))) for s in samples])
    npz(f"unet_{name}.npz", **res)
    return m


def dump_unets():
    unet_case("tiny", TINY_CFG, 700, 2, 16, store_all_grads=False)
    unet_case("tiny_attnblocks", TINY_ATTN_CFG, 710, 2, 16, store_all_grads=False)
    for c, B in [("fid_test", 2), ("rf_mnist", 2), ("cos_cifar", 2)]:
        with open(os.path.join(REF, "arg_json", c + ".json")) as f:
            cfg = json.load(f)
        hw = 28 if c == "rf_mnist" else 32
        unet_case(c, cfg["model_cfg"], 720 + len(c), B, hw, with_grads=True, store_all_grads=False)

    # classifier-free guidance closure (predictor.py): three branches on the tiny model
    m = get_model("unet", **TINY_CFG)
    load_det_weights(m, 700)
    m.eval()
    rs = np.random.RandomState(77)
    z = rnd(rs, 4, 3, 16, 16)
    t = torch.full((4,), 321, dtype=torch.int64)
    cls = torch.tensor([1, 5, 9, 0])
    with torch.no_grad():
        npz("unet_tiny_cfg.npz", z=z, t=t, cls=cls,
            g0=m.get_pred_fn(cond=cls, guidance_scale=0.0)(z, t),
            g1=m.get_pred_fn(cond=cls, guidance_scale=1.0)(z, t),
            g2=m.get_pred_fn(cond=cls, guidance_scale=2.0)(z, t),
            gnone=m.get_pred_fn(cond=None, guidance_scale=3.0)(z, t))


def dump_stage_blocks():
    """The stage classes no shipped configuration instantiates: the embedding-free Encoder / Decoder stages
    (modeling/blocks.py:422-632) and a MidBlock without attention, as stand-alone modules."""
    from modeling.blocks import get_down_block, get_up_block, MidBlock
    rs = np.random.RandomState(4321)
    cases = [
        ("stage_enc_down", get_down_block("EncoderDownBlock", in_channels=64, out_channels=128, n_blocks=2, norm_groups=32,
                                          activation="silu", dropout=0.0, downsample=True, downsample_use_conv=True),
         (2, 64, 8, 8)),
        ("stage_enc_attn_down", get_down_block("EncoderAttnDownBlock", in_channels=64, out_channels=64, n_blocks=1,
                                               n_dim_head=8, norm_groups=32, activation="silu", dropout=0.0,
                                               downsample=False), (2, 64, 8, 8)),
        ("stage_dec_up", get_up_block("DecoderUpBlock", in_channels=128, out_channels=64, next_out_channels=None, n_blocks=2,
                                      norm_groups=32, activation="silu", dropout=0.0, upsample=True,
                                      upsample_use_conv=True), (2, 128, 4, 4)),
        ("stage_dec_attn_up", get_up_block("DecoderAttnUpBlock", in_channels=64, out_channels=64, next_out_channels=None,
                                           n_blocks=1, n_dim_head=16, norm_groups=32, activation="silu", dropout=0.0,
                                           upsample=False), (2, 64, 8, 8)),
        ("stage_mid_noattn", MidBlock(64, n_blocks=1, norm_groups=32, activation="silu", embedding_dim=None,
                                      add_attention=False), (2, 64, 8, 8)),
    ]
    for i, (name, mod, shape) in enumerate(cases):
        load_det_weights(mod, 800 + i)
        mod.train()
        r = run_fwd_bwd(mod, {"x": rnd(rs, *shape)}, 900 + i)
        r["meta"] = np.array(list(shape) + [800 + i])
        npz(f"layer_{name}.npz", **r)


VARIANT_CFG = dict(
    in_channels=1, out_channels=2,
    down_block_types=["UNetDownBlock", "UNetAttnDownBlock", "UNetDownBlock"],
    up_block_types=["UNetUpBlock", "UNetAttnUpBlock", "UNetUpBlock"],
    block_out_channels=[64, 64, 128], n_blocks_per_layer=2, n_attention_heads=4, mid_attention=False,
    norm_groups=16, activation="silu", time_encoding_dim=32, embedding_dim=96, embedding_mixing="addition",
    n_class_embeddings=None, dropout=0.0, downsample_use_conv=False, upsample_use_conv=False,
)


def dump_unet_variant():
    """The constructor switches no other fixture exercises, in one network: other channel counts at both ends, three
    levels with equal widths (identity residuals), two blocks per level, attention by head COUNT, no mid attention, 16
    groups, additive mixing, no class table (labels and mask are ignored), pooling / plain nearest resampling, and a
    20x20 image (10x10, 5x5 below)."""
    unet_case("tiny_variant", VARIANT_CFG, 750, 2, 20, store_all_grads=False)


def dump_unet_activations():
    """The block activations other than SiLU that get_activation offers (modeling/activation.py:4-12): ReLU with
    scale/shift mixing, GELU with additive mixing and attention stages."""
    unet_case("tiny_relu", dict(TINY_CFG, activation="relu"), 730, 2, 16, store_all_grads=False)
    unet_case("tiny_gelu", dict(TINY_ATTN_CFG, activation="gelu"), 740, 2, 16, store_all_grads=False)


# ----------------------------------------------------------------------------------------------
# 4. scheduler losses + samplers with analytic models
def dump_diffusion():
    torch.manual_seed(123)
    z = torch.randn(4, 3, 32, 32)
    x = torch.rand(8, 3, 32, 32) * 2 - 1
    res = {"z": z, "x": x}
    half = lambda x_, t_, **kw: 0.5 * x_
    scheds = {
        "lin": get_scheduler("beta", schedule="linear", n_steps=1000),
        "cos": get_scheduler("beta", schedule="cosine", cos_s=0.008, cos_max_beta=0.999, n_steps=1000),
        "rf": get_scheduler("rectified_flow", n_steps=1000),
    }
    for k, s in scheds.items():
        gen = torch.Generator().manual_seed(42)
        res[f"loss_{k}"] = s.get_loss(x, half, gen=gen)
        gen = torch.Generator().manual_seed(42)
        eps = torch.randn_like(x, generator=gen)
        t = torch.randint(0, 1000, size=(8,), generator=gen)
        res[f"eps_{k}"], res[f"t_{k}"] = eps, t
        res[f"xt_{k}"] = s.diffuse(x, t, eps)
    pred = lambda z_, t_: 0.1 * z_ + 1e-6 * t_.view(-1, 1, 1, 1)
    res["euler50"] = get_sampler("euler", n_steps=50).sample(z, scheds["rf"], pred)
    res["euler7"] = get_sampler("euler", n_steps=7).sample(z, scheds["rf"], pred)
    res["ddim50_lin"] = get_sampler("ddim", n_steps=50).sample(z, scheds["lin"], pred)
    res["ddim50_cos"] = get_sampler("ddim", n_steps=50).sample(z, scheds["cos"], pred)
    res["ddim10_cos_noclip"] = get_sampler("ddim", n_steps=10, clip_latent=False).sample(z, scheds["cos"], pred)
    res["ddpm1000_lin"] = get_sampler("ddpm", n_steps=1000).sample(z, scheds["lin"], pred)
    res["ddpm1000_lin_seed7"] = get_sampler("ddpm", n_steps=1000).sample(z, scheds["lin"], pred, seed=7)
    res["ddpm20_cos_seed3"] = get_sampler("ddpm", n_steps=20).sample(z, scheds["cos"], pred, seed=3)
    # the noise the DDPM sampler draws (CPU generator stream), so a device path can be fed the same noise
    for tag, n, seed in [("ddpm20_cos_seed3", 20, 3)]:
        gen = torch.Generator()
        gen.manual_seed(seed)
        res[f"noise_{tag}"] = torch.stack([torch.randn_like(z, generator=gen) for _ in range(n - 1)])
    npz("diffusion.npz", **res)


# ----------------------------------------------------------------------------------------------
# 5. train-step fixture: the caller semantics of train.py:429-452 re-stated around the imported reference
#    (train.py itself needs torchvision/wandb/fire/diffusers, which are absent here).
def dump_train_steps(kind="vp"):
    """kind 'vp': cosine-beta noise prediction (train_steps_tiny.npz); 'rf': rectified flow, velocity target x - eps
    (train_steps_tiny_rf.npz) - the objective rf_cifar_long.json / rf_mnist.json train with."""
    m = get_model("unet", **TINY_CFG)
    load_det_weights(m, 800 if kind == "vp" else 810)
    m.train()
    if kind == "vp":
        sched = get_scheduler("beta", schedule="cosine", cos_s=0.008, cos_max_beta=0.999, n_steps=1000)
    else:
        sched = get_scheduler("rectified_flow", n_steps=1000)
    opt = torch.optim.AdamW(m.parameters(), lr=2e-4, betas=(0.9, 0.99))
    rs = np.random.RandomState(801 if kind == "vp" else 811)
    B, steps = 8, 3
    res = {"steps": np.array(steps), "seed": np.array(800 if kind == "vp" else 810)}
    names = [k for k, _ in m.named_parameters()]
    for s in range(steps):
        x = torch.from_numpy(rs.uniform(-1, 1, size=(B, 3, 16, 16)).astype(np.float32))
        cls = torch.from_numpy(rs.randint(0, 10, size=(B,)).astype(np.int64))
        mask = torch.from_numpy((rs.uniform(size=(B,)) < 0.2).astype(np.float32))
        eps = rnd(rs, B, 3, 16, 16)
        t = torch.from_numpy(rs.randint(0, 1000, size=(B,)).astype(np.int64))
        x_t = sched.diffuse(x, t, eps)
        pred = m(x_t, t, cls=cls, uncond_mask=mask)
        loss = torch.nn.functional.mse_loss(pred, eps if kind == "vp" else x - eps)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        m.zero_grad()
        res.update({f"x{s}": x, f"cls{s}": cls, f"mask{s}": mask, f"eps{s}": eps, f"t{s}": t,
                    f"loss{s}": loss.detach(), f"gnorm{s}": gnorm.detach(), f"xt{s}": x_t})
        res[f"pnorm{s}"] = np.array([float(p.detach().double().norm()) for p in m.parameters()])
        res[f"psample{s}"] = np.stack([p.detach().flatten()[:8].numpy().copy() if p.numel() >= 8
                                       else np.pad(p.detach().flatten().numpy(), (0, 8 - p.numel()))
                                       for p in m.parameters()])
    if kind == "rf":
        # the scheduler's own get_loss on the first batch, with its RNG order (eps, then t) from a seeded generator
        m2 = get_model("unet", **TINY_CFG)
        load_det_weights(m2, 810)
        m2.train()
        gen = torch.Generator().manual_seed(5)
        res["getloss_x"], res["getloss_cls"] = res["x0"], res["cls0"]
        res["getloss_value"] = sched.get_loss(res["x0"], m2, gen=gen, cls=res["cls0"]).detach()
        gen = torch.Generator().manual_seed(5)
        res["getloss_eps"] = torch.randn_like(res["x0"], generator=gen)
        res["getloss_t"] = torch.randint(0, 1000, size=(B,), generator=gen)
    res["param_names"] = np.array(names)
    npz("train_steps_tiny.npz" if kind == "vp" else "train_steps_tiny_rf.npz", **res)


# ----------------------------------------------------------------------------------------------
# 5b. images generated THROUGH THE NETWORK (train.py:476-504 / save_fid_eval.py:163-172: pred_fn = model.get_pred_fn,
#     sampler.sample, clip) for every sampler family and both beta schedules, fp32 on the CPU as the reference
#     samples. An untrained network is a chaotic map under 50 sampler steps (fp32 summation-order noise of 1e-6 grows
#     to 3e-2), so the networks here are TRAINED by the reference first (the train.py:429-452 step, 600 steps on the
#     structured toy data of parity.py) - a denoiser is a contraction towards the data - and their weights are stored
#     rounded to bf16 (exactly representable in both implementations). The DDPM noise stream is stored as well.
MINI_CFG = dict(TINY_CFG, block_out_channels=[64, 64])
SAMPLE_TRAIN = dict(steps=600, batch=16, lr=5e-4, weights_seed=920)


def _train_mini(kind):
    sys.path.insert(0, os.path.dirname(HERE))
    from parity import curve_batches
    from itertools import islice
    m = get_model("unet", **MINI_CFG)
    load_det_weights(m, SAMPLE_TRAIN["weights_seed"])
    m.train()
    sched = get_scheduler("beta", schedule="linear", n_steps=1000) if kind == "vp" else get_scheduler("rectified_flow", n_steps=1000)
    opt = torch.optim.AdamW(m.parameters(), lr=SAMPLE_TRAIN["lr"], betas=(0.9, 0.99))
    for s, (x, cls, mask, eps, t) in enumerate(islice(curve_batches(), SAMPLE_TRAIN["steps"])):
        pred = m(sched.diffuse(x, t, eps), t, cls=cls, uncond_mask=mask)
        loss = torch.nn.functional.mse_loss(pred, eps if kind == "vp" else x - eps)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        m.zero_grad()
        if s % 100 == 0:
            print("  ", kind, "train step", s, float(loss), flush=True)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    return m.eval(), float(loss)


def dump_samples():
    res = {}
    gen = torch.Generator().manual_seed(42)
    z = torch.randn(8, 3, 16, 16, generator=gen)
    cls = torch.randint(0, 10, (8,), generator=torch.Generator().manual_seed(42))
    res["z"], res["cls"] = z, cls
    lin = get_scheduler("beta", schedule="linear", n_steps=1000)
    cos = get_scheduler("beta", schedule="cosine", cos_s=0.008, cos_max_beta=0.999, n_steps=1000)
    rf = get_scheduler("rectified_flow", n_steps=1000)
    for kind in ("vp", "rf"):
        m, last = _train_mini(kind)
        res[f"{kind}_final_train_loss"] = np.array(last)
        for k, p in m.named_parameters():
            res[f"w_{kind}_{k}"] = p.detach().to(torch.bfloat16).view(torch.int16).numpy()
        with torch.no_grad():
            pred = m.get_pred_fn(cond=cls, guidance_scale=1.0)
            if kind == "vp":
                res["ddim50_lin"] = torch.clip(get_sampler("ddim", n_steps=50).sample(z, lin, pred), -1, 1)
                res["ddim50_cos"] = torch.clip(get_sampler("ddim", n_steps=50).sample(z, cos, pred), -1, 1)
                z4 = z[:4].clone()   # (the stored noise stream is the bulk of this fixture: 4 images, 50 steps)
                pred4 = m.get_pred_fn(cond=cls[:4], guidance_scale=1.0)
                res["ddpm50_lin"] = torch.clip(get_sampler("ddpm", n_steps=50).sample(z4, lin, pred4, seed=9), -1, 1)
                g = torch.Generator()
                g.manual_seed(9)
                res["noise_ddpm50_lin"] = torch.stack([torch.randn_like(z4, generator=g) for _ in range(49)])
                res["ddim20_lin_g2"] = torch.clip(
                    get_sampler("ddim", n_steps=20).sample(z, lin, m.get_pred_fn(cond=cls, guidance_scale=2.0)), -1, 1)
            else:
                res["euler50_rf"] = torch.clip(get_sampler("euler", n_steps=50).sample(z, rf, pred), -1, 1)
    npz("samples.npz", **res)


# 5c. ancestral sampling over ALL 1000 steps through the network (diffusion/sampler.py:50-86; what BASELINE's
#     "sample imgs/sec @1000 steps" row runs): the variance-preserving mini network of samples.npz (its stored
#     weights, not a retrained copy), linear beta, 4 images, sampler seed 11. The noise stream is not stored: the
#     sampler draws it from torch.Generator().manual_seed(11) with randn_like(z) once per step but the last, which a
#     test regenerates.
def dump_samples_ddpm1000():
    sys.path.insert(0, os.path.dirname(HERE))
    from parity import trained_mini_state
    g = np.load(os.path.join(HERE, "samples.npz"))
    m = get_model("unet", **MINI_CFG)
    missing = m.load_state_dict(trained_mini_state(g, "vp"), strict=False)
    assert not missing.unexpected_keys
    m.eval()
    z4, cls4 = torch.from_numpy(g["z"][:4]).clone(), torch.from_numpy(g["cls"][:4]).clone()
    lin = get_scheduler("beta", schedule="linear", n_steps=1000)
    with torch.no_grad():
        out = get_sampler("ddpm", n_steps=1000, pbar=False).sample(z4, lin, m.get_pred_fn(cond=cls4, guidance_scale=1.0), seed=11)
    npz("samples_ddpm1000.npz", ddpm1000_lin=torch.clip(out, -1, 1), ddpm1000_lin_raw=out, seed=np.array(11))


# ----------------------------------------------------------------------------------------------
# 6. loss curve over 1000 training steps of the tiny model on a small structured data set. Only the recipe's
#    parameters and the reference's losses are stored; tests/parity.py:curve_batches() rebuilds the batches.
def dump_train_curve():
    sys.path.insert(0, os.path.dirname(HERE))
    from parity import CURVE, curve_batches
    m = get_model("unet", **TINY_CFG)
    load_det_weights(m, CURVE["weights_seed"])
    m.train()
    sched = get_scheduler("beta", schedule="cosine", cos_s=0.008, cos_max_beta=0.999, n_steps=1000)
    opt = torch.optim.AdamW(m.parameters(), lr=CURVE["lr"], betas=(0.9, 0.99))
    losses, gnorms = [], []
    for s, (x, cls, mask, eps, t) in enumerate(curve_batches()):
        x_t = sched.diffuse(x, t, eps)
        loss = torch.nn.functional.mse_loss(m(x_t, t, cls=cls, uncond_mask=mask), eps)
        loss.backward()
        gnorms.append(float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)))
        opt.step()
        m.zero_grad()
        losses.append(float(loss))
        if s % 100 == 0:
            print("  curve step", s, losses[-1], flush=True)
    npz("train_curve_tiny.npz", loss=np.array(losses, dtype=np.float32), gnorm=np.array(gnorms, dtype=np.float32))


if __name__ == "__main__":
    which = sys.argv[1:] or ["configs", "schedules", "layers", "unets", "diffusion", "train"]
    if "configs" in which:
        dump_configs()
    if "schedules" in which:
        dump_schedules()
    if "layers" in which:
        dump_layers()
    if "unets" in which:
        dump_unets()
    if "variant" in which:
        dump_unet_variant()
    if "stages" in which:
        dump_stage_blocks()
    if "activations" in which:
        dump_unet_activations()
    if "diffusion" in which:
        dump_diffusion()
    if "train" in which:
        dump_train_steps("vp")
    if "train_rf" in which:
        dump_train_steps("rf")
    if "samples" in which:
        dump_samples()
    if "samples1000" in which:
        dump_samples_ddpm1000()
    if "curve" in which:
        dump_train_curve()
