"""CPU oracle: a restatement of the reference's UNet training step and sampling loop.

TEST INFRASTRUCTURE ONLY. Nothing in the product path (diffusion-models_amd/) imports this file; it is used by
tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg as the checker / timed CPU baseline.

It is a functional restatement (no nn.Module tree): the network is evaluated straight from a flat state dict
and the model_cfg dictionary, with stock torch CPU fp32 primitives for the arithmetic, and autograd for the
backward. Every function cites the reference lines it follows (paths relative to the reference tree).

Pinning: the reference has no tests or golden vectors of its own, so this oracle is pinned against fixtures
generated in the build container by importing the unmodified reference (tests/golden/make_golden.py);
tests/test_oracle_golden.py checks every function here against them.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------------
# schedules (diffusion/scheduler.py:86-132)
def beta_linear(beta_start=1e-4, beta_end=2e-2, n_steps=1000):
    """scheduler.py:96-97: fp64 linspace."""
    return torch.linspace(beta_start, beta_end, steps=n_steps, dtype=torch.float64)


def beta_cosine(s=0.008, max_beta=0.999, n_steps=1000):
    """scheduler.py:111-121: python-float loop, beta_i = min(1 - abar((i+1)/N)/abar(i/N), max_beta)."""
    def abar(u):
        return math.cos((u + s) / (1 + s) * math.pi / 2) ** 2
    return torch.tensor([min(1 - abar((i + 1) / n_steps) / abar(i / n_steps), max_beta) for i in range(n_steps)],
                        dtype=torch.float64)


def alpha_sq_from_beta(beta):
    """scheduler.py:105."""
    return torch.cumprod(1 - beta, dim=0)


def rf_sigmas(n_steps=1000):
    """scheduler.py:132: fp32 linspace(1, N, N) / N."""
    return torch.linspace(1, n_steps, steps=n_steps) / n_steps


def timestep_grid(n_sched, n_steps):
    """sampler.py:63-68 (same in :104-109, :145-150): int64 linspace with the last point dropped.
    Restated in numpy: torch evaluates an integer linspace as start + step*i in the first half and
    end - step*(steps-1-i) in the second half, with a float step, then truncates."""
    steps = n_steps + 1
    start, end = float(n_sched - 1), 0.0
    if steps == 1:
        return np.array([n_sched - 1], dtype=np.int64)[:0]
    step = (end - start) / (steps - 1)
    half = steps // 2
    out = np.empty(steps, dtype=np.int64)
    for i in range(steps):
        v = start + step * i if i < half else end - step * (steps - i - 1)
        out[i] = int(v)  # truncation toward zero
    return out[:-1]


# ----------------------------------------------------------------------------------------------------------
# embeddings (modeling/embeddings.py)
def periodic_freq(n_dim, max_period=10000):
    """embeddings.py:19-21."""
    return torch.exp(-math.log(max_period) * torch.linspace(0, 1, steps=n_dim // 2, dtype=torch.float32))


def periodic_encoding(t, freq):
    """embeddings.py:24-27: sin first, then cos."""
    a = t[:, None] * freq[None, :]
    return torch.cat([torch.sin(a), torch.cos(a)], dim=-1)


def _lin(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def _gn(sd, prefix, x, groups):
    return F.group_norm(x, groups, sd[prefix + ".weight"], sd[prefix + ".bias"], eps=1e-5)


def _conv(sd, prefix, x, stride=1, padding=1):
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride=stride, padding=padding)


# ----------------------------------------------------------------------------------------------------------
# blocks (modeling/blocks.py, modeling/attention.py)
def conv_block(sd, p, x, emb, groups, mixing):
    """blocks.py:47-75. `emb` None <=> the block has no emb_proj."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups)))
    if (p + ".emb_proj.weight") in sd:
        e = _lin(sd, p + ".emb_proj", F.silu(emb))[:, :, None, None]
        if mixing == "addition":
            h = _gn(sd, p + ".norm2", h + e, groups)
        else:
            scale, shift = torch.chunk(e, 2, dim=1)
            h = (1 + scale) * _gn(sd, p + ".norm2", h, groups) + shift
    else:
        h = _gn(sd, p + ".norm2", h, groups)
    h = _conv(sd, p + ".conv2", F.silu(h))
    if (p + ".res_proj.weight") in sd:
        x = _conv(sd, p + ".res_proj", x, padding=0)
    return h + x


def attention2d(sd, p, x, groups, head_dim=None, n_heads=None):
    """attention.py:62-84 (scaled_dot_product_attention restated as softmax(q k^T / sqrt(d)) v)."""
    B, C, H, W = x.shape
    inner = sd[p + ".qkv_proj.weight"].shape[0] // 3
    d = head_dim if head_dim is not None else inner // n_heads
    nh = inner // d
    h = _gn(sd, p + ".norm", x, groups).view(B, C, H * W).transpose(-1, -2)
    qkv = _lin(sd, p + ".qkv_proj", h).view(B, H * W, 3, nh, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    w = torch.softmax(q @ k.transpose(-1, -2) * (d ** -0.5), dim=-1)
    a = (w @ v).transpose(1, 2).reshape(B, H * W, inner)
    out = _lin(sd, p + ".out_proj", a).transpose(-1, -2).reshape(B, C, H, W)
    return out + x


def unet_forward(sd, cfg, x, t, cls=None, uncond_mask=None):
    """modeling/unet.py:129-163 evaluated from a flat state dict `sd` and the arg_json `model_cfg` dict."""
    groups = cfg.get("norm_groups", 32)
    mixing = cfg.get("embedding_mixing", "addition")
    hd, nh = cfg.get("n_dim_attention_head"), cfg.get("n_attention_heads")
    widths = list(cfg["block_out_channels"])
    n_levels = len(widths)
    nb = cfg.get("n_blocks_per_layer", 2)

    emb = periodic_encoding(t, sd["time_embedding.0.freq"])
    emb = _lin(sd, "time_embedding.1.linear2", F.silu(_lin(sd, "time_embedding.1.linear1", emb)))
    if "class_embedding.weight" in sd:
        table = sd["class_embedding.weight"]
        n_cls = table.shape[0] - 1
        if cls is not None:
            idx = cls if uncond_mask is None else torch.where(uncond_mask.bool(), torch.full_like(cls, n_cls), cls)
        else:
            idx = torch.full((x.shape[0],), n_cls, dtype=torch.int64)
        # row n_cls is the padding row: zero at init and excluded from the gradient (unet.py:51-55, 138-145)
        rows = table[idx]
        emb = emb + torch.where((idx != n_cls)[:, None], rows, rows.detach())

    h = _conv(sd, "in_conv", x)
    skips = [h]
    for i, kind in enumerate(cfg["down_block_types"]):
        attn = "Attn" in kind
        for j in range(nb):
            h = conv_block(sd, f"down.{i}.blocks.{j}", h, emb, groups, mixing)
            if attn:
                h = attention2d(sd, f"down.{i}.attentions.{j}", h, groups, hd, nh)
            skips.append(h)
        if i < n_levels - 1:
            if cfg.get("downsample_use_conv", True):
                h = _conv(sd, f"down.{i}.downsample.conv", h, stride=2, padding=1)
            else:
                h = F.avg_pool2d(h, 2, 2)
            skips.append(h)

    h = conv_block(sd, "mid.blocks.0", h, emb, groups, mixing)
    if cfg.get("mid_attention", True):
        h = attention2d(sd, "mid.attentions.0", h, groups, hd, nh)
    h = conv_block(sd, "mid.blocks.1", h, emb, groups, mixing)

    for i, kind in enumerate(cfg["up_block_types"]):
        attn = "Attn" in kind
        for j in range(nb + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = conv_block(sd, f"up.{i}.blocks.{j}", h, emb, groups, mixing)
            if attn:
                h = attention2d(sd, f"up.{i}.attentions.{j}", h, groups, hd, nh)
        if i < n_levels - 1:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if cfg.get("upsample_use_conv", True):
                h = _conv(sd, f"up.{i}.upsample.conv", h)

    h = F.silu(_gn(sd, "out_norm", h, groups))
    return _conv(sd, "out_conv", h)


def guided_pred(sd, cfg, z, t, cond=None, guidance_scale=1.0):
    """modeling/predictor.py:15-32, including the interleave/first-half-mask pairing as written."""
    if cond is None or guidance_scale == 0.0:
        return unet_forward(sd, cfg, z, t)
    if guidance_scale == 1.0:
        return unet_forward(sd, cfg, z, t, cls=cond)
    both = unet_forward(sd, cfg, torch.repeat_interleave(z, 2, dim=0), torch.repeat_interleave(t, 2, dim=0),
                        cls=torch.repeat_interleave(cond, 2, dim=0),
                        uncond_mask=torch.cat([torch.ones_like(cond), torch.zeros_like(cond)], dim=0))
    u, c = torch.chunk(both, 2, dim=0)
    return u + guidance_scale * (c - u)


# ----------------------------------------------------------------------------------------------------------
# noising and losses (diffusion/scheduler.py:60-83, 134-153)
def diffuse_vp(x, t, eps, alpha_sq):
    """scheduler.py:75-83: square roots in fp64, cast to x.dtype, broadcast per sample."""
    a = torch.sqrt(alpha_sq[t]).to(x.dtype).view(-1, *([1] * (x.dim() - 1)))
    s = torch.sqrt(1 - alpha_sq[t]).to(x.dtype).view(-1, *([1] * (x.dim() - 1)))
    return a * x + s * eps


def diffuse_rf(x, t, eps, sigmas):
    """scheduler.py:149-153."""
    s = sigmas.to(x.dtype)[t].view(-1, *([1] * (x.dim() - 1)))
    return (1 - s) * x + s * eps


def loss_vp(model, x, t, eps, alpha_sq, **kw):
    """scheduler.py:60-73 with (eps, t) given: mean squared error against the noise."""
    return F.mse_loss(model(diffuse_vp(x, t, eps, alpha_sq), t, **kw), eps)


def loss_rf(model, x, t, eps, sigmas, **kw):
    """scheduler.py:134-147 with (eps, t) given: mean squared error against x - eps."""
    return F.mse_loss(model(diffuse_rf(x, t, eps, sigmas), t, **kw), x - eps)


def draw_eps_t(x, n_steps, gen):
    """RNG order of scheduler.py:62-67 / 136-141: randn_like first, then randint."""
    eps = torch.randn_like(x, generator=gen)
    t = torch.randint(0, n_steps, size=(x.size(0),), generator=gen)
    return eps, t


# ----------------------------------------------------------------------------------------------------------
# samplers (diffusion/sampler.py)
def sample_ddpm(z, beta, alpha_sq, n_steps, pred_fn, noise_fn):
    """sampler.py:50-86. noise_fn(i) returns the step-i noise (the reference draws randn_like from one generator)."""
    ts = timestep_grid(len(beta), n_steps)
    b32 = beta.to(z.dtype)
    coef = b32 / torch.sqrt(1 - alpha_sq).to(z.dtype)
    inv = 1 / torch.sqrt(1 - b32)
    sig = torch.sqrt(b32)
    for i in range(n_steps):
        t = int(ts[i])
        e = pred_fn(z, torch.full((z.size(0),), t, dtype=torch.int64))
        z = inv[t] * (z - coef[t] * e)
        if i < n_steps - 1:
            z = z + sig[t] * noise_fn(i)
    return z


def sample_ddim(z, alpha_sq, n_steps, pred_fn, clip_latent=True):
    """sampler.py:90-132 (eta is never used by the reference)."""
    ts = timestep_grid(len(alpha_sq), n_steps)
    for i in range(n_steps):
        t = int(ts[i])
        e = pred_fn(z, torch.full((z.size(0),), t, dtype=torch.int64))
        a = torch.sqrt(alpha_sq[t]).to(z.dtype)
        s = torch.sqrt(1 - alpha_sq[t]).to(z.dtype)
        x0 = (z - s * e) / a
        if clip_latent:
            x0 = torch.clip(x0, -1, 1)
        if i < n_steps - 1:
            tn = int(ts[i + 1])
            z = torch.sqrt(alpha_sq[tn]).to(z.dtype) * x0 + torch.sqrt(1 - alpha_sq[tn]).to(z.dtype) * e
        else:
            z = x0
    return z


def sample_euler(z, sigmas, n_steps, pred_fn):
    """sampler.py:135-164."""
    ts = timestep_grid(len(sigmas), n_steps)
    for i in range(n_steps):
        t = int(ts[i])
        v = pred_fn(z, torch.full((z.size(0),), t, dtype=torch.int64))
        nxt = sigmas[int(ts[i + 1])] if i < n_steps - 1 else 0
        z = z - (nxt - sigmas[t]) * v
    return z


# ----------------------------------------------------------------------------------------------------------
# the training step (the caller semantics of train.py:429-452 and :265-279)
class AdamWState:
    """torch.optim.AdamW defaults used by the reference (train.py:273-277): eps 1e-8, weight_decay 1e-2."""

    def __init__(self, params, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.step = 0

    def apply(self, params, grads):
        self.step += 1
        bc1, bc2 = 1 - self.b1 ** self.step, 1 - self.b2 ** self.step
        for k, p in params.items():
            g = grads[k]
            p.mul_(1 - self.lr * self.wd)
            self.m[k].lerp_(g, 1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def clip_grads_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (train.py:441-444): global L2 norm, coefficient clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def ema_decay(optimization_step, inv_gamma=1.0, power=0.75, max_decay=0.9999, min_decay=0.0):
    """diffusers.training_utils.EMAModel.get_decay with use_ema_warmup=True (train.py:213-218). The package is
    absent from the build container, so this follows its published formula; parity unpinned."""
    step = max(0, optimization_step - 1)
    if step <= 0:
        return 0.0
    return max(min(1 - (1 + step / inv_gamma) ** -power, max_decay), min_decay)


def param_names(sd):
    return [k for k in sd if not k.endswith(("uncond_class", ".freq", "dropout_p"))]


def train_step(sd, cfg, opt, batch, sched_kind, table, clip=1.0, ema=None, ema_step=None, ema_kw=None):
    """One optimisation step on the flat state dict `sd` (updated in place): loss (given eps, t), backward,
    clip, AdamW, optional EMA. Returns (loss, grad_norm). batch = (x, cls, uncond_mask, eps, t)."""
    x, cls, mask, eps, t = batch
    names = param_names(sd)
    params = {k: sd[k].detach().requires_grad_(True) for k in names}
    live = dict(sd)
    live.update(params)
    model = lambda xt, tt, **kw: unet_forward(live, cfg, xt, tt, **kw)  # noqa: E731
    if sched_kind == "vp":
        loss = loss_vp(model, x, t, eps, table, cls=cls, uncond_mask=mask)
    else:
        loss = loss_rf(model, x, t, eps, table, cls=cls, uncond_mask=mask)
    gl = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(names, gl)}
    gnorm = clip_grads_(grads, clip) if clip and clip > 0 else None
    with torch.no_grad():
        opt.apply({k: sd[k] for k in names}, grads)
        if ema is not None:
            d = ema_decay(ema_step, **(ema_kw or {}))
            for k in names:
                ema[k].sub_((1 - d) * (ema[k] - sd[k]))
    return loss.detach(), gnorm, grads
