"""CPU: the oracle (oracle/unet_oracle.py) against the golden vectors generated from the reference."""
import numpy as np
import pytest
import torch

from tests.parity import TINY_ATTN_CFG, TINY_CFG, T, contract, golden, load_cfg, max_abs, rel_err
from detweights import det_state_dict
from oracle import unet_oracle as O


def test_schedules_bitexact():
    g = golden("schedules.npz")
    assert np.array_equal(O.beta_linear().numpy(), g["lin_beta"])
    assert np.array_equal(O.alpha_sq_from_beta(O.beta_linear()).numpy(), g["lin_alpha_sq"])
    assert np.array_equal(O.beta_cosine().numpy(), g["cos_beta"])
    assert np.array_equal(O.alpha_sq_from_beta(O.beta_cosine()).numpy(), g["cos_alpha_sq"])
    assert np.array_equal(O.beta_linear(1e-3, 5e-2, 37).numpy(), g["lin_small_beta"])
    assert np.array_equal(O.beta_cosine(0.01, 0.9, 50).numpy(), g["cos_small_beta"])
    assert np.array_equal(O.rf_sigmas().numpy(), g["rf_sigmas"])
    assert np.array_equal(O.rf_sigmas(13).numpy(), g["rf_small_sigmas"])
    # known answers probed from the reference (linear and cosine schedules)
    a = O.alpha_sq_from_beta(O.beta_linear())
    assert float(a[500]) == 0.07779665836502389 and float(a[999]) == 4.0358297653756754e-05


@pytest.mark.parametrize("n_sched,n", [(1000, 10), (1000, 50), (1000, 1000), (1000, 7), (37, 5), (13, 13), (1000, 1)])
def test_timestep_grid(n_sched, n):
    assert np.array_equal(O.timestep_grid(n_sched, n), golden("schedules.npz")[f"tsteps_{n_sched}_{n}"])


def test_periodic_encoding():
    g = golden("embeddings.npz")
    for dim in (64, 128):
        f = O.periodic_freq(dim)
        assert np.array_equal(f.numpy(), g[f"freq_{dim}"])
        assert np.array_equal(O.periodic_encoding(T(g["t"]), f).numpy(), g[f"enc_{dim}"])


def _unet_case(name, cfg):
    g = golden(f"unet_{name}.npz")
    shapes = [(k, s) for k, s, _ in _contract_keys(cfg)]
    vals = det_state_dict(shapes, int(g["seed"]))
    sd = {}
    for k, s in shapes:
        if vals[k] is not None:
            sd[k] = T(vals[k])
    sd["time_embedding.0.freq"] = O.periodic_freq(cfg["time_encoding_dim"])
    return g, sd


def _contract_keys(cfg):
    for name, c in contract().items():
        if load_cfg(name)["model_cfg"] == cfg:
            return c["keys"]
    # tiny configs are not among the arg_json files: derive the layout from the build's own module tree
    from modeling import get_model
    m = get_model("unet", **cfg)
    return [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]


@pytest.mark.parametrize("name,cfg", [("tiny", TINY_CFG), ("tiny_attnblocks", TINY_ATTN_CFG),
                                      ("fid_test", None), ("rf_mnist", None)])
def test_unet_forward_backward(name, cfg):
    cfg = cfg or load_cfg(name)["model_cfg"]
    g, sd = _unet_case(name, cfg)
    names = O.param_names(sd)
    for k in names:
        sd[k].requires_grad_(True)
    x, t, cls, mask = T(g["x"]), T(g["t"]), T(g["cls"]), T(g["mask"])
    out = O.unet_forward(sd, cfg, x, t, cls=cls, uncond_mask=mask)
    assert max_abs(out, T(g["out"])) < 2e-4
    with torch.no_grad():
        assert max_abs(O.unet_forward(sd, cfg, x, t), T(g["out_nocls"])) < 2e-4
        assert max_abs(O.unet_forward(sd, cfg, x, t, cls=cls), T(g["out_nomask"])) < 2e-4
    grads = torch.autograd.grad(out, [sd[k] for k in names], grad_outputs=T(g["gy"]), allow_unused=True)
    gnames = [str(s) for s in g["grad_names"]]
    assert gnames == names
    for k, gr, ref_norm in zip(names, grads, g["grad_norms"]):
        n = 0.0 if gr is None else float(gr.double().norm())
        assert abs(n - ref_norm) <= 2e-4 * max(1.0, ref_norm), k


def test_unet_cos_cifar_forward():
    cfg = load_cfg("cos_cifar")["model_cfg"]
    g, sd = _unet_case("cos_cifar", cfg)
    with torch.no_grad():
        out = O.unet_forward(sd, cfg, T(g["x"]), T(g["t"]), cls=T(g["cls"]), uncond_mask=T(g["mask"]))
    assert max_abs(out, T(g["out"])) < 5e-4


def test_guided_pred():
    g = golden("unet_tiny_cfg.npz")
    _, sd = _unet_case("tiny", TINY_CFG)
    z, t, cls = T(g["z"]), T(g["t"]), T(g["cls"])
    with torch.no_grad():
        for key, kw in [("g0", dict(cond=cls, guidance_scale=0.0)), ("g1", dict(cond=cls, guidance_scale=1.0)),
                        ("g2", dict(cond=cls, guidance_scale=2.0)), ("gnone", dict(cond=None, guidance_scale=3.0))]:
            assert max_abs(O.guided_pred(sd, TINY_CFG, z, t, **kw), T(g[key])) < 2e-4, key


def test_losses_and_noising():
    g = golden("diffusion.npz")
    x = T(g["x"])
    half = lambda x_, t_, **kw: 0.5 * x_  # noqa: E731
    tabs = {"lin": O.alpha_sq_from_beta(O.beta_linear()), "cos": O.alpha_sq_from_beta(O.beta_cosine()), "rf": O.rf_sigmas()}
    for k in ("lin", "cos", "rf"):
        gen = torch.Generator().manual_seed(42)
        eps, t = O.draw_eps_t(x, 1000, gen)
        assert np.array_equal(eps.numpy(), g[f"eps_{k}"]) and np.array_equal(t.numpy(), g[f"t_{k}"])
        if k == "rf":
            assert np.array_equal(O.diffuse_rf(x, t, eps, tabs[k]).numpy(), g[f"xt_{k}"])
            loss = O.loss_rf(half, x, t, eps, tabs[k])
        else:
            assert np.array_equal(O.diffuse_vp(x, t, eps, tabs[k]).numpy(), g[f"xt_{k}"])
            loss = O.loss_vp(half, x, t, eps, tabs[k])
        assert float(loss) == float(g[f"loss_{k}"])
    # known answers probed from the reference (linear and cosine schedules)
    assert abs(float(g["loss_lin"]) - 0.40292478) < 1e-7 and abs(float(g["loss_rf"]) - 1.79405022) < 1e-7


def _same(out, ref):
    # bit-exact on the CPU model that generated the vectors; torch's vectorised fp32 sqrt/reciprocal are not
    # correctly rounded everywhere, so on another CPU model the coefficient tables may move by an ulp
    ref = T(ref)
    return torch.equal(out, ref) or rel_err(out, ref) < 5e-6


def test_samplers_bitexact():
    g = golden("diffusion.npz")
    z = T(g["z"])
    pred = lambda z_, t_: 0.1 * z_ + 1e-6 * t_.view(-1, 1, 1, 1)  # noqa: E731
    lin, cos, rf = O.alpha_sq_from_beta(O.beta_linear()), O.alpha_sq_from_beta(O.beta_cosine()), O.rf_sigmas()
    assert _same(O.sample_euler(z, rf, 50, pred), g["euler50"])
    assert _same(O.sample_euler(z, rf, 7, pred), g["euler7"])
    assert _same(O.sample_ddim(z, lin, 50, pred), g["ddim50_lin"])
    assert _same(O.sample_ddim(z, cos, 50, pred), g["ddim50_cos"])
    assert _same(O.sample_ddim(z, cos, 10, pred, clip_latent=False), g["ddim10_cos_noclip"])
    noise = T(g["noise_ddpm20_cos_seed3"])
    out = O.sample_ddpm(z, O.beta_cosine(), cos, 20, pred, lambda i: noise[i])
    assert _same(out, g["ddpm20_cos_seed3"])
    gen = torch.Generator()
    gen.manual_seed(7)
    out = O.sample_ddpm(z, O.beta_linear(), lin, 1000, pred, lambda i: torch.randn_like(z, generator=gen))
    assert _same(out, g["ddpm1000_lin_seed7"])
    assert abs(float(out.double().sum()) - (-8037.306618)) < 1e-3  # known answer probed from the reference


@pytest.mark.parametrize("fixture,kind", [("train_steps_tiny.npz", "vp"), ("train_steps_tiny_rf.npz", "rf")])
def test_train_steps(fixture, kind):
    g = golden(fixture)
    from modeling import get_model
    m = get_model("unet", **TINY_CFG)
    shapes = [(k, v.shape) for k, v in m.state_dict().items()]
    vals = det_state_dict(shapes, int(g["seed"]))
    sd = {k: (T(vals[k]) if vals[k] is not None else v.clone()) for (k, _), v in zip(shapes, m.state_dict().values())}
    names = O.param_names(sd)
    assert names == [str(s) for s in g["param_names"]]
    opt = O.AdamWState({k: sd[k] for k in names}, lr=2e-4)
    table = O.alpha_sq_from_beta(O.beta_cosine()) if kind == "vp" else O.rf_sigmas()
    if kind == "rf":
        # RectifiedFlowScheduler.get_loss with its own RNG order (eps, then t) on the initial weights
        gen = torch.Generator().manual_seed(5)
        eps, t = O.draw_eps_t(T(g["getloss_x"]), 1000, gen)
        assert np.array_equal(eps.numpy(), g["getloss_eps"]) and np.array_equal(t.numpy(), g["getloss_t"])
        with torch.no_grad():
            model = lambda xt, tt, **kw: O.unet_forward(sd, TINY_CFG, xt, tt, **kw)  # noqa: E731
            assert abs(float(O.loss_rf(model, T(g["getloss_x"]), t, eps, table, cls=T(g["getloss_cls"]))) -
                       float(g["getloss_value"])) < 1e-5
    for s in range(int(g["steps"])):
        batch = tuple(T(g[f"{k}{s}"]) for k in ("x", "cls", "mask", "eps", "t"))
        loss, gnorm, _ = O.train_step(sd, TINY_CFG, opt, batch, kind, table, clip=1.0)
        assert abs(float(loss) - float(g[f"loss{s}"])) < 1e-5
        assert abs(float(gnorm) - float(g[f"gnorm{s}"])) < 1e-4 * float(g[f"gnorm{s}"])
        pn = np.array([float(sd[k].double().norm()) for k in names])
        assert np.allclose(pn, g[f"pnorm{s}"], rtol=1e-5, atol=1e-6)
        ps = np.stack([np.pad(sd[k].flatten()[:8].numpy(), (0, max(0, 8 - sd[k].numel()))) for k in names])
        assert np.allclose(ps, g[f"psample{s}"], rtol=2e-4, atol=2e-6)


def test_samples_through_the_network():
    """Images generated through reference-trained UNets (pred_fn = guided_pred) by every sampler family, against the
    reference's images (fp32 on both sides)."""
    from modeling import get_model
    from tests.parity import MINI_CFG, trained_mini_state
    g = golden("samples.npz")
    z, cls = T(g["z"]), T(g["cls"])
    lin, cos, rf = O.alpha_sq_from_beta(O.beta_linear()), O.alpha_sq_from_beta(O.beta_cosine()), O.rf_sigmas()
    base = {k: v.clone() for k, v in get_model("unet", **MINI_CFG).state_dict().items()}
    outs = {}
    with torch.no_grad():
        sd = dict(base, **trained_mini_state(g, "vp"))
        pred = lambda z_, t_: O.guided_pred(sd, MINI_CFG, z_, t_, cond=cls, guidance_scale=1.0)  # noqa: E731
        pred4 = lambda z_, t_: O.guided_pred(sd, MINI_CFG, z_, t_, cond=cls[:4], guidance_scale=1.0)  # noqa: E731
        pred2 = lambda z_, t_: O.guided_pred(sd, MINI_CFG, z_, t_, cond=cls, guidance_scale=2.0)  # noqa: E731
        noise = T(g["noise_ddpm50_lin"])
        outs["ddim50_lin"] = O.sample_ddim(z, lin, 50, pred)
        outs["ddim50_cos"] = O.sample_ddim(z, cos, 50, pred)
        outs["ddpm50_lin"] = O.sample_ddpm(z[:4], O.beta_linear(), lin, 50, pred4, lambda i: noise[i])
        outs["ddim20_lin_g2"] = O.sample_ddim(z, lin, 20, pred2)
        sd_rf = dict(base, **trained_mini_state(g, "rf"))
        outs["euler50_rf"] = O.sample_euler(z, rf, 50, lambda z_, t_: O.guided_pred(sd_rf, MINI_CFG, z_, t_, cond=cls,
                                                                                     guidance_scale=1.0))
    for k, out in outs.items():
        err = max_abs(torch.clip(out, -1, 1), T(g[k]))
        assert err < 1e-3, (k, err)   # summation-order differences of the fp32 kernels over 20-50 network evaluations


def test_ddpm_1000_steps_through_the_network():
    """Ancestral sampling over all 1000 steps (diffusion/sampler.py:50-86) through the reference-trained mini network,
    with the sampler's own noise stream (generator seeded 11, one randn_like per step but the last) regenerated here:
    oracle against the reference's images, fp32 on both sides."""
    from modeling import get_model
    from tests.parity import MINI_CFG, ddpm_noise_stream, trained_mini_state
    g, ref = golden("samples.npz"), golden("samples_ddpm1000.npz")
    z4, cls4 = T(g["z"])[:4], T(g["cls"])[:4]
    base = {k: v.clone() for k, v in get_model("unet", **MINI_CFG).state_dict().items()}
    sd = dict(base, **trained_mini_state(g, "vp"))
    noise = ddpm_noise_stream(z4, int(ref["seed"]), 1000)
    with torch.no_grad():
        out = O.sample_ddpm(z4, O.beta_linear(), O.alpha_sq_from_beta(O.beta_linear()), 1000,
                            lambda z_, t_: O.guided_pred(sd, MINI_CFG, z_, t_, cond=cls4, guidance_scale=1.0),
                            lambda i: noise[i])
    err = max_abs(torch.clip(out, -1, 1), T(ref["ddpm1000_lin"]))
    assert err < 2e-3, err      # fp32 summation-order differences over 1000 network evaluations


def test_loss_curve_head():
    """The oracle follows the reference's loss curve (first 40 of the fixture's 1000 steps; fp32 on both sides)."""
    from itertools import islice
    from modeling import get_model
    from tests.parity import CURVE, curve_batches
    g = golden("train_curve_tiny.npz")
    m = get_model("unet", **TINY_CFG)
    shapes = [(k, v.shape) for k, v in m.state_dict().items()]
    vals = det_state_dict(shapes, CURVE["weights_seed"])
    sd = {k: (T(vals[k]) if vals[k] is not None else v.clone()) for (k, _), v in zip(shapes, m.state_dict().values())}
    opt = O.AdamWState({k: sd[k] for k in O.param_names(sd)}, lr=CURVE["lr"])
    table = O.alpha_sq_from_beta(O.beta_cosine())
    for s, batch in enumerate(islice(curve_batches(), 40)):
        loss, gnorm, _ = O.train_step(sd, TINY_CFG, opt, batch, "vp", table, clip=1.0)
        assert abs(float(loss) - float(g["loss"][s])) < 2e-3 * max(1.0, float(g["loss"][s])), s
