"""CPU: host-side logic of the drop-in packages and the C-ABI surface (no kernel is launched)."""
import ctypes
import re
import os

import numpy as np
import pytest
import torch

from tests.parity import ROOT, contract, golden, load_cfg


def test_library_exports_every_declared_symbol():
    import dm_hip
    header = open(os.path.join(ROOT, "include", "dm_hip.h")).read()
    declared = set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", header))
    declared -= {"dm_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(dm_hip.lib, name), f"{name} declared in include/dm_hip.h but not exported"
    assert set(dm_hip.EXPORTS) == declared, set(dm_hip.EXPORTS) ^ declared
    assert dm_hip.lib.dm_abi_version() == 3


def test_no_packed_fp32_instruction_takes_its_low_lane_from_the_second_sources_high_register():
    """The built library's device code holds no v_pk_{fma,mul,add}_f32 with op_sel:[x,1,..]: the form that lost a
    product beside matrix-core kernels on MI355X (tools/check_packed_forms.py has the story; the GPU suite's
    test_fp32_operators_beside_matrix_core_kernels is the run-time side of this check)."""
    import importlib.util
    import shutil
    import dm_hip
    spec = importlib.util.spec_from_file_location("check_packed_forms", os.path.join(ROOT, "tools", "check_packed_forms.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not shutil.which(os.path.join(mod.LLVM, "llvm-objdump")):
        pytest.skip("no ROCm LLVM tools to disassemble with")
    hits, objects = mod.find_forms(dm_hip.LIB_PATH)
    assert objects >= 10, objects
    assert not hits, hits[:8]


def test_status_and_error_text_without_gpu():
    import dm_hip
    rc = dm_hip.lib.dm_timestep_embed(None, None, None, 0, 0, None)
    assert rc == 1
    assert b"dm_timestep_embed" in dm_hip.lib.dm_last_error()
    d = dm_hip.ConvDesc(1, 8, 8, 48, 0, 64, 3, 3, 1, 1, 0, 0, 8, 8)
    rc = dm_hip.lib.dm_conv2d_igemm(ctypes.byref(d), 16, None, 16, None, None, 16, None)
    assert rc == 2 and b"multiples of 64" in dm_hip.lib.dm_last_error()
    with pytest.raises(dm_hip.DmHipError):
        dm_hip.check(rc)


@pytest.mark.parametrize("name", ["beta_cifar", "cos_cifar", "fid_test", "rf_cifar", "rf_cifar_long", "rf_mnist"])
def test_configs_load_unchanged_and_state_dict_contract(name):
    from modeling import get_model
    from diffusion.scheduler import get_scheduler
    from diffusion.sampler import get_sampler
    cfg = load_cfg(name)
    model = get_model(cfg["model_type"], **cfg["model_cfg"])
    sched = get_scheduler(cfg["scheduler_type"], **cfg["scheduler_cfg"])
    sampler = get_sampler(cfg["sampler_type"], **cfg["sampler_cfg"])
    assert sampler.pred_type == sched.pred_type
    c = contract()[name]
    keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]
    assert keys == c["keys"]
    assert [k for k, _ in model.named_parameters()] == c["param_names"]
    assert sum(p.numel() for p in model.parameters()) == c["n_params"]
    # a reference-layout checkpoint loads with all keys matched
    sd = {k: torch.zeros(s, dtype=getattr(torch, dt)) for k, s, dt in c["keys"]}
    assert str(model.load_state_dict(sd)) == "<All keys matched successfully>"


def test_factories_reject_unknown_names():
    from modeling import get_model
    from modeling.activation import get_activation
    from modeling.blocks import get_down_block, get_up_block, ConvBlock
    from diffusion.scheduler import get_scheduler
    from diffusion.sampler import get_sampler
    for fn, arg in [(get_model, "resnet"), (get_activation, "tanh"), (get_scheduler, "edm"), (get_sampler, "heun")]:
        with pytest.raises(ValueError):
            fn(arg)
    with pytest.raises(ValueError):
        get_down_block("NoSuchBlock", 64, 64)
    with pytest.raises(ValueError):
        get_up_block("NoSuchBlock", 64, 64, 64)
    with pytest.raises(ValueError):
        ConvBlock(64, 64, embedding_dim=64, embedding_mixing="concat")
    with pytest.raises(ValueError):
        get_scheduler("beta", schedule="sigmoid")


def test_all_block_types_constructible():
    from modeling.blocks import get_down_block, get_up_block
    for kind in ("UNetDownBlock", "UNetAttnDownBlock", "EncoderDownBlock", "EncoderAttnDownBlock"):
        b = get_down_block(kind, 64, 128, n_dim_head=8, embedding_dim=64)
        assert len(b.blocks) == 2 and b.downsample is not None
    for kind in ("UNetUpBlock", "UNetAttnUpBlock", "DecoderUpBlock", "DecoderAttnUpBlock"):
        b = get_up_block(kind, 128, 64, next_out_channels=64, n_dim_head=8, embedding_dim=64)
        assert len(b.blocks) == 2 and b.upsample is not None


def test_schedule_tables_and_grids_match_reference_vectors():
    from diffusion.scheduler import get_scheduler
    from diffusion.sampler import timestep_grid
    g = golden("schedules.npz")
    lin = get_scheduler("beta", schedule="linear", n_steps=1000)
    cos = get_scheduler("beta", schedule="cosine", cos_s=0.008, cos_max_beta=0.999, n_steps=1000)
    rf = get_scheduler("rectified_flow", n_steps=1000)
    assert np.array_equal(lin.beta.numpy(), g["lin_beta"]) and np.array_equal(lin.alpha_sq.numpy(), g["lin_alpha_sq"])
    assert np.array_equal(cos.beta.numpy(), g["cos_beta"]) and np.array_equal(cos.alpha_sq.numpy(), g["cos_alpha_sq"])
    assert np.array_equal(rf.sigmas.numpy(), g["rf_sigmas"])
    assert lin.n_steps == 1000 and lin.pred_type == "noise" and rf.pred_type == "velocity"
    sch = lin.get_schedule(torch.from_numpy(g["sched_t"]))
    assert np.array_equal(sch.alpha_sq.numpy(), g["sched_alpha_sq"])
    assert np.array_equal(sch.sigma_sq.numpy(), g["sched_sigma_sq"])
    assert np.array_equal(sch.log_snr.numpy(), g["sched_log_snr"])
    for a, b in [(1000, 10), (1000, 50), (1000, 1000), (1000, 7), (37, 5), (13, 13), (1000, 1)]:
        assert np.array_equal(timestep_grid(a, b).numpy(), g[f"tsteps_{a}_{b}"])


def test_sampler_api_surface():
    from diffusion.sampler import BaseSampler, get_sampler
    s = get_sampler("ddim", n_steps=50, eta=0.0, clip_latent=True, pbar=True)
    assert s.n_steps == 50 and s.clip_latent and s.pbar and s.pbar_kwargs == {"leave": False}
    s.set_steps(10)
    assert s.n_steps == 10
    assert list(s.prepare_iterator(range(3))) == [0, 1, 2]
    with pytest.raises(NotImplementedError):
        BaseSampler(5).sample(torch.zeros(1), None, None)


def test_operators_refuse_cpu_tensors():
    from modeling import get_model
    from tests.parity import TINY_CFG
    m = get_model("unet", **TINY_CFG)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 3, 16, 16), torch.zeros(1, dtype=torch.int64))


def test_epoch_permutation_is_the_dataloaders_order():
    """harness.data.epoch_permutation restates what DataLoader(shuffle=True, generator=g) draws per epoch (train.py:
    316-327): checked against torch's own DataLoader over two epochs."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from harness.data import epoch_permutation
    n = 103
    ds = TensorDataset(torch.arange(n))
    rng = torch.Generator().manual_seed(42)
    loader = DataLoader(ds, batch_size=16, shuffle=True, drop_last=False, num_workers=0, generator=rng)
    mine = torch.Generator().manual_seed(42)
    for _ in range(2):
        order = torch.cat([b[0] for b in loader])
        assert torch.equal(order, epoch_permutation(n, mine))


def test_fid_extrapolation_and_frechet_distance():
    """utils/fid_infinity.py: the regression of subset FIDs on 1/n; Frechet distance known answers."""
    import numpy as np
    import torch
    from scipy.stats import linregress
    from harness.evaluation import calculate_frechet_distance, fid_extrapolation
    mu, sig = np.array([1.0, -2.0, 0.5]), np.diag([1.0, 4.0, 9.0])
    assert abs(calculate_frechet_distance(mu, sig, mu, sig)) < 1e-9
    # diagonal Gaussians: ||dmu||^2 + sum (sqrt(a) - sqrt(b))^2
    sig2 = np.diag([4.0, 1.0, 9.0])
    assert abs(calculate_frechet_distance(mu, sig, mu + 1, sig2) - (3.0 + 1.0 + 1.0 + 0.0)) < 1e-9
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(2000, 8, generator=g) * 1.5 + 0.3
    ref_mu, ref_sigma = np.zeros(8), np.eye(8)
    sizes = [500, 1000, 2000]
    r = fid_extrapolation(feats, ref_mu, ref_sigma, sizes, target_n=50_000, seed=42)
    # the same subsets by hand
    rng = np.random.default_rng(42)
    f = feats.numpy()
    fids = []
    for n in sizes:
        sub = f[rng.choice(2000, size=n, replace=False)]
        fids.append(calculate_frechet_distance(sub.mean(0), np.cov(sub, rowvar=False), ref_mu, ref_sigma))
    fit = linregress([1 / n for n in sizes], fids)
    assert np.allclose(r["fids"], fids) and abs(r["fid_infinity"] - fit.intercept) < 1e-12
    assert abs(r["fid_target"] - (fit.slope / 50_000 + fit.intercept)) < 1e-12
    assert r["subset_sizes"] == sizes and 0 <= r["regression"]["r_squared"] <= 1


def test_bench_command_line_parses():
    """bench.py's argument parser builds (a duplicated option would only show when the driver starts the benchmark)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def test_latest_ckpt_written_in_the_reference_form_is_read_without_unpickling_code(tmp_path):
    """The reference stores np.random.get_state() - a tuple holding a numpy array - in latest.pt (train.py:386). Such
    a file (written here, in that form) is read back with weights_only=True plus the handful of numpy globals the array
    needs; a file holding anything else that would have to be executed is still refused."""
    import random
    import numpy as np
    from harness.checkpoint import _np_rng_state_plain, _read_latest
    path = str(tmp_path / "latest.pt")
    state = np.random.get_state()
    torch.save({"global_steps": 7, "np_rng_state": state, "rd_rng_state": random.getstate(),
                "rng_state": torch.get_rng_state(), "model_state_dict": {"w": torch.ones(3)}}, path)
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)                    # the plain safe loader refuses the numpy array ...
    ck = _read_latest(path)                                    # ... the allow-listed one reads it
    assert ck["global_steps"] == 7 and np.array_equal(ck["np_rng_state"][1], state[1])
    # the form this harness writes needs no allow-list and np.random accepts it as it accepts the array form
    plain = _np_rng_state_plain()
    torch.save({"np_rng_state": plain}, path)
    back = torch.load(path, weights_only=True)["np_rng_state"]
    before = np.random.get_state()
    try:
        np.random.set_state(back)
        assert np.array_equal(np.random.get_state()[1], before[1])
    finally:
        np.random.set_state(before)

    class Evil:
        def __reduce__(self):
            return (print, ("executed",))
    torch.save({"x": Evil()}, path)
    with pytest.raises(Exception):
        _read_latest(path)


def test_lr_schedules_match_transformers_get_scheduler():
    """The learning-rate rules evaluated by dm_optimizer_tick_lr (host twin dm_lr_multiplier) against the schedulers
    the reference builds (train.py:281-291: transformers.get_scheduler -> LambdaLR), for every optimizer step up to
    and beyond num_training_steps; argument defaults and errors as transformers has them."""
    transformers = pytest.importorskip("transformers")
    from dm_hip import lib
    from harness.train_step import lr_schedule_args
    base = 2e-4
    cases = [("constant", 0, None, {}), ("constant_with_warmup", 7, None, {}), ("linear", 5, 40, {}), ("linear", 0, 40, {}),
             ("cosine", 5, 40, {}), ("cosine", 3, 40, {"num_cycles": 1.5}), ("cosine_with_restarts", 4, 40, {"num_cycles": 3}),
             ("polynomial", 5, 40, {}), ("polynomial", 5, 40, {"power": 2.0, "lr_end": 1e-5}), ("inverse_sqrt", 6, None, {}),
             ("inverse_sqrt", 0, None, {"timescale": 20})]
    for name, warmup, total, cfg in cases:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=base)
        cfg = {"scheduler_specific_kwargs": cfg} if cfg else {}      # TrainArgs.lr_schduler_cfg, splatted as train.py:284-290 does
        ref = transformers.get_scheduler(name, optimizer=opt, num_warmup_steps=warmup, num_training_steps=total, **cfg)
        args = lr_schedule_args(name, base, warmup, total, cfg)
        for k in range(50):
            want = opt.param_groups[0]["lr"]                    # the rate optimizer step k runs at
            got = base * lib.dm_lr_multiplier(args[0], k, *args[1:])
            assert abs(got - want) <= 1e-12 * base + 1e-15 * abs(want), (name, cfg, k, got, want)
            opt.step()
            ref.step()
    assert lr_schedule_args(None, base) == (0, 0, 0, 0.0, 0.0)
    with pytest.raises(ValueError):
        lr_schedule_args("linear", base, 5, None)
    with pytest.raises(ValueError):
        lr_schedule_args("no_such_rule", base, 5, 10)
    with pytest.raises(ValueError):
        lr_schedule_args("polynomial", base, 5, 10, {"scheduler_specific_kwargs": {"lr_end": 1.0}})
    with pytest.raises(TypeError):
        lr_schedule_args("cosine", base, 5, 10, {"num_cycles": 2})
    with pytest.raises(TypeError):
        lr_schedule_args("cosine", base, 5, 10, {"scheduler_specific_kwargs": {"cycles": 2}})
