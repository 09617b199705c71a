"""On-disk layout of a training run, as the reference writes and reads it (train.py:204-205, 363-415;
save_fid_eval.py:126-140):

    <output_dir>/train_args.json                      the flat config (arg_json format)
    <output_dir>/ckpts/ckpt-%06d/model.pt             model.state_dict()
    <output_dir>/ckpts/ckpt-%06d/ema_model.pt         the EMA weights as a model state dict
    <output_dir>/latest.pt                            everything needed to resume
    <output_dir>/train_log.jsonl, fid_evaluations.jsonl

State dicts are written in the reference's tensor layout (contiguous [Cout, Cin, KH, KW] conv weights), so a
checkpoint written here loads into the reference's UNet and vice versa. latest.pt carries every key the reference's
load_latest_ckpt indexes (train.py:398-413), the two host RNG states included, so the reference can resume a run
written here; a latest.pt written by the reference (whose numpy RNG state is a pickled ndarray) is read back with the
few numpy globals that array needs allow-listed, nothing else unpickled. No learning-rate scheduler state is kept:
every shipped config trains without one (train.py:281-291 yields None), and the reference reads that key only when a
scheduler is configured.
"""
import json
import os
import random
from typing import Optional

import numpy as np
import torch

from diffusion.sampler import get_sampler
from diffusion.scheduler import get_scheduler
from modeling import get_model


def ckpt_dir_for(output_dir: str, global_steps: int) -> str:
    return os.path.join(output_dir, "ckpts", f"ckpt-{global_steps:06d}")   # train.py:651


def write_train_args(output_dir: str, cfg: dict) -> str:
    """train.py:204-205: the run's config next to its outputs; save_fid_eval.py:127-129 reads it back."""
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "train_args.json")
    with open(path, "w") as f:
        json.dump(cfg, f, indent=2)
    return path


def save_ckpt(step, output_dir: str, global_steps: Optional[int] = None) -> str:
    """train.py:363-373: model.pt (+ ema_model.pt when the run keeps an EMA) under ckpts/ckpt-%06d."""
    d = ckpt_dir_for(output_dir, step.steps if global_steps is None else global_steps)
    os.makedirs(d, exist_ok=True)
    torch.save({k: v.cpu() for k, v in step.model_state_dict().items()}, os.path.join(d, "model.pt"))
    if step.ema is not None:
        torch.save({k: v.cpu() for k, v in step.ema_state_dict().items()}, os.path.join(d, "ema_model.pt"))
    return d


def _cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.cpu()
    if isinstance(obj, dict):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


def _np_rng_state_plain():
    """np.random.get_state() with the key vector as a plain list of ints: np.random.set_state accepts it as it
    accepts the ndarray (train.py:412), and the file stays loadable with weights_only=True."""
    name, keys, pos, has_gauss, cached = np.random.get_state()
    return (str(name), [int(k) for k in keys], int(pos), int(has_gauss), float(cached))


def save_latest_ckpt(step, output_dir: str, epochs: int = 0, steps_in_epoch: int = 0, dataset=None) -> str:
    """train.py:375-392. dataset: the DeviceDataset feeding the run; the state of its flip generator is kept too
    (the reference's flips come from its DataLoader workers' generators, which it cannot save), under a key of this
    harness's own that the reference ignores."""
    os.makedirs(output_dir, exist_ok=True)
    ckpt = dict(_cpu(step.state_dict()), epochs=epochs, steps_in_epoch=steps_in_epoch,
                rng_state=torch.get_rng_state(), cuda_rng_state=torch.cuda.get_rng_state(),
                np_rng_state=_np_rng_state_plain(), rd_rng_state=random.getstate())
    if dataset is not None:
        ckpt["flip_rng_state"] = dataset.flip_gen.get_state().cpu()
    path = os.path.join(output_dir, "latest.pt")
    torch.save(ckpt, path)
    return path


def _read_latest(path):
    """latest.pt with nothing executed from the file: tensors and plain containers only; if that refuses the file
    because of a numpy array inside (the reference stores np.random.get_state() as is), once more with exactly the
    globals a numpy array needs allow-listed."""
    try:
        return torch.load(path, weights_only=True)
    except Exception as first:   # noqa: BLE001  (pickle.UnpicklingError carrying the refused global's name)
        if "numpy" not in str(first):
            raise
        allowed = [np.ndarray, np.dtype, type(np.dtype(np.uint32))]
        for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
            try:
                allowed.append(getattr(__import__(mod, fromlist=["_reconstruct"]), "_reconstruct"))
            except (ImportError, AttributeError):
                pass
        with torch.serialization.safe_globals(allowed):
            return torch.load(path, weights_only=True)


def load_latest_ckpt(step, ckpt_dir: str, dataset=None) -> dict:
    """train.py:394-415. Returns the checkpoint dict; its epochs / steps_in_epoch are where DeviceDataset.epochs
    (start_epoch, start_step) takes the data order up again (the reference's skip_previous_steps, train.py:598-605)."""
    ckpt = _read_latest(os.path.join(ckpt_dir, "latest.pt"))
    step.load_state_dict(ckpt)
    if "rng_state" in ckpt:
        torch.set_rng_state(ckpt["rng_state"])
    if "cuda_rng_state" in ckpt:
        torch.cuda.set_rng_state(ckpt["cuda_rng_state"])
    if "np_rng_state" in ckpt:
        name, keys, pos, has_gauss, cached = ckpt["np_rng_state"]
        np.random.set_state((name, np.asarray(keys, dtype=np.uint32), int(pos), int(has_gauss), float(cached)))
    if "rd_rng_state" in ckpt:
        version, internal, gauss_next = ckpt["rd_rng_state"]
        random.setstate((version, tuple(internal), gauss_next))
    if dataset is not None and "flip_rng_state" in ckpt:
        dataset.flip_gen.set_state(ckpt["flip_rng_state"].cpu())
    return ckpt


def load_for_eval(run_dir: str, ckpt_name: str, use_ema: bool = True, sampling_steps: Optional[int] = None, device=None,
                  model=None):
    """save_fid_eval.py:126-142: (model, scheduler, sampler, config) of a checkpoint of a finished run.
    model: the model a previous call returned for the same run. The checkpoint is then loaded INTO it (parameters keep
    their addresses), so a loop over checkpoints (save_fid_eval.py:main_all_ckpt) goes on replaying the HIP graphs that
    generate() captured for that model instead of capturing - and keeping alive - a new set per checkpoint."""
    with open(os.path.join(run_dir, "train_args.json")) as f:
        config = json.load(f)
    if model is None:
        model = get_model(config["model_type"], **config["model_cfg"])
        if device is not None:
            model.to(device)     # before loading: load_state_dict copies into the device tensors
    scheduler = get_scheduler(config["scheduler_type"], **config["scheduler_cfg"])
    sampler_cfg = dict(config["sampler_cfg"])
    if sampling_steps is not None:
        sampler_cfg["n_steps"] = sampling_steps
    sampler = get_sampler(config["sampler_type"], **sampler_cfg)
    ckpt_file = "ema_model.pt" if use_ema else "model.pt"
    sd = torch.load(os.path.join(run_dir, "ckpts", ckpt_name, ckpt_file), weights_only=True)
    result = model.load_state_dict(sd)
    assert not result.missing_keys and not result.unexpected_keys, result
    model.eval()
    return model, scheduler, sampler, config


def log_train(output_dir: str, steps: int, **logs):
    """train.py:418-420."""
    with open(os.path.join(output_dir, "train_log.jsonl"), "a") as f:
        f.write(json.dumps(dict(steps=steps, **logs)) + "\n")
