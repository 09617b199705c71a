"""On-disk layout of a training run, as the reference writes and reads it (train.py:204-205, 363-415;
save_fid_eval.py:126-140):

    <output_dir>/train_args.json                      the flat config (arg_json format)
    <output_dir>/ckpts/ckpt-%06d/model.pt             model.state_dict()
    <output_dir>/ckpts/ckpt-%06d/ema_model.pt         the EMA weights as a model state dict
    <output_dir>/latest.pt                            everything needed to resume
    <output_dir>/train_log.jsonl, fid_evaluations.jsonl

State dicts are written in the reference's tensor layout (contiguous [Cout, Cin, KH, KW] conv weights), so a
checkpoint written here loads into the reference's UNet and vice versa.
"""
import json
import os
from typing import Optional

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


def save_latest_ckpt(step, output_dir: str, epochs: int = 0, steps_in_epoch: int = 0) -> str:
    """train.py:375-392 (the host RNG states the reference adds belong to its DataLoader workers and are not kept)."""
    os.makedirs(output_dir, exist_ok=True)
    ckpt = dict(_cpu(step.state_dict()), epochs=epochs, steps_in_epoch=steps_in_epoch,
                rng_state=torch.get_rng_state(), cuda_rng_state=torch.cuda.get_rng_state())
    path = os.path.join(output_dir, "latest.pt")
    torch.save(ckpt, path)
    return path


def load_latest_ckpt(step, ckpt_dir: str) -> dict:
    """train.py:394-415. Only tensors and plain containers are read back (weights_only=True)."""
    ckpt = torch.load(os.path.join(ckpt_dir, "latest.pt"), weights_only=True)
    step.load_state_dict(ckpt)
    if "rng_state" in ckpt:
        torch.set_rng_state(ckpt["rng_state"])
    if "cuda_rng_state" in ckpt:
        torch.cuda.set_rng_state(ckpt["cuda_rng_state"])
    return ckpt


def load_for_eval(run_dir: str, ckpt_name: str, use_ema: bool = True, sampling_steps: Optional[int] = None, device=None):
    """save_fid_eval.py:126-142: (model, scheduler, sampler, config) of a checkpoint of a finished run."""
    with open(os.path.join(run_dir, "train_args.json")) as f:
        config = json.load(f)
    model = get_model(config["model_type"], **config["model_cfg"])
    scheduler = get_scheduler(config["scheduler_type"], **config["scheduler_cfg"])
    sampler_cfg = dict(config["sampler_cfg"])
    if sampling_steps is not None:
        sampler_cfg["n_steps"] = sampling_steps
    sampler = get_sampler(config["sampler_type"], **sampler_cfg)
    ckpt_file = "ema_model.pt" if use_ema else "model.pt"
    sd = torch.load(os.path.join(run_dir, "ckpts", ckpt_name, ckpt_file), weights_only=True)
    result = model.load_state_dict(sd)
    assert not result.missing_keys and not result.unexpected_keys, result
    if device is not None:
        model.to(device)
    model.eval()
    return model, scheduler, sampler, config


def log_train(output_dir: str, steps: int, **logs):
    """train.py:418-420."""
    with open(os.path.join(output_dir, "train_log.jsonl"), "a") as f:
        f.write(json.dumps(dict(steps=steps, **logs)) + "\n")
