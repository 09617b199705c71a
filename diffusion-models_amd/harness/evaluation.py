"""Evaluation post-processing (reference: utils/validate_memo.py, utils/fid_infinity.py, train.py:506-541).

The memorisation metric runs on the GPU through dm_nn2_distances (fp32 MFMA nearest-neighbour kernel). FID-infinity
is host-side numpy/scipy on Inception features; the feature extractor itself (pytorch_fid's InceptionV3 and its
downloaded weights) does not exist offline, so `features` are an input here.
"""
import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from dm_hip import check, lib


@torch.no_grad()
def calc_memorization_metric(generated: torch.Tensor, reference: torch.Tensor, k: float = 1 / 3, batch_size: int = 4096,
                             device=None):
    """utils/validate_memo.py:5-28: the fraction of generated images whose nearest training image is closer than k
    times the second nearest. Returns (memorized, dist_1nn, dist_2nn) with the distances on the host, as the reference.
    `batch_size` bounds the generated rows per kernel call (the reference's 256 is a cdist-matrix memory bound that
    does not apply: no distance matrix is materialised)."""
    device = torch.device(device) if device is not None else (generated.device if generated.is_cuda else reference.device)
    if device.type != "cuda":
        raise RuntimeError("calc_memorization_metric: libdm_hip operators run on the GPU only (there is no CPU fallback)")
    n_gen = generated.size(0)
    ref = reference.reshape(reference.size(0), -1).to(device=device, dtype=torch.float32).contiguous()
    n_ref, D = ref.shape
    st = torch.cuda.current_stream().cuda_stream
    d1s, d2s = [], []
    for i in range(0, n_gen, batch_size):
        gen = generated[i:i + batch_size].reshape(-1, D).to(device=device, dtype=torch.float32).contiguous()
        n = gen.size(0)
        d1 = torch.empty(n, dtype=torch.float32, device=device)
        d2 = torch.empty(n, dtype=torch.float32, device=device)
        ws = torch.empty(lib.dm_nn2_ws_floats(n, n_ref), dtype=torch.float32, device=device)
        check(lib.dm_nn2_distances(gen.data_ptr(), ref.data_ptr(), d1.data_ptr(), d2.data_ptr(), n, n_ref, D,
                                   ws.data_ptr(), st))
        d1s.append(d1.cpu())
        d2s.append(d2.cpu())
    dist_1nn, dist_2nn = torch.cat(d1s), torch.cat(d2s)
    memorized = torch.sum(dist_1nn / dist_2nn < k) / n_gen
    return memorized, dist_1nn, dist_2nn


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    """Frechet distance between two Gaussians, ||mu1 - mu2||^2 + Tr(s1) + Tr(s2) - 2 Tr(sqrt(s1 s2)), with the
    regularisation pytorch_fid applies when the product is near singular. (pytorch_fid, which utils/fid.py:6 imports
    this from, is not installed here: restated from its published formula; parity unpinned.)"""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def fid_extrapolation(features, ref_mu: np.ndarray, ref_sigma: np.ndarray,
                      subset_sizes: Sequence[int] = (2048, 3032, 4016, 5000), target_n: Optional[int] = None,
                      seed: int = 42) -> Dict:
    """utils/fid_infinity.py:11-68 (Chong & Forsyth 2020): FID of random subsets against 1/n, the intercept of the
    linear fit is FID at infinitely many samples; slope / target_n + intercept is the FID expected at target_n."""
    from scipy.stats import linregress
    if isinstance(features, torch.Tensor):
        features = features.cpu().numpy()
    total_n = len(features)
    assert total_n >= max(subset_sizes)
    features = features.reshape(total_n, -1)
    fids, inv_n = [], []
    rng = np.random.default_rng(seed)
    for n in subset_sizes:
        subset = features[rng.choice(total_n, size=n, replace=False)]
        fids.append(calculate_frechet_distance(np.mean(subset, axis=0), np.cov(subset, rowvar=False), ref_mu, ref_sigma))
        inv_n.append(1.0 / n)
    slope, intercept, r, *_ = linregress(inv_n, fids)
    slope, intercept, r = float(slope), float(intercept), float(r)
    ret = {"subset_sizes": list(subset_sizes), "fids": fids, "fid_infinity": intercept,
           "regression": {"slope": slope, "intercept": intercept, "r_squared": r ** 2}}
    if target_n is not None:
        ret["fid_target"] = slope / target_n + intercept
    return ret


def log_fid_evaluation(output_dir: str, steps: int, generated: torch.Tensor, train_data_tensor: torch.Tensor,
                       inception_features=None, fid_reference=None, fid_adjust_subsets: Optional[List[int]] = None,
                       device=None) -> Dict:
    """train.py:506-541 evaluate_fid minus the Inception network: the memorisation ratio always; FID / FID@inf /
    FID@50k when the caller brings Inception features of `generated` and the reference statistics (mu, sigma).
    Appends {"steps": ..., ...} to <output_dir>/fid_evaluations.jsonl like the reference."""
    ret = {}
    if inception_features is not None and fid_reference is not None:
        if fid_adjust_subsets:
            r = fid_extrapolation(inception_features, fid_reference[0], fid_reference[1], fid_adjust_subsets, target_n=50_000)
            ret["FID"], ret["FID@inf"], ret["FID@50k"] = r["fids"][-1], r["fid_infinity"], r["fid_target"]
        else:
            f = inception_features.double() if isinstance(inception_features, torch.Tensor) else torch.from_numpy(inception_features).double()
            mu, sigma = torch.mean(f, dim=0).numpy(), torch.cov(f.T).numpy()   # utils/fid.py:77-82
            ret["FID"] = calculate_frechet_distance(mu, sigma, fid_reference[0], fid_reference[1])
    mem_ratio, *_ = calc_memorization_metric(generated, train_data_tensor, device=device)
    ret["memorization_ratio"] = mem_ratio.item()
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "fid_evaluations.jsonl"), "a") as f:
        f.write(json.dumps(dict(steps=steps, **ret)) + "\n")
    return ret
