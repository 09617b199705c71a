"""GPU: the entry point that allocates nothing (dm_groupnorm_act_fwd_noalloc, scratch inside the library) against
dm_groupnorm_act_fwd with a caller workspace, through the C ABI: bit-identical outputs on two calls in a row, the
second with a smaller input (the shared scratch keeps the first call's contents), and the size limit is an error."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, C1, C2, H, W, ss, seed):
    from dm_hip import GnDesc, check, lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    C, G = C1 + C2, 32
    bf = torch.bfloat16
    x1 = (torch.randn(B, H, W, C1, generator=g) + 0.3).to(bf).to(dev)
    x2 = (2 * torch.randn(B, H, W, C2, generator=g)).to(bf).to(dev) if C2 else None
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(dev)
    beta = (0.2 * torch.randn(C, generator=g)).to(dev)
    sst = (0.5 * torch.randn(B, 2 * C, generator=g)).to(dev) if ss else None
    d = GnDesc(B, H * W, C1, C2, G, 1e-5, 1, int(ss), 0)
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for which in ("workspace", "noalloc"):
        y = torch.full((B, H, W, C), float("nan"), dtype=bf, device=dev)
        mean = torch.full((B, G), float("nan"), device=dev)
        rstd = torch.full((B, G), float("nan"), device=dev)
        if which == "workspace":
            ws = torch.full((lib.dm_groupnorm_ws_floats(ctypes.byref(d)),), float("nan"), device=dev)
            check(lib.dm_groupnorm_act_fwd(ctypes.byref(d), p(x1), p(x2), p(gamma), p(beta), p(sst), None, p(y), p(mean),
                                           p(rstd), p(ws), st))
        else:
            check(lib.dm_groupnorm_act_fwd_noalloc(ctypes.byref(d), p(x1), p(x2), p(gamma), p(beta), p(sst), None, p(y),
                                                   p(mean), p(rstd), st))
        torch.cuda.synchronize()
        outs.append((y, mean, rstd))
    (ya, ma, ra), (yb, mb, rb) = outs
    assert bool(torch.isfinite(ya.float()).all()) and bool(torch.isfinite(ma).all())
    assert torch.equal(ya, yb) and torch.equal(ma, mb) and torch.equal(ra, rb)
    # and the operator is right: against torch's fp32 GroupNorm + SiLU on the same bf16-rounded input
    xin = (x1 if x2 is None else torch.cat([x1, x2], -1)).float().permute(0, 3, 1, 2)
    v = torch.nn.functional.group_norm(xin, G, gamma, beta, eps=1e-5)
    if ss:
        sc, sh = torch.chunk(sst[:, :, None, None], 2, dim=1)
        v = (1 + sc) * v + sh
    ref = torch.nn.functional.silu(v).permute(0, 2, 3, 1)
    err = float((yb.float() - ref).norm() / ref.norm())
    assert err < 6e-3, err   # one bf16 rounding of the output


def test_noalloc_groupnorm_equals_workspace_form_two_calls_in_a_row():
    from dm_hip import check, lib
    check(lib.dm_noalloc_init())
    # both shapes take the two-sweep kernels, which are the ones that use the scratch (more than 1024 pixels per
    # sample exceed the single-pass kernels); the second call is smaller
    _case(6, 256, 128, 36, 36, True, 1)
    _case(3, 64, 0, 40, 40, False, 2)
    # single-pass shape (scratch unused) still goes through the same entry point
    _case(2, 128, 0, 8, 8, True, 3)


def test_noalloc_groupnorm_rejects_sizes_beyond_its_scratch():
    from dm_hip import GnDesc, lib
    cap = lib.dm_groupnorm_noalloc_max_ws_floats()
    assert cap == 1 << 20
    B, C = 512, 1024                     # 3 * B * C floats > cap
    d = GnDesc(B, 4, C, 0, 32, 1e-5, 1, 0, 0)
    assert lib.dm_groupnorm_ws_floats(ctypes.byref(d)) > cap
    dev = torch.device("cuda:0")
    x = torch.zeros(B, 2, 2, C, dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(x)
    gb = torch.ones(C, device=dev)
    ms = torch.empty(B, 32, device=dev)
    rc = lib.dm_groupnorm_act_fwd_noalloc(ctypes.byref(d), x.data_ptr(), None, gb.data_ptr(), gb.data_ptr(), None, None,
                                          y.data_ptr(), ms.data_ptr(), ms.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
    assert rc == 4 and b"scratch holds" in lib.dm_last_error()   # DM_ERR_UNSUPPORTED
