"""Which tensor of a captured forward pass is the first to differ when the graph is launched right behind another
launch of itself (no host synchronisation in between)? Copies of the intermediate tensors are captured into the
graph; after every pair of launches they are compared with the copies an eager pass made.

  DM_HIP_LIBRARY=tools/bin/libdm_hip_fast.so python tools/diag_localise.py [pairs] [variant]
variant: plain | sync (synchronise between the two launches) | joinearly (the main stream waits for the embedding
branch right after the fork)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]

import torch  # noqa: E402

from dm_hip import ops  # noqa: E402
from modeling import blocks as blocks_mod  # noqa: E402
from modeling.blocks import ConvBlock  # noqa: E402
from tests.test_gpu_harness import _batch, _make_step, dev  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
variant = sys.argv[2] if len(sys.argv) > 2 else "plain"

DUMP = None   # list of (name, tensor) being filled
HOOKS = os.environ.get("DIAG_HOOKS", "emb,main").split(",")   # which groups of tensors to copy


def keep(name, t):
    if DUMP is None:
        return
    if isinstance(t, (tuple, list)):
        for k, v in enumerate(t):
            if torch.is_tensor(v):
                keep(f"{name}[{k}]", v)
    elif torch.is_tensor(t):
        DUMP.append((name, t.detach().clone()))


_orig_init = blocks_mod.ProjectedEmbeddings.__init__


def _init(self, embedding, blks):
    _orig_init(self, embedding, blks)
    if "emb" not in HOOKS:
        return
    keep("emb.raw", self._raw)
    keep("emb.proj", torch.cat([o.flatten() for o in self._proj.values()]))


blocks_mod.ProjectedEmbeddings.__init__ = _init
if variant == "joinearly":
    _orig_produced = blocks_mod.ProjectedEmbeddings.produced_on

    def _produced(self, stream):
        _orig_produced(self, stream)
        self.join()

    blocks_mod.ProjectedEmbeddings.produced_on = _produced


def hook_model(m):
    if "emb" in HOOKS:
        m.time_embedding.register_forward_hook(lambda mod, i, o: keep("time_embedding", o))
    if "main" not in HOOKS:
        return 0
    m.register_forward_pre_hook(lambda mod, args: keep("x_t", args[0]))
    m.in_conv.register_forward_hook(lambda mod, i, o: keep("in_conv", o))
    n = 0
    for name, sub in m.named_modules():
        if isinstance(sub, ConvBlock):
            sub.norm1.register_forward_hook(lambda mod, i, o, name=name: keep(name + ".norm1", o))
            sub.conv1.register_forward_hook(lambda mod, i, o, name=name: keep(name + ".conv1", o))
            sub.norm2.register_forward_hook(lambda mod, i, o, name=name: keep(name + ".norm2", o))
            sub.register_forward_hook(lambda mod, i, o, name=name: keep(name, o))
            n += 1
    m.out_conv.register_forward_hook(lambda mod, i, o: keep("out_conv", o))
    return n


x, cls = _batch()
g = torch.Generator().manual_seed(5)
B = x.shape[0]
mask = (torch.rand(B, generator=g) < 0.25).float().to(dev())
eps = torch.randn(B, 3, 16, 16, generator=g).to(dev())
t = torch.randint(0, 1000, (B,), generator=g).to(dev())
args = (x, cls, mask, eps, t)

a, c = _make_step(bucket_mb=1), _make_step(bucket_mb=1)
hook_model(a.model)
hook_model(c.model)

DUMP = ref = []
la = a.loss_given_noise(*args)
a._backward(la, overlap=False)
torch.cuda.synchronize()
DUMP = None
print("eager pass kept", len(ref), "tensors; loss", float(la))

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
ops.FORCE_REPACK = True
with torch.cuda.stream(side):
    c._backward(c.loss_given_noise(*args), overlap=False)
    c.arena.grad.zero_()
    torch.cuda.synchronize()
g1 = ops.new_graph()
DUMP = got = []
with torch.cuda.graph(g1, stream=side):
    lc = c.loss_given_noise(*args)
    c._backward(lc, overlap=False)
DUMP = None
ops.FORCE_REPACK = False
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
assert [n for n, _ in got] == [n for n, _ in ref]

import ctypes  # noqa: E402
import numpy as np  # noqa: E402
from dm_hip import lib  # noqa: E402
HAVE_DBG = hasattr(lib, "dm_debug_read")


def read_dbg():
    out = np.zeros(128 * 32, dtype=np.float32)
    rc = lib.dm_debug_read(out.ctypes.data_as(ctypes.c_void_p), out.size)
    assert rc == 0, rc
    return out


if HAVE_DBG:
    lib.dm_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    g1.replay()
    torch.cuda.synchronize()
    dbg_ref = read_dbg()
    g1.replay()
    torch.cuda.synchronize()
    assert np.array_equal(dbg_ref, read_dbg())
    print("tile checksums recorded:", int((dbg_ref != 0).sum()), "non-zero entries")

bad_pairs = 0
first_names = {}
for i in range(pairs):
    g1.replay()
    if variant == "sync":
        torch.cuda.synchronize()
    g1.replay()
    torch.cuda.synchronize()
    diff = [(n, float((u.float() - v.float()).abs().max()), int((u != v).sum()), v.numel())
            for (n, u), (_, v) in zip(got, ref) if not torch.equal(u, v)]
    loss_ok = float(lc) == float(la)
    if HAVE_DBG:
        d = read_dbg()
        wrong = np.nonzero(d != dbg_ref)[0]
        if len(wrong) or not loss_ok:
            print(f"pair {i}: loss ok {loss_ok}; staged-tile checksums that differ (workgroup, slot<16: x row, 16+q: w rows 16q..): "
                  f"{[(int(w) // 32, int(w) % 32) for w in wrong[:12]]}")
    if diff or not loss_ok:
        bad_pairs += 1
        first_names[diff[0][0] if diff else "loss only"] = first_names.get(diff[0][0] if diff else "loss only", 0) + 1
        if bad_pairs <= int(os.environ.get('DIAG_SHOW', '2')):
            print(f"pair {i}: {[(r[0], r[2]) for r in diff[:4]]}")
            if diff and diff[0][0] == "emb.proj":
                u, v = dict(got)["emb.proj"], dict(ref)["emb.proj"]
                raw = dict(ref)["emb.raw"]
                M = raw.shape[0]
                widths = [b.emb_proj.weight.shape[0] for b in c.model._conv_blocks if b.emb_proj is not None]
                idx = (u != v).nonzero().flatten().tolist()
                off = 0
                bounds = []
                for pi, N in enumerate(widths):
                    bounds.append((off, off + M * N, pi, N))
                    off += M * N
                cells = {}
                for j in idx:
                    lo, hi, pi, N = next(bb for bb in bounds if bb[0] <= j < bb[1])
                    m, n = divmod(j - lo, N)
                    cells.setdefault((pi, N, m), []).append(n)
                blks = [b for b in c.model._conv_blocks if b.emb_proj is not None]
                for (pi, N, m), ns in cells.items():
                    lo = next(bb for bb in bounds if bb[2] == pi)[0]
                    W = blks[pi].emb_proj.weight.detach().double().cpu().numpy()[ns]          # [16][K]
                    d = np.array([float(u[lo + m * N + n]) - float(v[lo + m * N + n]) for n in ns])
                    xs_row = torch.nn.functional.silu(raw[m].double()).cpu().numpy()
                    fits = []
                    for k4 in range(W.shape[1] // 4):
                        A = W[:, k4 * 4:k4 * 4 + 4]
                        sol, *_ = np.linalg.lstsq(A, d, rcond=None)
                        fits.append((float(np.linalg.norm(A @ sol - d) / np.linalg.norm(d)), k4, sol))
                    fits.sort(key=lambda f: f[0])
                    res, k4, sol = fits[0]
                    print(f"      layer {pi} row {m}: best single 4-wide k block {k4}: relative residual {res:.2e} (next best {fits[1][0]:.2e}); "
                          f"x' - x = {np.round(sol, 5).tolist()}, x = {np.round(xs_row[k4 * 4:k4 * 4 + 4], 5).tolist()}, "
                          f"raw = {np.round(raw[m, k4 * 4:k4 * 4 + 4].cpu().numpy(), 5).tolist()}")
                print("   ", "; ".join(f"layer {pi} N={N} row {m} cols {min(ns)}-{max(ns)} ({len(ns)})" for (pi, N, m), ns in cells.items()))
print(f"variant {variant}: {bad_pairs} of {pairs} pairs differ; first differing tensor: {first_names}")
