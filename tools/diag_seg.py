"""Runs GPU tests back to back in one process, with a native call stack printed on SIGSEGV (tools/segv_bt.c):
    python tools/diag_seg.py <steps...>      steps: ddim ddpm euler (generation tests), h1 h2 h3 (harness tests), gc
DM_DESTROY_GRAPHS=1 python tools/diag_seg.py ddim ddpm euler h3 reproduces the hipGraphLaunch fault described in
profiles/r03_graph_destroy_segfault.md (HIP graphs destroyed when their Python objects die)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "diffusion-models_amd"))
import torch, ctypes
torch.zeros(1, device="cuda")
_bt = os.path.join(ROOT, "tools", "bin", "libsegv_bt.so")   # gcc -shared -fPIC -o tools/bin/libsegv_bt.so tools/segv_bt.c
if os.path.exists(_bt):
    ctypes.CDLL(_bt).dm_install_segv_bt()
from tests import test_gpu_harness as H
from tests import test_generate as G
kw = {"ddim": {"n_steps": 4}, "ddpm": {"n_steps": 5}, "euler": {"n_steps": 3}}
for s in sys.argv[1:]:
    print("run", s, flush=True)
    if s in kw:
        G.test_generate_graph_replay_equals_eager(s, kw[s])
    elif s == "h1":
        H.test_training_step_is_bitwise_reproducible()
    elif s == "h2":
        H.test_step_with_grouped_weight_gradients()
    elif s == "h3":
        H.test_step_in_phases_equals_step()
    elif s == "gc":
        import gc; gc.collect(); torch.cuda.synchronize()
print("all done", flush=True)
