"""The gradient buckets of a config as the data-parallel reducer cuts them (CPU only): per bucket its size and the
parameters it holds, in the order backward completes them. usage: python tools/ddp_buckets.py [config] [bucket MiB]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
from harness.ddp import cut_buckets
from harness.train_step import arena_layout, late_region
from modeling import get_model

name = sys.argv[1] if len(sys.argv) > 1 else "cos_cifar"
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "arg_json", name + ".json")))
model = get_model(cfg["model_type"], **cfg["model_cfg"])
params = [(n, p.numel()) for n, p in model.named_parameters()]
late = set(model.late_gradient_parameter_names())
offsets, total = arena_layout([n for n, _ in params], [k for _, k in params], late)
buckets = cut_buckets(offsets, total, (mb << 20) // 4, late_region([n for n, _ in params], offsets, late))
print(f"{name}: {len(params)} parameters, arena {total * 4 / 2**20:.1f} MiB, {len(buckets)} buckets of >= {mb} MiB")
for b, (lo, hi) in enumerate(buckets):
    inside = sorted(((o, n) for (n, _), o in zip(params, offsets) if lo <= o < hi), reverse=True)
    mods = []
    for _, n in inside:
        m = ".".join(n.split(".")[:3]) if n.startswith(("down", "up", "mid")) else n.split(".")[0]
        if not mods or mods[-1] != m:
            mods.append(m)
    n_late = sum(n in late for _, n in inside)
    print(f"bucket {b}: {(hi - lo) * 4 / 2**20:6.1f} MiB, {len(inside):3d} tensors ({n_late} of the embedding path): {mods[0]} ... {mods[-1]}")
