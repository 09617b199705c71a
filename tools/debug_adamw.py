import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diffusion-models_amd")]
import numpy as np, torch
from dm_hip import lib, check
rs = np.random.RandomState(21)
n = 100003
p0 = torch.from_numpy(rs.standard_normal(n).astype(np.float32)); g = torch.from_numpy((rs.standard_normal(n)*3).astype(np.float32))
ref = p0.clone().requires_grad_(True)
opt = torch.optim.AdamW([ref], lr=2e-4, betas=(0.9, 0.99))
d = torch.device("cuda:0")
p, m, v = p0.to(d), torch.zeros(n, device=d), torch.zeros(n, device=d)
partials = torch.empty(lib.dm_sumsq_partials(), device=d); norm = torch.empty(1, device=d)
st = torch.cuda.current_stream().cuda_stream
for step in range(1, 4):
    gs = g * (0.5 ** step)
    ref.grad = gs.clone()
    tn = torch.nn.utils.clip_grad_norm_([ref], 1.0)
    opt.step()
    gd = gs.to(d)
    check(lib.dm_grad_norm(gd.data_ptr(), n, partials.data_ptr(), norm.data_ptr(), st))
    check(lib.dm_adamw_ema_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), None, None, norm.data_ptr(), 1.0, n, 2e-4, 0.9, 0.99, 1e-8, 1e-2, 1 - 0.9 ** step, 1 - 0.99 ** step, 0.0, None, st))
    diff = (p.cpu() - ref.detach()).abs()
    st_ = opt.state[ref]
    print(step, "norm", float(norm), float(tn), "max p diff", diff.max().item(), "at", diff.argmax().item(),
          "m diff", (m.cpu() - st_["exp_avg"]).abs().max().item(), "v diff", (v.cpu() - st_["exp_avg_sq"]).abs().max().item(),
          "rel upd diff", ((p.cpu() - p0) - (ref.detach() - p0)).abs().max().item())
