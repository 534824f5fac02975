import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from bench import make_config
from torch.optim.optimizer import _default_to_fused_or_foreach
dev = torch.device("cuda:0")
cfg = make_config("10m")
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
ps = list(m.parameters())
print("n params", len(ps), "types", {type(p).__name__ for p in ps})
print("fused/foreach:", _default_to_fused_or_foreach(ps, False, False))
opt = torch.optim.AdamW(ps, lr=1e-3)
toks = torch.randint(0, 255, (8, 256), device=dev)
for i in range(3):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
import ironcortex_amd.training as T
# time sections of train_step
import cProfile, pstats, io
for _ in range(2):
    t0 = time.perf_counter(); opt.step(); torch.cuda.synchronize(); print("opt.step ms", (time.perf_counter() - t0) * 1e3)
n_grad = sum(p.grad is not None for p in ps)
print("params with grad", n_grad)
