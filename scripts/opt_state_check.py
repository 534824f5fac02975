"""Does the optimiser state persist across train_steps (parameters keep their identity)?"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
toks = torch.randint(0, 255, (8, 256), device=dev)
n_z = [0]
oz = torch.zeros_like
def zl(*a, **k):
    n_z[0] += 1
    return oz(*a, **k)
torch.zeros_like = zl
for i in range(4):
    n_z[0] = 0
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
    steps = sorted({int(v["step"]) for v in opt.state.values() if "step" in v})
    ids_ok = all(any(p is q for q in m.parameters()) for p in opt.state)
    print(f"step {i}: zeros_like calls {n_z[0]}, params with state {len(opt.state)}, step counters {steps}, state keys are model params: {ids_ok}")
