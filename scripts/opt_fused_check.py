"""torch.optim.AdamW fused=True vs the default (foreach) on the arena parameters: same model, same seeds,
parameters after 1 and after 3 train_steps."""
import copy, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0)
m0 = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
toks = torch.randint(0, 255, (4, 128), device=dev)
res = {}
for fused in (None, True):
    m = copy.deepcopy(m0)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=fused)
    snaps, losses = [], []
    for i in range(3):
        torch.manual_seed(77); random.seed(77 + i)
        out = ic.train_step(m, opt, toks, ic.LossWeights(), dev)
        losses.append(out["total"])
        snaps.append({n: p.detach().clone() for n, p in m.named_parameters()})
    res[fused] = (snaps, losses)
for i in range(3):
    a, b = res[None][0][i], res[True][0][i]
    worst = max(((a[n] - b[n]).abs().max().item() / max(a[n].abs().max().item(), 1e-12), n) for n in a)
    print(f"after step {i + 1}: worst relative parameter difference {worst[0]:.3e} at {worst[1]}; losses {res[None][1][i]:.6f} {res[True][1][i]:.6f}")
