"""Host time of the pieces of LoopEngine.run_steered around the graph replay (perf_counter, best of 8 steps)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd import engine as E
from ironcortex_amd.hip import lib
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
toks = torch.randint(0, 255, (8, 256), device=dev)
L = lib()
T = {}
def timed(obj, attr, name):
    f = getattr(obj, attr)
    def w(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T.setdefault(name, []).append(time.perf_counter() - t0); return r
    setattr(obj, attr, w)
timed(L, "ic_graph_launch", "hipGraphLaunch (ic_graph_launch)")
timed(m.engine, "run_steered", "run_steered total")
timed(m.engine, "_heads_setup", "_heads_setup")
timed(m.router, "_cached_bias", "_cached_bias")
timed(m.router, "_params", "router._params")
timed(m.bank, "params", "bank.params")
timed(E.LoopBatchFn, "apply", "LoopBatchFn.apply")
for _ in range(4):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
T.clear()
for _ in range(8):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
for k, v in T.items():
    n = len(v) // 8
    print(f"{k:36s} calls/step {n}  best total per step {1e3 * min(sum(v[i * n:(i + 1) * n]) for i in range(8)):.3f} ms")
