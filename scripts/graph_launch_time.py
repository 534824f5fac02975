"""Host time of the steered loops' launch (graph replay vs call by call) and of the whole train_step's host side."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd import engine as E
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
toks = torch.randint(0, 255, (8, 256), device=dev)
for graph in (True, False):
    E.GRAPH = graph
    for _ in range(3):
        ic.train_step(m, opt, toks, ic.LossWeights(), dev)
    orig = m.engine.run_steered
    acc = []
    def timed(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); acc.append(time.perf_counter() - t0); return r
    m.engine.run_steered = timed
    orig_b = torch.Tensor.backward
    accb = []
    def tb(self, *a, **k):
        t0 = time.perf_counter(); r = orig_b(self, *a, **k); accb.append(time.perf_counter() - t0); return r
    torch.Tensor.backward = tb
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 10
    for _ in range(N):
        ic.train_step(m, opt, toks, ic.LossWeights(), dev)
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N
    torch.Tensor.backward = orig_b
    m.engine.run_steered = orig
    print(f"graph={graph}: train_step {tot*1e3:.2f} ms; host time inside run_steered (no sync) {1e3*sum(acc)/len(acc):.2f} ms; "
          f"host time inside backward() {1e3*sum(accb)/len(accb):.2f} ms")
