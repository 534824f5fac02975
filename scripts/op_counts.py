"""aten op counts / device kernel counts of one train_step (torch.profiler), strict or multi."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
import ironcortex_amd as ic
from ironcortex_amd import training as T
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config("10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
if len(sys.argv) > 1:
    m.stream_mode = sys.argv[1]
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
toks = torch.randint(0, 255, (8, 256), device=dev)
for _ in range(3):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
# label the phases
orig_rlb, orig_bwd, orig_step = m.reasoning_loop_batch, torch.Tensor.backward, opt.step
def rlb(*a, **k):
    with record_function("PHASE_loops"):
        return orig_rlb(*a, **k)
def bwd(self, *a, **k):
    with record_function("PHASE_backward"):
        return orig_bwd(self, *a, **k)
def step(*a, **k):
    with record_function("PHASE_opt"):
        return orig_step(*a, **k)
m.reasoning_loop_batch = rlb; torch.Tensor.backward = bwd; opt.step = step
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function("PHASE_train_step"):
        ic.train_step(m, opt, toks, ic.LossWeights(), dev)
    torch.cuda.synchronize()
ev = prof.events()
phases = {e.name: (e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("PHASE_")}
print({k: round((v[1] - v[0]) / 1e3, 2) for k, v in phases.items()})
def phase_of(t):
    for k in ("PHASE_loops", "PHASE_backward", "PHASE_opt"):
        if k in phases and phases[k][0] <= t <= phases[k][1]:
            return k
    return "PHASE_other"
import collections
cnt = collections.defaultdict(lambda: collections.Counter())
for e in ev:
    if e.name.startswith("aten::") and e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::"):
        par = e.cpu_parent.name
        key = e.name if par.startswith("PHASE_") else e.name + " <- " + par[:40]
        cnt[phase_of(e.time_range.start)][key] += 1
for ph, c in cnt.items():
    print(ph, sum(c.values()))
    for k_, v_ in c.most_common(40):
        print("   ", v_, k_)
