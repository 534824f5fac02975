"""Host time (no stream syncs) of the per-loop set-up pieces of LoopEngine.run, per loop."""
import sys, os, time, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd import engine as E, training as T
from ironcortex_amd.hip import ops
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config(sys.argv[1] if len(sys.argv) > 1 else "10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
toks = torch.randint(0, 255, (8, 256), device=dev)
acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] += time.perf_counter() - t0
        cnt[name] += 1
        return r
    return w
eng = m.engine
ops._region_args = timed("ops._region_args", ops._region_args)
ops._router_args = timed("ops._router_args", ops._router_args)
E.new = timed("new(struct)", E.new)
m.bank.params = timed("bank.params", m.bank.params)
m.router._params = timed("router._params", m.router._params)
m.router._edge_tables = timed("router._edge_tables", m.router._edge_tables)
m.router._cached_bias = timed("router._cached_bias", m.router._cached_bias)
m.bank.kwta_now = timed("bank.kwta_now", m.bank.kwta_now)
m.gate._tables = timed("gate._tables", m.gate._tables)
eng._heads_setup = timed("_heads_setup", eng._heads_setup)
eng._get_arena = timed("_get_arena", eng._get_arena)
eng._drive = timed("_drive", eng._drive)
eng.run = timed("run", eng.run)
eng.backward = timed("backward", eng.backward)
eng._finish_pass = timed("_finish_pass", eng._finish_pass)
eng._begin_pass = timed("_begin_pass", eng._begin_pass)
m.sensor_vector = timed("sensor_vector", m.sensor_vector)
m.reasoning_loop = timed("reasoning_loop", m.reasoning_loop)
m.reasoning_loop_batch = timed("reasoning_loop_batch", m.reasoning_loop_batch)
E.LoopFn.apply = timed("LoopFn.apply", E.LoopFn.apply)
for i in range(3):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
acc.clear(); cnt.clear()
N = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N
print(f"train_step {tot*1e3:.1f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v/N*1e3:8.3f} ms/step  {cnt[k]//N:5d} calls  {v/max(1,cnt[k])*1e6:8.1f} us/call")
