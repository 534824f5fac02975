"""Which part of a train_step launches how many device kernels / copies (torch.profiler, one step after warm-up):
runtime launch calls binned by the section of the step the host was in, and inside each section by the PyTorch
operator that issued them.  Host-side census, no timing claims.  usage: op_census.py [config] [--top N]"""
import os, sys, random, bisect
from collections import Counter, defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
import ironcortex_amd as ic
from ironcortex_amd import training as T
from bench import make_config
dev = torch.device("cuda:0")
argv = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--top"]
cfg = make_config(argv[0] if argv else "10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
toks = torch.randint(0, 255, (8, 256), device=dev)


def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def w(*a, **k):
        with record_function("SEC:" + name):
            return f(*a, **k)
    setattr(obj, attr, w)


wrap(T, "make_negatives", "make_negatives")
wrap(m, "sensor_vectors_batch", "mixer_fwd")
wrap(m.engine, "run_steered", "run_steered")
wrap(m.engine, "new_step", "new_step")
wrap(m.engine, "begin_batch", "begin_batch")
wrap(m, "_loop_inputs", "loop_inputs")
wrap(m.engine, "backward_batch", "backward_batch")
wrap(m._coord, "_end", "coord_end")
wrap(m._coord, "publish", "publish")
wrap(opt, "step", "opt.step")
wrap(opt, "zero_grad", "zero_grad")
wrap(m.engine, "prefetch_open", "prefetch_open")
ob = torch.Tensor.backward
def tb(self, *a, **k):
    with record_function("SEC:backward"):
        return ob(self, *a, **k)
torch.Tensor.backward = tb
for _ in range(4):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function("SEC:train_step"):
        ic.train_step(m, opt, toks, ic.LossWeights(), dev)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
secs = sorted([(e.time_range.start, e.time_range.end, e.name[4:]) for e in evs if e.name.startswith("SEC:")])
launches = [e for e in evs if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipGraphLaunch", "hipExtLaunchKernel", "hipModuleLaunchKernel")]
ops = sorted([(e.time_range.start, e.time_range.end, e.name) for e in evs if e.name.startswith("aten::") or e.name.endswith("Backward0") or "Fn" in e.name])


def section_of(t):
    best = None
    for s, e, n in secs:  # innermost section containing t
        if s <= t <= e and (best is None or s >= best[0]):
            best = (s, e, n)
    return best[2] if best else "?"


def op_of(t):
    best = None
    for s, e, n in ops:  # outermost aten op containing t
        if s <= t <= e and (best is None or s < best[0]):
            best = (s, e, n)
    return best[2] if best else "(library call)"


per = defaultdict(Counter)
for e in launches:
    t = e.time_range.start
    per[section_of(t)][op_of(t)] += 1
tot = sum(sum(c.values()) for c in per.values())
print(f"launch calls in one train_step: {tot}  (a captured graph counts as ONE hipGraphLaunch)")
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 8
for sec, c in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
    n = sum(c.values())
    lib = c.get("(library call)", 0)
    print(f"\n{sec:18s} {n:4d} launches ({lib} library)")
    for op, k in c.most_common(top):
        print(f"    {op[:56]:56s} {k:4d}")
