"""Host timeline of the reasoning loop's outer iterations: where the ~90 us per iteration go.
Wraps the library's entry points and Stream.synchronize with timestamps (perf_counter)."""
import sys, os, time, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd import hip
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config(sys.argv[1] if len(sys.argv) > 1 else "10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
toks = torch.randint(0, 255, (8, 256), device=dev)
L = hip.lib()
ev = []
on = [False]
def wrap(name):
    f = getattr(L, name)
    def w(*a):
        t0 = time.perf_counter()
        r = f(*a)
        if on[0]:
            ev.append((name, t0, time.perf_counter()))
        return r
    setattr(L, name, w)
for n in ("ic_loop_step_fwd", "ic_heads_gemv_fwd", "ic_heads_fwd", "ic_gemv_multi", "ic_gemv_chain", "ic_verify_score",
          "ic_steer_select", "ic_front_fwd", "ic_region_phase_fwd"):
    wrap(n)
S = torch.cuda.Stream
osync = S.synchronize
def sync(self):
    t0 = time.perf_counter()
    osync(self)
    if on[0]:
        ev.append(("sync", t0, time.perf_counter()))
S.synchronize = sync
for _ in range(3):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
on[0] = True
ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
on[0] = False
# iterations: sync -> ... -> next sync
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
for i, (n, t0, t1) in enumerate(ev):
    dur[n].append(t1 - t0)
    if i:
        gap[ev[i - 1][0] + " -> " + n].append(t0 - ev[i - 1][2])
print("call durations (host, us):")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:24s} n={len(v):5d} mean={1e6 * sum(v) / len(v):7.2f} total={1e3 * sum(v):7.3f} ms")
print("host time between calls (us):")
for n, v in sorted(gap.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 16:
        v2 = sorted(v)
        print(f"  {n:48s} n={len(v):5d} median={1e6 * v2[len(v2) // 2]:7.2f} mean={1e6 * sum(v) / len(v):7.2f} total={1e3 * sum(v):7.3f} ms")
big = sorted(gap["sync -> ic_loop_step_fwd"], reverse=True)[:17]
print("largest sync -> ic_loop_step_fwd gaps (loop boundaries, us):", [round(1e6 * g) for g in big])

# ---- second pass: where a loop boundary's host time goes (enter / exit stamps of the Python layers)
from ironcortex_amd import engine as E
marks = []
def stamp(cls, name):
    f = getattr(cls, name)
    def w(*a, **k):
        marks.append((name + ">", time.perf_counter()))
        r = f(*a, **k)
        marks.append((name + "<", time.perf_counter()))
        return r
    setattr(cls, name, w)
stamp(E.LoopEngine, "run"); stamp(E.LoopEngine, "_drive"); stamp(E.LoopEngine, "_heads_setup")
stamp(ic.CortexReasoner, "reasoning_loop"); stamp(ic.CortexReasoner, "detach_state")
ev.clear(); on[0] = True
ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize()
on[0] = False
allm = sorted(marks + [("sync<", t1) for n, t0, t1 in ev if n == "sync"] + [("step>", t0) for n, t0, t1 in ev if n == "ic_loop_step_fwd"],
              key=lambda x: x[1])
seg = collections.defaultdict(list)
for (a, ta), (b, tb) in zip(allm, allm[1:]):
    seg[a + " .. " + b].append(tb - ta)
print("segments between stamps (us): median, n")
for k, v in sorted(seg.items(), key=lambda kv: -sum(kv[1])):
    if "sync< .. step>" in k or len(v) < 8:
        continue
    v2 = sorted(v)
    print(f"  {k:44s} median={1e6 * v2[len(v2) // 2]:7.1f} n={len(v)}")
