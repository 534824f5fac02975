"""Host clock and GPU clock (HIP events) at the same markers of one train_step: where the GPU runs behind the host the
step is GPU-bound, where the two advance together the host is the limit.  No tracing tool involved."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd import engine as E, training as T
from bench import make_config
dev = torch.device("cuda:0")
MULTI = "--multi" in sys.argv
DP = "--dp" in sys.argv  # the data-parallel path on RCCL with a one-rank group (identity collectives, real calls)
dp = None
if "--pg" in sys.argv or "--pg-each" in sys.argv:  # the process group and its communicator exist, the step does not use them (A/B: what RCCL's mere presence costs)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29542")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    _t = torch.ones(4, device=dev); dist.all_reduce(_t); torch.cuda.synchronize()
if DP:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from ironcortex_amd.parallel import DataParallel
    dp = DataParallel()
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = make_config(argv[0] if argv else "10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=True)
if MULTI:
    m.stream_mode = "multi"
toks = torch.randint(0, 255, (8, 256), device=dev)
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    marks.append((name, time.perf_counter(), ev))
def wrap(obj, attr, name):
    f = getattr(obj, attr)
    def w(*a, **k):
        mark(name + ">"); r = f(*a, **k); mark(name + "<"); return r
    setattr(obj, attr, w)
wrap(T, "make_negatives", "make_negatives")
wrap(m, "sensor_vectors_batch", "mixer")
wrap(m.engine, "run_steered", "run_steered")
wrap(m.engine, "run_lockstep", "run_lockstep")
wrap(m.engine, "backward_batch", "backward_batch")
wrap(m._coord, "_end", "coord_end")
wrap(m._coord, "publish", "publish")
wrap(opt, "step", "opt.step")
if "--fine" in sys.argv:  # finer host-side markers inside the prologue
    wrap(m, "detach_state", "detach_state")
    wrap(m.engine, "new_step", "new_step")
    wrap(m.engine, "begin_batch", "begin_batch")
    wrap(m, "_loop_inputs", "loop_inputs")
    wrap(m.router, "_cached_bias", "cached_bias")
    wrap(m.engine, "_heads_setup", "heads_setup")
    wrap(m.engine, "_steer_bufs", "steer_bufs")
    wrap(m.router, "_edge_tables", "edge_tables")
    wrap(m.embed, "forward", "embed")
    wrap(m.local_mix, "forward", "local_mix")
    from ironcortex_amd.hip import lib
    L = lib()
    class _W:  # (ctypes functions cannot be wrapped by setattr on the CDLL for cached attribute lookups: patch the engine's view)
        pass

if dp is not None:
    if "--dp-sync" in sys.argv:  # experiment: the first collective of a step waits (on the host) for the loops to finish,
        # so nothing sits on the backend's queue while the loops run (DESIGN rule xiii)
        _mm = dp.mean_many_async
        def _mm_sync(ts):
            torch.cuda.current_stream().synchronize()
            return _mm(ts)
        dp.mean_many_async = _mm_sync
    for nm in ("mean", "any", "gather", "any_flags", "start_arena_reduce", "sync_grads"):
        wrap(dp, nm, "dp." + nm)
wrap(opt, "zero_grad", "zero_grad")
wrap(m, "finalize_grads", "finalize_grads")
ob = torch.Tensor.backward
def tb(self, *a, **k):
    mark("backward>"); r = ob(self, *a, **k); mark("backward<"); return r
torch.Tensor.backward = tb
if "--own-stream" in sys.argv:  # experiment: everything on a stream of the pool instead of the default (null) stream
    _own = torch.cuda.Stream(dev)
    torch.cuda.set_stream(_own)
for _ in range(4):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev, dp=dp)
torch.cuda.synchronize()
rows = []
for rep in range(5):
    marks.clear()
    if "--pg-each" in sys.argv:  # one collective between the steps, none inside
        dist.all_reduce(_t)
    torch.cuda.synchronize()
    mark("step>")
    ic.train_step(m, opt, toks, ic.LossWeights(), dev, dp=dp)
    mark("step<")
    torch.cuda.synchronize()
    t0, e0 = marks[0][1], marks[0][2]
    rows.append([(n, 1e3 * (t - t0), e0.elapsed_time(ev)) for n, t, ev in marks])
best = min(rows, key=lambda r: r[-1][1])
print(f"{'marker':22s} {'host ms':>9s} {'gpu ms':>9s} {'gpu behind':>10s}")
for n, h, g in best:
    print(f"{n:22s} {h:9.3f} {g:9.3f} {g - h:10.3f}")
