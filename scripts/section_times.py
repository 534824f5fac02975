"""Wall time of the sections of one train_step (stream drained at every boundary)."""
import sys, os, time, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ironcortex_amd as ic
from ironcortex_amd import engine as E, training as T
from bench import make_config
dev = torch.device("cuda:0")
cfg = make_config(sys.argv[1] if len(sys.argv) > 1 else "10m")
torch.manual_seed(0); random.seed(0)
m = ic.CortexReasoner(ic.hex_neighbors(cfg.R), ic.hex_axial_coords(cfg.R), {"sensor": 0, "motor": cfg.R - 1}, cfg).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
toks = torch.randint(0, 255, (8, 256), device=dev)
acc = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] += time.perf_counter() - t0
        return r
    return w
m.sensor_vector = timed("sensor_vector(mixer fwd)", m.sensor_vector)
m.engine._drive = timed("engine._drive(loop fwd)", m.engine._drive)
m.engine.backward = timed("engine.backward(loop bwd)", m.engine.backward)
m.engine.run = timed("engine.run(total)", m.engine.run)
m.engine.run_lockstep = timed("engine.run_lockstep(multi-stream fwd)", m.engine.run_lockstep)
m.engine.run_steered = timed("engine.run_steered(device-steered strict loops)", m.engine.run_steered)
m.engine.resolve_steered = timed("engine.resolve_steered", m.engine.resolve_steered)
if len(sys.argv) > 2:
    m.stream_mode = sys.argv[2]  # strict | multi
T.make_negatives = timed("make_negatives", T.make_negatives)
opt.step = timed("opt.step", opt.step)
orig_bwd = torch.Tensor.backward
torch.Tensor.backward = timed("total.backward", orig_bwd)
m.finalize_grads = timed("finalize_grads", m.finalize_grads)
m.engine.backward_batch = timed("engine.backward_batch", m.engine.backward_batch)
m._coord._end = timed("GradCoordinator._end (wgrad + per-pass reductions)", m._coord._end)
m.engine.begin_batch = timed("engine.begin_batch", m.engine.begin_batch)
m.engine.end_batch = timed("engine.end_batch", m.engine.end_batch)
m.sensor_vectors_batch = timed("sensor_vectors_batch(mixer, batched)", m.sensor_vectors_batch)
m.reasoning_loop_batch = timed("reasoning_loop_batch(total)", m.reasoning_loop_batch)
opt.zero_grad = timed("opt.zero_grad", opt.zero_grad)
m.detach_state = timed("detach_state", m.detach_state)
m.train = timed("model.train()", m.train)
from ironcortex_amd.hip import ops as _ops
_ops.hebbian_rank1 = timed("hebbian_rank1", _ops.hebbian_rank1)
T.ops.hebbian_rank1 = _ops.hebbian_rank1
m.lm_head.forward_rows = timed("lm_head.forward_rows", m.lm_head.forward_rows)
m.verify.forward = timed("verify.forward(x2)", m.verify.forward)
m.gate.update_homeo = timed("update_homeo", m.gate.update_homeo)
_fn = torch._foreach_norm
torch._foreach_norm = timed("foreach_norm", _fn)
for i in range(3):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
acc.clear()
N = 3
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
    ic.train_step(m, opt, toks, ic.LossWeights(), dev)
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / N
print(f"train_step {tot*1e3:.1f} ms (with section syncs)")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:32s} {v/N*1e3:8.2f} ms")
