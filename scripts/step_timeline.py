#!/usr/bin/env python3
"""GPU timeline of one train_step from a rocprofv3 kernel trace: the step is cut at the first k_loop_steer of each
step (the graph's first kernel); prints, for the LAST complete step, the busy time and the idle gaps > 15 us with the
kernels on both sides.  usage: step_timeline.py <kernel_trace.csv>"""
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
# one cycle = from the first k_loop_steer of a step (the graph's first kernel) to the first one of the next step:
# loops -> losses / backward / optimizer of that step -> host prologue + token mixer of the next step
steer = [i for i, r in enumerate(rows) if "k_loop_steer" in r[2]]
firsts = [i for j, i in enumerate(steer) if j == 0 or i - steer[j - 1] > 200]
if len(firsts) < 6:
    sys.exit("fewer than 6 steps in the trace")
lo, hi = firsts[4], firsts[5]
step = rows[lo:hi]
t0, t1 = step[0][0], rows[hi][0]
busy = sum(e - s for s, e, _ in step)
last_steer = max(i for i, r in enumerate(step) if "k_loop_steer" in r[2])
print(f"cycle: {len(step)} kernels, {1e-6 * (t1 - t0):.3f} ms from graph start to the next step's graph start, GPU busy {1e-6 * busy:.3f} ms")
print(f"loops: {1e-6 * (step[last_steer][1] - t0):.3f} ms ({last_steer + 1} kernels); rest of the cycle: {1e-6 * (t1 - step[last_steer][1]):.3f} ms, "
      f"busy {1e-6 * sum(e - s for s, e, _ in step[last_steer + 1:]):.3f} ms in {len(step) - last_steer - 1} kernels")
print("idle gaps > 15 us in the rest of the cycle:")
prev = step[last_steer]
tot = 0
for cur in step[last_steer + 1:] + [rows[hi]]:
    g = cur[0] - prev[1]
    if g > 15000:
        tot += g
        print(f"  {1e-3 * g:8.1f} us at +{1e-6 * (prev[1] - t0):7.3f} ms  after {prev[2]}  before {cur[2]}")
    prev = cur
print(f"sum of those gaps: {1e-6 * tot:.3f} ms")

# ---- the loops' chain: per position inside one outer iteration (kernels between two k_loop_steer), mean duration and
# mean gap to the kernel before, over the iterations of this step that have the most common length
loops = step[:last_steer + 1]
cuts = [i for i, r in enumerate(loops) if "k_loop_steer" in r[2]]
its = [loops[a + 1:b + 1] for a, b in zip(cuts[:-1], cuts[1:])]
prevs = [loops[a] for a in cuts[:-1]]
from collections import Counter
L = Counter(len(it) for it in its).most_common(1)[0][0]
sel = [(p, it) for p, it in zip(prevs, its) if len(it) == L]
print(f"\nloop chain: {len(its)} iterations, {len(sel)} of {L} kernels; per position: mean duration, mean gap before")
tot_d = tot_g = 0.0
for pos in range(L):
    ds = [it[pos][1] - it[pos][0] for _, it in sel]
    gs = [it[pos][0] - (it[pos - 1][1] if pos else p[1]) for p, it in sel]
    md, mg = sum(ds) / len(ds) / 1e3, sum(gs) / len(gs) / 1e3
    tot_d += md
    tot_g += mg
    print(f"  {sel[0][1][pos][2]:44s} {md:7.2f} us  gap {mg:6.2f} us")
print(f"  per iteration: kernels {tot_d:.2f} us + gaps {tot_g:.2f} us = {tot_d + tot_g:.2f} us")

# ---- what the GPU runs after the loops (losses, backward, optimizer, the next step's prologue): summed per kernel name
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for s_, e_, n_ in step[last_steer + 1:]:
    agg[n_][0] += 1
    agg[n_][1] += e_ - s_
print("\nafter the loops, by kernel (launches, total us):")
for n_, (c_, t_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {n_:44s} {c_:5d} {t_ / 1e3:9.1f}")
