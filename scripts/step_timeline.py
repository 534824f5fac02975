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
