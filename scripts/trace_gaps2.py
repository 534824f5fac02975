#!/usr/bin/env python3
"""Like trace_gaps.py, plus the launch properties rocprofv3 records (LDS, scratch, VGPRs, grid) of both
kernels of a pair and the quartiles of the gap: what distinguishes the pairs with a ~3.7 us gap from
those with none.  usage: trace_gaps2.py <kernel_trace.csv>"""
import collections
import csv
import sys


def short(n: str) -> str:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:34]


rows = []
with open(sys.argv[1]) as f:
    rd = csv.DictReader(f)
    print("columns:", rd.fieldnames)
    for r in rd:
        props = (r.get("LDS_Block_Size", "?"), r.get("Scratch_Size", "?"), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"),
                 r.get("SGPR_Count", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"),
                 r.get("Queue_Id", "?"))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), props))
rows.sort()
gaps = collections.defaultdict(list)
prop = {}
for (s0, e0, n0, p0), (s1, e1, n1, p1) in zip(rows, rows[1:]):
    gaps[(n0, n1)].append(s1 - e0)
    prop[n0], prop[n1] = p0, p1
q = lambda v, f: sorted(v)[min(len(v) - 1, int(len(v) * f))]
print("props = (LDS, scratch, VGPR, AGPR, SGPR, grid x, y, z, queue)")
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:40]:
    print(f"{a:34s} -> {b:34s} n={len(v):5d} gap q10/50/90 {q(v, .1) / 1e3:6.2f} {q(v, .5) / 1e3:6.2f} {q(v, .9) / 1e3:6.2f} us | {prop[a]} -> {prop[b]}")
