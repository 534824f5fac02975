#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace (CSV): median gap and
median duration per (previous kernel -> next kernel) pair, most frequent pairs first.
usage: trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import sys


def short(n: str) -> str:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0][:28]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[(n0, n1)].append((s1 - e0, e1 - s1))
med = lambda v: sorted(v)[len(v) // 2]
tot_gap = sum(max(0, g) for v in gaps.values() for g, _ in v if g < 50000)
print(f"kernels {len(rows)}, idle between consecutive kernels (gaps < 50 us) {tot_gap / 1e6:.2f} ms in total")
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:24]:
    print(f"{a:28s} -> {b:28s} n={len(v):5d} gap median {med([g for g, _ in v]) / 1e3:6.2f} us   next kernel runs {med([d for _, d in v]) / 1e3:6.2f} us")
