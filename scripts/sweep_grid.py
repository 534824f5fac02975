#!/usr/bin/env python3
"""Split ONE rocprofv3 kernel trace of `kernel_sweep.py --no-events --markers` (many sweep points in one process) into
its points by the marker launches, and price every library kernel of every point: average duration (rocprofv3
timestamps), algorithmic bytes (kernel_sweep's rows), GB/s and fraction of the 8 TB/s HBM peak.
usage: sweep_grid.py <alg.jsonl> <kernel_trace.csv>  -> one JSON line per point"""
import collections
import csv
import json
import sys

from sweep_profile import HBM_PEAK_GBS, short

MARK_BASE = 1000
points = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
rows = []
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])))
rows.sort()
cur, acc = None, collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for s, e, name, gx in rows:
    if "k_kwta_rows" in name:
        idx = gx // 256 - MARK_BASE
        cur = idx if 0 <= idx < 4000 else None
        continue
    if cur is None:
        continue
    k = short(name)
    if any(k.startswith(p) for p in ("region_", "router_", "wgrad_", "gate_")):
        a = acc[cur][k]
        a[0] += 1
        a[1] += e - s
for i, pt in enumerate(points):
    alg = pt.pop("alg_bytes", {})
    pt.pop("kernels", None)
    ks = {}
    for k, (n, ns) in sorted(acc.get(i, {}).items(), key=lambda kv: -kv[1][1]):
        us = ns / n / 1e3
        ab = alg.get(k, 0.0)
        ks[k] = {"calls": n, "avg_us": round(us, 2), "alg_MB": round(ab / 1e6, 3),
                 "GBs": round(ab / (us * 1e-6) / 1e9, 1) if ab else None,
                 "frac_of_8TBs": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 3) if ab else None}
    pt["kernels"] = ks
    pt["source"] = "rocprofv3 --kernel-trace timestamps, one process per R, points split by marker launches"
    print(json.dumps(pt))
