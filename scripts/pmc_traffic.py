#!/usr/bin/env python3
"""HBM-side bytes per launch and kernel from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE).
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> "<workload text>"
Units/corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3
reports KB per dispatch; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes, so
fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact."""
import collections
import csv
import json
import sys


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    n = n[2:] if n.startswith("k_") else n
    return {"region_rkv_routed": "region_rkv"}.get(n, n)  # same kernel id in the library's timing table


def avg(path: str, counter: str):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            c = acc[short(row["Kernel_Name"])]
            c[0] += 1
            c[1] += float(row["Counter_Value"])
    return {k: (n, tot / n) for k, (n, tot) in acc.items()}


fetch, write = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
kernels = {}
for k in sorted(set(fetch) & set(write)):
    if k.startswith("at::") or k.startswith("Cijk") or k.startswith("__amd"):
        continue
    (n, f), (_, w) = fetch[k], write[k]
    kernels[k] = {"fetch_kb_raw": round(f, 1), "write_kb": round(w, 1), "dispatches": n,
                  "traffic_bytes_per_launch": int(round((2 * f + w) * 1024))}
print(json.dumps({
    "workload": sys.argv[3] if len(sys.argv) > 3 else "",
    "units": "rocprofv3 reports KB per dispatch; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE "
             "counts 128-B requests at 64 B, so fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact",
    "kernels": kernels}, indent=1))
