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


# kernel function -> the library's kernel id (ic_common.h IcKid; bench.py reports by id).  k_front is the
# gate + router-edge launch: id router_edge_fwd with precision routing (the bench config), router_fwd without.
ALIAS = {"region_rkv_routed": "region_rkv", "region_rkvG": "region_rkv", "router_fwd_ms": "router_fwd", "region_fin_chain": "region_fin",
         "router_comb_bwd_serial": "router_comb_bwd", "region_rkvT_bwd": "region_rkv_bwd", "router_matT_bwd": "router_mat_bwd",
         "region_chan_bwd": "region_mid_bwd", "router_combine_serial": "router_combine",
         "gemv_multi": "gemv", "gemv_reg": "gemv", "gemv_chain1": "gemv", "heads": "token_rows",
         "combine_gemv": "logits_combine", "front": "router_edge_fwd", "gate_fused": "gate_select"}


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
    n = n[2:] if n.startswith("k_") else n
    return ALIAS.get(n, n)


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
