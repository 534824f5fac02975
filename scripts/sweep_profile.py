#!/usr/bin/env python3
"""Join one kernel-sweep point's rocprofv3 tables with its algorithmic bytes.
usage: sweep_profile.py <alg.jsonl from kernel_sweep.py --no-events> <kernel_stats.csv> <FETCH counter_collection.csv>
       <WRITE counter_collection.csv>   -> JSON on stdout
Per library kernel: average duration (rocprofv3 --kernel-trace --stats), algorithmic bytes per launch
(bench.algorithmic_bytes x the per-launch unit counts of the point), achieved GB/s and fraction of the
8 TB/s HBM peak, HBM-side traffic per launch from the two PMC passes (gfx950: fetch bytes = 2 x
FETCH_SIZE x 1024, MI355X_MICROARCH.md HBM section) and traffic / algorithmic."""
import collections
import csv
import json
import os
import sys

HBM_PEAK_GBS = 8000.0
# kernel function -> the library's kernel id (ic_common.h IcKid)
ALIAS = {"region_rkv_routed": "region_rkv", "region_rkvG": "region_rkv", "router_fwd_ms": "router_fwd", "region_fin_chain": "region_fin",
         "router_comb_bwd_serial": "router_comb_bwd", "region_rkvT_bwd": "region_rkv_bwd", "router_matT_bwd": "router_mat_bwd",
         "region_chan_bwd": "region_mid_bwd", "router_combine_serial": "router_combine"}


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
    n = n[2:] if n.startswith("k_") else n
    return ALIAS.get(n, n)


def pmc_avg(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):  # a PMC pass that failed or was killed: the column stays empty
        return {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                c = acc[short(row["Kernel_Name"])]
                c[0] += 1
                c[1] += float(row["Counter_Value"])
    return {k: tot / n for k, (n, tot) in acc.items()}


def main():
    alg_path, ks_path, f_path, w_path = sys.argv[1:5]
    point = None
    for line in open(alg_path):
        line = line.strip()
        if line.startswith("{"):
            point = json.loads(line)
    alg = point.pop("alg_bytes")
    point.pop("kernels", None)
    dur = collections.defaultdict(lambda: [0, 0.0])
    with open(ks_path) as f:
        for row in csv.DictReader(f):
            k = short(row["Name"])
            dur[k][0] += int(row["Calls"])
            dur[k][1] += float(row["TotalDurationNs"])
    fetch, write = pmc_avg(f_path, "FETCH_SIZE"), pmc_avg(w_path, "WRITE_SIZE")
    kernels = {}
    for k, (calls, tot_ns) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        if k not in alg and not any(k.startswith(p) for p in ("region_", "router_", "wgrad_", "gate_")):
            continue
        us = tot_ns / calls / 1e3
        ab = float(alg.get(k, 0.0))
        traffic = (2 * fetch[k] + write[k]) * 1024 if k in fetch and k in write else None
        fetch_only = 2 * fetch[k] * 1024 if (k in fetch and k not in write) else None
        kernels[k] = {"calls": calls, "avg_us": round(us, 2), "alg_MB": round(ab / 1e6, 3),
                      "GBs": round(ab / (us * 1e-6) / 1e9, 1) if ab else None,
                      "frac_of_8TBs": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if ab else None,
                      "traffic_MB": round(traffic / 1e6, 3) if traffic is not None else None,
                      "traffic_over_alg": round(traffic / ab, 3) if (traffic is not None and ab) else None}
        if fetch_only is not None:  # the WRITE_SIZE pass is missing: reads only
            kernels[k]["fetch_MB"] = round(fetch_only / 1e6, 3)
    print(json.dumps({"point": point, "source": "rocprofv3 --kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE (three runs)",
                      "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
