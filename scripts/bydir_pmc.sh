#!/bin/bash
# MFMA-busy counters of the by_direction kernels (separate PMC pass, no tracing): bydir_pmc.sh [active fraction, default 0.5]
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-0.5}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bd_pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/bd_pmc -- python3 $R/scripts/bydir_bench.py --no-events --S 64 --iters 6 --p $P > /dev/null 2> $R/gpurun_out/bydir_pmc.err || { echo "pmc pass failed"; tail -3 $R/gpurun_out/bydir_pmc.err; exit 1; }
python3 $R/scripts/pmc_summary.py $(find /tmp/bd_pmc -name '*counter_collection.csv' | head -1) > $R/gpurun_out/bydir_pmc_p$P.json
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/bydir_pmc_p$P.json"))
for k, v in d.items():
    if "mfma" in k:
        print(k[:80], {c: round(x["avg"]) for c, x in v.items()}, "n", v["GRBM_GUI_ACTIVE"]["dispatches"])
PY
