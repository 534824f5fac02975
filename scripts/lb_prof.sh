#!/bin/bash
# per-kernel durations of scripts/linear_bwd_bench.py (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lb
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/lb -- python3 $R/scripts/linear_bwd_bench.py > $R/gpurun_out/lb_prof.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/lb/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "linear_bwd" in n:
        d[(n.split("(")[0][-24:], r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    v.sort(); print(k, len(v), "median us", v[len(v)//2]/1e3)
PY
