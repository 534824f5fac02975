#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-kernel-sweep --no-multi-stream > $OUT/prof_tl.log 2>&1 || { tail -5 $OUT/prof_tl.log; exit 1; }
python3 $R/scripts/step_timeline.py $(find /tmp/prof_tl -name '*kernel_trace.csv' | head -1) > $OUT/step_timeline.txt
cat $OUT/step_timeline.txt
grep -o '"ms_per_step": [0-9.]*' $OUT/prof_tl.log | head -1
