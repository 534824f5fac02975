#!/bin/bash
# kernel trace of two strict bench steps -> gap analysis with launch properties (gpurun_out/gaps2.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-kernel-sweep --no-multi-stream"
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_kt -- python3 $R/bench.py $ARGS > $OUT/prof_kt.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $OUT/prof_kt.log; exit 1; }
python3 $R/scripts/trace_gaps2.py $(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1) > $OUT/gaps2.txt
tail -3 $OUT/prof_kt.log
