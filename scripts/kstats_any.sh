#!/bin/bash
# rocprofv3 kernel statistics of any python script of this repository:
#   kstats_any.sh <out-name> <script.py> <args...>  -> gpurun_out/<out-name>_kernel_stats.csv (+ top 16 printed)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
name=$1; script=$2; shift 2
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$name
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/$script "$@" > $OUT/${name}_prof.log 2>&1 || { tail -5 $OUT/${name}_prof.log; exit 1; }
cp $(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1) $OUT/${name}_kernel_stats.csv
head -17 $OUT/${name}_kernel_stats.csv | cut -c1-130
