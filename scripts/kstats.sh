#!/bin/bash
# rocprofv3 kernel statistics of a bench.py run: kstats.sh <out-name> <bench args...>  -> gpurun_out/<out-name>_kernel_stats.csv (+ top 25 printed)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
name=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$name
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" > $OUT/${name}_prof.log 2>&1 || { tail -5 $OUT/${name}_prof.log; exit 1; }
cp $(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1) $OUT/${name}_kernel_stats.csv
head -26 $OUT/${name}_kernel_stats.csv | cut -c1-150
grep -o '"ms_per_step": [0-9.]*' $OUT/${name}_prof.log | head -1
