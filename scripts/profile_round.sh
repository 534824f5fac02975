#!/bin/bash
# rocprofv3 kernel statistics and the two PMC passes (separate runs) of the bench command (strict mode);
# summaries -> gpurun_out/ (raw traces stay in /tmp: gpurun_out/ is copied back only up to 64 MiB)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( while true; do sleep 45; echo "[profile_round] alive $(date +%T)"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-kernel-sweep --no-multi-stream"  # (+ 3 unit-counting steps after the timed ones: 9 train_steps in the trace)
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python3 $R/bench.py $ARGS > $OUT/prof_ks.log 2>&1 || { echo "kernel-trace pass failed"; exit 1; }
cp $(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
PARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-kernel-sweep --no-multi-stream"
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -- python3 $R/bench.py $PARGS > $OUT/prof_f.log 2>&1 || { echo "FETCH_SIZE pass failed"; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -- python3 $R/bench.py $PARGS > $OUT/prof_w.log 2>&1 || { echo "WRITE_SIZE pass failed"; exit 1; }
python3 $R/scripts/pmc_traffic.py $(find /tmp/prof_f -name '*counter_collection.csv' | head -1) $(find /tmp/prof_w -name '*counter_collection.csv' | head -1) \
  "bench.py --config 10m (B=8,T=256, strict mode), 1 step after 1 warm-up, two separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE)" > $OUT/pmc_traffic.json
echo profile_round done
