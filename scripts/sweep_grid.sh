#!/bin/bash
# BASELINE configs[2]: the whole R x k x S forward grid (d = 256) with the CURRENT kernels, rocprofv3 kernel-trace,
# one process per R (points separated by marker launches).  Output: gpurun_out/<tag>_grid_R<R>.jsonl
# usage: sweep_grid.sh <tag> "<R list>" [extra kernel_sweep args]
TAG=$1; RL=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for RR in $RL; do
  P=/tmp/sg_${TAG}_$RR
  rm -rf $P; mkdir -p $P
  echo "$(date +%T) R=$RR" >> $OUT/${TAG}_grid_progress.log
  timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $P/kt -- python3 $R/scripts/kernel_sweep.py --R $RR --k 4 8 16 32 --S 1 8 64 --rotate 16 --iters 16 --no-events --markers "$@" > $P/alg.jsonl 2> $OUT/${TAG}_grid_R${RR}.err || { echo "R=$RR failed"; tail -3 $OUT/${TAG}_grid_R${RR}.err; exit 1; }
  KT=$(find $P/kt -name '*kernel_trace.csv' | head -1)
  (cd $R/scripts && python3 sweep_grid.py $P/alg.jsonl $KT) > $OUT/${TAG}_grid_R${RR}.jsonl
  echo "done R=$RR: $(wc -l < $OUT/${TAG}_grid_R${RR}.jsonl) points"
done
