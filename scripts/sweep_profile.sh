#!/bin/bash
# Roofline evidence for the kernel sweep (BASELINE configs[2]) from rocprofv3 alone: per point
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel average duration
#   2. rocprofv3 --pmc FETCH_SIZE         -> read traffic per launch   (separate passes: the TCC block
#   3. rocprofv3 --pmc WRITE_SIZE         -> write traffic per launch   holds only one of them at a time)
# of `python3 scripts/kernel_sweep.py --no-events --rotate 16` (active sets rotate through all the regions, so
# no weight byte an iteration needs is left in the 256 MiB Infinity Cache by the previous ones), then
# scripts/sweep_profile.py joins them with the algorithmic bytes.  Output: gpurun_out/<tag>_*.
# usage: sweep_profile.sh <tag> <R> <k> "<S list>" [fwd|bwd|both]
TAG=$1; RR=$2; K=$3; SL=$4; MODE=${5:-both}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( while sleep 45; do echo "$(date +%T) alive" >> $OUT/${TAG}_progress.log; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
for S in $SL; do
  for DIR in fwd bwd; do
    if [ "$MODE" != both ] && [ "$MODE" != $DIR ]; then continue; fi
    B=""; FLUSH=""; [ $DIR = bwd ] && B="--backward" && FLUSH="${SWEEP_FLUSH---flush-bwd}"
    ARGS="$R/scripts/kernel_sweep.py --R $RR --k $K --S $S --rotate 16 --iters 32 --no-events $B $FLUSH"
    P=/tmp/sp_${TAG}_S${S}_${DIR}
    rm -rf $P; mkdir -p $P
    echo "$(date +%T) S=$S $DIR: kernel-trace pass" >> $OUT/${TAG}_progress.log
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $P/ks -- python3 $ARGS > $P/alg.jsonl 2> $OUT/${TAG}_S${S}_${DIR}_ks.err
    echo "$(date +%T) S=$S $DIR: FETCH_SIZE pass" >> $OUT/${TAG}_progress.log
    timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/f -- python3 $ARGS > /dev/null 2> $OUT/${TAG}_S${S}_${DIR}_f.err
    echo "$(date +%T) S=$S $DIR: WRITE_SIZE pass" >> $OUT/${TAG}_progress.log
    timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/w -- python3 $ARGS > /dev/null 2> $OUT/${TAG}_S${S}_${DIR}_w.err
    KS=$(find $P/ks -name '*kernel_stats.csv' | head -1)
    if [ -z "$KS" ]; then echo "S=$S $DIR: kernel-trace pass failed" >> $OUT/${TAG}_progress.log; continue; fi
    cp $KS $OUT/${TAG}_S${S}_${DIR}_kernel_stats.csv
    FC=$(find $P/f -name '*counter_collection.csv' 2>/dev/null | head -1); WC=$(find $P/w -name '*counter_collection.csv' 2>/dev/null | head -1)
    python3 $R/scripts/sweep_profile.py $P/alg.jsonl $KS "${FC:-none}" "${WC:-none}" > $OUT/${TAG}_S${S}_${DIR}.json
    echo "done S=$S $DIR"
  done
done
