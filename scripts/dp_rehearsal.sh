#!/bin/bash
# 2-rank rehearsal of bench.py's data-parallel path on ONE GPU over gloo (RCCL needs a GPU per rank), variants by env.
# usage: dp_rehearsal.sh "VAR=val VAR2=val" ...   (one run per argument; prints ok / HUNG and the hang's Python frame)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
i=0
for envs in "$@"; do
  i=$((i+1))
  log=$OUT/dp_rehearsal_$i.log
  env $envs IC_HANG_DUMP=45 IC_DIST_BACKEND=gloo timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port $((29520+i)) bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-sweep --no-multi-stream $BENCH_ARGS > $log 2>&1
  if grep -q '"metric"' $log; then echo "[$envs] ok: $(grep -o '"ms_per_step": [0-9.]*' $log | head -1)"; else echo "[$envs] HUNG/FAILED:"; grep -A8 "most recent call first" $log | grep "File" | head -6; fi
done
