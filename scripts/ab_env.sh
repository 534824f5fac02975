#!/bin/bash
# A/B on ONE box of an environment toggle: bench.py as is (A) and with "$1" exported (B, e.g.
# IC_FUSE_STEP=0), interleaved twice; prints tokens/s and ms/step of each run.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in A B; do
    if [ $v = B ]; then export "$1"; else unset "${1%%=*}"; fi
    timeout -k 10 200 python3 -u $R/bench.py --no-cpu-baseline --no-kernel-sweep --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_ms_per_step']
print('$v', round(j['value']), round(j['ms_per_step'],2), {n:k[n] for n in list(k)[:9]})"
  done
done
