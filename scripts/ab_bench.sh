#!/bin/bash
# A/B on ONE box: bench.py with the in-tree library (A) and with another build of it (B, path in $1),
# interleaved twice; prints tokens/s and ms/step of each run.  Build B beforehand, e.g.
#   make -C ironcortex_amd/csrc && cp ironcortex_amd/libironcortex_hip.so /root/repo/gpurun_out/lib_b.so
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$1
for i in 1 2; do
  for v in A B; do
    if [ $v = B ]; then export IC_LIB_PATH=$B; else unset IC_LIB_PATH; fi
    timeout -k 10 200 python3 -u $R/bench.py --no-cpu-baseline --no-kernel-sweep --steps 30 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels_ms_per_step']
print('$v', round(j['value']), round(j['ms_per_step'],2), {n:k[n] for n in list(k)[:9]})"
  done
done
