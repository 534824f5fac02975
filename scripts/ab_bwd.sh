#!/bin/bash
# backward kernels at the sweep point with / without the harness's dirty lines (HIP events, rotating active sets)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for S in ${SLIST:-1 8 64}; do
  for fl in ${FLS:-"--flush-bwd"}; do
    echo "== S=$S $fl"
    timeout -k 10 120 python -u scripts/kernel_sweep.py --R 512 --k 32 --S $S --backward --rotate 4 --iters 12 $fl 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d['kernels'].items():
        if 'bwd' in k or 'wgrad' in k: print('  ',k, v.get('avg_us'), v.get('frac'))
" || exit 1
  done
done
