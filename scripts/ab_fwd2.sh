#!/bin/bash
# forward kernels at the sweep point (HIP events)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for S in ${SLIST:-1 8 64}; do
  echo "== S=$S"
  timeout -k 10 120 python -u scripts/kernel_sweep.py --R 512 --k 32 --S $S --rotate 4 --iters 12 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d['kernels'].items():
        print('  ',k, v.get('avg_us'), v.get('frac'))
" || exit 1
done
