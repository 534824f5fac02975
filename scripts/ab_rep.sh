#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 0 1 3; do
  echo "== IC_SWEEP_REPEAT=$rep"
  IC_SWEEP_REPEAT=$rep timeout -k 10 120 python -u scripts/kernel_sweep.py --R 512 --k 32 --S 8 --rotate 4 --iters 12 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d['kernels'].items(): print('  ',k, v.get('avg_us'), v.get('GBs'))
" || exit 1
done
