#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for fl in 0 1; do
  echo "== IC_SWEEP_FLUSH=$fl"
  IC_SWEEP_FLUSH=$fl timeout -k 10 120 python -u scripts/kernel_sweep.py --R 512 --k 32 --S 8 --backward --rotate 4 --iters 12 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d['kernels'].items():
        print('  ',k, v.get('avg_us'), v.get('GBs'))
" || exit 1
done
