#!/bin/bash
# forward region kernels at the sweep point, fused (IC_FWD_SPLIT=0) against split (=1) r/k/v path (HIP events)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for S in ${SLIST:-8 32 64}; do
  for sp in 0 1; do
    echo "== S=$S IC_FWD_SPLIT=$sp"
    IC_FWD_SPLIT=$sp timeout -k 10 120 python -u scripts/kernel_sweep.py --R 512 --k 32 --S $S --rotate 4 --iters 12 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d['kernels'].items():
        if k.startswith('region'): print('  ',k, v.get('avg_us'), v.get('frac'))
" || exit 1
  done
done
