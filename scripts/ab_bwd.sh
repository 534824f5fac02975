#!/bin/bash
# A/B of the backward variants at the sweep point (HIP events, rotating active sets)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for S in ${SLIST:-8 64}; do
  for m in 1 0; do
    echo "== S=$S IC_BWD_SPLIT=$m"
    IC_BWD_SPLIT=$m timeout -k 10 120 python -u scripts/kernel_sweep.py --R 512 --k 32 --S $S --backward --rotate 4 --iters 12 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d['kernels'].items():
        if 'bwd' in k: print('  ',k, v.get('avg_us'), v.get('GBs'))
" || exit 1
  done
done
