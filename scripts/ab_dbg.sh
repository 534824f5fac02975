#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for dbg in ${DBGS:-64}; do
  echo "== IC_DBG=$dbg"
  IC_DBG=$dbg IC_BWD_RPW=${RPW:-8} timeout -k 10 120 python -u scripts/kernel_sweep.py --R 512 --k 32 --S 8 --backward --rotate 4 --iters 12 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    for k,v in d['kernels'].items():
        if k in ('region_rkv_bwd','region_rkv','region_out_bwd','region_out','region_fin_bwd','kwta_rows'): print('  ',k, v.get('avg_us'), v.get('GBs'))
" || exit 1
done
