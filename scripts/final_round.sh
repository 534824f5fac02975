#!/bin/bash
# End-of-round evidence: GPU tests, the bench line (both configs), generation timings, rocprofv3 kernel statistics of
# the strict and the multi-stream step, PMC traffic passes.  Every GPU step under its own time limit; stops at the first failure.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $OUT/gpu_tests.log 2>&1 || { tail -20 $OUT/gpu_tests.log; exit 1; }
tail -2 $OUT/gpu_tests.log
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 500 python -u bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err || { tail -5 $OUT/bench_final.err; exit 1; }
tail -c 600 $OUT/bench_final.json; echo
timeout -k 10 400 python -u bench.py --config default --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
echo "default done"
timeout -k 10 300 python -u scripts/gen_bench.py > $OUT/gen_final.jsonl 2> $OUT/gen_final.err || { tail -5 $OUT/gen_final.err; exit 1; }
tail -2 $OUT/gen_final.jsonl
bash scripts/profile_round.sh || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_multi
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_multi -- python3 $R/scripts/multi_steps.py multi 6 > $OUT/prof_multi.log 2>&1 || { echo "multi kernel-trace pass failed"; exit 1; }
cp $(find /tmp/prof_multi -name '*kernel_stats.csv' | head -1) $OUT/multi_kernel_stats.csv
echo final_round done
