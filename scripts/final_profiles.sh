#!/bin/bash
# Round-end evidence in one call: the default bench line, the host/GPU clock timelines (strict, multi-stream), the
# rocprofv3 step timeline (loop chain per kernel), then scripts/profile_round.sh (kernel stats + the two PMC passes).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout -k 10 400 python bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err || { tail -5 $OUT/final_bench.err; exit 1; }
{ echo "# scripts/host_gpu_timeline.py (strict): host clock and HIP-event clock at the same markers of one train_step, best of 5"; timeout -k 10 200 python scripts/host_gpu_timeline.py 2>/dev/null | grep -v amdgpu.ids;
  echo; echo "# scripts/host_gpu_timeline.py --multi (multi-stream mode)"; timeout -k 10 200 python scripts/host_gpu_timeline.py --multi 2>/dev/null | grep -v amdgpu.ids; } > $OUT/final_host_gpu_timeline.txt || exit 1
timeout -k 10 300 bash scripts/timeline.sh > /dev/null 2>&1 || exit 1
cp $OUT/step_timeline.txt $OUT/final_step_timeline.txt
timeout -k 10 200 python scripts/op_census.py --top 6 2>/dev/null | grep -v "amdgpu.ids\|Warning\|_warn_once" > $OUT/final_op_census.txt || exit 1
timeout -k 10 900 bash scripts/profile_round.sh || exit 1
echo final_profiles done
