#!/bin/bash
# per-kernel averages of SQ / TCC counters over the FORWARD sweep at R=512,k=32 (S from env, default 64); separate passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="$R/scripts/kernel_sweep.py --R 512 --k 32 --S ${S:-64} --rotate 4 --iters 4 --no-events"
rm -rf /tmp/pf1 /tmp/pf2 /tmp/pf3
timeout -s KILL 100 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pf1 -- python3 $ARGS > /dev/null 2> $R/gpurun_out/pf1.err || { echo pass1 failed; tail -3 $R/gpurun_out/pf1.err; exit 1; }
echo pass1 done
timeout -s KILL 100 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY --output-format csv -d /tmp/pf3 -- python3 $ARGS > /dev/null 2> $R/gpurun_out/pf3.err || { echo pass3 failed; tail -3 $R/gpurun_out/pf3.err; exit 1; }
echo pass3 done
timeout -s KILL 100 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv -d /tmp/pf2 -- python3 $ARGS > /dev/null 2> $R/gpurun_out/pf2.err || { echo pass2 failed; tail -3 $R/gpurun_out/pf2.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pf1", "/tmp/pf3", "/tmp/pf2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not (k.startswith("k_region") or k.startswith("k_router")): continue
        c = acc[k][row["Counter_Name"]]; c[0] += 1; c[1] += float(row["Counter_Value"])
    for k in sorted(acc):
        print(k, {n: round(t / c, 1) for n, (c, t) in acc[k].items()})
PY
