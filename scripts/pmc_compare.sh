#!/bin/bash
# per-kernel averages of a few SQ / TCC counters over the bwd sweep at R=512,k=32,S=8 (two separate passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="$R/scripts/kernel_sweep.py --R 512 --k 32 --S ${S:-8} --rotate 4 --iters 8 --no-events --backward"
rm -rf /tmp/pc1 /tmp/pc2
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc1 -- python3 $ARGS > /dev/null 2> $R/gpurun_out/pc1.err || { echo pass1 failed; tail -3 $R/gpurun_out/pc1.err; exit 1; }
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv -d /tmp/pc2 -- python3 $ARGS > /dev/null 2> $R/gpurun_out/pc2.err || { echo pass2 failed; tail -3 $R/gpurun_out/pc2.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pc1", "/tmp/pc2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not k.startswith("k_region"): continue
        c = acc[k][row["Counter_Name"]]; c[0] += 1; c[1] += float(row["Counter_Value"])
    for k in sorted(acc):
        print(k, {n: round(t / c, 1) for n, (c, t) in acc[k].items()})
PY
