#!/bin/bash
# rocprofv3 kernel statistics (and one PMC pass: MFMA busy cycles) of scripts/mixer_bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mx_ks /tmp/mx_pmc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mx_ks -- python3 $R/scripts/mixer_bench.py > $OUT/mixer_prof_run.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $OUT/mixer_prof_run.log; exit 1; }
KS=$(find /tmp/mx_ks -name '*kernel_stats.csv' | head -1)
cp $KS $OUT/mixer_kernel_stats.csv
head -12 $OUT/mixer_kernel_stats.csv
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/mx_pmc -- python3 $R/scripts/mixer_bench.py > /dev/null 2> $OUT/mixer_pmc.err || { echo "pmc pass failed"; tail -3 $OUT/mixer_pmc.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections, json, os
f = glob.glob("/tmp/mx_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if "attn" not in k: continue
    c = acc[k][row["Counter_Name"]]; c[0] += 1; c[1] += float(row["Counter_Value"])
out = {}
for k in sorted(acc):
    d = {n: t / c for n, (c, t) in acc[k].items()}
    # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum over the 1024 SIMDs
    d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8) if d.get("GRBM_GUI_ACTIVE") else None
    out[k] = d
    print(k, {n: round(v, 3) for n, v in d.items()})
json.dump(out, open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/mixer_pmc.json", "w"), indent=1)
PY
