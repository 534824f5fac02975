#!/usr/bin/env python3
"""Average per-dispatch PMC counter values per kernel from rocprofv3 --pmc CSV output.
usage: pmc_summary.py <counter_collection.csv> [<more.csv> ...]  -> JSON on stdout"""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            c = acc[name][row["Counter_Name"]]
            c[0] += 1
            c[1] += float(row["Counter_Value"])
out = {}
for name, ctrs in acc.items():
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    out[short] = {c: {"dispatches": n, "avg": tot / n} for c, (n, tot) in ctrs.items()}
print(json.dumps(out, indent=1, sort_keys=True))
