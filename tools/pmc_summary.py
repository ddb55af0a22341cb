#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch and kernel from a counter_collection CSV (development aid).
usage: pmc_summary.py DIR [substring]"""
import csv
import glob
import json
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if want not in k:
            continue
        k = k.split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
out = {k: {"dispatches": len(disp[k]), **{c: v / len(disp[k]) for c, v in sorted(cs.items())}} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
