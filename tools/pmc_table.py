#!/usr/bin/env python3
"""Pivot a rocprofv3 --pmc counter_collection.csv: per kernel, mean of each counter over its dispatches."""
import csv, glob, os, re, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_"
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
