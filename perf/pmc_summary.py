#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: mean counter value per kernel name (+ per grid size)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "qpal"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if filt not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0][-70:], r.get("Grid_Size", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, ctrs in acc.items():
    print(key)
    for c, v in sorted(ctrs.items()):
        print(f"   {c:32s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
