#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one dir with <name>_counter_collection.csv files) per kernel."""
import csv, glob, os, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "ddp_solve"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k:
            continue
        agg[k.split("(")[0][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
