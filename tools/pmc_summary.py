#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel (optionally filtered by regex)."""
import csv, glob, re, sys, collections
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        if pat and not pat.search(row["Kernel_Name"]): continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
