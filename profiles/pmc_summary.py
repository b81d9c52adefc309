#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch.  usage: pmc_summary.py <dir>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()[-60:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if not any(s in k for s in ("gemm", "xpass", "prep", "fakequant")):
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
