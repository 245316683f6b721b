#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (one dir per pass) per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections, os
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]                 # template instantiations are printed with their return type
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if not k.startswith("k_"): continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:36s} mean/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})")
