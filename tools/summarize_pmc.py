#!/usr/bin/env python3
"""Sums rocprofv3 counter_collection CSVs per kernel and counter (per dispatch average)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        ndisp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in sorted(acc):
    if "render" not in k and "trace" not in k: continue
    print(k)
    for c in sorted(acc[k]):
        n = max(1, len(ndisp[(k, c)]))
        print("  %-32s %18.0f per dispatch (%d dispatches)" % (c, acc[k][c] / n, n))
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read())
