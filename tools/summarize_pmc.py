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
# derived figures for the render kernel (what bench.py's roofline_issue reports)
for k in sorted(acc):
    if "k_render" not in k or "k_render_init" in k: continue
    a = {c: acc[k][c] / max(1, len(ndisp[(k, c)])) for c in acc[k]}
    if a.get("SQ_INSTS_VALU"):
        print("derived for %s:" % k)
        if a.get("SQ_THREAD_CYCLES_VALU"): print("  lane utilisation (SQ_THREAD_CYCLES_VALU / 64 SQ_INSTS_VALU)  %.4f" % (a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_INSTS_VALU"])))
        if a.get("SQ_INSTS_SALU"): print("  SALU : VALU instructions                                      %.4f" % (a["SQ_INSTS_SALU"] / a["SQ_INSTS_VALU"]))
        if a.get("SQ_WAIT_ANY") and a.get("SQ_WAVE_CYCLES"): print("  SQ_WAIT_ANY / SQ_WAVE_CYCLES                                  %.4f" % (a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]))
        if a.get("SQ_WAIT_INST_ANY") and a.get("SQ_WAVE_CYCLES"): print("  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                             %.4f" % (a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
        if a.get("FETCH_SIZE") is not None and a.get("WRITE_SIZE") is not None: print("  FETCH_SIZE + WRITE_SIZE per launch (KB)                       %.0f + %.0f" % (a["FETCH_SIZE"], a["WRITE_SIZE"]))
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read())
