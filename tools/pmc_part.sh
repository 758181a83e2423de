#!/bin/bash
root=$GRAFT_REPO_ROOT
for np in 1 8; do
  d=/tmp/pp_$np; rm -rf $d
  cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $d -- python3 $root/tools/part_trace.py $np > /dev/null 2>&1
  python3 - $d $np <<'PY'
import sys, glob, csv, os
acc, n = {}, {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render<" not in r["Kernel_Name"]: continue
        c = r["Counter_Name"]; acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"]); n.setdefault(c, set()).add(r["Dispatch_Id"])
a = {c: acc[c] / len(n[c]) for c in acc}
print("part 0 of %s: VALU %.1f G, FETCH %.1f GB, wait_any share %.3f, busy cycles %.2f G" % (sys.argv[2], a["SQ_INSTS_VALU"] / 1e9, a["FETCH_SIZE"] * 1024 / 1e9, a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_BUSY_CYCLES"] / 1e9))
PY
done
