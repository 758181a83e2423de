#!/bin/bash
# tools/part_kernels.sh [variant]: per-kernel times (rocprofv3 kernel trace) of part 0 of 8 and of the whole C5 frame — what a rank's share of
# an 8-GPU frame spends outside k_render.  GPU box only.
root=$(cd "$(dirname "$0")/.." && pwd)
[ -n "$1" ] && export RT_AMD_LIB=$root/dd2360-raytracing_amd/variants/lib_$1.so
for np in 8 1; do
  d=$(mktemp -d /tmp/pk_XXXX)
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $root/tools/part_trace.py $np > $d/log 2>&1) || { tail -5 $d/log; exit 1; }
  python3 - $d $np <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
print("part 0 of %s:" % sys.argv[2])
for r in csv.DictReader(open(f)):
    print("  %-70s calls %4s avg %10.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
  rm -rf $d
done
