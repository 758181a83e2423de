#!/bin/bash
# tools/pmc_quick.sh CONFIG [variant]: SQ_INSTS_VALU / SALU / wave cycles of the render kernel for one config (one rocprofv3 --pmc pass)
cfg=$1; v=$2
root=$(cd "$(dirname "$0")/.." && pwd)
if [ -n "$v" ] && [ "$v" != base ]; then export RT_AMD_LIB=$root/dd2360-raytracing_amd/variants/lib_$v.so; fi
d=$(mktemp -d /tmp/pmcq_XXXX)
cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $d -- python3 $root/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2>&1
python3 - $d "$cfg" "$v" <<'PY'
import sys, glob, csv, os
acc, n = {}, {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_render" not in k or "init" in k: continue
        c = r["Counter_Name"]; acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"]); n.setdefault(c, set()).add(r["Dispatch_Id"])
print(sys.argv[2], sys.argv[3] or "base", {c: "%.3fG" % (acc[c] / len(n[c]) / 1e9) for c in sorted(acc)})
PY
rm -rf $d
