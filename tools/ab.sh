#!/bin/bash
# A/B timing of kernel variants on the GPU box: tools/ab.sh [-c CONFIG] [-p] name...
# "base" = the product library, any other name = dd2360-raytracing_amd/variants/lib_<name>.so (built by hand with -D switches).
# -p also runs the GPU parity suite against the LAST variant.  Prints Msamples/s, ms per step, kernel ms.
cfg=c3; parity=0
while getopts "c:p" o; do case $o in c) cfg=$OPTARG;; p) parity=1;; esac; done
shift $((OPTIND - 1))
root=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
    if [ "$v" = base ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$root/dd2360-raytracing_amd/variants/lib_$v.so; fi
    out=$(timeout -k 10 180 python "$root/bench.py" --config "$cfg" --steps 6 --no-cpu-baseline 2>/dev/null) || { echo "$v FAILED"; exit 1; }
    echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %9.1f Msamples/s  %8.3f ms/step  kernel %8.3f ms' % ('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
if [ $parity = 1 ]; then timeout -k 10 400 python -m pytest "$root/tests/test_gpu_parity.py" -x -q 2>&1 | tail -3; fi
