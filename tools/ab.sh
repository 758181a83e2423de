#!/bin/bash
# A/B timing of kernel variants on the GPU box: tools/ab.sh [-c CONFIG] [-r ROUNDS] [-p] name...
# "base" = the product library, any other name = dd2360-raytracing_amd/variants/lib_<name>.so (built by hand with -D switches).
# The variants are timed in ROUNDS interleaved passes (clock / thermal drift hits all alike); prints per variant the
# median and minimum kernel time and the median step time.  -p also runs the GPU parity suite against the LAST variant.
cfg=c3; parity=0; rounds=3
while getopts "c:pr:" o; do case $o in c) cfg=$OPTARG;; p) parity=1;; r) rounds=$OPTARG;; esac; done
shift $((OPTIND - 1))
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp)
for ((k = 0; k < rounds; k++)); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$root/dd2360-raytracing_amd/variants/lib_$v.so; fi
    out=$(timeout -k 10 180 python "$root/bench.py" --config "$cfg" --steps 8 --no-cpu-baseline --no-pmc 2>/dev/null) || { echo "$v FAILED"; exit 1; }
    echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $tmp
  done
done
python - $tmp "$@" <<'PY'
import sys, statistics as st
rows = [l.split() for l in open(sys.argv[1])]
for v in dict.fromkeys(sys.argv[2:]):
    step = [float(r[1]) for r in rows if r[0] == v]; ker = [float(r[2]) for r in rows if r[0] == v]
    print("%-10s kernel median %7.3f min %7.3f ms   step median %7.3f ms   (%d runs)" % (v, st.median(ker), min(ker), st.median(step), len(ker)))
PY
rm -f $tmp
if [ $parity = 1 ]; then timeout -k 10 400 python -m pytest "$root/tests/test_gpu_parity.py" -x -q 2>&1 | tail -3; fi
