#!/bin/bash
# tools/collect_profiles.sh TAG — on the GPU box: for every BASELINE config the bench line (python bench.py --config cN, the
# driver's form), the rocprofv3 kernel trace and the PMC passes (tools/profile_pmc.sh), gathered under gpurun_out/profiles_TAG/
# in the layout of profiles/rN/ (bench_cN.json, kernel_stats_cN.csv, pmc_summary_cN.txt), plus the progressive bench, the set-up
# times and the two-rank rehearsal of bench.py's N > 1 flow on this box's one GPU.
TAG=${1:-r3}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $ROOT
for c in c3 c2 c4 c5; do
  extra=""; [ $c != c3 ] && extra="--no-cpu-baseline"
  steps=10; [ $c = c5 ] && steps=3
  timeout -k 10 400 python3 bench.py --config $c --steps $steps --warmup 2 $extra > $OUT/bench_$c.json 2> $OUT/bench_$c.err || echo "bench $c failed"
  timeout -k 10 500 tools/profile_pmc.sh ${TAG}_$c --config $c > $OUT/profile_$c.log 2>&1 || echo "profile $c failed"
  cp gpurun_out/pmc_${TAG}_$c/summary.txt $OUT/pmc_summary_$c.txt 2>/dev/null
  f=$(find gpurun_out/pmc_${TAG}_$c/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_$c.csv
  rm -rf gpurun_out/pmc_${TAG}_$c
  echo "$c done"
done
timeout -k 10 120 python3 tools/progressive_bench.py 2>/dev/null | tail -1 > $OUT/progressive_c3.txt
timeout -k 10 120 python3 tools/setup_time.py > $OUT/setup_time.txt 2>&1
RT_BENCH_SAME_GPU=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_c5_2ranks_one_gpu_rehearsal.json 2> $OUT/bench_c5_2ranks.err || echo "2-rank rehearsal failed"
ls -la $OUT
