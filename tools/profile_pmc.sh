#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box via gpurun).  Counters are collected in their own runs,
# never together with sys/hip/hsa tracing.  Usage: tools/profile_pmc.sh <tag> [bench args...]
set -e
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-pmc $@"
# kernel trace over 2 + 8 launches (the average is then the warm kernel time, as bench.py's HIP events measure it)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc $@ > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || echo "pass $i failed: $PMC"
  echo "pass $i done: $PMC"
done
python3 $ROOT/tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
