#!/bin/bash
# Rehearsal of bench.py's N>1 flow on a ONE-GPU box: 2 ranks on GPU 0, rt_multi_render with the custom-gather form over gloo
# (RCCL refuses two ranks on one device).  The bench line carries single_gpu_same_frame.frame_equals_multi_gpu_frame.
# Usage (GPU box): tools/rehearse_2rank.sh [extra bench args]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
RT_BENCH_SAME_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 "$@"
