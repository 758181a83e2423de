#!/bin/bash
# Rehearsal of bench.py's N>1 path on a ONE-GPU box: 2 ranks, both on GPU 0, gloo instead of RCCL (test-only switches),
# and a bit-for-bit check of the assembled frame against a single-process render.  The real runs use nccl, one GPU per rank.
set -e
export RT_BENCH_SAME_GPU=1 RT_BENCH_BACKEND=gloo RT_BENCH_CHECK=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 "$@"
