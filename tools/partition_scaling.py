#!/usr/bin/env python3
"""What the tile split can scale to, measured on ONE GPU: part 0 of N of the C5 frame (the tiles rank 0 of an N-GPU rt_multi_render
renders) against the whole frame — kernel time of the part x N / whole frame = the efficiency the split allows before any exchange
(a part has 1/N of the pixels: per-lane granularity and each GPU's tail weigh more).  GPU box only.
usage: partition_scaling.py [spp] [nparts,nparts,...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import rt_amd as rt
nx, ny, n, spl = int(os.environ.get('RT_NX', 3840)), 2160, 100000, 320      # RT_NX: another frame width (how the runs of a part stack from row to row depends on it)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
base = None
ALL = os.environ.get('RT_ALL_PARTS') == '1'      # every part of N (the slowest one counts), not only part 0
PARTS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4, 8]
for nparts in PARTS:
  worst = 0.0
  for pidx in (range(nparts) if ALL else [0]):
    part = rt.Partition(pidx, nparts)
    st = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
    ts = []
    for rep in range(3):
        rt.render_init(nx, ny, st, part)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rt.render(fb, nx, ny, spp, W, st, O, part)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    worst = max(worst, min(ts[1:]))
    if ALL: print("  part %d of %d: %8.2f ms" % (pidx, nparts, min(ts[1:])), flush=True)
  if True:
    t = worst
    base = base or t
    print("%s of %d: %8.2f ms  (whole / %d = %7.2f ms)  -> split efficiency %.3f, i.e. %.2fx at %d GPUs before the exchange" % (
        "slowest part" if ALL else "part 0", nparts, t, nparts, base / nparts, base / (nparts * t), base / t, nparts), flush=True)
