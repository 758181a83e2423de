#!/usr/bin/env python3
"""Per-part kernel times of the C5 frame for a given split, on ONE GPU (GPU box only): every part of N rendered alone, its tiles, the
wall time of rt_render (pilot pass + selection + kernel) and the render kernel alone.  The library is RT_AMD_LIB (a variant whose
RT_PART_RUN_BUILD makes each part one contiguous band of tiles, or the product's runs of 64).  With "balanced" the parts are the bands of
rt_split_balanced (RT_SPLIT_WB / RT_SPLIT_WT in the environment override the weights).  usage: band_probe.py nparts [spp] [balanced]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import rt_amd as rt
nx, ny, n, spl = 3840, 2160, 100000, 320
nparts = int(sys.argv[1]); spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
balanced = len(sys.argv) > 3 and sys.argv[3] == "balanced"
print("lib %s, %d parts, %d spp%s" % (os.path.basename(rt.LIB_PATH), nparts, spp, ", balanced bands" if balanced else ""), flush=True)
parts = [rt.Partition(p, nparts) for p in range(nparts)]
if balanced:
    starts = rt.split_balanced(W, O, nx, ny, nparts)
    print("starts", starts, "tile rows", [s // 480 for s in starts], flush=True)
    parts = rt.split_parts(starts)
tot_k = 0.0
for pidx in range(nparts):
    part = parts[pidx]
    st = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
    wall, ker = [], []
    for rep in range(3):
        rt.render_init(nx, ny, st, part)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rt.render(fb, nx, ny, spp, W, st, O, part)
        torch.cuda.synchronize(); wall.append((time.perf_counter() - t0) * 1e3)
        ker.append(W.render_times()[-1])
    tot_k += min(ker[1:])
    print("  part %2d of %d: pixels %8d  wall %8.2f ms  kernel %8.2f ms  counters %s" % (pidx, nparts, fb.numel() // 3, min(wall[1:]), min(ker[1:]), W.render_counters()), flush=True)
    del st, fb
print("sum of the parts' kernels: %.2f ms" % tot_k)

