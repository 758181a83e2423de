#!/usr/bin/env python3
"""MI355X counterpart of the reference's analysis/run_experiment.sh + evaluations.ipynb sweep:
NUM_SPHERES in {488, 1000..9000} x SPHERE_RADIUS in {0.1, 0.2} x {hitable_list, octree}, 1200x800, ns = 10
(main.cu:348-350), 5 repetitions each (run_experiment.sh:30), render kernel time from HIP events (the reference reads
Nsight-Compute kernel durations).  Prints a markdown table with the octree speed-up and writes profiles/experiment_r1.json.
GPU box only."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import rt_amd as rt

NX, NY, NS, REPS = 1200, 800, 10, 5
SIZES = [488, 1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000]


def time_render(W, O):
    st = rt.alloc_rand_state(NX, NY); fb = rt.alloc_fb(NX, NY)
    ts = []
    for rep in range(REPS + 1):
        rt.render_init(NX, NY, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rt.render(fb, NX, NY, NS, W, st, O); e1.record()
        torch.cuda.synchronize()
        if rep:
            ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def spl_for(n, radius):
    # the reference adjusts SPHERES_PER_LEAF by hand "when NUM_SPHERES is changed" (acceleration_structure.h:15):
    # smallest multiple of 10 >= 30 with no dropped sphere
    spl = 30
    while True:
        W = rt.World(n, NX, NY, sphere_radius=radius)
        O = rt.Octree(W, spl)
        if O.info()["dropped_full"] == 0:
            return W, O, spl
        spl += 10


rows = []
for radius in (0.1, 0.2):
    for n in SIZES:
        W, O, spl = spl_for(n, radius)
        W.upload(); O.upload()
        W.set_list_traversal(rt.TRAVERSAL_REFERENCE)               # hitable_list::hit as written: every sphere, list order
        t_list = time_render(W, None)
        W.set_list_traversal(rt.TRAVERSAL_FAST)                    # the default: the list through the candidate grid
        t_grid = time_render(W, None)
        t_tree = time_render(W, O)
        rows.append(dict(radius=radius, n=n, spl=spl, list_ms=round(t_list, 3), list_grid_ms=round(t_grid, 3), octree_ms=round(t_tree, 3),
                         speedup=round(t_list / t_tree, 2)))
        print("r=%.1f N=%5d SPL=%3d  list scan %8.3f ms  list via grid %7.3f ms  octree %7.3f ms  octree vs scan %5.2fx  (%.0f / %.0f / %.0f Msamples/s)" % (
            radius, n, spl, t_list, t_grid, t_tree, t_list / t_tree, NX * NY * NS / t_list / 1e3, NX * NY * NS / t_grid / 1e3, NX * NY * NS / t_tree / 1e3), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "experiment_r1.json"), "w"), indent=1)
