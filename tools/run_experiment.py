#!/usr/bin/env python3
"""MI355X counterpart of the reference's experiment harness (analysis/run_experiment.sh:26-57 + the evaluation notebook):
NUM_SPHERES in {488, 1000..9000} x SPHERE_RADIUS in {0.1, 0.2} x {hitable_list ("BASELINE"), octree}, 1200x800, ns = 10
(main.cu:348-350), FIVE runs per cell (run_experiment.sh:30), render kernel time from HIP events (the reference reads
Nsight-Compute kernel durations), and per (N, radius) Welch's t-test between the five baseline and the five octree durations
(evaluations.ipynb:1640-1651: scipy.stats.ttest_ind(..., equal_var=False)).  Every run's duration is kept.

  python tools/run_experiment.py [--sizes 488,1000,...] [--radii 0.1,0.2] [--out profiles/experiment_r2.json]

Prints one line per cell and writes the JSON (per-run data + statistics).  GPU box only.  The statistics live in
welch_t() / summarise() so that the CPU test suite can check them without a GPU."""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))

NX, NY, NS, REPS = 1200, 800, 10, 5
SIZES = [488, 1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000]


def welch_t(a, b):
    """Welch's unequal-variance t-test, two-sided (what scipy.stats.ttest_ind(a, b, equal_var=False) computes):
    returns (t, degrees of freedom by Welch-Satterthwaite, p)."""
    from scipy import stats
    na, nb = len(a), len(b)
    ma, mb = sum(a) / na, sum(b) / nb
    va = sum((x - ma) ** 2 for x in a) / (na - 1)
    vb = sum((x - mb) ** 2 for x in b) / (nb - 1)
    se2 = va / na + vb / nb
    if se2 == 0.0:
        return (0.0 if ma == mb else math.copysign(math.inf, ma - mb)), float(na + nb - 2), (1.0 if ma == mb else 0.0)
    t = (ma - mb) / math.sqrt(se2)
    df = se2 ** 2 / ((va / na) ** 2 / (na - 1) + (vb / nb) ** 2 / (nb - 1))
    return t, df, float(2.0 * stats.t.sf(abs(t), df))


def summarise(cell):
    """adds mean / speed-up / Welch statistics to one cell {list_runs_ms, octree_runs_ms, ...}"""
    a, b = cell["list_runs_ms"], cell["octree_runs_ms"]
    t, df, p = welch_t(a, b)
    cell.update(list_ms=round(sum(a) / len(a), 4), octree_ms=round(sum(b) / len(b), 4),
                speedup=round((sum(a) / len(a)) / (sum(b) / len(b)), 3),
                welch_t=round(t, 3) if math.isfinite(t) else str(t), welch_df=round(df, 2), welch_p=p, significant_5pct=bool(p < 0.05))
    return cell


def time_runs(rt, torch, W, O, reps=REPS):
    """`reps` timed runs after one warm-up; each run is render_init + render like the reference's timed region"""
    st = rt.alloc_rand_state(NX, NY)
    fb = rt.alloc_fb(NX, NY)
    ts = []
    for rep in range(reps + 1):
        rt.render_init(NX, NY, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rt.render(fb, NX, NY, NS, W, st, O)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            ts.append(round(e0.elapsed_time(e1), 4))
    return ts


def spl_for(rt, n, radius):
    # the reference adjusts SPHERES_PER_LEAF by hand "when NUM_SPHERES is changed" (acceleration_structure.h:15):
    # smallest multiple of 10 >= 30 with no dropped sphere
    spl = 30
    while True:
        W = rt.World(n, NX, NY, sphere_radius=radius)
        O = rt.Octree(W, spl)
        if O.info()["dropped_full"] == 0:
            return W, O, spl
        spl += 10


def run(sizes, radii, reps=REPS, verbose=True):
    import torch
    import rt_amd as rt
    cells = []
    for radius in radii:
        for n in sizes:
            W, O, spl = spl_for(rt, n, radius)
            W.upload(); O.upload()
            W.set_list_traversal(rt.TRAVERSAL_REFERENCE)           # hitable_list::hit as written: every sphere, list order
            t_list = time_runs(rt, torch, W, None, reps)
            W.set_list_traversal(rt.TRAVERSAL_FAST)                # the default: the list through the candidate grid
            t_grid = time_runs(rt, torch, W, None, reps)
            t_tree = time_runs(rt, torch, W, O, reps)
            c = summarise(dict(radius=radius, n=n, spl=spl, list_runs_ms=t_list, list_grid_runs_ms=t_grid, octree_runs_ms=t_tree))
            cells.append(c)
            if verbose:
                print("r=%.1f N=%5d SPL=%3d  list scan %8.3f ms  list via grid %7.3f ms  octree %7.3f ms  octree vs scan %6.2fx  Welch t = %s, df = %.1f, p = %.2e%s"
                      % (radius, n, spl, c["list_ms"], sum(t_grid) / len(t_grid), c["octree_ms"], c["speedup"], c["welch_t"], c["welch_df"], c["welch_p"],
                         "" if c["significant_5pct"] else "  (not significant)"), flush=True)
    return cells


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default=",".join(str(s) for s in SIZES))
    ap.add_argument("--radii", default="0.1,0.2")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "experiment_r2.json"))
    a = ap.parse_args()
    cells = run([int(x) for x in a.sizes.split(",")], [float(x) for x in a.radii.split(",")])
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"frame": [NX, NY, NS], "runs_per_cell": REPS, "protocol": "analysis/run_experiment.sh:26-57; Welch t-test as evaluations.ipynb:1640-1651",
               "cells": cells}, open(a.out, "w"), indent=1)
    print("wrote", a.out)
