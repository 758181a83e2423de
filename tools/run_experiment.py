#!/usr/bin/env python3
"""MI355X counterpart of the reference's experiment harness (analysis/run_experiment.sh:26-57 + the evaluation notebook):
NUM_SPHERES in {488, 1000..9000} x SPHERE_RADIUS in {0.1, 0.2} x {hitable_list ("BASELINE"), octree}, 1200x800, ns = 10
(main.cu:348-350), FIVE runs per cell (run_experiment.sh:30), render kernel time from HIP events (the reference reads
Nsight-Compute kernel durations), and per (N, radius) Welch's t-test between the five baseline and the five octree durations
(evaluations.ipynb:1640-1651: scipy.stats.ttest_ind(..., equal_var=False)).  Every run's duration is kept.

  python tools/run_experiment.py [--sizes 488,1000,...] [--radii 0.1,0.2] [--out profiles/experiment_r2.json]

Per cell it also keeps what the reference's harness keeps beside the durations: the first run's image of either variant
(run_experiment.sh:45-49 keeps output.ppm; here a binary P6 under --image-dir plus the md5 of the ASCII P3 the reference would have
written), the notebook's image comparison of the pair (evaluations.ipynb:1021-1027: greyscale SSIM / PSNR of the PPM levels,
tools/image_metrics.py), the device memory the cell holds (run_experiment.sh:39-42 polls nvidia-smi; here hipMemGetInfo around the
cell), and with --pmc one rocprofv3 counter pass per variant (run_experiment.sh:34-35 runs every binary under ncu; the profiled
program — this script in --cell mode — stands directly behind `--`).

Prints one line per cell and writes the JSON (per-run data + statistics).  GPU box only.  The statistics live in
welch_t() / summarise() so that the CPU test suite can check them without a GPU."""
import argparse
import csv
import glob
import hashlib
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile

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


def time_runs(rt, torch, W, O, reps=REPS, keep=None):
    """`reps` timed runs after one warm-up; each run is render_init + render like the reference's timed region.
    keep: a dict that receives the first timed run's frame as a host array under "fb" """
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
        if rep == 1 and keep is not None:
            keep["fb"] = fb.cpu().numpy().reshape(NY, NX, 3).copy()
    return ts


PMC_COUNTERS = "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"


def pmc_pass(n, radius, spl, variant):
    """one rocprofv3 --pmc pass over a single render of this cell's variant (a child process: this script in --cell mode, the
    program directly behind `--`); returns {counter: value of the render kernel's dispatch} or {"error": ...}"""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    d = tempfile.mkdtemp(prefix="rt_exp_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc"] + PMC_COUNTERS.split() + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                                                       "--cell", "%d,%g,%d,%s" % (n, radius, spl, variant)]
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        if p.returncode != 0:
            return {"error": "rc %d" % p.returncode, "stderr_tail": p.stderr.decode(errors="replace")[-500:]}
        out = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_render" in row["Kernel_Name"] and "k_render_init" not in row["Kernel_Name"]:
                    out[row["Counter_Name"]] = out.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    out["kernel"] = row["Kernel_Name"].replace("rt::", "").replace("void ", "")
        if "SQ_INSTS_VALU" in out and out.get("SQ_WAVE_CYCLES"):
            out["valu_busy_share"] = round(4.0 * out.get("SQ_ACTIVE_INST_VALU", 0.0) / out["SQ_WAVE_CYCLES"], 4)
        return out or {"error": "no render dispatch in the counter file"}
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_cell(spec):
    """--cell N,radius,spl,variant: one untimed render of one variant (the process rocprofv3 profiles)"""
    import torch
    import rt_amd as rt
    n, radius, spl, variant = spec.split(",")
    n, radius, spl = int(n), float(radius), int(spl)
    W = rt.World(n, NX, NY, sphere_radius=radius).upload()
    O = rt.Octree(W, spl).upload() if variant == "octree" else None
    if variant == "list":
        W.set_list_traversal(rt.TRAVERSAL_REFERENCE)
    st = rt.alloc_rand_state(NX, NY)
    fb = rt.alloc_fb(NX, NY)
    rt.render_init(NX, NY, st)
    rt.render(fb, NX, NY, NS, W, st, O)
    torch.cuda.synchronize()


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


def run(sizes, radii, reps=REPS, verbose=True, image_dir=None, pmc=False):
    import torch
    import rt_amd as rt
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from image_metrics import compare_frames
    cells = []
    # the counter passes first, for every cell, while this process has not touched the GPU (World / Octree are host objects until
    # they are uploaded): the profiled children then have the device to themselves, as bench.py's have
    pmc_of = {}
    if pmc:
        for radius in radii:
            for n in sizes:
                _, _, spl = spl_for(rt, n, radius)
                pmc_of[(radius, n)] = {"list": pmc_pass(n, radius, spl, "list"), "octree": pmc_pass(n, radius, spl, "octree")}
    for radius in radii:
        for n in sizes:
            torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
            free0, _ = torch.cuda.mem_get_info()
            W, O, spl = spl_for(rt, n, radius)
            W.upload(); O.upload()
            free1, _ = torch.cuda.mem_get_info()
            k_list, k_tree = {}, {}
            W.set_list_traversal(rt.TRAVERSAL_REFERENCE)           # hitable_list::hit as written: every sphere, list order
            t_list = time_runs(rt, torch, W, None, reps, keep=k_list)
            W.set_list_traversal(rt.TRAVERSAL_FAST)                # the default: the list through the candidate grid
            t_grid = time_runs(rt, torch, W, None, reps)
            t_tree = time_runs(rt, torch, W, O, reps, keep=k_tree)
            c = summarise(dict(radius=radius, n=n, spl=spl, list_runs_ms=t_list, list_grid_runs_ms=t_grid, octree_runs_ms=t_tree))
            # the first run's image of either variant, and the notebook's comparison of the pair
            c["ppm_md5"] = {}
            for name, k in (("list", k_list), ("octree", k_tree)):
                host = rt.format_ppm(k["fb"], NX, NY)
                c["ppm_md5"][name] = hashlib.md5(host).hexdigest()
                if image_dir:
                    os.makedirs(image_dir, exist_ok=True)
                    rt.write_image(os.path.join(image_dir, "%s_N%d_r%g.ppm" % (name, n, radius)), k["fb"], NX, NY, fmt=rt.IMAGE_P6)
            c["image_list_vs_octree"] = compare_frames(k_list["fb"], k_tree["fb"])
            # what the cell holds on the device: scene + tree + grid (the library's allocations: hipMemGetInfo around their upload) and
            # the peak of the caller's buffers (RNG states, frame: the torch allocator's own peak — it caches freed blocks)
            c["device_memory_mb"] = round(max(0, free0 - free1) / 2.0 ** 20 + torch.cuda.max_memory_allocated() / 2.0 ** 20, 2)
            if pmc:
                c["pmc"] = pmc_of[(radius, n)]
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
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "experiment_r3.json"))
    ap.add_argument("--image-dir", default=None, help="keep the first run's image of either variant of every cell here (binary P6)")
    ap.add_argument("--pmc", action="store_true", help="one rocprofv3 counter pass per cell and variant")
    ap.add_argument("--cell", default=None, help="internal: N,radius,spl,variant — one render, for the profiler")
    a = ap.parse_args()
    if a.cell:
        run_cell(a.cell)
        sys.exit(0)
    cells = run([int(x) for x in a.sizes.split(",")], [float(x) for x in a.radii.split(",")], image_dir=a.image_dir, pmc=a.pmc)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump({"frame": [NX, NY, NS], "runs_per_cell": REPS, "protocol": "analysis/run_experiment.sh:26-57; Welch t-test as evaluations.ipynb:1640-1651; image comparison as evaluations.ipynb:1021-1027",
               "pmc_counters": PMC_COUNTERS if a.pmc else None,
               "cells": cells}, open(a.out, "w"), indent=1)
    print("wrote", a.out)
