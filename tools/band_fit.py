#!/usr/bin/env python3
"""Calibration of rt_split_balanced's tile cost (GPU box only): the pilot's per-tile counts of the C5 frame (bounces, grid entries tested)
against the measured kernel times of contiguous bands of tiles — 8 and 16 bands of equal tile count, rendered one after the other on
ONE GPU.  Least squares T_band = a x bounces + b x tests + c; prints the fit, its residuals and integer weights for rt_tuning.h
(RT_SPLIT_WB / RT_SPLIT_WT).  usage: band_fit.py [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import numpy as np, torch
import rt_amd as rt
nx, ny, n, spl = 3840, 2160, 100000, 320
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
t0 = time.perf_counter()
starts, b, t, c = rt.split_balanced(W, O, nx, ny, 8, counts=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
starts2 = rt.split_balanced(W, O, nx, ny, 8)
torch.cuda.synchronize(); t2 = time.perf_counter()
print("rt_split_balanced: %.2f ms the first call, %.2f ms the second; starts %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, starts))
tiles = len(b)
print("pilot: bounces total %d, tests total %d, columns total %d" % (b.sum(), t.sum(), c.sum()))
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "c5_pilot_counts.npz"), bounces=b, tests=t, columns=c)
rows = []
for nb in (8, 12, 16, 24):
    per = tiles // nb
    for k in range(nb):
        lo, hi = k * per, (k + 1) * per if k < nb - 1 else tiles
        part = rt.Partition(0, 1, lo, hi)
        st = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
        ker = []
        for rep in range(3):
            rt.render_init(nx, ny, st, part); rt.render(fb, nx, ny, spp, W, st, O, part); torch.cuda.synchronize()
            ker.append(W.render_times()[-1])
        rows.append((lo, hi, float(b[lo:hi].sum()), float(t[lo:hi].sum()), min(ker[1:]), float(c[lo:hi].sum()), nb))
        print("  band [%6d, %6d): bounces %9d tests %11d columns %10d kernel %7.2f ms" % (lo, hi, rows[-1][2], rows[-1][3], rows[-1][5], rows[-1][4]), flush=True)
        del st, fb
R = np.array(rows)
A = np.stack([R[:, 2], R[:, 3], R[:, 5], np.ones(len(R))], axis=1)
for name, cols in (("bounces only", [0, 3]), ("bounces + tests", [0, 1, 3]), ("bounces + tests + columns", [0, 1, 2, 3]), ("bounces + columns", [0, 2, 3])):
    x, res, rk, sv = np.linalg.lstsq(A[:, cols], R[:, 4], rcond=None)
    pred = A[:, cols] @ x
    print("%s: coefficients %s; residuals ms: max |%.2f|, rms %.2f; relative: %s" % (name, x, np.abs(pred - R[:, 4]).max(), np.sqrt(((pred - R[:, 4]) ** 2).mean()),
          " ".join("%+.0f" % (100 * (p - m) / m) for p, m in zip(pred, R[:, 4]))))
# integer weights for rt_tuning.h from the three-term fit of the ground bands (a band of sky costs next to nothing either way)
g = R[:, 3] > 0
x, *_ = np.linalg.lstsq(A[g][:, [0, 1, 2, 3]], R[g, 4], rcond=None)
x = np.maximum(x[:3], 0.0)
s = 1000.0 / max(x[0], 1e-12) if x[0] > 0 else 1.0
print("WEIGHTS RT_SPLIT_WB=%d RT_SPLIT_WT=%d RT_SPLIT_WC=%d" % (round(x[0] * s), round(x[1] * s), round(x[2] * s)))
