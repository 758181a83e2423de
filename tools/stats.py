#!/usr/bin/env python3
"""Work counters of the render kernel (diagnostic build librt_amd_stats.so, -DRT_STATS).  GPU box only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import rt_amd as rt
rt.LIB_PATH = os.path.join(ROOT, "dd2360-raytracing_amd", os.environ.get("RT_STATS_LIB", "librt_amd_stats.so"))   # RT_STATS_LIB=librt_amd_wpass.so: wave passes
L = rt.lib()
L.rt_debug_stats.restype = C.c_int
L.rt_debug_stats.argtypes = [C.c_void_p, C.c_int]
names = ["rays", "fast", "slow", "tie", "cols", "tests", "discpos", "offers", "elig", "elig_nodes", "A_iters_wave", "B_rounds_wave",
         "loop_iters_wave", "A_lane_steps", "B_lanes", "waves", "live_ge56", "live_32_55", "live_8_31", "live_lt8", "switches"]
n, nx, ny, ns, spl = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (10000, 1200, 800, 8, 32)))
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload() if spl > 0 else None          # spl 0: the hitable_list path (through its grid)
st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
buf = (C.c_ulonglong * 64)()
rt.render_init(nx, ny, st); torch.cuda.synchronize()
L.rt_debug_stats(buf, 1)
rt.render(fb, nx, ny, ns, W, st, O); torch.cuda.synchronize()
L.rt_debug_stats(buf, 1)
raw = list(buf)
v = dict(zip(names, raw))
rays = max(1, v["rays"]); samples = nx * ny * ns; waves = max(1, v["waves"])
print("accel", O.accel_info() if O is not None else W.list_accel_info())
print("samples %d rays/sample %.3f fast %.4f slow %.5f ties %d" % (samples, rays / samples, v["fast"] / rays, v["slow"] / rays, v["tie"]))
for k in ("cols", "tests", "discpos", "offers", "elig", "elig_nodes", "A_lane_steps", "B_lanes"):
    print("  %-14s %8.3f per ray" % (k, v[k] / rays))
print("per wave: loop iters %.1f, A iters %.1f (%.1f per loop iter), B rounds %.1f (%.2f per loop iter)" %
      (v["loop_iters_wave"] / waves, v["A_iters_wave"] / waves, v["A_iters_wave"] / max(1, v["loop_iters_wave"]),
       v["B_rounds_wave"] / waves, v["B_rounds_wave"] / max(1, v["loop_iters_wave"])))
print("lane utilisation: phase A %.3f, phase B %.3f, rays per loop iter %.2f of 64" %
      (v["A_lane_steps"] / max(1, 64 * v["A_iters_wave"]), v["B_lanes"] / max(1, 64 * v["B_rounds_wave"]), rays / max(1, v["loop_iters_wave"])))
print("waves %d; loop iterations by live lanes: >=56: %.3f  32-55: %.3f  8-31: %.3f  <8: %.3f; pixel switches %d" % (
    waves, *[v[k] / max(1, v["loop_iters_wave"]) for k in ("live_ge56", "live_32_55", "live_8_31", "live_lt8")], v["switches"]))

import numpy as np
img = fb.cpu().numpy().reshape(ny, nx, 3)
it = img[:, :, 0]
print("per-pixel loop iterations: mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f" % (it.mean(), *np.percentile(it, [50, 90, 99, 99.9]), it.max()))
tend, tstart = img[:, :, 1].astype(np.float64), img[:, :, 2].astype(np.float64)
t0_ = tstart.min()
tend = ((tend - t0_) % 2**24) * 16 / 1e5; tstart = ((tstart - t0_) % 2**24) * 16 / 1e5          # ms since the first pixel started (the kernel stores 100 MHz ticks / 16)
T = tend.max()
print("pixel timing: frame ends at %.2f ms; pixels ending in the last 2 ms: %d, last 1 ms: %d" % (T, (tend > T - 2).sum(), (tend > T - 1).sum()))
late = tend > T - 1.5
if late.any():
    print("  those ending in the last 1.5 ms: started at ms p10 %.1f p50 %.1f p90 %.1f; iterations p10 %.0f p50 %.0f p90 %.0f; duration ms p50 %.1f" % (
        *np.percentile(tstart[late], [10, 50, 90]), *np.percentile(it[late], [10, 50, 90]), np.percentile((tend - tstart)[late], 50)))
for lo, hi in ((0, 200), (200, 400), (400, 800), (800, 1280), (1280, 4000)):
    m = (it >= lo) & (it < hi)
    if m.any():
        print("  pixels with %4d-%4d iterations: %7d, start p50 %.1f p90 %.1f max %.1f ms; end p50 %.1f p99 %.1f max %.1f ms; us/iteration p50 %.1f" % (
            lo, hi, m.sum(), *np.percentile(tstart[m], [50, 90, 100]), *np.percentile(tend[m], [50, 99, 100]), np.percentile(((tend - tstart)[m] / np.maximum(it[m], 1)) * 1e3, 50)))
if os.environ.get("RT_STATS_DUMP"):           # per-pixel iterations + the pilot's per-block counts, for tools/predictor.py
    nt = ((nx + 7) // 8) * ((ny + 7) // 8)
    pb = (C.c_int * (nt * 16))()
    L.rt_debug_pilot.restype = C.c_int; L.rt_debug_pilot.argtypes = [C.c_void_p, C.c_int]
    L.rt_debug_pilot(pb, nt * 16)
    np.savez_compressed(os.environ["RT_STATS_DUMP"], it=it.astype(np.uint16), pilot=np.array(pb, dtype=np.int16).reshape(nt, 16),
                        tstart=tstart.astype(np.float32), tend=tend.astype(np.float32))
rows = it.mean(axis=1)
print("row means of iterations (every 50 rows from bottom):", " ".join("%.0f" % rows[k] for k in range(0, ny, 50)))
worst = np.argsort(it.ravel())[-5:]
print("worst pixels (row, col, iters):", [(int(w // nx), int(w % nx), int(it.ravel()[w])) for w in worst])

wb = (C.c_ulonglong * (8192 * 4))()
L.rt_debug_waves.restype = C.c_int; L.rt_debug_waves.argtypes = [C.c_void_p]
L.rt_debug_waves(wb)
w = np.array(list(wb), dtype=np.float64).reshape(8192, 4)[:waves]
t = (w[:, 0] - w[:, 0].min()) / 100.0 / 1000.0     # ms relative to the first wave to end
print("wave end times (ms after the first wave ended): p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(t, [10, 50, 90, 99, 100])))
print("waves that went thin: %d; long pixels detected %d; thin iterations: mean %.0f max %.0f" % ((w[:, 2] > 0).sum(), w[:, 3].sum(), w[:, 2][w[:, 2] > 0].mean() if (w[:, 2] > 0).any() else 0, w[:, 2].max()))
late = np.argsort(t)[-8:]
print("last waves (end ms, loop iters, thin iters, long px):", [(round(t[k], 1), int(w[k, 1]), int(w[k, 2]), int(w[k, 3])) for k in late])

sp = raw[28:33]
if sp[4]:
    print("thin waves under load: %.1f k cycles per thin iteration (closest %.1f k); with <= 2 live lanes: %.1f k cycles per iteration (%d iterations)" % (
        sp[0] / sp[4] / 1e3, sp[1] / sp[4] / 1e3, sp[2] / max(1, sp[3]) / 1e3, sp[3]))
cyc = raw[21:28]
print("wave-cycles (summed over waves): total %.0fM, closest-hit %.1f%% (fast path %.1f%%, ground %.1f%%, scan %.1f%%), shade+loop %.1f%%" % (
    cyc[0] / 1e6, 100 * cyc[1] / cyc[0], 100 * cyc[2] / cyc[0], 100 * cyc[3] / cyc[0], 100 * cyc[4] / cyc[0], 100 * cyc[5] / cyc[0]))
print("  inside the fast path: large spheres + set-up %.1f%% of total wave-cycles" % (100 * raw[33] / cyc[0]))
print("  phase B (roots + offer) of the per-lane walk: %.1f%% of total wave-cycles" % (100 * raw[34] / cyc[0]))

if sp[3]:
    n12 = sp[3]
    print("iterations of thin waves with <= 2 live lanes: %.1f k cycles each = ground %.1f k + large spheres/set-up %.1f k + grid walk %.1f k + scan %.1f k + shade/loop %.1f k" % (
        sp[2] / n12 / 1e3, raw[59] / n12 / 1e3, raw[60] / n12 / 1e3, raw[61] / n12 / 1e3, raw[62] / n12 / 1e3, (sp[2] - raw[59] - raw[60] - raw[61] - raw[62]) / n12 / 1e3))
wp = ["ground", "large_k", "large_exact", "offer_node", "offer_raybox", "elig_fn", "elig_list", "setup", "A_col", "A_batch", "A_hold",
      "B_offer", "B_clip", "coop_chunk", "scan", "scatter", "sc_lambert", "sc_metal", "sc_dielectric", "rej_iter", "primary", "disk_iter", "sky", "end_pixel"]
li = max(1, v["loop_iters_wave"])
print("wave passes per main-loop iteration (a block executed by >= 1 lane of the wave):")
for k, name in enumerate(wp):
    print("  %-14s %8.3f" % (name, raw[35 + k] / li))
