#!/usr/bin/env python3
"""Per-pixel timing of ONE PART of the C5 frame (diagnostic build librt_amd_stats.so): when the pixels of each cost class start and
end, and when the waves end — what the tail of a rank's share of an N-GPU frame is made of.  usage: stats_part.py [nparts] [spp] [part]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import numpy as np, torch
import rt_amd as rt
rt.LIB_PATH = os.environ.get("RT_STATS_LIB", os.path.join(ROOT, "dd2360-raytracing_amd", "librt_amd_stats.so"))
L = rt.lib()
nparts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nx, ny, n, spl = 3840, 2160, 100000, 320
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
pidx = int(sys.argv[3]) if len(sys.argv) > 3 else 0
part = rt.Partition(pidx, nparts)
st = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
buf = (C.c_ulonglong * 64)()
rt.render_init(nx, ny, st, part); torch.cuda.synchronize(); L.rt_debug_stats(buf, 1)
rt.render(fb, nx, ny, spp, W, st, O, part); torch.cuda.synchronize()
L.rt_debug_stats(buf, 1); raw = list(buf)
img = fb.cpu().numpy().reshape(-1, 3)
it = img[:, 0]; ok = it > 0
tend, tstart = img[:, 1].astype(np.float64), img[:, 2].astype(np.float64)
raw_s, raw_e = tstart, tend                      # 24-bit tick / 16 stamps: they wrap every 2.68 s — take them relative to the longest
ref = raw_s[np.argmax(np.where(ok, it, 0))] - 6250.0      # pixel's start (long chains start with the kernel), minus 1 ms
tstart = ((raw_s - ref) % 2**24) * 16 / 1e5 - 1.0; tend = ((raw_e - ref) % 2**24) * 16 / 1e5 - 1.0
T = tend[ok].max()
print("part %d of %d, %d spp (instrumented): kernel %.1f ms; last pixel ends at %.1f ms; iterations per pixel mean %.0f p99 %.0f max %.0f" % (
    pidx, nparts, spp, W.render_times()[-1], T, it[ok].mean(), np.percentile(it[ok], 99), it[ok].max()))
for lo, hi in ((0, 400), (400, 800), (800, 1280), (1280, 2500), (2500, 100000)):
    m = ok & (it >= lo) & (it < hi)
    if m.any():
        print("  %5d-%6d iterations: %8d pixels, start p50 %6.1f p99 %6.1f max %6.1f | end p50 %6.1f p99 %6.1f max %6.1f | us/iteration p50 %5.1f" % (
            lo, hi, m.sum(), *np.percentile(tstart[m], [50, 99, 100]), *np.percentile(tend[m], [50, 99, 100]), np.percentile((tend - tstart)[m] / it[m] * 1e3, 50)))
for q in (0.5, 0.9, 0.99, 0.999):
    print("  %.1f %% of the pixels have ended by %.1f ms" % (100 * q, np.quantile(tend[ok], q)))
wb = (C.c_ulonglong * (8192 * 4))()
L.rt_debug_waves.restype = C.c_int; L.rt_debug_waves.argtypes = [C.c_void_p]
L.rt_debug_waves(wb)
w = np.array(list(wb), dtype=np.float64).reshape(8192, 4)[:4096]
t = (w[:, 0] - w[:, 0].min()) / 1e5
print("wave end times, ms after the first wave to end: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f; waves that went thin: %d" % (*np.percentile(t, [10, 50, 90, 99, 100]), (w[:, 2] > 0).sum()))
sp = raw[28:33]
if sp[3]:
    n12 = sp[3]
    print("iterations of thin waves with <= 2 live lanes (%d): %.1f k cycles each = ground %.1f k + large spheres/set-up %.1f k + grid walk %.1f k + scan %.1f k + shade/loop %.1f k" % (
        n12, sp[2] / n12 / 1e3, raw[59] / n12 / 1e3, raw[60] / n12 / 1e3, raw[61] / n12 / 1e3, raw[62] / n12 / 1e3, (sp[2] - raw[59] - raw[60] - raw[61] - raw[62]) / n12 / 1e3))
if sp[4]:
    print("thin waves: %.1f k cycles per thin iteration (closest %.1f k)" % (sp[0] / sp[4] / 1e3, sp[1] / sp[4] / 1e3))
# iterations in flight over time (every pixel's iterations spread evenly over its life): where the frame runs below its peak rate
nb = 24
edges = np.linspace(0.0, T, nb + 1)
ts_, te_, it_ = tstart[ok], np.maximum(tend[ok], tstart[ok] + 1e-6), it[ok]
rate = np.zeros(nb); lanes = np.zeros(nb)
for b in range(nb):
    ov = np.clip(np.minimum(te_, edges[b + 1]) - np.maximum(ts_, edges[b]), 0.0, None)
    rate[b] = (it_ * ov / (te_ - ts_)).sum(); lanes[b] = ov.sum() / (edges[b + 1] - edges[b])
print("per %.1f ms bin: iterations done, relative to the best bin:   " % (T / nb) + " ".join("%.2f" % (r / rate.max()) for r in rate))
print("per bin: pixels in flight (of %d lanes):                       " % (4096 * 64) + " ".join("%.2f" % (l / (4096 * 64)) for l in lanes))
top = np.argsort(it)[-12:][::-1]
print("longest pixels (iterations, start ms, end ms, us per iteration):", [(int(it[k]), round(float(tstart[k]), 1), round(float(tend[k]), 1), round(float((tend[k] - tstart[k]) / it[k] * 1e3), 1)) for k in top])
lastp = np.argsort(np.where(ok, tend, -1))[-12:][::-1]
print("last pixels to end (iterations, start ms, end ms, us per iteration):", [(int(it[k]), round(float(tstart[k]), 1), round(float(tend[k]), 1), round(float((tend[k] - tstart[k]) / it[k] * 1e3), 1)) for k in lastp])
late = np.argsort(t)[-8:]
print("last waves (end ms after the first, loop iterations, thin iterations, long pixels):", [(round(float(t[k]), 1), int(w[k, 1]), int(w[k, 2]), int(w[k, 3])) for k in late])
