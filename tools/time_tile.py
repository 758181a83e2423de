#!/usr/bin/env python3
"""time one 8x8 tile alone (critical-path probe): tools/time_tile.py row col"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import ctypes as C
import rt_amd as rt
if os.environ.get("RT_STATS"):
    rt.LIB_PATH = os.path.join(ROOT, "dd2360-raytracing_amd", "librt_amd_stats.so")
    L = rt.lib(); L.rt_debug_stats.restype = C.c_int; L.rt_debug_stats.argtypes = [C.c_void_p, C.c_int]
nx, ny, ns, n, spl = 1200, 800, 64, 10000, 32
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
tiles_x = (nx + 7) // 8
for row, col in [(317, 802), (100, 600), (700, 600)]:
    tile = (row // 8) * tiles_x + col // 8
    part = rt.Partition(tile, tiles_x * ((ny + 7) // 8))
    st = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
    for rep in range(3 if not os.environ.get('RT_STATS') else 1):
        rt.render_init(nx, ny, st, part); torch.cuda.synchronize()
        t0 = time.perf_counter(); rt.render(fb, nx, ny, ns, W, st, O, part); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("tile of pixel (%d,%d): %.2f ms alone" % (row, col, dt * 1e3))
    if os.environ.get("RT_STATS"):
        buf = (C.c_ulonglong * 64)(); L.rt_debug_stats(buf, 1)
        v = list(buf)
        names = ["total", "closest", "fastpath(large+setup+walk)", "ground test", "scan", "shade+loop", "realtime(100MHz)"]
        cyc = v[21:28]
        print("   cycles: " + ", ".join("%s %.2fM" % (n, c / 1e6) for n, c in zip(names, cyc)), " -> clock %.2f GHz" % (cyc[0] / max(1, cyc[6]) * 0.1))
        print("   loop iters %d, A iters %d, B rounds %d, rays %d, cols %d tests %d held %d elig %d elig_nodes %d" % (v[12], v[10], v[11], v[0], v[4], v[5], v[6], v[8], v[9]))

