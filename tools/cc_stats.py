#!/usr/bin/env python3
"""chain-cache counters of one frame (diagnostic variant -DRT_CC_STATS through RT_AMD_LIB): tools/cc_stats.py [c2|c3]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import rt_amd as rt
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
nx, ny, ns = 1200, 800, 64
n, spl = (500, 0) if cfg == "c2" else (10000, 32)
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload() if spl else None
st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
L = rt.lib(); out = (C.c_ulonglong * 4)()
for k in range(2):
    rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O); torch.cuda.synchronize()
    L.rt_debug_cc(out, 1)
v = list(out)
print("%s: bounces of a wave's only ray %d, blocks fetched %d, bounces served from the cache %d, refused %d; kernel %.2f ms" % (cfg, v[3], v[0], v[1], v[2], W.render_times()[-1]))
