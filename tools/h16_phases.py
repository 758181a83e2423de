#!/usr/bin/env python3
"""Where a C4 frame's wave-cycles go (diagnostic variant built with -DRT_H16_STATS, loaded through RT_AMD_LIB):
tools/mkvariant.sh h16stats -DRT_H16_STATS && RT_AMD_LIB=.../variants/lib_h16stats.so python tools/h16_phases.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import numpy as np, torch
import rt_amd as rt
nx, ny, ns, n, spl = 1200, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 10000, 32
W = rt.World(n, nx, ny, precision=rt.FP16); O = rt.Octree(W, spl)
st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny, precision=rt.FP16)
L = rt.lib(); out = (C.c_ulonglong * 8)()
for k in range(2):
    rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O); torch.cuda.synchronize()
    L.rt_debug_h16(out, 1)
v = list(out)
tot = v[3] or 1
print("walk %.1f %%  prefix+search %.1f %%  tests %.1f %%  rest (ground, shade, bookkeeping) %.1f %%   kernel %.2f ms" % (
    100.0 * v[0] / tot, 100.0 * v[1] / tot, 100.0 * v[2] / tot, 100.0 * (tot - v[0] - v[1] - v[2]) / tot, W.render_times()[-1]))
print("inside tests: waiting for the pair loads %.1f %%, push block %.1f %%, drains in the loop %.1f %% of the kernel's wave-cycles" % (100.0 * v[6] / tot, 100.0 * v[4] / tot, 100.0 * v[5] / tot))
if False: print("pair slots %.3g (of which wave iterations %.3g -> %.1f lanes busy), positive discriminants %.3g (%.1f %% of sphere slots), exact root evaluations %.3g (%.1f %% of positives)" % (
    v[4], v[7], v[4] / 4.0 / max(1, v[7]), v[5], 100.0 * v[5] / max(1, 2 * v[4]), v[6], 100.0 * v[6] / max(1, v[5])))
