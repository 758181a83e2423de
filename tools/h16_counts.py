"""Candidate and walk-loop counts of a C4 frame (counts build: tools/mkvariant.sh h16counts -DRT_H16_STATS -DRT_H16_COUNTS, loaded through RT_AMD_LIB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dd2360-raytracing_amd"))
import torch, rt_amd as rt
nx, ny, ns, n, spl = 1200, 800, 16, 10000, 32
W = rt.World(n, nx, ny, precision=rt.FP16); O = rt.Octree(W, spl)
st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny, precision=rt.FP16)
L = rt.lib(); out = (C.c_ulonglong * 8)()
for k in range(2):
    rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O); torch.cuda.synchronize()
    L.rt_debug_h16(out, 1)
v = list(out)
samples = nx * ny * ns
print("per sample: candidates evaluated %.2f, past the float filter (exact roots) %.2f (%.1f %%)" % (v[5] / samples, v[6] / samples, 100.0 * v[6] / max(1, v[5])))
if v[7]: print("walk loop: %.3g lane trips in %.3g wave trips -> %.1f of 64 lanes busy; %.1f lane trips per ray" % (v[4], v[7], v[4] / v[7], v[4] / (samples * 3.0)))
print("closest_tree: %.3g calls of a wave, %.3g rounds -> %.2f rounds (plane tables, walks) per call" % (v[2], v[1], v[1] / max(1, v[2])))
