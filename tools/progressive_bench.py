#!/usr/bin/env python3
"""render_progressive (main.cu:119-142: one sample per launch) over 64 passes of the C3 frame: direct calls vs a hipGraph
replay of one captured pass; the render kernel's own device time per pass (HIP events) and the scheduling counters of the last
pass (how many pixels the kept pilot schedule started as long chains).  GPU box only.
usage: progressive_bench.py [nosched]   (nosched: start the sequence at current_sample 2 — no pilot pass, no kept schedule)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import rt_amd as rt
nosched = len(sys.argv) > 1 and sys.argv[1] == "nosched"
nx, ny, n, spl, passes = 1200, 800, 10000, 32, 64
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)


def run(graph):
    rt.render_init(nx, ny, st)
    if not nosched:
        rt.render_progressive(fb, nx, ny, 1, W, st, O)
    rt.render_progressive(fb, nx, ny, 2, W, st, O)
    g = None
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            rt.render_progressive(fb, nx, ny, 3, W, st, O)
    torch.cuda.synchronize(); W.render_times(); t0 = time.perf_counter()
    for k in range(3, passes + 1):
        if g: g.replay()
        else: rt.render_progressive(fb, nx, ny, k, W, st, O)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3 / (passes - 2)
    kt = W.render_times()
    return dt, (sum(kt) / len(kt) if kt else float("nan"))


for rep in range(2):
    (a, ka), (b, _) = run(False), run(True)
print("render_progressive, 1200x800, N=10000 octree, %s (%s): %.3f ms per pass direct (kernel %.3f ms), %.3f ms per pass from a hipGraph (%.0f / %.0f Msamples/s); last pass: %s" % (
    rt.render_kernel_name(W, O, 1), "no kept schedule" if nosched else "schedule kept from pass 1", a, ka, b, nx * ny / a / 1e3, nx * ny / b / 1e3, W.render_counters()))
