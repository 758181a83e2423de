#!/usr/bin/env python3
"""render_progressive (main.cu:119-142: one sample per launch) over 64 passes of the C3 frame: direct calls vs a hipGraph
replay of one captured pass.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import torch
import rt_amd as rt
nx, ny, n, spl, passes = 1200, 800, 10000, 32, 64
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)


def run(graph):
    rt.render_init(nx, ny, st)
    rt.render_progressive(fb, nx, ny, 1, W, st, O); rt.render_progressive(fb, nx, ny, 2, W, st, O)
    g = None
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            rt.render_progressive(fb, nx, ny, 3, W, st, O)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(3, passes + 1):
        if g: g.replay()
        else: rt.render_progressive(fb, nx, ny, k, W, st, O)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (passes - 2)


for rep in range(2):
    a, b = run(False), run(True)
print("render_progressive, 1200x800, N=10000 octree: %.3f ms per pass direct, %.3f ms per pass from a hipGraph (%.0f / %.0f Msamples/s)" % (
    a, b, nx * ny / a / 1e3, nx * ny / b / 1e3))
