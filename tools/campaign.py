#!/usr/bin/env python3
"""Exactness campaign (GPU box): the fast traversal against the exact reference scan on many random worlds and rays,
and on full frames of several scene families.  Any mismatch is printed and the exit code is 1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rt_amd as rt
from test_gpu_parity import random_rays, random_world

bad = 0
t0 = time.time()
total_rays = 0
list_worlds = 0
N_WORLDS = int(sys.argv[1]) if len(sys.argv) > 1 else 24          # tools/campaign.py 240: ten times the default campaign
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0               # tools/campaign.py 2400 2400: the next 2400 worlds
for seed in range(FIRST, FIRST + N_WORLDS):
    n = [300, 1500, 6000, 20000][seed % 4]
    sp, cam = random_world(rt, 1000 + seed, n, 96, 64, big=3 + seed % 5, air=0.1 + 0.1 * (seed % 4))
    W = rt.World(n, 96, 64, spheres=sp, camera=cam); O = rt.Octree(W, 30 + 10 * (seed % 6))
    nr = 1_000_000
    rays = random_rays(nr, 7000 + seed)
    rng = np.random.default_rng(seed)
    k = nr // 3                                           # secondary-ray like: origins on / near sphere surfaces
    pick = rng.integers(1, n, k)
    dirs = rng.normal(size=(k, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    rays[:k, 0:3] = sp["center"][pick] + dirs * (sp["radius"][pick][:, None] * rng.uniform(0.99, 1.01, (k, 1)))
    rays[:k, 3:6] = rng.normal(size=(k, 3))
    d = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).cuda()
    outs = []
    for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_FAST):
        O.set_traversal(mode)
        o = torch.zeros(nr * 32, dtype=torch.uint8, device="cuda")
        rt.trace_rays(W, O, d, nr, o); torch.cuda.synchronize()
        outs.append(o.cpu().numpy().view(np.uint32).reshape(nr, 8))
    diff = np.nonzero((outs[0] != outs[1]).any(axis=1))[0]
    total_rays += nr
    if diff.size:
        bad += diff.size
        print("world seed %d n=%d: %d differing rays, e.g. %s" % (seed, n, diff.size, rays[diff[0]]))
    # hitable_list::hit: every sphere in list order against the candidate grid (where the world has one)
    if n <= 6000 and W.list_accel_info()["enabled"]:
        outs = []
        for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_FAST):
            W.set_list_traversal(mode)
            o = torch.zeros(nr * 32, dtype=torch.uint8, device="cuda")
            rt.trace_rays(W, None, d, nr, o); torch.cuda.synchronize()
            outs.append(o.cpu().numpy().view(np.uint32).reshape(nr, 8))
        diff = np.nonzero((outs[0] != outs[1]).any(axis=1))[0]
        total_rays += nr; list_worlds += 1
        if diff.size:
            bad += diff.size
            print("LIST world seed %d n=%d: %d differing rays, e.g. %s" % (seed, n, diff.size, rays[diff[0]]))
print("list path checked on %d of the worlds" % list_worlds)
print("random worlds: %d rays, %d mismatches, %.0f s" % (total_rays, bad, time.time() - t0), flush=True)

def frame(n, radius, spl, nx, ny, ns):
    global bad
    W = rt.World(n, nx, ny, sphere_radius=radius); O = rt.Octree(W, spl)
    res = []
    for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_FAST):
        O.set_traversal(mode)
        st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
        rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O); torch.cuda.synchronize()
        res.append((fb, st))
    same = torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[0][1], res[1][1])
    print("frame N=%d r=%.2f SPL=%d %dx%dx%d: %s" % (n, radius, spl, nx, ny, ns, "identical" if same else "DIFFERENT"), flush=True)
    if not same:
        bad += 1

def list_frame(n, radius, nx, ny, ns):
    global bad
    W = rt.World(n, nx, ny, sphere_radius=radius)
    res = []
    for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_FAST):
        W.set_list_traversal(mode)
        st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
        rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, None); torch.cuda.synchronize()
        res.append((fb, st))
    same = torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[0][1], res[1][1])
    print("list frame N=%d r=%.2f %dx%dx%d: %s" % (n, radius, nx, ny, ns, "identical" if same else "DIFFERENT"), flush=True)
    if not same:
        bad += 1

list_frame(500, 0.1, 1200, 800, 64)
list_frame(2000, 0.2, 1200, 800, 16)
list_frame(488, 0.2, 1200, 800, 32)
frame(500, 0.1, 30, 1200, 800, 64)
frame(2000, 0.2, 30, 1200, 800, 32)
frame(8000, 0.1, 30, 1200, 800, 32)
frame(8000, 0.2, 30, 1200, 800, 16)
frame(100000, 0.1, 320, 1920, 1080, 8)
frame(10000, 0.05, 32, 800, 600, 32)
frame(100000, 0.1, 320, 960, 540, 32)          # dense grid, long-chain pre-classification on: the pooled walk of dense grids
frame(40000, 0.2, 100, 960, 540, 16)
sys.exit(1 if bad else 0)
