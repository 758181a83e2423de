#!/usr/bin/env python3
"""Start-up cost outside the timed region: create_world, then either buildOctree + candidate grid on the host +
upload (rt_build_octree, rt_octree_upload) or everything on the device (rt_build_octree_gpu).
usage: python tools/setup_time.py [num_spheres spheres_per_leaf]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dd2360-raytracing_amd"))
import rt_amd as rt

def main():
    cases = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(10000, 32), (100000, 320)]
    rt.lib()
    gpu = False
    try:
        import torch
        gpu = torch.cuda.is_available()
        if gpu: torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    except Exception:
        pass
    for n, spl in cases:
        for rep in range(4):
            t0 = time.perf_counter(); w = rt.World(n, 3840, 2160)
            t1 = time.perf_counter(); o = rt.Octree(w, spl)
            t2 = time.perf_counter()
            if gpu: w.upload(); o.upload(); torch.cuda.synchronize()
            t3 = time.perf_counter()
            line = "N=%d spl=%d: create_world %.2f ms | host: build_octree + grid %.2f ms, upload (world + tree) %s" % (
                n, spl, (t1 - t0) * 1e3, (t2 - t1) * 1e3, "%.2f ms" % ((t3 - t2) * 1e3) if gpu else "skipped (no GPU)")
            o.close()
            if gpu:
                w2 = rt.World(n, 3840, 2160)
                t4 = time.perf_counter(); w2.upload(); torch.cuda.synchronize()
                t5 = time.perf_counter(); g = rt.Octree(w2, spl, gpu=True); torch.cuda.synchronize()
                t6 = time.perf_counter()
                line += " | device: world upload %.2f ms, rt_build_octree_gpu %.2f ms -> set-up %.2f ms instead of %.2f ms" % (
                    (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t1 - t0 + t6 - t4) * 1e3, (t3 - t0) * 1e3)
                g.close(); w2.close()
            print(line)
            w.close()

if __name__ == "__main__":
    main()
