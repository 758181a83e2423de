"""Three renders of part 0 of N of the C5 frame (for rocprofv3: tools/part_kernels.sh, tools/pmc_part.sh).  usage: part_trace.py [nparts]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dd2360-raytracing_amd"))
import torch, rt_amd as rt
nx, ny, n, spl, spp = 3840, 2160, 100000, 320, 256
nparts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = rt.World(n, nx, ny).upload(); O = rt.Octree(W, spl).upload()
part = rt.Partition(0, nparts)
st = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
for rep in range(3):
    rt.render_init(nx, ny, st, part); rt.render(fb, nx, ny, spp, W, st, O, part); torch.cuda.synchronize()
print("kernel ms", W.render_times(), W.render_counters())
