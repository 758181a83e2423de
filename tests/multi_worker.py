"""Child process of tests/test_gpu_multi.py: one rank of an N-rank rt_multi_render job, all ranks on GPU 0.

RCCL refuses two ranks on one device, so the exchange is the custom-gather form of rt_multi (rt_multi_init_custom): the
callback copies this rank's part to the host (hipMemcpy through the HIP runtime already in the process), gathers over gloo
on 127.0.0.1 and, on the root, copies every other rank's part into its staging slot.  Everything else — partition, render
kernels, staging layout, rt_assemble — is the code path the RCCL form runs.  Rank 0 compares the assembled frame with a
single-process rt_render of the whole frame and exits 0 on equality.

usage: multi_worker.py RANK WORLD PORT NX NY NS SPHERES SPL(0 = no octree) FP16 [SPLIT 0 runs | 1 balanced | 2 balanced, cached]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))


def make_gloo_gather(rt, torch, dist, rank, world, nx, ny, px):
    """an rt_gather_fn (include/rt_amd.h) that moves the parts through host memory and a gloo gather"""
    import numpy as np
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    holder = {}                                                  # 'M': the rt.Multi this callback serves (set by main once it exists)

    def part_bytes(r):
        M = holder["M"]
        if M.split_mode != rt.SPLIT_RUNS:                        # bands of rt_split_balanced: every rank computed the same starts
            st = M.last_split()
            return (st[r + 1] - st[r]) * 64 * px
        return rt.part_pixels(nx, ny, rt.Partition(r, world)) * px

    def gather(user, d_send, send_bytes, d_parts, stride, root, stream):
        try:
            assert send_bytes == part_bytes(rank) and stride >= max(part_bytes(r) for r in range(world))
            if holder["M"].split_mode == rt.SPLIT_RUNS:
                assert stride == rt.part_pixels(nx, ny, rt.Partition(0, world)) * px
            if hip.hipStreamSynchronize(stream) != 0:
                return 1
            host = np.zeros(stride, np.uint8)
            if send_bytes and hip.hipMemcpy(host.ctypes.data, d_send, send_bytes, 2) != 0:          # device -> host
                return 2
            mine = torch.from_numpy(host)
            got = [torch.empty_like(mine) for _ in range(world)] if rank == root else None
            dist.gather(mine, got, dst=root)
            if rank == root:
                for r in range(world):
                    nbytes = part_bytes(r)
                    if r != root and nbytes and hip.hipMemcpy(d_parts + r * stride, got[r].numpy().ctypes.data, nbytes, 1) != 0:
                        return 3
            return 0
        except Exception as e:                                   # nothing may propagate through the C frame
            print("gather callback:", repr(e), file=sys.stderr, flush=True)
            return 9

    return gather, holder


def main():
    rank, world, port, nx, ny, ns, n, spl, fp16 = [int(a) for a in sys.argv[1:10]]
    split = int(sys.argv[10]) if len(sys.argv) > 10 else 1       # rt.SPLIT_RUNS 0 | SPLIT_BALANCED 1 (the library's default) | SPLIT_BALANCED_CACHED 2
    import torch
    import torch.distributed as dist
    import rt_amd as rt
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    precision = rt.FP16 if fp16 else rt.FP32
    gather, holder = make_gloo_gather(rt, torch, dist, rank, world, nx, ny, 6 if fp16 else 12)

    W = rt.World(n, nx, ny, precision=precision)
    O = rt.Octree(W, spl) if spl > 0 else None
    M = rt.Multi(rank, world, gather=gather).set_split(split)
    holder["M"] = M
    dt = torch.float16 if fp16 else torch.float32
    full = torch.zeros(nx * ny * 3, dtype=dt, device="cuda") if rank == 0 else None
    for _ in range(2):                                           # twice: buffers are reused, the RNG starts over (render_init)
        M.render(full, nx, ny, ns, W, O, root=0)
        torch.cuda.synchronize()
    call_ms, kernel_ms = M.last_render_ms()
    assert call_ms > 0 and kernel_ms > 0
    rc = 0
    if rank == 0:
        st = rt.alloc_rand_state(nx, ny)
        fb = rt.alloc_fb(nx, ny, precision=precision)
        rt.render_init(nx, ny, st)
        rt.render(fb, nx, ny, ns, W, st, O)
        torch.cuda.synchronize()
        it = torch.int16 if fp16 else torch.int32
        same = torch.equal(full.view(it), fb.view(it))
        print("multi_worker: %d-rank frame %s the single-process frame" % (world, "EQUALS" if same else "DIFFERS FROM"), flush=True)
        rc = 0 if same else 3
    dist.barrier()
    M.close()
    dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
