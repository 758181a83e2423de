"""Multi-GPU entry points of the C-ABI on the one GPU of the test box (-m gpu): rt_multi_render with an RCCL communicator of
one rank, RCCL bound at run time moving device memory (self send/recv), N ranks as N fresh child processes on GPU 0 through
the custom-gather form (RCCL refuses two ranks on one device), and render contexts on concurrent streams."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def whole_frame(rt, torch, W, O, nx, ny, ns, precision=None):
    precision = rt.FP32 if precision is None else precision
    st = rt.alloc_rand_state(nx, ny)
    fb = rt.alloc_fb(nx, ny, precision=precision)
    rt.render_init(nx, ny, st)
    rt.render(fb, nx, ny, ns, W, st, O)
    torch.cuda.synchronize()
    return fb, st


def test_multi_render_one_rank_over_rccl(rt, cuda):
    """rt_multi_unique_id + rt_multi_init (ncclCommInitRank, RCCL dlopen'ed) with one rank: rt_multi_render equals rt_render, and
    a grouped ncclSend/ncclRecv to self moves device bytes on the caller's stream."""
    torch = cuda
    nx, ny, ns, n, spl = 200, 120, 8, 500, 30
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl)
    M = rt.Multi(0, 1, unique_id=rt.multi_unique_id())
    full = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
    M.render(full, nx, ny, ns, W, O)
    torch.cuda.synchronize()
    fb, _ = whole_frame(rt, torch, W, O, nx, ny, ns)
    assert torch.equal(full.view(torch.int32), fb.view(torch.int32))
    call_ms, kernel_ms = M.last_render_ms()
    assert 0 < kernel_ms <= call_ms
    src = torch.arange(1 << 20, dtype=torch.int32, device="cuda")
    dst = torch.zeros_like(src)
    M.selftest(src, dst, src.numel() * 4)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    M.close()


@pytest.mark.parametrize("world,nx,ny,ns,n,spl,fp16,split", [
    (2, 400, 232, 16, 10000, 32, 0, 1),      # octree, long-chain classification on; bands of equal predicted cost (the default split)
    (3, 203, 117, 4, 500, 0, 0, 1),          # ragged frame, three ranks, hitable_list path
    (2, 200, 120, 4, 500, 30, 1, 1),         # USE_FP16
    (2, 400, 232, 16, 10000, 32, 0, 0),      # the same three through runs of RT_PART_RUN tiles dealt round-robin (RT_SPLIT_RUNS)
    (3, 203, 117, 4, 500, 0, 0, 0),
    (2, 200, 120, 4, 500, 30, 1, 0),
    (3, 400, 232, 16, 10000, 32, 0, 2),      # the split kept from the first frame to the second (RT_SPLIT_BALANCED_CACHED)
])
def test_multi_render_child_processes_on_one_gpu(rt, cuda, world, nx, ny, ns, n, spl, fp16, split):
    """N fresh child processes, one rank each, all on GPU 0: the real rt_multi_render (split, render, staging slots,
    rt_assemble / rt_assemble_split) with a gloo exchange; rank 0 checks the assembled frame against a single-process render bit for bit."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(ROOT, "tests", "multi_worker.py")
    args = [str(v) for v in (world, port, nx, ny, ns, n, spl, fp16)]
    procs = [subprocess.Popen([sys.executable, script, str(r)] + args + [str(split)], stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=300)
            outs.append((p.returncode, o.decode(), e.decode()))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rc, o, e in outs:
        assert rc == 0, (rc, o[-2000:], e[-2000:])
    assert "EQUALS" in outs[0][1]


def test_multi_render_rejects_a_precision_that_is_not_the_worlds(rt, cuda):
    """rt_multi_render sizes its buffers and the exchange by `precision`, the kernel is chosen by the world's: a mismatch
    (an fp32 world into binary16-sized buffers = an out-of-bounds write) is refused before anything is launched."""
    torch = cuda
    nx, ny = 64, 40
    W32 = rt.World(22, nx, ny)
    W16 = rt.World(22, nx, ny, precision=rt.FP16)
    M = rt.Multi(0, 1, gather=lambda *a: 0)
    full = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
    for W, wrong in ((W32, rt.FP16), (W16, rt.FP32)):
        with pytest.raises(rt.RtError):
            M.render(full, nx, ny, 2, W, None, precision=wrong)
    M.render(full, nx, ny, 2, W32, None)
    torch.cuda.synchronize()
    assert rt.multi_probe() == 0                                 # RCCL bound, context + events on this device: no communication involved
    M.close()


def test_bench_self_launch_two_ranks_one_gpu(cuda):
    """plain `python3 bench.py --gpus 2` (no launcher, no WORLD_SIZE) starts its own ranks as fresh child processes,
    runs the strong-scaled C5 frame through rt_multi_render and prints ONE result line — here with both ranks on the box's one GPU
    (RT_BENCH_SAME_GPU=1: the exchange goes through the custom-gather form over gloo; RCCL refuses two ranks on one device)."""
    import json
    env = dict(os.environ, RT_BENCH_SAME_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, (p.returncode, p.stderr.decode()[-3000:])
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert "C5: 3840x2160, 256 spp, NUM_SPHERES=100000" in d["config"]["workload"]
    assert d["single_gpu_same_frame"]["frame_equals_multi_gpu_frame"] is True
    assert d["speedup_vs_single_gpu_same_frame"] > 0
    assert len(d["per_rank_render_ms"]["kernel"]) == 2 and d["per_rank_render_ms"]["imbalance_max_over_mean"] >= 1.0
    assert "roofline_issue" in d
    # a line that survives a SCALE run — a fraction that is one (C5's algorithmic figure exceeds the machine's peak: the
    # machine fraction takes its place, the reference-visit-set figure stays beside it), the like-for-like one-GPU value at top
    # level, and the CPU baseline on the N > 1 line too
    assert 0 < d["roofline"]["frac"] <= 1.0 and d["roofline"]["frac_reference_work"] > 0
    assert d["roofline"]["frac"] == d["roofline_issue"]["lane_op_frac"] or d["roofline"]["frac_reference_work"] <= 1.0
    assert 0 < d["roofline_issue"]["lane_op_frac"] <= 1.0
    assert d["value_1gpu_same_workload"] == d["single_gpu_same_frame"]["msamples_per_s"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["kind"] == "port"


def test_render_calls_on_two_streams(rt, cuda):
    """two partitions of one frame rendered on two streams.  Through the world's own context the calls are
    ordered by the library; with a context each they overlap.  Both ways every part equals the part rendered alone."""
    torch = cuda
    nx, ny, ns, n, spl = 400, 232, 16, 10000, 32
    W = rt.World(n, nx, ny).upload()
    O = rt.Octree(W, spl).upload()
    parts = [rt.Partition(p, 2) for p in range(2)]
    ref = []
    for part in parts:
        st = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
        rt.render_init(nx, ny, st, part); rt.render(fb, nx, ny, ns, W, st, O, part)
        torch.cuda.synchronize()
        ref.append((fb.clone(), st.clone()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for mode in ("shared", "own"):
        ctxs = [rt.RenderCtx().reserve(nx, ny, part) for part in parts] if mode == "own" else [None, None]
        bufs = []
        for _ in range(3):                                           # a few rounds: races do not show every time
            bufs = [(rt.alloc_fb(nx, ny, part), rt.alloc_rand_state(nx, ny, part)) for part in parts]
            torch.cuda.synchronize()
            for part, s, ctx, (fb, st) in zip(parts, streams, ctxs, bufs):   # both parts in flight before either is waited for
                with torch.cuda.stream(s):
                    rt.render_init(nx, ny, st, part)
                    if ctx is None:
                        rt.render(fb, nx, ny, ns, W, st, O, part)
                    else:
                        ctx.render(fb, nx, ny, ns, W, st, O, part)
            torch.cuda.synchronize()
            for (fb, st), (rfb, rst) in zip(bufs, ref):
                assert torch.equal(fb.view(torch.int32), rfb.view(torch.int32)) and torch.equal(st, rst), mode
        if mode == "own":
            assert all(len(c.times()) == 3 for c in ctxs)
