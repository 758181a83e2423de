#!/usr/bin/env python3
"""bench.py — Msamples/s of the render() hot path on MI355X (BASELINE.json metric).

One "step" = one frame through the hot path: render_init + render (the reference's timed region, main.cu:419-431),
inputs (scene, octree) already resident in HBM.  N=1 workload = BASELINE config 3: 1200x800, 64 spp,
NUM_SPHERES=10000, USE_OCTREE on, SPHERES_PER_LEAF=32, fp32.  N>1 (one process per GPU, torch.distributed over
RCCL): weak scaling — the frame keeps its 3:2 aspect and grows to N x 960 000 pixels, 8x8-pixel tiles are dealt
round-robin to the ranks (tile t -> rank t % N), each rank renders its tiles into a compact buffer and ONE gather
over xGMI brings the framebuffer to rank 0, which reassembles it (inside the timed region).

Prints ONE JSON line on rank 0.  --config c2|c3 selects another BASELINE config for manual runs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic work per sample of the reference's visit set (unfused flops), used for roofline.achieved
CONFIGS = {
    "c2": dict(name="C2", nx=1200, ny=800, spp=64, spheres=500, octree=False, spl=30, flops_per_sample=18.2e3),
    "c3": dict(name="C3", nx=1200, ny=800, spp=64, spheres=10000, octree=True, spl=32, flops_per_sample=13.6e3),
    "c5": dict(name="C5 (one GPU's share: full 4K frame at 32 spp)", nx=3840, ny=2160, spp=32, spheres=100000, octree=True, spl=320, flops_per_sample=78e3),
    "c4": dict(name="C4", nx=1200, ny=800, spp=64, spheres=10000, octree=True, spl=32, flops_per_sample=9.8e3, fp16=True),
}
PEAK_FP32_VECTOR_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector (= FP32 matrix) rate, spec
PEAK_UNFUSED_TOPS = 78.6              # the parity mode's own ceiling (no FMA contraction): PEAK_FP32_VECTOR / 2.  A wave64 fp32 instruction
                                      # issues in 2 cycles on gfx950 whether packed or not (profiles/micro_pk_rate_r1.txt: v_pk_* run at half
                                      # the instruction rate of the scalar forms; a mul+add stream sustains 54 T op/s), so packing buys nothing
                                      # and SURVEY 8d's 39.3 T (one op per lane and cycle on 64 lanes per CU) was a factor 2 too low
PEAK_HBM_GBS = 8000.0


def cpu_baseline(cfg, rt_cores, use_octree, rows, spp):
    """the oracle ("port" of the reference algorithm) timed on this host's cores on a bounded sample of the workload"""
    from oracle_lib import OracleScene
    S = OracleScene(cfg["spheres"], cfg["nx"], cfg["ny"], fp16=bool(cfg.get("fp16")), use_octree=use_octree, spl=cfg["spl"])
    # rows spread over the frame so the sample sees sky, spheres and ground like the whole frame does
    picks = [int((k + 0.5) * cfg["ny"] / rows) for k in range(rows)]
    t0 = time.perf_counter()
    import threading
    def work(r):
        S.render(spp, row0=r, rows=1, nthreads=1)
    # one python thread per row batch; the C call releases the GIL (ctypes), rows run concurrently
    chunks = [picks[i::rt_cores] for i in range(rt_cores)]
    def run(chunk):
        for r in chunk:
            work(r)
    th = [threading.Thread(target=run, args=(c,)) for c in chunks if c]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    samples = rows * cfg["nx"] * spp
    return samples / dt / 1e6, dt, samples


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--list-reference", action="store_true", help="octree-off configs: plain list-order scan instead of the candidate grid")
    args = ap.parse_args()

    import torch
    import rt_amd as rt
    import rt_dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # test-only switches (tools/rehearse_2rank.sh): all ranks on GPU 0 with the gloo backend, so the whole N>1 flow can be
    # rehearsed on a one-GPU box; the driver's runs never set them
    same_gpu = os.environ.get("RT_BENCH_SAME_GPU") == "1"
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    rc, _ = rt.device_check()
    if rc != 0:
        raise SystemExit("rt_device_check failed: %d" % rc)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = CONFIGS[args.config]
    nx, ny = rt_dist.scaled_frame(cfg["nx"], cfg["ny"], world)
    spp = cfg["spp"]
    part = rt.Partition(rank, world)

    # scene: generated on the host exactly as create_world does (seed 1984), resident in HBM before the timed region
    precision = rt.FP16 if cfg.get("fp16") else rt.FP32
    W = rt.World(cfg["spheres"], nx, ny, precision=precision).upload()
    if args.list_reference:
        W.set_list_traversal(rt.TRAVERSAL_REFERENCE)
    O = rt.Octree(W, cfg["spl"]).upload() if cfg["octree"] else None
    st = rt.alloc_rand_state(nx, ny, part)
    fb = rt.alloc_fb(nx, ny, part, precision=precision)
    per = rt.part_pixels(nx, ny, rt.Partition(0, world))          # padded part size (largest part)
    if world > 1:
        send = torch.zeros(per * 3, dtype=fb.dtype, device="cuda")
        parts = torch.zeros(world * per * 3, dtype=fb.dtype, device="cuda") if rank == 0 else None
        full = torch.zeros(nx * ny * 3, dtype=fb.dtype, device="cuda") if rank == 0 else None

    ev = []

    def step(timed):
        rt.render_init(nx, ny, st, part)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rt.render(fb, nx, ny, spp, W, st, O, part)
        if timed:
            e1.record()
            ev.append((e0, e1))
        if world > 1:
            send[: fb.numel()].copy_(fb)
            if backend == "nccl":
                gathered = rt_dist.gather_parts(dist, send, rank, world, dst=0)   # the single framebuffer exchange over xGMI
            else:                                                                 # rehearsal: gloo moves host tensors
                g = rt_dist.gather_parts(dist, send.cpu(), rank, world, dst=0)
                gathered = [x.cuda() for x in g] if rank == 0 else None
            if rank == 0:
                torch.cat(gathered, out=parts)
                rt.assemble(full, parts, nx, ny, world, precision=precision)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    W.render_times()                                               # forget the warm-up launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if os.environ.get("RT_BENCH_CHECK") == "1" and rank == 0:
            # rehearsal only: the assembled frame must equal a single-process render of the same frame
            W1 = rt.World(cfg["spheres"], nx, ny, precision=precision).upload()
            O1 = rt.Octree(W1, cfg["spl"]).upload() if cfg["octree"] else None
            st1 = rt.alloc_rand_state(nx, ny); fb1 = rt.alloc_fb(nx, ny, precision=precision)
            rt.render_init(nx, ny, st1); rt.render(fb1, nx, ny, spp, W1, st1, O1); torch.cuda.synchronize()
            same = torch.equal(full.view(torch.int16 if cfg.get("fp16") else torch.int32), fb1.view(torch.int16 if cfg.get("fp16") else torch.int32))
            print("rehearsal: assembled %d-rank frame %s the single-process frame" % (world, "EQUALS" if same else "DIFFERS FROM"), file=sys.stderr, flush=True)
            if not same:
                raise SystemExit(3)

    call_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))       # whole rt_render call (pre-pass + kernel)
    kt = W.render_times()                                            # HIP events around the render kernel itself, on its stream
    kernel_ms = sum(kt) / max(1, len(kt))
    samples_step = nx * ny * spp                                  # whole job, all ranks
    local_samples = rt.part_pixels(nx, ny, part) * spp if world > 1 else samples_step
    value = samples_step * args.steps / dt / 1e6

    if rank == 0:
        # the render kernel that ran (rt_kernels.hip launch_render: sparse grids take the variant with grouped cooperative walks)
        if cfg.get("fp16"):
            kernel_name = "k_render_h<%s,0>" % ("true" if cfg["octree"] else "false")
        elif cfg["octree"]:
            ai = O.accel_info()
            kernel_name = "k_render<true,0,%d>" % (4 if ai["grid_entries"] <= 8 * ai["grid_dim"] ** 2 else 1)
        else:
            ai = W.list_accel_info()
            if ai["enabled"] and not args.list_reference:             # the list as a one-node tree through the candidate grid
                kernel_name = "k_render<true,0,%d>" % (4 if ai["grid_entries"] <= 8 * ai["grid_dim"] ** 2 else 1)
            else:
                kernel_name = "k_render<false,0,1>"
        flops_launch = cfg["flops_per_sample"] * local_samples
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")     # written from the rocprofv3 --pmc passes
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(cfg["name"], {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s (W*H*spp/render_time) at 1200x800, 10k spheres", "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if cfg.get("fp16") else "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d spp, NUM_SPHERES=%d, USE_OCTREE %s, SPHERES_PER_LEAF=%d, %s, create_world seed 1984%s"
                       % (cfg["name"], nx, ny, spp, cfg["spheres"], "on" if cfg["octree"] else "off", cfg["spl"], "USE_FP16" if cfg.get("fp16") else "fp32",
                          "" if world == 1 else "; frame grown to %d x 960000 px, 8x8 tiles round-robin over %d GPUs, one RCCL gather" % (world, world)),
                       "timed_region": "render_init + render (+ gather + assemble when n_gpus>1), scene resident in HBM"},
            "roofline": {"bound": "valu", "kernel": kernel_name,
                         "achieved": round(achieved, 4), "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 5), "traffic": traffic,
                         "frac_unfused": round(achieved / PEAK_UNFUSED_TOPS, 5), "peak_unfused": PEAK_UNFUSED_TOPS,
                         "kernel_ms": round(kernel_ms, 4), "render_call_ms": round(call_ms, 4), "flops_per_sample": cfg["flops_per_sample"],
                         "note": "algorithmic unfused flops of the reference's visit set (SURVEY 8d) / device time of the render kernel "
                                 "(HIP events on its stream, last <=64 launches); render_call_ms adds the scheduling pre-pass; "
                                 "VALU-bound path, HBM traffic is ~1.7 B/sample"},
        }
        # the same kernel against the HBM roof (for the record: it is nowhere near it): algorithmic bytes = RNG state in + out
        # (2 x 48 B) and the vec3 written (12 B) per pixel this rank renders
        alg_bytes = (96.0 + (6.0 if cfg.get("fp16") else 12.0)) * (local_samples / spp)
        hbm = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out["roofline_hbm"] = {"bound": "hbm", "kernel": kernel_name, "achieved": round(hbm, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(hbm / PEAK_HBM_GBS, 6), "traffic": traffic,
                               "note": "96 B curandState in+out + vec3 out (12 B, fp16: 6 B) per pixel / kernel time; traffic = FETCH_SIZE + WRITE_SIZE per launch (profiles/hbm_traffic.json)"}
        if world == 1 and not args.no_cpu_baseline:
            cores = max(1, min(16, os.cpu_count() or 1))
            v, secs, smp = cpu_baseline(cfg, cores, cfg["octree"], rows=64 if cfg["octree"] else 4, spp=spp if cfg["octree"] else 16)
            out["cpu_baseline"] = {"value": round(v, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                   "sample": "%d rows x %d px of the same frame (%d samples), oracle %s path, %.1f s wall on %d threads"
                                             % (64 if cfg["octree"] else 4, cfg["nx"], smp, "hitTree" if cfg["octree"] else "hitable_list", secs, cores)}
            if cfg["octree"]:
                v2, secs2, smp2 = cpu_baseline(cfg, cores, False, rows=16, spp=8)
                out["cpu_baseline_hitable_list"] = {"value": round(v2, 5), "unit": "Msamples/s", "cores": cores, "kind": "port",
                                                    "sample": "%d rows x %d px x %d spp (%d samples), oracle hitable_list path, %.1f s wall" % (16, cfg["nx"], 8, smp2, secs2)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
