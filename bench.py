#!/usr/bin/env python3
"""bench.py — Msamples/s of the render() hot path on MI355X (BASELINE.json metric).

One "step" = one frame through the hot path: render_init + render (the reference's timed region, main.cu:419-431),
inputs (scene, octree) already resident in HBM.

N = 1   BASELINE config 3 (the config the metric is quoted on): 1200x800, 64 spp, NUM_SPHERES=10000, USE_OCTREE on,
        SPHERES_PER_LEAF=32, fp32.  --config c2|c4|c5 selects another BASELINE config for manual runs and profiles.
N > 1   BASELINE config 5, STRONG scaling: the fixed 3840x2160 frame at 256 spp, NUM_SPHERES=100000, SPHERES_PER_LEAF=320 is
        split over the N ranks (one process per GPU) by rt_multi_render of the C-ABI: 8x8 tiles dealt round-robin in runs of 64, each rank
        renders its tiles into a compact buffer, ONE RCCL exchange (grouped ncclSend/ncclRecv over xGMI, issued by
        librt_amd.so on the render stream) brings the framebuffer to rank 0, which reassembles it — all inside the timed
        region.  torch.distributed (gloo, 127.0.0.1) only carries the RCCL id, the barriers and the max over ranks.
        --scaling weak keeps round 1's weak-scaled C3 frame (N x 960 000 pixels) for comparison.
        Started WITHOUT a launcher (`python3 bench.py --gpus N`, no WORLD_SIZE in the environment) the script launches its own
        N ranks: a parent that never touches torch or the GPU starts `python -m torch.distributed.run --nproc-per-node N
        bench.py ...` as a child process group, relays rank 0's JSON line, and kills the group and exits non-zero when the job
        exceeds --timeout (a rank lost before the exchange must not leave the root waiting for ever).  Every rank also carries
        its own watchdog for the launcher-started form.

Prints ONE JSON line on rank 0.  `roofline` is the algorithmic VALU roofline of the render kernel (SURVEY 8d flops / kernel
time from HIP events on the launch stream), `roofline_issue` the machine's own view from rocprofv3 PMC passes collected in
this run (VALU issue rate against the measured 888 G wave-instructions/s, lane utilisation, SALU:VALU, wait share),
`roofline_hbm` the HBM side with measured FETCH_SIZE / WRITE_SIZE, `cpu_baseline` the CPU oracle's hitable_list path on this
host's cores.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic work per sample of the reference's visit set (unfused flops), used for roofline.achieved
CONFIGS = {
    "c2": dict(name="C2", nx=1200, ny=800, spp=64, spheres=500, octree=False, spl=30, flops_per_sample=18.2e3),
    "c3": dict(name="C3", nx=1200, ny=800, spp=64, spheres=10000, octree=True, spl=32, flops_per_sample=13.6e3),
    "c4": dict(name="C4", nx=1200, ny=800, spp=64, spheres=10000, octree=True, spl=32, flops_per_sample=9.8e3, fp16=True),
    "c5": dict(name="C5", nx=3840, ny=2160, spp=256, spheres=100000, octree=True, spl=320, flops_per_sample=78e3),
}
PEAK_FP32_VECTOR_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector (= FP32 matrix) rate, spec
PEAK_UNFUSED_TOPS = 78.6              # the parity mode's own ceiling (no FMA contraction): PEAK_FP32_VECTOR / 2
PEAK_VALU_ISSUE_G = 888.0             # G wave-instructions/s the chip sustains on an independent v_fma_f32 stream
                                      # (tools/micro/pk_rate.hip, profiles/micro_pk_rate_r1.txt)
PEAK_VALU_ISSUE_SPEC_G = PEAK_FP32_VECTOR_TFLOPS * 1e3 / 128.0   # the same from the spec sheet: 157.3 TFLOP/s / (64 lanes x 2 flops) = 1228.9 G
PEAK_HBM_GBS = 8000.0

PMC_PASSES = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM",
    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES",
    "FETCH_SIZE",
    "WRITE_SIZE",
]


def usable_cores():
    """threads this process may really run at once: the affinity mask, cut by a cgroup CPU quota when there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(cfg, threads, use_octree, rows, spp):
    """the oracle ("port" of the reference algorithm) timed on this host's cores on a bounded sample of the workload"""
    import threading
    from oracle_lib import OracleScene
    S = OracleScene(cfg["spheres"], cfg["nx"], cfg["ny"], fp16=bool(cfg.get("fp16")), use_octree=use_octree, spl=cfg["spl"])
    # rows spread over the frame so the sample sees sky, spheres and ground like the whole frame does
    picks = [int((k + 0.5) * cfg["ny"] / rows) for k in range(rows)]
    chunks = [picks[i::threads] for i in range(threads)]

    def run(chunk):                       # the C call releases the GIL (ctypes): rows run concurrently
        for r in chunk:
            S.render(spp, row0=r, rows=1, nthreads=1)

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(c,)) for c in chunks if c]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    samples = rows * cfg["nx"] * spp
    return samples / dt / 1e6, dt, samples


def parse_pmc_dir(d):
    """{counter: per-dispatch average over the render kernel's dispatches} from the counter_collection CSVs under d"""
    acc, disp = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "k_render" not in k or "k_render_init" in k:
                continue
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            disp.setdefault(c, set()).add(row["Dispatch_Id"])
    return {c: acc[c] / max(1, len(disp[c])) for c in acc}


def collect_pmc(config, list_reference, arith="ieee", timeout=240, deadline=None):
    """rocprofv3 --pmc passes over a short run of this same script (child processes, before this process touches the GPU).
    Returns ({counter: per-dispatch average for the render kernel}, note).  Counters come in their own runs, never together
    with tracing; the profiled program stands directly behind `--`."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    out = {}
    tmp = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)                                     # the child is a one-GPU run of its own, whoever started us
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc"]
    if list_reference:
        child.append("--list-reference")
    if arith != "ieee":
        child += ["--arith", arith]                          # (the parsed value: `--arith=contract` and prefixes profile the right kernels too)
    note = None
    try:
        for i, pmc in enumerate(PMC_PASSES):
            left = timeout if deadline is None else min(timeout, deadline - time.monotonic())
            if left < 45:
                note = "pmc passes %d.. skipped: the job's --timeout leaves no room for them" % i      # optional extras never cost the result line
                break
            d = os.path.join(tmp, "p%d" % i)
            cmd = [exe, "--pmc"] + pmc.split() + ["--output-format", "csv", "-d", d, "--"] + child
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=left)
            except subprocess.TimeoutExpired:
                note = "pmc pass %d timed out" % i
                break
            if p.returncode != 0:
                note = "pmc pass %d failed (rc %d)" % (i, p.returncode)
                continue
            out.update(parse_pmc_dir(d))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, note


def alive(pid):
    try:
        return open("/proc/%d/stat" % pid).read().rsplit(")", 1)[1].split()[0] != "Z"      # a zombie is not a survivor
    except (FileNotFoundError, ProcessLookupError, IndexError):
        return False


def descendants(root):
    """pids of every process below `root` right now (children of children included), from /proc — never root itself"""
    parent = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                parent[int(d)] = int(open("/proc/%s/stat" % d).read().rsplit(")", 1)[1].split()[1])
            except (FileNotFoundError, ProcessLookupError, IndexError, ValueError):
                pass
    out, frontier = [], [root]
    while frontier:
        cur = frontier.pop()
        for pid, pp in parent.items():
            if pp == cur and pid != root and pid not in out:
                out.append(pid); frontier.append(pid)
    return out


def self_launch(n, argv, timeout):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as a fresh process group (torch.distributed.run, one
    process per GPU), relay rank 0's JSON line, enforce the timeout.  This parent imports neither torch nor the library and
    never touches a GPU; it never execs.  Returns the exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["RT_BENCH_SELF_LAUNCHED"] = "1"                           # the ranks' own watchdogs stand back: this parent's timeout comes first
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)      # a process group of its own: killable as one
    lines = []

    def pump():
        for raw in p.stdout:
            lines.append(raw.decode(errors="replace"))

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        rc = p.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        print("bench.py: the %d-rank job did not finish within %d s — stopping it" % (n, timeout), file=sys.stderr, flush=True)
        # p is the elastic agent, alone in the group started above: it starts every rank in a session of its OWN
        # (start_new_session), so signalling p's group reaches the agent only.  The ranks are recorded by pid first (the
        # descendants of p, from /proc), the agent gets SIGTERM and its own 30 s grace to stop them, and whatever of the
        # recorded pids is still alive afterwards is killed — those exact processes and their sessions, nothing else.
        ranks = descendants(p.pid)
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        try:
            p.wait(timeout=40)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        deadline = time.monotonic() + 5
        for pid in ranks:
            while alive(pid) and time.monotonic() < deadline:
                time.sleep(0.1)
            if alive(pid):
                print("bench.py: rank process %d survived the agent — killing it" % pid, file=sys.stderr, flush=True)
                try:
                    os.killpg(os.getpgid(pid), signal.SIGKILL)     # its own session / group (torch.distributed.run made it the leader)
                except (ProcessLookupError, PermissionError):
                    try:
                        os.kill(pid, signal.SIGKILL)
                    except ProcessLookupError:
                        pass
        try:
            p.wait(timeout=5)
        except subprocess.TimeoutExpired:
            pass
        rc = 124
    t.join(timeout=5)
    found = None
    for ln in lines:
        st = ln.strip()
        if st.startswith("{") and '"metric"' in st:
            found = st
        elif st:
            print(st, file=sys.stderr)
    if found is not None and rc == 0:
        print(found, flush=True)
        return 0
    if rc == 0:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr, flush=True)
        rc = 1
    return rc


def arm_watchdog(seconds, rank):
    """launcher-started ranks: a rank that is still here after `seconds` leaves with 124 (torch.distributed.run then stops the
    others) instead of sitting in a collective for ever"""
    def fire():
        print("bench.py rank %d: watchdog after %d s — exiting" % (rank, seconds), file=sys.stderr, flush=True)
        os._exit(124)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default: c3 on one GPU, c5 (strong-scaled) on several")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N>1 only: strong = fixed C5 frame (default), weak = C3 frame grown N-fold")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline_issue / traffic become null)")
    ap.add_argument("--list-reference", action="store_true", help="octree-off configs: plain list-order scan instead of the candidate grid")
    ap.add_argument("--arith", default="ieee", choices=["ieee", "contract"], help="contract: the opt-in tolerance mode with FMA contraction allowed (rt_world_set_arith) — never the parity mode, reported separately")
    ap.add_argument("--split", default="runs", choices=["runs", "balanced", "balanced-cached"], help="N>1: how rt_multi_render divides the frame — runs of 64 tiles dealt round-robin (default), "
                    "bands of equal predicted cost from a whole-frame pilot pass on every rank and frame, or those bands kept from frame to frame (rt_multi_set_split)")
    ap.add_argument("--timeout", type=int, default=900, help="N>1: seconds after which the job is killed (self-launched: by the parent; every rank also watches itself)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:], args.timeout))      # before torch, before any GPU call

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or without a launcher: bench.py starts its own ranks)" % (args.gpus, world, args.gpus))
    watchdog = arm_watchdog(args.timeout + (60 if os.environ.get("RT_BENCH_SELF_LAUNCHED") == "1" else 0), rank) if world > 1 else None
    if os.environ.get("RT_BENCH_TEST_HANG") == str(rank):          # tests/test_bench_launch.py: a rank that never arrives
        if watchdog is not None:
            watchdog.cancel()
        time.sleep(3600)
    weak = world > 1 and args.scaling == "weak"
    cfg_key = args.config or ("c3" if (world == 1 or weak) else "c5")
    cfg = CONFIGS[cfg_key]

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # single node: the control plane stays on the loopback
        dist.init_process_group("gloo")                            # (no GPU involved: the ranks meet before any of them touches one)

    # counters first, in child processes, while NO rank of this job has touched a GPU yet.  N = 1: the workload itself.
    # N > 1 (strong-scaled C5): rank 0 profiles the same frame on one GPU; a rank's share of those instructions over its own
    # kernel time is its issue rate (the tile split deals the frame's work out evenly: per_rank_render_ms shows how evenly).
    pmc, pmc_note = ({}, "skipped")
    if not args.no_pmc and not weak and rank == 0:
        # (N > 1: the other ranks wait at the barrier below with their watchdogs armed: the passes get a third of --timeout at most)
        pmc, pmc_note = collect_pmc(cfg_key, args.list_reference, args.arith, timeout=240 if world == 1 else 400,
                                    deadline=None if world == 1 else time.monotonic() + args.timeout / 3.0)
    if world > 1:
        dist.barrier()

    import rt_amd as rt

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # test-only switch (tools/rehearse_2rank.sh): all ranks on GPU 0 and the exchange through rt_multi's custom-gather form
    # over gloo (RCCL refuses two ranks on one device), so the whole N>1 flow can be rehearsed on a one-GPU box; the driver's
    # runs never set it
    same_gpu = os.environ.get("RT_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    local_rank %= max(1, torch.cuda.device_count())                # fewer devices than ranks: ranks share (RCCL then declines, see below)
    torch.cuda.set_device(local_rank)
    rc, _ = rt.device_check()
    if rc != 0:
        raise SystemExit("rt_device_check failed: %d" % rc)

    nx, ny = cfg["nx"], cfg["ny"]
    if weak:                                                       # same aspect, ~world times the pixels
        nx, ny = int(round(nx * world ** 0.5)), int(round(ny * world ** 0.5))
    spp = cfg["spp"]
    precision = rt.FP16 if cfg.get("fp16") else rt.FP32

    # scene: generated on the host exactly as create_world does (seed 1984), resident in HBM before the timed region
    W = rt.World(cfg["spheres"], nx, ny, precision=precision).upload()
    if args.list_reference:
        W.set_list_traversal(rt.TRAVERSAL_REFERENCE)
    if args.arith == "contract":
        W.set_arith(rt.ARITH_CONTRACT)
    O = rt.Octree(W, cfg["spl"]).upload() if cfg["octree"] else None
    kernel_name = rt.render_kernel_name(W, O, 0)                   # the library's own selection, as rocprofv3 names it
    M = None
    if world > 1:
        transport = "rccl"
        why_not = "RT_BENCH_SAME_GPU=1" if same_gpu else None
        if not same_gpu:
            # RCCL inside librt_amd.so.  ncclCommInitRank is collective: a rank that cannot get there (no RCCL, no context, a
            # device shared with another rank — RCCL refuses two ranks on one device) must be known BEFORE any rank enters it,
            # or the others wait inside it for ever.  So: every rank probes locally (rt_multi_probe, no communication) and says
            # which device it sits on; the answers are agreed over gloo; then ALL ranks join the communicator — or none does
            # and the job takes the host-staged gloo exchange below, and the line says so.
            mine = [rt.multi_probe(), local_rank, os.environ.get("HIP_VISIBLE_DEVICES", "") + "|" + os.environ.get("ROCR_VISIBLE_DEVICES", "")]
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            devices = [(e[2], e[1]) for e in everyone]
            if any(e[0] != 0 for e in everyone):
                why_not = "rt_multi_probe failed on rank(s) %s" % [r for r, e in enumerate(everyone) if e[0] != 0]
            elif len(set(devices)) < world:
                why_not = "%d ranks on %d device(s)" % (world, len(set(devices)))
            if why_not is None:
                ids = [None]
                try:
                    ids = [rt.multi_unique_id() if rank == 0 else None]
                except rt.RtError as e:
                    print("rank 0: %s" % e, file=sys.stderr, flush=True)
                dist.broadcast_object_list(ids, src=0)
                if ids[0] is None:
                    why_not = "rt_multi_unique_id failed on rank 0"
            if why_not is None:
                try:
                    M = rt.Multi(rank, world, unique_id=ids[0])      # collective; the watchdog bounds it
                except rt.RtError as e:
                    print("rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
                    M = None
                ok = torch.tensor([1 if M is not None else 0])
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    if M is not None:
                        M.close()
                    M = None
                    why_not = "ncclCommInitRank failed on a rank"
        if M is None:
            from multi_worker import make_gloo_gather
            transport = "gloo (host-staged) — %s" % why_not
            gather, holder = make_gloo_gather(rt, torch, dist, rank, world, nx, ny, 6 if cfg.get("fp16") else 12)
            M = rt.Multi(rank, world, gather=gather)
            holder["M"] = M
        M.set_split({"runs": rt.SPLIT_RUNS, "balanced": rt.SPLIT_BALANCED, "balanced-cached": rt.SPLIT_BALANCED_CACHED}[args.split])
        M.reserve(nx, ny, precision, 0)
        full = torch.zeros(nx * ny * 3, dtype=torch.float16 if cfg.get("fp16") else torch.float32, device="cuda") if rank == 0 else None
    else:
        st = rt.alloc_rand_state(nx, ny)
        fb = rt.alloc_fb(nx, ny, precision=precision)

    ev, rank_ms = [], []

    def step(timed):
        if M is not None:
            M.render(full, nx, ny, spp, W, O, root=0)            # render_init + render + RCCL exchange + assemble, one call
            if timed:
                rank_ms.append(M.last_render_ms())                 # (synchronises with this rank's render, not with the exchange)
            return
        rt.render_init(nx, ny, st)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rt.render(fb, nx, ny, spp, W, st, O)
        if timed:
            e1.record()
            ev.append((e0, e1))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    if M is None:
        W.render_times()                                           # forget the warm-up launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([sum(a for a, _ in rank_ms) / max(1, len(rank_ms)), sum(b for _, b in rank_ms) / max(1, len(rank_ms))], dtype=torch.float64)
        allr = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(x[0]), 3), round(float(x[1]), 3)] for x in allr]
        kernel_ms = float(mine[1])
        call_ms = float(mine[0])
    else:
        call_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))       # whole rt_render call (pre-pass + kernel)
        kt = W.render_times()                                        # HIP events around the render kernel itself, on its stream
        kernel_ms = sum(kt) / max(1, len(kt))
    long_chains = None if world > 1 else W.render_counters()["long_chains"]      # pixels the pilot pass started as long chains (last frame)
    samples_step = nx * ny * spp                                  # whole job, all ranks
    if world > 1 and args.split != "runs":
        stt = M.last_split()                                       # the bands of the last frame (every rank computed the same ones)
        rank_pixels = [(stt[r + 1] - stt[r]) * 64 for r in range(world)]
    else:
        rank_pixels = [rt.part_pixels(nx, ny, rt.Partition(r, world)) for r in range(world)] if world > 1 else [nx * ny]
    local_samples = rank_pixels[rank] * spp if world > 1 else samples_step
    value = samples_step * args.steps / dt / 1e6

    # strong scaling: the same frame on ONE GPU (rank 0 alone, after the timed region) as the reference point of the curve
    single = None
    if world > 1 and not weak:
        fence()
        if rank == 0:
            st1 = rt.alloc_rand_state(nx, ny); fb1 = rt.alloc_fb(nx, ny, precision=precision)
            rt.render_init(nx, ny, st1); rt.render(fb1, nx, ny, spp, W, st1, O); torch.cuda.synchronize()      # warm-up (workspace)
            t1 = time.perf_counter()
            rt.render_init(nx, ny, st1); rt.render(fb1, nx, ny, spp, W, st1, O); torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            it = torch.int16 if cfg.get("fp16") else torch.int32
            single = {"ms_per_step": round(d1 * 1e3, 3), "msamples_per_s": round(samples_step / d1 / 1e6, 3),
                      "frame_equals_multi_gpu_frame": bool(torch.equal(fb1.view(it), full.view(it)))}
        fence()

    if rank == 0:
        flops_launch = cfg["flops_per_sample"] * local_samples
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        fetch_kb, write_kb = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")
        if world > 1 and fetch_kb is not None and write_kb is not None:      # counters of the one-GPU frame: this rank's share of them
            fetch_kb, write_kb = fetch_kb * local_samples / float(samples_step), write_kb * local_samples / float(samples_step)
        traffic = int((fetch_kb + write_kb) * 1024) if fetch_kb is not None and write_kb is not None else None
        workload = "%s: %dx%d, %d spp, NUM_SPHERES=%d, USE_OCTREE %s, SPHERES_PER_LEAF=%d, %s, create_world seed 1984" % (
            cfg["name"], nx, ny, spp, cfg["spheres"], "on" if cfg["octree"] else "off", cfg["spl"], "USE_FP16" if cfg.get("fp16") else "fp32")
        if args.arith == "contract":
            workload += "; ARITHMETIC: FMA contraction allowed (RT_ARITH_CONTRACT) - a tolerance mode, NOT the pixel-identical parity mode"
        if world > 1:
            workload += ("; frame grown to %d x 960000 px" % world if weak else "; the fixed frame") + \
                        ", %s over %d GPUs (rt_multi_render), one exchange to rank 0 over %s" % (
                            {"runs": "runs of 64 8x8 tiles round-robin", "balanced": "bands of equal predicted cost (rt_split_balanced: a whole-frame pilot pass on every rank, every frame, inside the timed region)",
                             "balanced-cached": "bands of equal predicted cost, kept from the first frame on (rt_split_balanced once, outside the timed steps)"}[args.split], world, transport)
        out = {
            "metric": "Msamples/s (W*H*spp/render_time) at %dx%d, %d spheres" % (nx, ny, cfg["spheres"]),
            "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak" if (world == 1 or weak) else "strong", "vs_baseline": None,
            "dtype": "f16" if cfg.get("fp16") else "f32", "data": "synthetic",
            "config": {"workload": workload,
                       "timed_region": "render_init + render" + (" + framebuffer exchange + rt_assemble (one rt_multi_render call per frame)" if world > 1 else "") + ", scene resident in HBM"},
            "roofline": {"bound": "valu", "kernel": kernel_name,
                         "achieved": round(achieved, 4), "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 5), "traffic": traffic,
                         "frac_unfused": round(achieved / PEAK_UNFUSED_TOPS, 5), "peak_unfused": PEAK_UNFUSED_TOPS,
                         "kernel_ms": round(kernel_ms, 4), "render_call_ms": round(call_ms, 4), "flops_per_sample": cfg["flops_per_sample"],
                         "long_chains_preclassified": long_chains,
                         "note": "ALGORITHMIC unfused flops of the reference's visit set (SURVEY 8d: every sphere of every visited bucket, "
                                 "or of the whole list) / device time of the render kernel (HIP events on its stream%s).  The fp32 kernels "
                                 "find the same hits with ~20x fewer sphere tests (exact culling grid), so this is delivered algorithmic "
                                 "work, not issued instructions, and can exceed a machine peak: roofline_issue is the machine-side figure."
                                 % ("" if world == 1 else ", this rank's share")},
        }
        out["roofline"]["frac_reference_work"] = out["roofline"]["frac"]
        vi, tc = pmc.get("SQ_INSTS_VALU"), pmc.get("SQ_THREAD_CYCLES_VALU")
        if vi:
            share = local_samples / float(samples_step)           # N > 1: this rank's part of the profiled one-GPU frame
            rate = vi * share / (kernel_ms * 1e-3) / 1e9
            wc = pmc.get("SQ_WAVE_CYCLES")
            out["roofline_issue"] = {
                "bound": "valu_issue", "kernel": kernel_name, "achieved": round(rate, 2), "peak": PEAK_VALU_ISSUE_G, "unit": "G wave-instructions/s",
                "frac": round(rate / PEAK_VALU_ISSUE_G, 4),
                "peak_spec": round(PEAK_VALU_ISSUE_SPEC_G, 1), "frac_of_spec": round(rate / PEAK_VALU_ISSUE_SPEC_G, 4),
                "valu_wave_insts_per_launch": int(vi * share),
                "lane_utilisation": round(tc / (vi * 64.0), 4) if tc else None,
                "lane_op_frac": round(rate / PEAK_VALU_ISSUE_SPEC_G * tc / (vi * 64.0), 4) if tc else None,
                "salu_to_valu": round(pmc["SQ_INSTS_SALU"] / vi, 4) if pmc.get("SQ_INSTS_SALU") else None,
                "wait_any_share": round(pmc["SQ_WAIT_ANY"] / wc, 4) if pmc.get("SQ_WAIT_ANY") and wc else None,
                "wait_inst_share": round(pmc["SQ_WAIT_INST_ANY"] / wc, 4) if pmc.get("SQ_WAIT_INST_ANY") and wc else None,
                "note": "rocprofv3 --pmc passes over 3 launches of this workload in child processes of this run (per-launch averages of the "
                        "render kernel); SQ_INSTS_VALU / kernel_ms.  `peak` is the issue rate an independent v_fma_f32 stream was MEASURED to sustain "
                        "on this chip (tools/micro/pk_rate.hip), `peak_spec` the spec sheet's 157.3 TFLOP/s / 128 flops per wave-instruction; "
                        "lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU); lane_op_frac = frac_of_spec x lane_utilisation: busy lane-slots of the "
                        "spec sheet's lane-slots — the one machine fraction that cannot exceed 1" + ("; " + pmc_note if pmc_note else "")}
            if world > 1:
                out["roofline_issue"]["note"] += ("; N > 1: the counters are those of the SAME frame rendered on one GPU (rank 0, before the job touched "
                                                  "a GPU), a rank's instructions = its share of the frame's pixels (%.4f) of them, over its own kernel time" % share)
                out["roofline_issue"]["per_rank_achieved"] = [round(vi * (rank_pixels[r] * spp / float(samples_step)) / (per_rank[r][1] * 1e-3) / 1e9, 2)
                                                              for r in range(world)]
        else:
            out["roofline_issue"] = {"note": "no counters: " + str(pmc_note)}
        # A fraction above 1 is not a fraction.  Where the culling grid leaves the reference's visit set so far behind that its algorithmic
        # figure exceeds the machine's peak (C5: 2 150 sphere tests per ray in the reference, 82 here), `frac` is the MACHINE fraction
        # (lane_op_frac: busy VALU lane-slots of the peak's) and the algorithmic figure stays beside it as frac_reference_work.
        if out["roofline"]["frac_reference_work"] > 1.0:
            mf = out["roofline_issue"].get("lane_op_frac")
            out["roofline"]["frac"] = mf
            out["roofline"]["note"] += ("  HERE the algorithmic figure exceeds the peak (frac_reference_work %.3f): `frac` is roofline_issue.lane_op_frac%s, "
                                        "`achieved` / `peak` still the algorithmic TFLOP/s against the spec sheet." % (out["roofline"]["frac_reference_work"], "" if mf is not None else " (null: no counter passes in this run)"))
        # the same kernel against the HBM roof (for the record: it is nowhere near it): algorithmic bytes = RNG state in + out
        # (2 x 48 B) and the vec3 written (12 B) per pixel this rank renders
        alg_bytes = (96.0 + (6.0 if cfg.get("fp16") else 12.0)) * (local_samples / spp)
        hbm = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out["roofline_hbm"] = {"bound": "hbm", "kernel": kernel_name, "achieved": round(hbm, 3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(hbm / PEAK_HBM_GBS, 6), "traffic": traffic,
                               "fetch_bytes": int(fetch_kb * 1024) if fetch_kb is not None else None,
                               "write_bytes": int(write_kb * 1024) if write_kb is not None else None,
                               "note": "96 B curandState in+out + vec3 out (12 B, fp16: 6 B) per pixel / kernel time; traffic = FETCH_SIZE + WRITE_SIZE "
                                       "per launch, measured in this run (rocprofv3 --pmc, one pass each).  FETCH_SIZE is taken as reported: the guide's x2 correction is "
                                       "calibrated for 16 B/lane streaming reads, this kernel's reads are 24 B-of-48 B strided states and L2-resident scene data "
                                       "(uncalibrated); traffic_fetch_doubled is the upper reading"}
        out["roofline_hbm"]["traffic_fetch_doubled"] = int((2 * fetch_kb + write_kb) * 1024) if traffic is not None else None
        if traffic is None:
            out["roofline_hbm"]["note"] = "96 B curandState in+out + vec3 out (12 B, fp16: 6 B) per pixel / kernel time; no counter passes in this run (%s)" % (
                pmc_note)
        if per_rank is not None:
            ks = [p[1] for p in per_rank]
            out["per_rank_render_ms"] = {"call": [p[0] for p in per_rank], "kernel": ks,
                                         "imbalance_max_over_mean": round(max(ks) / (sum(ks) / len(ks)), 4) if min(ks) > 0 else None,
                                         "note": "device time of each rank's own render_init + render (call) and render kernel, mean over the timed steps"}
        if single is not None:
            out["single_gpu_same_frame"] = single
            out["speedup_vs_single_gpu_same_frame"] = round(single["ms_per_step"] / (dt / args.steps * 1e3), 4)
            out["value_1gpu_same_workload"] = single["msamples_per_s"]      # the like-for-like reference point of `value` (the driver's N = 1 line is C3, another workload)
        if not args.no_cpu_baseline:
            threads = usable_cores()
            # hitable_list is O(N) per ray: the sample is cut so that it stays ~10 s of wall time on any core count
            if cfg.get("fp16"):
                rows, bspp = threads, 1                                # binary16 emulation on the CPU is ~30x slower
            elif cfg["spheres"] > 20000:
                rows, bspp = threads, 1
            elif cfg["octree"]:
                rows, bspp = 4 * threads, 16
            else:
                rows, bspp = 4 * threads, spp
            v, secs, smp = cpu_baseline(cfg, threads, False, rows=rows, spp=bspp)
            out["cpu_baseline"] = {"value": round(v, 5), "unit": "Msamples/s", "cores": threads, "kind": "port",
                                   "sample": "%d rows x %d px x %d spp of the same frame (%d samples), oracle hitable_list path (hitable_list.h:16-31), "
                                             "%.1f s wall on %d threads (os.cpu_count() = %s, usable = %d)"
                                             % (rows, cfg["nx"], bspp, smp, secs, threads, os.cpu_count(), threads)}
            if cfg["octree"]:
                trows, tspp = (threads, 16) if cfg["spheres"] > 20000 else (4 * threads, 8 if cfg.get("fp16") else spp)
                v2, secs2, smp2 = cpu_baseline(cfg, threads, True, rows=trows, spp=tspp)
                out["cpu_baseline_hitTree"] = {"value": round(v2, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                                               "sample": "%d rows x %d px x %d spp (%d samples), oracle hitTree path, %.1f s wall" % (trows, cfg["nx"], tspp, smp2, secs2)}
        print(json.dumps(out), flush=True)
    if watchdog is not None:
        watchdog.cancel()
    if M is not None:
        fence()
        M.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
