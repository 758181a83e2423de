"""N>1 path on CPU (not gpu): two gloo ranks split a frame by rt_multi_render's tile rule (runs of tiles round-robin), gather the compact
part buffers (equal padded sizes, one gather to rank 0 — the layout rt_multi_render's staging slots have) and rank 0 reassembles
the frame.  The renderer is replaced by a pixel-id fill (the render itself needs a GPU: tests/test_gpu_multi.py runs the real
rt_multi_render with 2 and 3 ranks); a numpy restatement of rt_assemble's index arithmetic (tests only) checks that every pixel
of the frame arrives exactly once at the right place, and the partition arithmetic is checked against rt_part_pixels."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))


PART_RUN = 64          # include/rt_amd.h RT_PART_RUN: tiles are dealt to the parts in runs of this many consecutive tiles


def tile_owner(t, nparts):
    """(part, local tile) of global tile t — the split rt_amd.h's rt_partition describes"""
    run = t // PART_RUN
    return run % nparts, (run // nparts) * PART_RUN + t % PART_RUN


def part_pixels(nx, ny, part, nparts):
    """element count of the compact buffer of one part (what rt_part_pixels returns)"""
    if nparts == 1:
        return nx * ny
    tx, ty = (nx + 7) // 8, (ny + 7) // 8
    return sum(1 for t in range(tx * ty) if tile_owner(t, nparts)[0] == part) * 64


def padded_part_pixels(nx, ny, nparts):
    """the staging slot size: part 0 is the largest"""
    return part_pixels(nx, ny, 0, nparts)


def local_pixel_ids(nx, ny, part, nparts, padded):
    """tile-major compact buffer of pixel_index values for one part (-1 = padding / outside the frame)"""
    tx, ty = (nx + 7) // 8, (ny + 7) // 8
    buf = np.full(padded, -1, np.int64)
    for t in range(tx * ty):
        owner, lt = tile_owner(t, nparts)
        if owner != part:
            continue
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        for lane in range(64):
            i, j = x0 + (lane & 7), y0 + (lane >> 3)
            if i < nx and j < ny:
                buf[lt * 64 + lane] = j * nx + i
    return buf


def assemble_numpy(parts, nx, ny, nparts, per):
    """numpy restatement of k_assemble (tests only)"""
    tx, ty = (nx + 7) // 8, (ny + 7) // 8
    full = np.full(nx * ny, -7, np.int64)
    for t in range(tx * ty):
        for lane in range(64):
            i, j = (t % tx) * 8 + (lane & 7), (t // tx) * 8 + (lane >> 3)
            if i < nx and j < ny:
                owner, lt = tile_owner(t, nparts)
                full[j * nx + i] = parts[owner * per + lt * 64 + lane]
    return full


def _worker(rank, world, port, nx, ny, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per = padded_part_pixels(nx, ny, world)
        mine = part_pixels(nx, ny, rank, world)
        ids = local_pixel_ids(nx, ny, rank, world, per)
        assert (ids[mine:] == -1).all()
        send = torch.from_numpy(ids.astype(np.float64))
        got = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, got, dst=0)
        ok = True
        if rank == 0:
            parts = torch.cat(got).numpy().astype(np.int64)
            full = assemble_numpy(parts, nx, ny, world, per)
            ok = bool(np.array_equal(full, np.arange(nx * ny)))
        else:
            ok = got is None
        dist.barrier()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


# two ranks: 1.4 runs of tiles, less than one run (rank 1 owns nothing), 31 runs; eight ranks (the north star's node): 31 runs = three
# full rounds and an incomplete one, parts of different sizes
@pytest.mark.parametrize("world,nx,ny", [(2, 100, 52), (2, 61, 35), (2, 640, 200), (8, 640, 200)])
def test_tile_split_gather_assemble_over_gloo(world, nx, ny):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nx, ny, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def test_partition_sizes_match_library(rt):
    assert rt.PART_RUN == PART_RUN
    for nx, ny in ((1200, 800), (61, 35), (3840, 2160), (8, 8), (300, 9)):
        for nparts in (1, 2, 3, 4, 8):
            sizes = [rt.part_pixels(nx, ny, rt.Partition(p, nparts)) for p in range(nparts)]
            assert sizes == [part_pixels(nx, ny, p, nparts) for p in range(nparts)]
            assert max(sizes) == sizes[0] == padded_part_pixels(nx, ny, nparts)
            tiles = ((nx + 7) // 8) * ((ny + 7) // 8)
            assert sum(sizes) == (tiles * 64 if nparts > 1 else nx * ny)
