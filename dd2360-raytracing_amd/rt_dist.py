"""Multi-GPU split of one frame: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU
node, "gloo" in the CPU tests).  No reference counterpart exists (the reference is single-GPU); the unit of work is the
reference's 8x8 pixel block (main.cu:351-352).

Pixels are independent (the per-pixel RNG is keyed by the absolute pixel_index, main.cu:93), so any partition gives the
same bits.  Tiles are dealt round-robin (tile t -> rank t % world) so that cheap sky tiles and expensive ground tiles mix
on every GPU; each rank renders its tiles into a compact tile-major buffer (include/rt_amd.h, rt_partition) and ONE
gather brings the buffers to rank 0, where rt_assemble restores the row-major frame.
"""
import math


def scaled_frame(nx, ny, n):
    """weak scaling: same aspect, ~n times the pixels"""
    if n == 1:
        return nx, ny
    s = math.sqrt(n)
    return int(round(nx * s)), int(round(ny * s))


def tiles(nx, ny):
    return (nx + 7) // 8, (ny + 7) // 8


def part_pixels(nx, ny, part, nparts):
    """element count of the compact buffer of one part (mirrors rt_part_pixels)"""
    if nparts == 1:
        return nx * ny
    tx, ty = tiles(nx, ny)
    return (tx * ty - part + nparts - 1) // nparts * 64


def padded_part_pixels(nx, ny, nparts):
    """every rank sends this many pixels (the size of part 0, the largest) so the gather has equal counts"""
    return part_pixels(nx, ny, 0, nparts)


def gather_parts(dist, send, rank, world, dst=0):
    """the single framebuffer exchange: equal-size part buffers -> list on dst (None elsewhere)"""
    import torch
    out = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, out, dst=dst)
    return out
