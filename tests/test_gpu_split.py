"""rt_split_balanced / range partitions / rt_assemble_split on the GPU (-m gpu): a frame cut into bands of equal predicted cost gives
the bits of the undivided frame (the RNG is keyed by the absolute pixel_index, main.cu:93), the cuts are reproducible, every band is
non-empty, and rt_assemble_split restores the row-major frame from the bands' compact buffers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def whole_frame(rt, torch, W, O, nx, ny, ns, precision=None):
    precision = rt.FP32 if precision is None else precision
    st = rt.alloc_rand_state(nx, ny)
    fb = rt.alloc_fb(nx, ny, precision=precision)
    rt.render_init(nx, ny, st)
    rt.render(fb, nx, ny, ns, W, st, O)
    torch.cuda.synchronize()
    return fb, st


@pytest.mark.parametrize("nx,ny,ns,n,spl,nparts,fp16", [
    (400, 232, 16, 10000, 32, 4, 0),          # octree, long-chain classification and the sorted tail on
    (203, 117, 4, 500, 0, 3, 0),              # ragged frame, hitable_list path
    (200, 120, 4, 500, 30, 2, 1),             # USE_FP16 (bounces only: the binary16 walk has no candidate grid)
    (640, 360, 32, 100000, 320, 8, 0),        # C5's scene (dense grid): eight bands
])
def test_balanced_bands_render_the_bits_of_the_whole_frame(rt, cuda, nx, ny, ns, n, spl, nparts, fp16):
    torch = cuda
    precision = rt.FP16 if fp16 else rt.FP32
    W = rt.World(n, nx, ny, precision=precision)
    O = rt.Octree(W, spl) if spl else None
    tiles = ((nx + 7) // 8) * ((ny + 7) // 8)
    starts, b, t, c = rt.split_balanced(W, O, nx, ny, nparts, counts=True)
    assert starts == rt.split_balanced(W, O, nx, ny, nparts)                     # the same cuts every time (integer arithmetic on reproducible counts)
    assert starts[0] == 0 and starts[-1] == tiles and all(starts[p + 1] > starts[p] for p in range(nparts))
    assert b.min() >= 0 and b.sum() >= 32 * (tiles - ((nx + 7) // 8) - ((ny + 7) // 8))     # every pilot sample has at least one bounce
    if spl and not fp16:
        assert t.sum() > 0 and c.sum() >= b.sum()                                # trees with a candidate grid: tests counted, a loop pass per bounce at least
    full, st_full = whole_frame(rt, torch, W, O, nx, ny, ns, precision)
    parts = rt.split_parts(starts)
    per = max(rt.part_pixels(nx, ny, p) for p in parts)
    dt = torch.float16 if fp16 else torch.float32
    staged = torch.zeros(per * 3 * nparts, dtype=dt, device="cuda")
    states = []
    for p in parts:
        st = rt.alloc_rand_state(nx, ny, p)
        fb = staged[p.part * per * 3:(p.part * per + rt.part_pixels(nx, ny, p)) * 3]
        rt.render_init(nx, ny, st, p)
        rt.render(fb, nx, ny, ns, W, st, O, p)
        states.append(st)
    out = torch.zeros(nx * ny * 3, dtype=dt, device="cuda")
    rt.assemble_split(out, staged, nx, ny, starts, per, precision)
    torch.cuda.synchronize()
    it = torch.int16 if fp16 else torch.int32
    assert torch.equal(out.view(it), full.view(it)), "the assembled bands differ from the whole frame"
    # the written-back RNG state of a band's pixel equals the whole frame's (tile-major compact against row-major)
    sf = st_full.cpu().numpy().view(np.uint32).reshape(ny, nx, 12)
    p = parts[-1]
    sp = states[-1].cpu().numpy().view(np.uint32).reshape(-1, 64, 12)
    tx = (nx + 7) // 8
    for lt in (0, sp.shape[0] // 2, sp.shape[0] - 1):
        tile = p.tile_begin + lt
        for l in (0, 27, 63):
            i, j = (tile % tx) * 8 + (l & 7), (tile // tx) * 8 + (l >> 3)
            if i < nx and j < ny:
                assert np.array_equal(sp[lt, l, :6], sf[j, i, :6])


def test_split_arguments_and_single_band(rt, cuda):
    torch = cuda
    nx, ny = 64, 40
    W = rt.World(22, nx, ny)
    tiles = 8 * 5
    assert rt.split_balanced(W, None, nx, ny, 1) == [0, tiles]
    with pytest.raises(rt.RtError):
        rt.split_balanced(W, None, nx, ny, tiles + 1)                             # more parts than tiles: no band may be empty
    st = rt.split_balanced(W, None, nx, ny, tiles)                                # exactly one tile each
    assert st == list(range(tiles + 1))
    # a range as the only part of a "split into one": compact layout, the same pixels as the whole frame
    full, _ = whole_frame(rt, torch, W, None, nx, ny, 4)
    part = rt.Partition(0, 1, 0, tiles)
    s2 = rt.alloc_rand_state(nx, ny, part); fb = rt.alloc_fb(nx, ny, part)
    rt.render_init(nx, ny, s2, part); rt.render(fb, nx, ny, 4, W, s2, None, part)
    out = torch.zeros_like(full)
    rt.assemble_split(out, fb, nx, ny, [0, tiles], tiles * 64)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int32), full.view(torch.int32))
