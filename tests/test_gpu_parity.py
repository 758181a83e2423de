"""GPU parity tests (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same seeds.
Bar: bit-exact for fp32 (hit records, float framebuffers, RNG state, PPM bytes)."""
import hashlib

import numpy as np
import pytest

from oracle_lib import OracleScene, ppm_bytes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def gpu_render(rt, torch, W, O, nx, ny, ns, part=None):
    part = part or rt.WHOLE
    st = rt.alloc_rand_state(nx, ny, part)
    fb = rt.alloc_fb(nx, ny, part)
    rt.render_init(nx, ny, st, part)
    rt.render(fb, nx, ny, ns, W, st, O, part)
    torch.cuda.synchronize()
    return fb, st


def random_rays(n, seed):
    """rays that exercise the scene: origins around/inside the sphere field, at the camera, far away on the ground;
    directions random, some axis-aligned (zero components -> inf/NaN slab arithmetic)."""
    rng = np.random.default_rng(seed)
    o = np.empty((n, 3), np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    k = n // 4
    o[:k] = rng.uniform([-11, 0, -11], [11, 2, 11], (k, 3))
    o[k:2 * k] = np.array([13, 2, 3], np.float32) + rng.normal(scale=0.05, size=(k, 3))
    tgt = rng.uniform([-11, 0, -11], [11, 0.3, 11], (k, 3))
    d[k:2 * k] = tgt - o[k:2 * k]
    o[2 * k:3 * k] = rng.uniform([-300, 0, -300], [300, 40, 300], (k, 3))
    tgt = rng.uniform([-11, 0, -11], [11, 1, 11], (k, 3))
    d[2 * k:3 * k] = tgt - o[2 * k:3 * k]
    o[3 * k:] = rng.uniform([-12, -0.5, -12], [12, 3, 12], (n - 3 * k, 3))
    # axis-aligned / zero-component directions
    z = rng.integers(0, n, n // 50)
    d[z, rng.integers(0, 3, z.size)] = 0.0
    z = rng.integers(0, n, n // 100)
    d[z] = 0.0
    d[z, rng.integers(0, 3, z.size)] = rng.choice([-1.0, 1.0], z.size)
    return np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)


@pytest.mark.parametrize("n,spl", [(22, 30), (500, 30), (10000, 32)])
@pytest.mark.parametrize("tree", ["list", "list_reference", "tree_reference", "tree_fast"])
def test_trace_hit_records(rt, cuda, n, spl, tree):
    """hitTree / hitable_list::hit (both traversal modes each): per-ray hit records bit-identical to the oracle."""
    torch = cuda
    nrays = 200_000 if n <= 500 else 60_000
    rays = random_rays(nrays, 1234 + n)
    W = rt.World(n, 1200, 800)
    O = rt.Octree(W, spl) if not tree.startswith("list") else None
    if O is not None:
        O.set_traversal(rt.TRAVERSAL_FAST if tree == "tree_fast" else rt.TRAVERSAL_REFERENCE)
    else:
        assert W.list_accel_info()["enabled"] == (n >= 500)        # the candidate grid serves lists from 64 spheres on
        W.set_list_traversal(rt.TRAVERSAL_REFERENCE if tree == "list_reference" else rt.TRAVERSAL_FAST)
    tree = not tree.startswith("list")
    d_rays = torch.from_numpy(rays).cuda()
    d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
    rt.trace_rays(W, O, d_rays, nrays, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(rt.hit_record_dtype)
    ref = OracleScene(n, 1200, 800, use_octree=tree, spl=spl).trace(rays, mode=2 if tree else 1)
    assert np.array_equal(got["sphere"], ref["sphere"])
    assert ref["hit"].sum() > nrays // 20
    assert np.array_equal(bits(got["t"]), bits(ref["t"]))
    assert np.array_equal(bits(got["p"]), bits(ref["p"]))
    assert np.array_equal(bits(got["normal"]), bits(ref["normal"]))


def test_trace_hit_records_on_a_dense_grid(rt, cuda):
    """C5's world (100 000 spheres, SPHERES_PER_LEAF 320: ~30 entries per grid cell) takes the pooled walk of dense grids
    (walk_pool_dense through k_trace<true,2>; the kernel name says which variant the library chose): hit records bit-identical to the
    oracle's hitTree, and to the library's own reference scan."""
    torch = cuda
    n, spl, nrays = 100000, 320, 40_000
    rays = random_rays(nrays, 99)
    W = rt.World(n, 3840, 2160)
    O = rt.Octree(W, spl)
    assert rt.render_kernel_name(W, O, 0) == "k_render<true,0,2>"
    d_rays = torch.from_numpy(rays).cuda()
    outs = []
    for mode in (rt.TRAVERSAL_FAST, rt.TRAVERSAL_REFERENCE):
        O.set_traversal(mode)
        d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
        rt.trace_rays(W, O, d_rays, nrays, d_out)
        torch.cuda.synchronize()
        outs.append(d_out.cpu().numpy().view(rt.hit_record_dtype))
    ref = OracleScene(n, 3840, 2160, use_octree=True, spl=spl).trace(rays, mode=2)
    for got in outs:
        assert np.array_equal(got["sphere"], ref["sphere"])
        assert np.array_equal(bits(got["t"]), bits(ref["t"]))
        assert np.array_equal(bits(got["p"]), bits(ref["p"]))
        assert np.array_equal(bits(got["normal"]), bits(ref["normal"]))
    assert ref["hit"].sum() > nrays // 20


@pytest.mark.parametrize("n,nx,ny,ns,tree,spl", [
    (22, 64, 36, 4, False, 30), (22, 64, 36, 4, True, 30),
    (500, 64, 36, 4, False, 30), (500, 61, 35, 3, True, 30),      # ragged: not a multiple of the 8x8 tile
    (10000, 48, 32, 2, True, 32), (10000, 48, 32, 2, False, 32),
    (5, 40, 24, 16, True, 30), (5, 40, 24, 16, False, 30),       # the smallest world create_world accepts; 16 spp: long-chain pass on
])
def test_render_small_frames(rt, cuda, n, nx, ny, ns, tree, spl):
    """render(): float framebuffer and written-back RNG state bit-identical to the oracle."""
    torch = cuda
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl) if tree else None
    fb, st = gpu_render(rt, torch, W, O, nx, ny, ns)
    S = OracleScene(n, nx, ny, use_octree=tree, spl=spl)
    ref, ref_st = S.render(ns, nthreads=8)
    got = fb.cpu().numpy().reshape(ny, nx, 3)
    assert np.array_equal(bits(got), bits(ref))
    got_st = st.cpu().numpy().view(np.uint32).reshape(-1, 12)
    assert np.array_equal(got_st[:, :6], ref_st[:, :6])


def test_fast_traversal_equals_reference_scan_many_rays(rt, cuda):
    """4M rays (scene-like, grazing, axis-aligned, far origins) through both traversal modes on the GPU: identical records.
    The reference-scan mode itself is checked against the oracle above."""
    torch = cuda
    for n, spl, radius in ((10000, 32, 0.1), (2000, 30, 0.2), (100000, 320, 0.1)):
        W = rt.World(n, 1200, 800, sphere_radius=radius)
        O = rt.Octree(W, spl)
        nrays = 1_000_000
        rays = random_rays(nrays, 99 + n)
        # add rays skimming the sphere layer and rays starting on sphere surfaces (secondary-ray like)
        rng = np.random.default_rng(5)
        k = nrays // 4
        rays[:k, 0:3] = rng.uniform([-11, 0.0, -11], [11, 0.25, 11], (k, 3))
        rays[:k, 3:6] = rng.normal(size=(k, 3)) * np.array([1, 0.05, 1])
        d_rays = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).cuda()
        outs = []
        for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_FAST):
            O.set_traversal(mode)
            d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
            rt.trace_rays(W, O, d_rays, nrays, d_out)
            torch.cuda.synchronize()
            outs.append(d_out.cpu().numpy().view(np.uint32).reshape(nrays, 8))
        bad = np.nonzero((outs[0] != outs[1]).any(axis=1))[0]
        assert bad.size == 0, "n=%d: %d rays differ, first %s" % (n, bad.size, rays[bad[:3]])
        hits = (outs[0][:, 7].view(np.int32) >= 0).mean()
        assert hits > 0.2
        if n > 10000:
            continue
        # ... and hitable_list::hit: every sphere in list order against the candidate grid
        assert W.list_accel_info()["enabled"]
        outs = []
        for mode in (rt.TRAVERSAL_REFERENCE, rt.TRAVERSAL_FAST):
            W.set_list_traversal(mode)
            d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
            rt.trace_rays(W, None, d_rays, nrays, d_out)
            torch.cuda.synchronize()
            outs.append(d_out.cpu().numpy().view(np.uint32).reshape(nrays, 8))
        bad = np.nonzero((outs[0] != outs[1]).any(axis=1))[0]
        assert bad.size == 0, "list, n=%d: %d rays differ, first %s" % (n, bad.size, rays[bad[:3]])


def test_c1_ppm_md5(rt, cuda):
    """BASELINE config 1 (400x225, 4 spp, N=22, list): PPM bytes equal the oracle's and the SURVEY §8c probe md5."""
    torch = cuda
    nx, ny, ns = 400, 225, 4
    W = rt.World(22, nx, ny)
    fb, _ = gpu_render(rt, torch, W, None, nx, ny, ns)
    got = fb.cpu().numpy().reshape(ny, nx, 3)
    ppm = rt.format_ppm(got, nx, ny)
    assert hashlib.md5(ppm).hexdigest() == "bb5ebdd40d476c6a48e7de6a3af3e3ed"
    ref, _ = OracleScene(22, nx, ny).render(ns, nthreads=8)
    assert ppm == ppm_bytes(ref)


def test_render_init_states(rt, cuda):
    torch = cuda
    nx, ny = 70, 33
    st = rt.alloc_rand_state(nx, ny)
    rt.render_init(nx, ny, st)
    torch.cuda.synchronize()
    got = st.cpu().numpy().view(np.uint32).reshape(-1, 12)
    ref = OracleScene(22, nx, ny).render_init()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("nparts", [2, 3, 8])
def test_partition_assemble_equals_whole(rt, cuda, nparts):
    """tile partition (multi-GPU split) + rt_assemble reproduces the single-call frame bit for bit.  700 x 330 = 3696 tiles = 57.75
    runs of RT_PART_RUN tiles: several rounds of runs, an incomplete last round and a cut last run, ragged edge tiles."""
    torch = cuda
    nx, ny, ns, n = 700, 330, 3, 500
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, 30)
    whole, _ = gpu_render(rt, torch, W, O, nx, ny, ns)
    per = rt.part_pixels(nx, ny, rt.Partition(0, nparts))
    parts = torch.zeros(nparts * per * 3, dtype=torch.float32, device="cuda")
    for p in range(nparts):
        part = rt.Partition(p, nparts)
        fb, _ = gpu_render(rt, torch, W, O, nx, ny, ns, part)
        parts[p * per * 3: p * per * 3 + fb.numel()] = fb
    full = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
    rt.assemble(full, parts, nx, ny, nparts)
    torch.cuda.synchronize()
    assert torch.equal(full.view(torch.int32), whole.view(torch.int32))


@pytest.mark.parametrize("nparts", [1, 3])
def test_long_chain_selection_on_ragged_and_partitioned_frames(rt, cuda, nparts):
    """From 16 samples per pixel on, the pilot pass and k_long_select pre-classify long chains from the pilot counts of a 2x2 block
    and its eight neighbours.  A ragged frame (603 x 403: edge tiles with pixels outside), whole and as three parts whose tiles'
    neighbours belong to the other parts: every pixel is rendered exactly once, bit-equal to the oracle (frame and RNG state),
    and the scheduling counters show that chains were pre-classified and all handed out."""
    torch = cuda
    nx, ny, ns, n = 603, 403, 16, 500
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, 30)
    ref, ref_st = OracleScene(n, nx, ny, use_octree=True, spl=30).render(ns, nthreads=8)
    per = rt.part_pixels(nx, ny, rt.Partition(0, nparts))
    parts = torch.zeros(nparts * per * 3, dtype=torch.float32, device="cuda")
    n_long = 0
    for p in range(nparts):
        part = rt.Partition(p, nparts)
        fb, st = gpu_render(rt, torch, W, O, nx, ny, ns, part)
        c = W.render_counters()
        assert c["thin_waves"] == 0
        assert c["long_handles"] >= c["long_chains"]              # every pre-classified chain was taken (handles past the end find the list empty)
        n_long += c["long_chains"]
        if nparts == 1:
            got = fb.cpu().numpy().reshape(ny, nx, 3)
            assert np.array_equal(bits(got), bits(ref))
            assert np.array_equal(st.cpu().numpy().view(np.uint32).reshape(-1, 12)[:, :6], ref_st[:, :6])
        else:
            parts[p * per * 3: p * per * 3 + fb.numel()] = fb
    if nparts > 1:
        full = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
        rt.assemble(full, parts, nx, ny, nparts)
        torch.cuda.synchronize()
        assert np.array_equal(bits(full.cpu().numpy().reshape(ny, nx, 3)), bits(ref))
    assert 0 < n_long <= nx * ny // 64, n_long


def test_render_progressive(rt, cuda):
    """render_progressive: accumulation and RNG continuation equal the oracle's after 3 passes."""
    torch = cuda
    nx, ny, n = 56, 40, 500
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, 30)
    st = rt.alloc_rand_state(nx, ny)
    fb = rt.alloc_fb(nx, ny)
    rt.render_init(nx, ny, st)
    S = OracleScene(n, nx, ny, use_octree=True, spl=30)
    ref_st = S.render_init()
    ref = np.zeros((ny, nx, 3), np.float32)
    for k in range(1, 4):
        rt.render_progressive(fb, nx, ny, k, W, st, O)
        S.render_progressive(ref, k, ref_st, nthreads=8)
    torch.cuda.synchronize()
    assert np.array_equal(bits(fb.cpu().numpy().reshape(ny, nx, 3)), bits(ref))
    assert np.array_equal(st.cpu().numpy().view(np.uint32).reshape(-1, 12)[:, :6], ref_st[:, :6])


def test_full_size_properties_c3(rt, cuda):
    """BASELINE config 3 at full size (1200x800x64, N=10000, octree SPL 32): properties that do not need the whole
    oracle frame — sampled rows equal the oracle bit for bit, the run is deterministic, the partitioned render
    equals the whole one, and the RNG state after render(64) equals the state after 64 progressive passes on
    a sub-frame."""
    torch = cuda
    nx, ny, ns, n, spl = 1200, 800, 64, 10000, 32
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl)
    fb, st = gpu_render(rt, torch, W, O, nx, ny, ns)
    fb2, _ = gpu_render(rt, torch, W, O, nx, ny, ns)
    assert torch.equal(fb.view(torch.int32), fb2.view(torch.int32))
    O.set_traversal(rt.TRAVERSAL_REFERENCE)                    # the exact bucket scan gives the same 61 M samples
    fb3, st3 = gpu_render(rt, torch, W, O, nx, ny, ns)
    O.set_traversal(rt.TRAVERSAL_FAST)
    assert torch.equal(fb.view(torch.int32), fb3.view(torch.int32)) and torch.equal(st, st3)
    got = fb.cpu().numpy().reshape(ny, nx, 3)
    # NaN pixels are legitimate reference semantics (dielectric::scatter takes sqrt of a negative number,
    # material.h:95) — they must simply match the oracle bit for bit like everything else.
    finite = got[np.isfinite(got)]
    assert finite.min() >= 0.0 and finite.max() <= 1.0 and finite.size > 0.99 * got.size
    S = OracleScene(n, nx, ny, use_octree=True, spl=spl)
    # rows 316/317 hold the frame's longest chains (crevices: 2500 bounces per pixel) — the pixels that travel through the
    # thin-wave / cooperative-walk machinery of the render kernel
    for row in (3, 100, 250, 316, 317, 431, 600, 797):
        ref, _ = S.render(ns, row0=row, rows=1, nthreads=4)
        assert np.array_equal(bits(got[row]), bits(ref[0])), "row %d differs" % row
    st_host = st.cpu().numpy().view(np.uint32).reshape(-1, 12)
    _, st_ref = S.render(ns, row0=316, rows=2, nthreads=4)
    assert np.array_equal(st_host[316 * nx: 318 * nx, :6], st_ref[:, :6])          # RNG state written back (main.cu:110)
    nparts = 4
    per = rt.part_pixels(nx, ny, rt.Partition(0, nparts))
    parts = torch.zeros(nparts * per * 3, dtype=torch.float32, device="cuda")
    for p in range(nparts):
        f, _ = gpu_render(rt, torch, W, O, nx, ny, ns, rt.Partition(p, nparts))
        parts[p * per * 3: p * per * 3 + f.numel()] = f
    full = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
    rt.assemble(full, parts, nx, ny, nparts)
    torch.cuda.synchronize()
    assert torch.equal(full.view(torch.int32), fb.view(torch.int32))


def test_list_equals_octree_c2_scene(rt, cuda):
    """with fp32 the octree image equals the linear-list image (N=500, full width, 8 spp)."""
    torch = cuda
    nx, ny, ns, n = 1200, 800, 8, 500
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, 30)
    a, _ = gpu_render(rt, torch, W, None, nx, ny, ns)
    b, _ = gpu_render(rt, torch, W, O, nx, ny, ns)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # the list path again, every sphere in list order instead of the candidate grid: the same frame and RNG states
    W.set_list_traversal(rt.TRAVERSAL_REFERENCE)
    c, st_c = gpu_render(rt, torch, W, None, nx, ny, ns)
    W.set_list_traversal(rt.TRAVERSAL_FAST)
    d, st_d = gpu_render(rt, torch, W, None, nx, ny, ns)
    assert torch.equal(a.view(torch.int32), c.view(torch.int32)) and torch.equal(c.view(torch.int32), d.view(torch.int32))
    assert torch.equal(st_c.view(torch.uint8), st_d.view(torch.uint8))


# ---------------------------------------------------------------------------------------------------- USE_FP16
# Tolerance statement (BASELINE config 4): the GPU fp16 path and the oracle implement the same contract (every
# real_t operator = float op + one rounding to binary16), so the test bar is 0 ulp — bit-exact binary16 channels.
# Against a real CUDA build of the reference no bit-exactness can be claimed (hsin/hcos/__hdiv are approximations
# there); the documented expectation is >= 99 % of channels within +-1/255.
def half_bits(t):
    return t.cpu().numpy().view(np.uint16)


def f32_to_half_bits(a):
    return np.ascontiguousarray(a, np.float32).astype(np.float16).view(np.uint16)


@pytest.mark.parametrize("n,nx,ny,ns,tree,spl", [
    (500, 48, 32, 2, False, 30), (500, 61, 35, 3, True, 30), (9805, 48, 32, 2, True, 32), (22, 64, 36, 4, True, 30),
])
def test_fp16_render_small_frames(rt, cuda, n, nx, ny, ns, tree, spl):
    torch = cuda
    W = rt.World(n, nx, ny, precision=rt.FP16)
    O = rt.Octree(W, spl) if tree else None
    st = rt.alloc_rand_state(nx, ny)
    fb = rt.alloc_fb(nx, ny, precision=rt.FP16)
    rt.render_init(nx, ny, st)
    rt.render(fb, nx, ny, ns, W, st, O)
    torch.cuda.synchronize()
    ref, ref_st = OracleScene(n, nx, ny, fp16=True, use_octree=tree, spl=spl).render(ns, nthreads=8)
    got = half_bits(fb).reshape(ny, nx, 3)
    want = f32_to_half_bits(ref)                      # the oracle returns exact float images of its binary16 values
    nan = np.isnan(ref)
    assert np.array_equal(got[~nan], want[~nan])
    assert np.isnan(got.view(np.float16)[nan]).all()
    assert np.array_equal(st.cpu().numpy().view(np.uint32).reshape(-1, 12)[:, :6], ref_st[:, :6])


@pytest.mark.parametrize("tree", [False, True])
def test_fp16_trace_hit_records(rt, cuda, tree):
    torch = cuda
    n, spl, nrays = 500, 30, 50_000
    rays = random_rays(nrays, 77).astype(np.float16).astype(np.float32)      # binary16-representable rays
    W = rt.World(n, 1200, 800, precision=rt.FP16)
    O = rt.Octree(W, spl) if tree else None
    d_rays = torch.from_numpy(rays).cuda()
    d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
    rt.trace_rays(W, O, d_rays, nrays, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(rt.hit_record_dtype)
    ref = OracleScene(n, 1200, 800, fp16=True, use_octree=tree, spl=spl).trace(rays, mode=2 if tree else 1)
    assert np.array_equal(got["sphere"], ref["sphere"])
    assert ref["hit"].sum() > nrays // 50
    assert np.array_equal(bits(got["t"]), bits(ref["t"]))
    assert np.array_equal(bits(got["normal"]), bits(ref["normal"]))


def test_fp16_rays_that_pass_every_node_of_the_tree(rt, cuda):
    """The binary16 walk pools the level-2 node expansions of a wave's rays (rt_kernels_fp16.hip, closest_tree): a task pool of 384
    entries, segment pools of 128 / 256.  Rays from (-49152, -49152, -49152): in binary16 every box plane of the tree lies at the same
    rounded distance from such an origin (spacing 32 up there), every slab interval degenerates to one point, and with equal direction
    components all three coincide — EVERY node passes intersect_ray_aabb (acceleration_structure.h:226-244: closed comparisons), every
    leaf is visited, whole waves of such rays overflow the task pool and the segment pools many times over.  The records (and those of
    ordinary rays in the same launch, and of rays degenerate on one or two axes only) equal the oracle's hitTree, bit for bit."""
    torch = cuda
    n, spl = 10000, 32
    rng = np.random.default_rng(5)
    far = np.float32(-49152.0)
    blocks = []
    a = np.tile(np.array([far, far, far, 1, 1, 1], np.float32), (512, 1)); blocks.append(a)                      # every node passes
    b = random_rays(2048, 11); b[:, 0] = far; b[:, 3] = np.abs(b[:, 3]) + 0.5; blocks.append(b)                  # degenerate on x only
    c = random_rays(2048, 12); c[:, 0] = far; c[:, 2] = far; c[:, 3] = 1.0; c[:, 5] = 1.0; blocks.append(c)       # on x and z, equal parameters
    d = np.tile(np.array([far, far, far, 1, 1, 1], np.float32), (256, 1)); d[:, 3:] *= rng.choice([0.5, 1.0, 2.0], (256, 3)).astype(np.float32); blocks.append(d)
    blocks.append(random_rays(20000, 13))
    rays = np.ascontiguousarray(np.concatenate(blocks), np.float32).astype(np.float16).astype(np.float32)
    nrays = rays.shape[0]
    W = rt.World(n, 1200, 800, precision=rt.FP16)
    O = rt.Octree(W, spl)
    d_rays = torch.from_numpy(rays).cuda()
    d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
    rt.trace_rays(W, O, d_rays, nrays, d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(rt.hit_record_dtype)
    ref = OracleScene(n, 1200, 800, fp16=True, use_octree=True, spl=spl).trace(rays, mode=2)
    assert np.array_equal(got["sphere"], ref["sphere"])
    assert np.array_equal(bits(got["t"]), bits(ref["t"])) and np.array_equal(bits(got["normal"]), bits(ref["normal"]))
    assert ref["hit"][512 + 4352:].sum() > 1000                         # the ordinary rays of the launch do hit


def test_fp16_progressive_and_partition(rt, cuda):
    torch = cuda
    nx, ny, n = 56, 40, 500
    W = rt.World(n, nx, ny, precision=rt.FP16)
    O = rt.Octree(W, 30)
    st = rt.alloc_rand_state(nx, ny)
    fb = rt.alloc_fb(nx, ny, precision=rt.FP16)
    rt.render_init(nx, ny, st)
    S = OracleScene(n, nx, ny, fp16=True, use_octree=True, spl=30)
    ref_st = S.render_init()
    ref = np.zeros((ny, nx, 3), np.float32)
    for k in range(1, 4):
        rt.render_progressive(fb, nx, ny, k, W, st, O)
        S.render_progressive(ref, k, ref_st, nthreads=8)
    torch.cuda.synchronize()
    nan = np.isnan(ref)
    assert np.array_equal(half_bits(fb).reshape(ny, nx, 3)[~nan], f32_to_half_bits(ref)[~nan])
    # partitioned render + assemble == whole
    ns, nparts = 3, 3
    whole = rt.alloc_fb(nx, ny, precision=rt.FP16)
    st2 = rt.alloc_rand_state(nx, ny)
    rt.render_init(nx, ny, st2)
    rt.render(whole, nx, ny, ns, W, st2, O)
    per = rt.part_pixels(nx, ny, rt.Partition(0, nparts))
    parts = torch.zeros(nparts * per * 3, dtype=torch.float16, device="cuda")
    for p in range(nparts):
        part = rt.Partition(p, nparts)
        stp = rt.alloc_rand_state(nx, ny, part)
        fbp = rt.alloc_fb(nx, ny, part, precision=rt.FP16)
        rt.render_init(nx, ny, stp, part)
        rt.render(fbp, nx, ny, ns, W, stp, O, part)
        parts[p * per * 3: p * per * 3 + fbp.numel()] = fbp
    full = torch.zeros(nx * ny * 3, dtype=torch.float16, device="cuda")
    rt.assemble(full, parts, nx, ny, nparts, precision=rt.FP16)
    torch.cuda.synchronize()
    assert torch.equal(full.view(torch.int16), whole.view(torch.int16))


@pytest.mark.parametrize("nparts", [1, 3])
def test_fp16_scheduled_render_on_ragged_and_partitioned_frames(rt, cuda, nparts):
    """USE_FP16 from 16 samples per pixel on: pilot pass in binary16 (k_tile_cost_h), longest-first order, interleaved slots, long
    chains first in thin waves.  A ragged frame, whole and in three parts: binary16 framebuffer and RNG state bit-equal to the fp16
    oracle, and chains were pre-classified."""
    torch = cuda
    nx, ny, ns, n = 403, 301, 16, 500
    W = rt.World(n, nx, ny, precision=rt.FP16)
    O = rt.Octree(W, 30)
    ref, ref_st = OracleScene(n, nx, ny, fp16=True, use_octree=True, spl=30).render(ns, nthreads=8)
    per = rt.part_pixels(nx, ny, rt.Partition(0, nparts))
    parts = torch.zeros(nparts * per * 3, dtype=torch.float16, device="cuda")
    n_long = 0
    for p in range(nparts):
        part = rt.Partition(p, nparts)
        st = rt.alloc_rand_state(nx, ny, part)
        fb = rt.alloc_fb(nx, ny, part, precision=rt.FP16)
        rt.render_init(nx, ny, st, part)
        rt.render(fb, nx, ny, ns, W, st, O, part)
        torch.cuda.synchronize()
        c = W.render_counters()
        assert c["long_handles"] >= c["long_chains"]
        n_long += c["long_chains"]
        if nparts == 1:
            nan = np.isnan(ref)
            assert np.array_equal(half_bits(fb).reshape(ny, nx, 3)[~nan], f32_to_half_bits(ref)[~nan])
            assert np.array_equal(st.cpu().numpy().view(np.uint32).reshape(-1, 12)[:, :6], ref_st[:, :6])
        else:
            parts[p * per * 3: p * per * 3 + fb.numel()] = fb
    if nparts > 1:
        full = torch.zeros(nx * ny * 3, dtype=torch.float16, device="cuda")
        rt.assemble(full, parts, nx, ny, nparts, precision=rt.FP16)
        torch.cuda.synchronize()
        nan = np.isnan(ref)
        assert np.array_equal(half_bits(full).reshape(ny, nx, 3)[~nan], f32_to_half_bits(ref)[~nan])
    assert 0 < n_long <= nx * ny // 64, n_long


def test_fp16_c4_properties_and_psnr(rt, cuda):
    """BASELINE config 4 geometry (1200x800, N=10000 octree SPL 32, fp16) at 8 spp: deterministic, sampled rows equal
    the oracle, and the fp16-vs-fp32 distance is in the band the reference reports (PSNR 12.4 dB, evaluations.ipynb:1076)."""
    torch = cuda
    nx, ny, ns, n, spl = 1200, 800, 8, 10000, 32
    W = rt.World(n, nx, ny, precision=rt.FP16)
    O = rt.Octree(W, spl)
    outs = []
    for _ in range(2):
        st = rt.alloc_rand_state(nx, ny)
        fb = rt.alloc_fb(nx, ny, precision=rt.FP16)
        rt.render_init(nx, ny, st)
        rt.render(fb, nx, ny, ns, W, st, O)
        torch.cuda.synchronize()
        outs.append(fb)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    got = half_bits(outs[0]).reshape(ny, nx, 3)
    S = OracleScene(n, nx, ny, fp16=True, use_octree=True, spl=spl)
    for row in (5, 300, 640):
        ref, _ = S.render(ns, row0=row, rows=1, nthreads=1)
        nan = np.isnan(ref[0])
        assert np.array_equal(got[row][~nan], f32_to_half_bits(ref[0])[~nan]), "row %d" % row
    W32 = rt.World(n, nx, ny)
    O32 = rt.Octree(W32, spl)
    a, _ = gpu_render(rt, torch, W32, O32, nx, ny, ns)
    a = np.clip(np.nan_to_num(a.cpu().numpy()), 0, 1)
    b = np.clip(np.nan_to_num(outs[0].float().cpu().numpy()), 0, 1)
    psnr = 10 * np.log10(1.0 / float(np.mean((a - b) ** 2)))
    assert 8.0 < psnr < 18.0, psnr


# ---------------------------------------------------------------------------------------------------- arbitrary worlds
def random_world(rt, seed, n, nx, ny, big=6, air=0.3, outside=0.05, ghosts=0.03):
    """a caller-built world (not create_world): spheres of many sizes, in the air, overlapping, some outside the
    octree's root box (dropped with the reference's 'not in range' message), some ghost slots, every material"""
    rng = np.random.default_rng(seed)
    sp = np.zeros(n, rt.sphere_dtype)
    sp["center"][:, 0] = rng.uniform(-10.5, 10.5, n)
    sp["center"][:, 2] = rng.uniform(-10.5, 10.5, n)
    sp["radius"] = rng.choice([0.03, 0.05, 0.1, 0.1, 0.1, 0.15, 0.2, 0.3], n)
    sp["center"][:, 1] = sp["radius"]
    up = rng.random(n) < air
    sp["center"][up, 1] = rng.uniform(0.0, 1.9, up.sum())
    bigs = rng.choice(np.arange(1, n), big, replace=False)
    sp["radius"][bigs] = rng.uniform(0.6, 1.4, big)
    sp["center"][bigs, 1] = rng.uniform(0.2, 1.2, big)
    out = rng.random(n) < outside
    sp["center"][out, 0] = rng.uniform(11.5, 14.0, out.sum())
    sp["material"] = rng.choice([rt.MAT_LAMBERTIAN, rt.MAT_LAMBERTIAN, rt.MAT_METAL, rt.MAT_DIELECTRIC], n)
    sp["albedo"] = rng.uniform(0.05, 1.0, (n, 3))
    sp["param"] = np.where(sp["material"] == rt.MAT_METAL, rng.uniform(0, 1, n), np.where(sp["material"] == rt.MAT_DIELECTRIC, 1.5, 0.0))
    g = rng.random(n) < ghosts
    g[0] = False
    sp["material"][g] = rt.MAT_NONE
    sp["center"][g] = 0; sp["radius"][g] = 0; sp["albedo"][g] = 0; sp["param"][g] = 0
    sp[0] = ((0.0, -1000.0, -1.0), 1000.0, rt.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)      # ground stays slot 0 (hitTree tests it first)
    cam = rt.camera_init((rng.uniform(8, 14), rng.uniform(1, 4), rng.uniform(-4, 4)), (0, 0.3, 0), (0, 1, 0), 35.0,
                         np.float32(nx) / np.float32(ny), 0.05, 10.0)
    return sp, cam


def oracle_of(sp, cam, n, nx, ny, tree, spl):
    geom = np.concatenate([sp["center"], sp["radius"][:, None]], 1)
    mat = np.concatenate([sp["albedo"], sp["param"][:, None]], 1)
    return OracleScene(n, nx, ny, use_octree=tree, spl=spl, custom=(geom, mat, sp["material"], cam.view(np.float32).ravel()))


@pytest.mark.parametrize("seed,n,spl", [(1, 300, 30), (2, 3000, 40), (3, 12000, 64)])
def test_arbitrary_world_hit_records_and_frames(rt, cuda, seed, n, spl):
    """Worlds the caller builds itself: hit records (list, tree scan, tree fast) and a small frame equal the oracle."""
    torch = cuda
    nx, ny, ns = 72, 48, 3
    sp, cam = random_world(rt, seed, n, nx, ny)
    W = rt.World(n, nx, ny, spheres=sp, camera=cam)
    O = rt.Octree(W, spl)
    assert O.info()["dropped_outside"] > 0                       # the "not in range" path is exercised
    nrays = 150_000
    rays = random_rays(nrays, 500 + seed)
    d_rays = torch.from_numpy(rays).cuda()
    S_tree = oracle_of(sp, cam, n, nx, ny, True, spl)
    ref_tree = S_tree.trace(rays, mode=2)
    ref_list = S_tree.trace(rays, mode=1)
    for name, oct_, mode, ref in (("list", None, rt.TRAVERSAL_FAST, ref_list), ("list_scan", None, rt.TRAVERSAL_REFERENCE, ref_list),
                                 ("scan", O, rt.TRAVERSAL_REFERENCE, ref_tree), ("fast", O, rt.TRAVERSAL_FAST, ref_tree)):
        if oct_ is not None:
            oct_.set_traversal(mode)
        else:
            W.set_list_traversal(mode)
        d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
        rt.trace_rays(W, oct_, d_rays, nrays, d_out)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(rt.hit_record_dtype)
        assert np.array_equal(got["sphere"], ref["sphere"]), name
        assert np.array_equal(bits(got["t"]), bits(ref["t"])), name
        assert np.array_equal(bits(got["normal"]), bits(ref["normal"])), name
    O.set_traversal(rt.TRAVERSAL_FAST)
    fb, st = gpu_render(rt, torch, W, O, nx, ny, ns)
    ref, ref_st = S_tree.render(ns, nthreads=8)
    assert np.array_equal(bits(fb.cpu().numpy().reshape(ny, nx, 3)), bits(ref))
    assert np.array_equal(st.cpu().numpy().view(np.uint32).reshape(-1, 12)[:, :6], ref_st[:, :6])


# ---------------------------------------------------------------------------------------------------- host program
@pytest.mark.parametrize("args,n,nx,ny,ns,tree,spl", [
    (["3", "500", "64", "40", "4", "1", "30"], 500, 64, 40, 4, True, 30),
    (["3", "22", "100", "56", "3", "0", "30"], 22, 100, 56, 3, False, 30),
    (["3", "500", "72", "40", "4", "0", "30"], 500, 72, 40, 4, False, 30),           # USE_OCTREE off: the list through the candidate grid
    (["3", "2000", "64", "40", "4", "1", "30", "0.1", "0", "1"], 2000, 64, 40, 4, True, 30),   # BUILD_ON_GPU: rt_build_octree_gpu instead of buildOctree + upload
])
def test_rt_main_host_program_writes_the_oracle_ppm(rt, cuda, tmp_path, args, n, nx, ny, ns, tree, spl):
    """rt_main (host/main.cpp, the counterpart of the reference's main(), main.cu:347-477) in output mode 3: the
    output.ppm it writes through the C-ABI equals the oracle's PPM byte for byte; stderr carries the reference's lines."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dd2360-raytracing_amd", "rt_main")
    p = subprocess.run([exe] + args, cwd=tmp_path, capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    err = p.stderr.decode()
    assert "Rendering a %dx%d image with %d samples per pixel in 8x8 blocks." % (nx, ny, ns) in err
    assert "Number of spheres: %d" % n in err and ("Use octree: ON" if tree else "Use octree: OFF") in err and "took " in err
    ref, _ = OracleScene(n, nx, ny, use_octree=tree, spl=spl).render(ns, nthreads=8)
    assert (tmp_path / "output.ppm").read_bytes() == ppm_bytes(ref)


def test_rt_main_error_convention(rt, cuda, tmp_path):
    """checkCudaErrors convention (main.cu:27-37): a failing call prints '... error = <code> at <file>:<line> ...' and exits 99."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dd2360-raytracing_amd", "rt_main")
    p = subprocess.run([exe, "1", "3", "64", "40", "2"], cwd=tmp_path, capture_output=True, timeout=120)   # NUM_SPHERES must be > 4
    assert p.returncode == 99
    assert "error = " in p.stderr.decode() and "main.cpp" in p.stderr.decode()


def test_reference_default_configuration(rt, cuda):
    """The reference's own compile-time defaults (main.cu:22-24, :348-350; acceleration_structure.h:15): 1200x800, ns = 10,
    NUM_SPHERES 8000, SPHERE_RADIUS 0.1, USE_OCTREE, SPHERES_PER_LEAF 30 — sampled rows against the oracle, both traversals equal."""
    torch = cuda
    nx, ny, ns, n, spl = 1200, 800, 10, 8000, 30
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl)
    assert O.info()["dropped_full"] == 0 and W.created == 8000
    fb, st = gpu_render(rt, torch, W, O, nx, ny, ns)
    O.set_traversal(rt.TRAVERSAL_REFERENCE)
    fb2, st2 = gpu_render(rt, torch, W, O, nx, ny, ns)
    assert torch.equal(fb.view(torch.int32), fb2.view(torch.int32)) and torch.equal(st, st2)
    got = fb.cpu().numpy().reshape(ny, nx, 3)
    S = OracleScene(n, nx, ny, use_octree=True, spl=spl)
    for row in (0, 127, 402, 640, 799):
        ref, _ = S.render(ns, row0=row, rows=1, nthreads=1)
        assert np.array_equal(bits(got[row]), bits(ref[0])), "row %d" % row


@pytest.mark.gpu
def test_calls_captured_in_a_hip_graph(rt, cuda):
    """after one warm-up call (workspace allocations) the launches of rt_render / rt_render_progressive can be
    captured into a hipGraph.  Replaying the graph gives the bits of the direct calls: a progressive loop (fb += col, the RNG
    state continuing) and whole renders (counter reset, pilot pass with and without long-chain classification, tile order,
    render kernel), each replayed three times — a stale work queue would show as an untouched RNG state."""
    torch = cuda
    nx, ny, n, spl, passes = 400, 225, 500, 30, 8
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl)

    def progressive(direct):
        st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
        rt.render_init(nx, ny, st)
        rt.render_progressive(fb, nx, ny, 1, W, st, O)                # fb = col
        rt.render_progressive(fb, nx, ny, 2, W, st, O)                # fb += col
        torch.cuda.synchronize()
        if direct:
            for k in range(3, passes + 1):
                rt.render_progressive(fb, nx, ny, k, W, st, O)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                rt.render_progressive(fb, nx, ny, 3, W, st, O)        # captured, not executed
            for k in range(3, passes + 1):
                g.replay()
        torch.cuda.synchronize()
        return fb.clone(), st.clone()

    a, sa = progressive(True)
    b, sb = progressive(False)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(sa, sb)

    for ns in (8, 16):
        fb0, st0 = gpu_render(rt, torch, W, O, nx, ny, ns)             # also the warm-up of this frame size
        st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            rt.render_init(nx, ny, st)
            rt.render(fb, nx, ny, ns, W, st, O)
        for _ in range(3):                                             # every replay starts from render_init again
            fb.zero_()
            g.replay(); torch.cuda.synchronize()
            assert torch.equal(fb.view(torch.int32), fb0.view(torch.int32)) and torch.equal(st, st0)

    # without an octree (the list through the candidate grid, built and uploaded by the first such call — the direct render here)
    fb0, st0 = gpu_render(rt, torch, W, None, nx, ny, 8)
    st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rt.render_init(nx, ny, st)
        rt.render(fb, nx, ny, 8, W, st, None)
    for _ in range(2):
        fb.zero_()
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(fb.view(torch.int32), fb0.view(torch.int32)) and torch.equal(st, st0)

    # USE_FP16: the binary16 render has a pilot pass, a long-chain list and a tile order of its own (16 spp: all of it on)
    Wh = rt.World(n, nx, ny, precision=rt.FP16)
    Oh = rt.Octree(Wh, spl)
    def render_h(fbh, sth):
        rt.render_init(nx, ny, sth)
        rt.render(fbh, nx, ny, 16, Wh, sth, Oh)
    fb0 = rt.alloc_fb(nx, ny, precision=rt.FP16); st0 = rt.alloc_rand_state(nx, ny)
    render_h(fb0, st0); torch.cuda.synchronize()                     # direct call = the warm-up of this frame size
    fb = rt.alloc_fb(nx, ny, precision=rt.FP16); st = rt.alloc_rand_state(nx, ny)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        render_h(fb, st)
    for _ in range(2):
        fb.zero_()
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(fb.view(torch.int16), fb0.view(torch.int16)) and torch.equal(st, st0)


@pytest.mark.gpu
def test_c5_geometry_rows_against_the_oracle(rt, cuda):
    """BASELINE config 5's scene and frame (3840x2160, N = 100000, SPL 320 — the dense grid, rendered by the plain kernel variant)
    at 8 spp: sampled rows equal the oracle bit for bit, fast traversal equals the reference scan over the whole frame."""
    torch = cuda
    nx, ny, ns, n, spl = 3840, 2160, 8, 100000, 320
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl)
    assert O.info()["dropped_full"] == 0
    fb, st = gpu_render(rt, torch, W, O, nx, ny, ns)
    O.set_traversal(rt.TRAVERSAL_REFERENCE)
    fb2, st2 = gpu_render(rt, torch, W, O, nx, ny, ns)
    O.set_traversal(rt.TRAVERSAL_FAST)
    assert torch.equal(fb.view(torch.int32), fb2.view(torch.int32)) and torch.equal(st, st2)
    got = fb.cpu().numpy().reshape(ny, nx, 3)
    S = OracleScene(n, nx, ny, use_octree=True, spl=spl)
    for row in (40, 700, 1100, 2100):
        ref, _ = S.render(ns, row0=row, rows=1, nthreads=8)
        assert np.array_equal(bits(got[row]), bits(ref[0])), "row %d differs" % row


def to_half_image(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("seed,n,spl", [(21, 400, 30), (22, 5000, 48), (23, 12000, 64)])
def test_fp16_arbitrary_world_hit_records_and_frames(rt, cuda, seed, n, spl):
    """USE_FP16 on worlds the caller builds itself (large spheres stored in dozens of cells: a wave's rays visit more bucket
    ranges than its pool of segments holds, so the scan runs in several rounds; ghosts; spheres outside the root box): hit
    records of the redistributed binary16 scan and a small frame equal the fp16 oracle."""
    torch = cuda
    nx, ny, ns = 64, 40, 3
    sp, _ = random_world(rt, seed, n, nx, ny, big=24)
    for f in ("center", "radius", "albedo", "param"):
        sp[f] = to_half_image(sp[f])
    sp[0] = ((0.0, -1000.0, -1.0), 1000.0, rt.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)
    rng = np.random.default_rng(seed)
    cam = rt.camera_init((rng.uniform(8, 14), rng.uniform(1, 4), rng.uniform(-4, 4)), (0, 0.3, 0), (0, 1, 0), 35.0,
                         float(np.float16(nx) / np.float16(ny)), 0.05, 10.0, precision=rt.FP16)
    W = rt.World(n, nx, ny, precision=rt.FP16, spheres=sp, camera=cam)
    O = rt.Octree(W, spl)
    geom = np.concatenate([sp["center"], sp["radius"][:, None]], 1)
    mat = np.concatenate([sp["albedo"], sp["param"][:, None]], 1)
    S = OracleScene(n, nx, ny, fp16=True, use_octree=True, spl=spl, custom=(geom, mat, sp["material"], cam.view(np.float32).ravel()))
    nrays = 40_000 if n < 5000 else 12_000                      # (the binary16 oracle is ~30x slower than the fp32 one)
    rays = to_half_image(random_rays(nrays, 900 + seed))
    d_rays = torch.from_numpy(rays).cuda()
    for tree in (True, False):
        d_out = torch.zeros(nrays * 32, dtype=torch.uint8, device="cuda")
        rt.trace_rays(W, O if tree else None, d_rays, nrays, d_out)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(rt.hit_record_dtype)
        ref = S.trace(rays, mode=2 if tree else 1)
        assert np.array_equal(got["sphere"], ref["sphere"]), tree
        assert np.array_equal(bits(got["t"]), bits(ref["t"])), tree
        assert np.array_equal(bits(got["normal"]), bits(ref["normal"])), tree
    st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny, precision=rt.FP16)
    rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O)
    torch.cuda.synchronize()
    ref, ref_st = S.render(ns, nthreads=8)
    got = half_bits(fb).reshape(ny, nx, 3)
    nan = np.isnan(ref)
    assert np.array_equal(got[~nan], f32_to_half_bits(ref)[~nan])
    assert np.array_equal(st.cpu().numpy().view(np.uint32).reshape(-1, 12)[:, :6], ref_st[:, :6])
