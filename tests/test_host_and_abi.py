"""CPU tests of the product's host side (not gpu): the C-ABI library loads and exports every symbol of
include/rt_amd.h, and its host-only entry points (world generation, camera, octree build, PPM writer, partition
arithmetic) agree bit for bit with the oracle.  No compute entry point is called here."""
import os
import re

import numpy as np
import pytest

from oracle_lib import OracleScene, ppm_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_abi_exports_every_declared_symbol(rt):
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(rt.SYMBOLS), "binding and header disagree: %s" % (declared ^ set(rt.SYMBOLS))
    L = rt.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.rt_abi_version() == 6


def test_pod_sizes_match_reference_structs(rt):
    # curandState 48, camera 88, OctNode 60
    assert rt.rand_state_dtype.itemsize == 48 and rt.camera_dtype.itemsize == 88 and rt.octnode_dtype.itemsize == 60


def test_compute_calls_reject_bad_arguments(rt):
    L = rt.lib()
    assert L.rt_render(None, 0, 0, 0, None, None, None, rt.WHOLE, None) == -1
    assert L.rt_render_init(8, 8, None, rt.WHOLE, None) == -1
    assert L.rt_part_pixels(8, 8, rt.Partition(3, 2)) == -1
    assert L.rt_create_world(None, 4, 0.1, None, 1, 1, None, 0, None) == -1
    assert rt.lib().rt_error_string(-1) == b"invalid argument"


def test_range_partitions_the_bands_of_a_balanced_split(rt):
    """rt_partition with a tile range (ABI 6): a band [tile_begin, tile_end) of the row-major tile numbering — element counts, argument
    checks, and the runs form (tile_begin == tile_end == 0) unchanged next to it."""
    L = rt.lib()
    nx, ny = 100, 50                                              # 13 x 7 = 91 tiles, ragged edges
    tiles = 13 * 7
    assert rt.part_pixels(nx, ny, rt.Partition(0, 1, 0, tiles)) == tiles * 64          # a range is always compact and tile-major ...
    assert rt.part_pixels(nx, ny, rt.Partition(0, 1)) == nx * ny                       # ... the undivided frame is row-major
    starts = [0, 10, 11, 60, tiles]
    parts = rt.split_parts(starts)
    assert [(p.part, p.nparts, p.tile_begin, p.tile_end) for p in parts] == [(0, 4, 0, 10), (1, 4, 10, 11), (2, 4, 11, 60), (3, 4, 60, tiles)]
    assert [rt.part_pixels(nx, ny, p) for p in parts] == [640, 64, 49 * 64, 31 * 64]
    assert L.rt_part_pixels(nx, ny, rt.Partition(0, 2, 5, 5)) == -1                    # an empty range that is not (0, 0)
    assert L.rt_part_pixels(nx, ny, rt.Partition(0, 2, 7, 3)) == -1
    assert L.rt_part_pixels(nx, ny, rt.Partition(0, 2, -1, 3)) == -1
    assert L.rt_part_pixels(nx, ny, rt.Partition(1, 2, 60, tiles + 1)) == -1          # past the frame's last tile
    assert L.rt_render_init(nx, ny, None, rt.Partition(1, 2, 60, tiles + 1), None) == -1
    # split entry points check their arguments without a device
    st = (__import__("ctypes").c_int64 * 3)(0, 40, tiles)
    assert L.rt_assemble_split(None, None, nx, ny, 2, st, 64 * 51, 0, None) == -1
    assert L.rt_split_balanced(None, None, None, nx, ny, 2, st, None, None, None, None) == -1
    assert L.rt_multi_set_split(None, 1) == -1 and L.rt_multi_last_split(None, st) == -1


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("n,spl", [(22, 30), (500, 30), (8000, 30), (10000, 32)])
def test_world_camera_octree_match_oracle(rt, precision, n, spl):
    W = rt.World(n, 1200, 800, precision=precision)
    O = rt.Octree(W, spl)
    S = OracleScene(n, 1200, 800, fp16=bool(precision), use_octree=True, spl=spl)
    geom, mat, kind = S.spheres()
    sp = W.spheres
    assert np.array_equal(u32(sp["center"]), u32(geom[:, :3])) and np.array_equal(u32(sp["radius"]), u32(geom[:, 3]))
    assert np.array_equal(sp["material"], kind)
    assert np.array_equal(u32(sp["albedo"]), u32(mat[:, :3])) and np.array_equal(u32(sp["param"]), u32(mat[:, 3]))
    assert W.created == S.info()["real"]
    assert np.array_equal(u32(W.camera).ravel(), u32(S.camera()))
    assert np.array_equal(u32(W.rand_state).ravel()[:6], S.world_rng()[:6])        # *rand_state = local_rand_state
    t = S.octree()
    nodes = O.nodes()
    counts, idx = O.leaves()
    assert np.array_equal(nodes["level"], t["level"]) and np.array_equal(u32(nodes["aabb"]), u32(t["box"]))
    assert np.array_equal(nodes["children"], t["children"])
    assert np.array_equal(counts, t["counts"]) and np.array_equal(idx, t["indices"])
    info, oinfo = O.info(), S.info()
    assert info["node_count"] == oinfo["node_count"] and info["leaf_count"] == oinfo["leaf_count"]
    assert info["dropped_full"] == oinfo["dropped_full"]
    # traversal copy: every used node once, every non-ghost entry once
    ghosts = set(np.nonzero(kind == -1)[0].tolist())
    live_entries = sum(1 for l in range(1, len(counts)) for k in range(counts[l]) if idx[l, k] not in ghosts)
    assert info["flat_nodes"] == info["node_count"] and info["flat_entries"] == live_entries


def test_part_without_tiles_is_a_no_op(rt):
    # 8x8 frame = one tile dealt to part 0 of 3: parts 1 and 2 own nothing, their calls succeed without touching the device
    L = rt.lib()
    W = rt.World(22, 8, 8)
    assert rt.part_pixels(8, 8, rt.Partition(1, 3)) == 0
    assert L.rt_render_init(8, 8, None, rt.Partition(1, 3), None) == 0
    assert L.rt_render(None, 8, 8, 4, W.h, None, None, rt.Partition(2, 3), None) == 0
    assert L.rt_render_progressive(None, 8, 8, 1, W.h, None, None, rt.Partition(2, 3), None) == 0
    assert L.rt_render(None, 8, 8, 4, W.h, None, None, rt.Partition(0, 3), None) == -1        # part 0 has pixels: buffers are required


def test_camera_init_matches_create_world(rt):
    W = rt.World(22, 1200, 800)
    cam = rt.camera_init((13, 2, 3), (0, 0, 0), (0, 1, 0), 30.0, np.float32(1200) / np.float32(800), np.float32(0.1), 10.0)
    assert np.array_equal(u32(cam), u32(W.camera))


def test_ppm_writer_matches_oracle(rt, tmp_path):
    rng = np.random.default_rng(1)
    fb = rng.uniform(0, 1, (9, 13, 3)).astype(np.float32)
    fb[0, 0] = [0.0, 1.0, 0.999999]
    assert rt.format_ppm(fb, 13, 9) == ppm_bytes(fb)
    p = tmp_path / "o.ppm"
    assert rt.lib().rt_write_ppm(str(p).encode(), 13, 9, fb.ctypes.data, 0) == 0
    assert p.read_bytes() == ppm_bytes(fb)
    # fp16 framebuffer (3 x binary16 per pixel)
    h = fb.astype(np.float16)
    assert rt.format_ppm(h, 13, 9, precision=1) == ppm_bytes(h.astype(np.float32))


def test_partition_arithmetic(rt):
    for nx, ny in ((1200, 800), (61, 35), (8, 8), (3394, 2263)):
        tiles = ((nx + 7) // 8) * ((ny + 7) // 8)
        assert rt.part_pixels(nx, ny) == nx * ny
        for nparts in (2, 3, 8):
            sizes = [rt.part_pixels(nx, ny, rt.Partition(p, nparts)) for p in range(nparts)]
            # the parts differ by at most one run of tiles, and part 0 is never the smaller one (the staging slot size)
            assert sum(sizes) == tiles * 64 and max(sizes) == sizes[0] and max(sizes) - min(sizes) <= 64 * rt.PART_RUN


def test_reference_host_program_builds():
    # rt_main (the main.cu counterpart) is part of build(); it needs a GPU to run, only its presence is checked here
    import __graft_entry__
    if not os.path.exists(os.path.join(ROOT, "dd2360-raytracing_amd", "rt_main")):
        __graft_entry__.build()
    assert os.access(os.path.join(ROOT, "dd2360-raytracing_amd", "rt_main"), os.X_OK)


def test_binary_image_writers(rt, tmp_path):
    """P6 and PFM companions of the P3 writer: same quantisation / row order rules, checked against numpy."""
    rng = np.random.default_rng(2)
    fb = rng.uniform(0, 1, (7, 11, 3)).astype(np.float32)
    fb[3, 4] = [0.0, 1.0, 0.5]
    p6 = tmp_path / "a.ppm"
    rt.write_image(p6, fb, 11, 7, fmt=rt.IMAGE_P6)
    raw = p6.read_bytes()
    head = b"P6\n11 7\n255\n"
    assert raw.startswith(head) and len(raw) == len(head) + 7 * 11 * 3
    want = np.clip((255.99 * fb[::-1].astype(np.float64)).astype(np.int64), 0, 255).astype(np.uint8)
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(7, 11, 3), want)
    # the P6 bytes are the P3 numbers
    p3 = [int(x) for x in rt.format_ppm(fb, 11, 7).split()[4:]]
    assert p3 == want.ravel().tolist()
    pf = tmp_path / "a.pfm"
    rt.write_image(pf, fb, 11, 7, fmt=rt.IMAGE_PFM)
    raw = pf.read_bytes()
    head = b"PF\n11 7\n-1.0\n"
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(7, 11, 3).view(np.uint32), fb.view(np.uint32))
    # fp16 framebuffers go through the same writers
    rt.write_image(tmp_path / "h.pfm", fb.astype(np.float16), 11, 7, precision=rt.FP16, fmt=rt.IMAGE_PFM)
    raw = (tmp_path / "h.pfm").read_bytes()
    assert np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(7, 11, 3), fb.astype(np.float16).astype(np.float32))
    assert rt.lib().rt_write_image(b"/nonexistent_dir/x.ppm", 11, 7, fb.ctypes.data, 0, 1) == -3


def test_list_traversal_switch_and_candidate_grid_rules(rt):
    """hitable_list::hit through the candidate grid (rt_world_set_list_traversal): host-side rules only.  The grid serves fp32
    lists of >= 64 hittable spheres with at most 64 of them outside its range; everything else keeps the list-order scan."""
    L = rt.lib()
    assert L.rt_world_set_list_traversal(None, rt.TRAVERSAL_FAST) == -1
    W = rt.World(500, 64, 36)
    assert L.rt_world_set_list_traversal(W.h, 7) == -1
    W.set_list_traversal(rt.TRAVERSAL_REFERENCE).set_list_traversal(rt.TRAVERSAL_FAST)
    info = W.list_accel_info()
    assert info["enabled"] and info["large_spheres"] == 3 and info["grid_entries"] >= 484       # the three big spheres are tested directly
    assert not rt.World(22, 64, 36).list_accel_info()["enabled"]                                 # too small to pay
    assert not rt.World(500, 64, 36, precision=rt.FP16).list_accel_info()["enabled"]             # the error bounds are binary32 bounds
    # a list whose spheres mostly lie outside the grid's range: direct tests would dominate
    sp = W.spheres.copy()
    sp["center"][100:300, 0] += 40.0
    far = rt.World(500, 64, 36, spheres=sp, camera=W.camera)
    assert not far.list_accel_info()["enabled"]
    sp = W.spheres.copy()
    sp["center"][100:140, 0] += 40.0
    near = rt.World(500, 64, 36, spheres=sp, camera=W.camera).list_accel_info()
    assert near["enabled"] and near["large_spheres"] == 43


def test_bench_counter_parsing_and_core_count(tmp_path):
    """bench.py's helpers that do not need a GPU: per-dispatch averages of the render kernel from rocprofv3's counter CSV (other
    kernels ignored), and a usable-core count of at least one"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = tmp_path / "p0" / "host"
    d.mkdir(parents=True)
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value",
            '1,"rt::k_render_init(rt_rand_state*, int)",SQ_INSTS_VALU,5',
            '2,"void rt::k_render<true, 0, 4>(rt::RenderArgs)",SQ_INSTS_VALU,100',
            '2,"void rt::k_render<true, 0, 4>(rt::RenderArgs)",SQ_INSTS_VALU,20',      # (one row per XCD / SE: summed per dispatch)
            '3,"void rt::k_render<true, 0, 4>(rt::RenderArgs)",SQ_INSTS_VALU,140',
            '3,"void rt::k_render<true, 0, 4>(rt::RenderArgs)",SQ_WAVES,7',
            '4,"rt::k_tile_order(int const*)",SQ_INSTS_VALU,999']
    (d / "1_counter_collection.csv").write_text("\n".join(rows) + "\n")
    got = bench.parse_pmc_dir(str(tmp_path / "p0"))
    assert got == {"SQ_INSTS_VALU": 130.0, "SQ_WAVES": 7.0}
    assert bench.usable_cores() >= 1
    assert set(bench.CONFIGS) == {"c2", "c3", "c4", "c5"} and bench.CONFIGS["c5"]["spp"] == 256
