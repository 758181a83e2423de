"""rt_build_octree_gpu (buildOctree + traversal copy + candidate grid on the device, csrc/rt_build.hip) against rt_build_octree
(the host build, itself bit-equal to the oracle's buildOctree: tests/test_host_and_abi.py): the reference-layout tree, every
device-resident array, and the frames rendered through both."""
import numpy as np
import pytest

from test_gpu_parity import random_world

pytestmark = pytest.mark.gpu

ARRAYS = ["nodes", "ent_hot", "ent_id", "large_hot", "large_brick", "cs", "hot", "brick", "memb_start", "memb_cell", "cellnode", "bits_index", "cellbits"]


def compare(rt, W, spl):
    H = rt.Octree(W, spl).upload()
    G = rt.Octree(W, spl, gpu=True)
    assert H.info() == G.info()
    assert H.accel_info() == G.accel_info()
    assert np.array_equal(H.nodes().view(np.uint8), G.nodes().view(np.uint8))                  # OctNode[585], reference numbering
    hc, hi = H.leaves(); gc, gi = G.leaves()
    assert np.array_equal(hc, gc) and np.array_equal(hi, gi)                                   # OctLeaf contents, reference numbering
    for k, name in enumerate(ARRAYS):
        a, b = H.device_array(k), G.device_array(k)
        assert a.size == b.size, name
        if name == "large_hot" and a.size == 0:
            continue
        assert np.array_equal(a, b), name
    return H, G


@pytest.mark.parametrize("n,spl", [(22, 30), (500, 30), (8000, 30), (10000, 32), (100000, 320), (2000, 3), (5, 30)])
def test_device_build_equals_host_build_create_world(rt, cuda, n, spl):
    """create_world scenes, including N = 100 000 / SPL 320 (C5), a tree whose buckets overflow (SPL 3: drops), and the smallest world"""
    W = rt.World(n, 1200, 800).upload()
    H, G = compare(rt, W, spl)
    if spl == 3:
        assert H.info()["dropped_full"] > 0


@pytest.mark.parametrize("seed,n,spl", [(11, 300, 30), (12, 3000, 40), (13, 12000, 64), (14, 4000, 4)])
def test_device_build_equals_host_build_arbitrary_worlds(rt, cuda, seed, n, spl):
    """caller-built worlds: spheres of many sizes, in the air, outside the root box, ghost slots, overflowing buckets"""
    sp, cam = random_world(rt, seed, n, 400, 240)
    W = rt.World(n, 400, 240, spheres=sp, camera=cam).upload()
    compare(rt, W, spl)


def test_frames_through_a_device_built_tree(rt, cuda):
    torch = cuda
    nx, ny, ns, n, spl = 400, 232, 16, 10000, 32
    W = rt.World(n, nx, ny).upload()
    H = rt.Octree(W, spl)
    G = rt.Octree(W, spl, gpu=True)
    out = []
    for O in (H, G):
        for mode in (rt.TRAVERSAL_FAST, rt.TRAVERSAL_REFERENCE):
            O.set_traversal(mode)
            st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny)
            rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O)
            torch.cuda.synchronize()
            out.append((fb, st))
    for fb, st in out[1:]:
        assert torch.equal(fb.view(torch.int32), out[0][0].view(torch.int32)) and torch.equal(st, out[0][1])


@pytest.mark.parametrize("n,spl,custom", [(500, 30, False), (10000, 32, False), (2000, 3, False), (3000, 40, True)])
def test_device_build_of_binary16_trees(rt, cuda, n, spl, custom):
    """USE_FP16: the reference layout in binary16 arithmetic, the pair layout of the bucket entries, pair-based node ranges and
    plane indices equal the host build's; frames through both are the same"""
    torch = cuda
    nx, ny, ns = 96, 56, 3
    if custom:
        sp, _ = random_world(rt, 31, n, nx, ny, big=12)
        for f in ("center", "radius", "albedo", "param"):
            sp[f] = np.asarray(sp[f], np.float32).astype(np.float16).astype(np.float32)
        sp[0] = ((0.0, -1000.0, -1.0), 1000.0, rt.MAT_LAMBERTIAN, (0.5, 0.5, 0.5), 0.0)
        cam = rt.camera_init((12, 2, 3), (0, 0.3, 0), (0, 1, 0), 35.0, float(np.float16(nx) / np.float16(ny)), 0.05, 10.0, precision=rt.FP16)
        W = rt.World(n, nx, ny, precision=rt.FP16, spheres=sp, camera=cam).upload()
    else:
        W = rt.World(n, nx, ny, precision=rt.FP16).upload()
    H = rt.Octree(W, spl).upload()
    G = rt.Octree(W, spl, gpu=True)
    assert H.info() == G.info()
    assert np.array_equal(H.nodes().view(np.uint8), G.nodes().view(np.uint8))
    hc, hi = H.leaves(); gc, gi = G.leaves()
    assert np.array_equal(hc, gc) and np.array_equal(hi, gi)
    for k in range(3):                                            # traversal nodes (pair ranges, plane indices), pairs, pair -> sphere
        a, b = H.device_array(k), G.device_array(k)
        assert a.size == b.size and np.array_equal(a, b), k
    out = []
    for O in (H, G):
        st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny, precision=rt.FP16)
        rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O)
        torch.cuda.synchronize()
        out.append((fb, st))
    assert torch.equal(out[0][0].view(torch.int16), out[1][0].view(torch.int16)) and torch.equal(out[0][1], out[1][1])
