"""The candidate strips (csrc/rt_accel.h) at their seams (-m gpu): rays built ON the structure's lattice — origins exactly on column
edges and fine-bin edges, directions along a column, at slope exactly +-1 (where the major axis flips), nearly flat and nearly
vertical, origins at sphere centres, on sphere surfaces, on the y-slab's faces — through the fast traversal, the library's own
reference scan and the oracle's hitTree / hitable_list::hit.  A walk that leaves out the column of a hit point, or a bin of a
centre, shows here as a missing or farther hit.  Random rays are the campaign's business (tools/campaign.py)."""
import numpy as np
import pytest

from oracle_lib import OracleScene

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def lattice_rays(info, centers, radii, n, seed):
    rng = np.random.default_rng(seed)
    G, h = info["grid_dim"], float(info["cell_size"])
    g0 = -(11.0 + 5.0 * h)                                    # build_accel: half = root half-width + 2 Rlim + 2 h, Rlim = 1.5 h
    F = 8
    o = np.zeros((n, 3), np.float64)
    d = np.zeros((n, 3), np.float64)
    # origins: x and z on column edges / fine-bin edges / an ulp off them, y anywhere in and around the spheres' slab
    i = rng.integers(0, G * F + 1, (n, 2))
    on_col = rng.random((n, 2)) < 0.5
    i = np.where(on_col, (i // F) * F, i)
    o[:, 0] = g0 + i[:, 0] * (h / F)
    o[:, 2] = g0 + i[:, 1] * (h / F)
    o[:, 1] = rng.choice([0.0, 1e-4, 0.1, 0.2, 0.2, 0.4, 0.4001, 0.7, 2.0], n)
    # a third of the origins at a sphere's centre or on its surface along an axis
    k = n // 3
    pick = rng.integers(1, len(radii), k)
    o[:k] = centers[pick]
    on_surface = rng.random(k) < 0.6
    axis = rng.integers(0, 3, k)
    sign = rng.choice([-1.0, 1.0], k)
    o[np.arange(k)[on_surface], axis[on_surface]] += (sign * radii[pick])[on_surface]
    # directions
    kind = rng.integers(0, 8, n)
    ang = rng.uniform(0, 2 * np.pi, n)
    d[:, 0] = np.cos(ang); d[:, 2] = np.sin(ang); d[:, 1] = rng.normal(scale=0.2, size=n)
    m = kind == 0; d[m, 2] = rng.choice([1e-30, -1e-30, 1e-12, -1e-7], m.sum()); d[m, 0] = rng.choice([-1.0, 1.0], m.sum())     # along x, all but flat in z
    m = kind == 1; d[m, 0] = rng.choice([1e-30, -1e-30, 1e-12, -1e-7], m.sum()); d[m, 2] = rng.choice([-1.0, 1.0], m.sum())     # along z
    m = kind == 2; d[m, 0] = rng.choice([-1.0, 1.0], m.sum()); d[m, 2] = d[m, 0] * rng.choice([-1.0, 1.0], m.sum())               # |dx| == |dz|: the major axis' tie
    m = kind == 3; d[m, 1] = rng.choice([1e-30, -1e-30, 1e-9, -1e-6, 1e-3], m.sum())                                               # all but horizontal
    m = kind == 4; d[m, 0] *= 1e-6; d[m, 2] *= 1e-6; d[m, 1] = rng.choice([-1.0, 1.0], m.sum())                                    # all but vertical
    m = kind == 5; d[m] *= rng.choice([1e-9, 1e-4, 1e4, 1e9], m.sum())[:, None]                                                    # |d|^2 at and beyond the walk's preconditions
    return np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)


def trace(rt, torch, W, O, rays):
    n = len(rays)
    d_rays = torch.from_numpy(rays).cuda()
    d_out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    rt.trace_rays(W, O, d_rays, n, d_out)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(rt.hit_record_dtype)


@pytest.mark.parametrize("n,spl,frame", [(10000, 32, (1200, 800)), (500, 0, (1200, 800)), (100000, 320, (3840, 2160))])
def test_rays_on_the_strips_lattice(rt, cuda, n, spl, frame):
    torch = cuda
    nx, ny = frame
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl) if spl else None
    info = O.accel_info() if O else W.list_accel_info()
    assert info["grid_dim"] > 0
    S = OracleScene(n, nx, ny, use_octree=bool(spl), spl=spl or 30)
    sp = W.spheres
    nrays = 120_000 if n <= 10000 else 40_000
    rays = lattice_rays(info, sp["center"].astype(np.float64), sp["radius"].astype(np.float64), nrays, 4242 + n)
    ref = S.trace(rays, mode=2 if spl else 1)
    outs = []
    for mode in (rt.TRAVERSAL_FAST, rt.TRAVERSAL_REFERENCE):
        if O: O.set_traversal(mode)
        else: W.set_list_traversal(mode)
        outs.append(trace(rt, torch, W, O, rays))
    assert ref["hit"].sum() > nrays // 10
    for got in outs:
        assert np.array_equal(got["sphere"], ref["sphere"])
        assert np.array_equal(bits(got["t"]), bits(ref["t"]))
        assert np.array_equal(bits(got["p"]), bits(ref["p"]))
        assert np.array_equal(bits(got["normal"]), bits(ref["normal"]))
