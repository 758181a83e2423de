"""GPU parity at the REAL workloads of BASELINE.json's configs (-m gpu): every config at its own frame size, sample count,
sphere count and traversal, rows compared with the CPU oracle bit for bit (fp32) / bit-exact binary16 (USE_FP16), RNG state
written back included.  The oracle renders single rows (a row is independent of the others: the RNG is keyed by the absolute
pixel_index, main.cu:93), several rows at a time on the host's cores.

  C2  1200x800x64,  N = 500,    USE_OCTREE off -> hitable_list::hit (hitable_list.h:16-31), default list traversal
  C3  1200x800x64,  N = 10000,  octree SPL 32: tests/test_gpu_parity.py::test_full_size_properties_c3
  C4  = C3 with USE_FP16 (precision_types.h:8)
  C5  3840x2160x256, N = 100000, octree SPL 320, split into 8 parts (the 8-GPU tile split) and reassembled
"""
import threading

import numpy as np
import pytest

from oracle_lib import OracleScene, ppm_bytes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def oracle_rows(S, ns, rows):
    """{row: (frame row, RNG states of the row)} — one oracle call per row, rows in parallel threads (ctypes drops the GIL)"""
    out = {}

    def work(r):
        out[r] = S.render(ns, row0=r, rows=1, nthreads=1)

    th = [threading.Thread(target=work, args=(r,)) for r in rows]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


def render(rt, torch, W, O, nx, ny, ns, part=None, precision=None):
    part = part or rt.WHOLE
    precision = rt.FP32 if precision is None else precision
    st = rt.alloc_rand_state(nx, ny, part)
    fb = rt.alloc_fb(nx, ny, part, precision=precision)
    rt.render_init(nx, ny, st, part)
    rt.render(fb, nx, ny, ns, W, st, O, part)
    torch.cuda.synchronize()
    return fb, st


def test_c2_real_workload_list_path_against_the_oracle(rt, cuda):
    """BASELINE config 2 as benched: 1200x800, 64 spp, NUM_SPHERES = 500, no octree passed, default list traversal (the list
    as one unbounded node through the candidate grid, long-chain classification on: ns >= 16).  Eight rows — sky, sphere field,
    the crevice rows with the frame's longest chains (316/317), ground — and their RNG states equal the oracle's
    hitable_list path; the plain list-order scan gives the same frame."""
    torch = cuda
    nx, ny, ns, n = 1200, 800, 64, 500
    W = rt.World(n, nx, ny)
    assert W.list_accel_info()["enabled"]
    fb, st = render(rt, torch, W, None, nx, ny, ns)
    got = fb.cpu().numpy().reshape(ny, nx, 3)
    st_host = st.cpu().numpy().view(np.uint32).reshape(ny, nx, 12)
    S = OracleScene(n, nx, ny, use_octree=False)
    rows = (2, 120, 250, 316, 317, 431, 640, 798)
    ref = oracle_rows(S, ns, rows)
    for r in rows:
        assert np.array_equal(bits(got[r]), bits(ref[r][0][0])), "row %d differs" % r
        assert np.array_equal(st_host[r][:, :6], ref[r][1][:, :6]), "RNG state of row %d differs" % r
    W.set_list_traversal(rt.TRAVERSAL_REFERENCE)
    fb2, st2 = render(rt, torch, W, None, nx, ny, ns)
    assert torch.equal(fb.view(torch.int32), fb2.view(torch.int32)) and torch.equal(st, st2)


def test_c4_real_workload_fp16_against_the_oracle(rt, cuda):
    """BASELINE config 4 as benched: 1200x800, 64 spp, NUM_SPHERES = 10000, octree SPL 32, USE_FP16.  Four rows and their RNG
    states equal the fp16 oracle bit for bit (binary16 channels; NaN where the oracle has NaN)."""
    torch = cuda
    nx, ny, ns, n, spl = 1200, 800, 64, 10000, 32
    W = rt.World(n, nx, ny, precision=rt.FP16)
    O = rt.Octree(W, spl)
    fb, st = render(rt, torch, W, O, nx, ny, ns, precision=rt.FP16)
    got = fb.cpu().numpy().view(np.uint16).reshape(ny, nx, 3)
    st_host = st.cpu().numpy().view(np.uint32).reshape(ny, nx, 12)
    S = OracleScene(n, nx, ny, fp16=True, use_octree=True, spl=spl)
    rows = (5, 300, 317, 640)
    ref = oracle_rows(S, ns, rows)
    for r in rows:
        want = ref[r][0][0]
        nan = np.isnan(want)
        assert np.array_equal(got[r][~nan], want.astype(np.float16).view(np.uint16)[~nan]), "row %d differs" % r
        assert np.isnan(got[r].view(np.float16)[nan]).all()
        assert np.array_equal(st_host[r][:, :6], ref[r][1][:, :6]), "RNG state of row %d differs" % r


def test_c5_real_workload_8_part_split_against_the_oracle(rt, cuda):
    """BASELINE config 5: 3840x2160, 256 spp, NUM_SPHERES = 100000, octree SPL 320 (2.1 G samples), rendered the way the 8-GPU
    job renders it — eight rt_partition parts (runs of RT_PART_RUN = 64 consecutive tiles dealt round-robin: run r -> part r % 8, the
    default split of rt_multi_render), compact tile-major buffers, rt_assemble — and as one whole
    frame: both equal bit for bit, and four rows of the frame and their RNG states equal the oracle."""
    torch = cuda
    nx, ny, ns, n, spl, nparts = 3840, 2160, 256, 100000, 320, 8
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl)
    assert O.info()["dropped_full"] == 0
    rows = (40, 700, 1100, 2100)
    S = OracleScene(n, nx, ny, use_octree=True, spl=spl)
    ref = {}
    bg = threading.Thread(target=lambda: ref.update(oracle_rows(S, ns, rows)))      # the oracle works while the GPU renders
    bg.start()
    per = rt.part_pixels(nx, ny, rt.Partition(0, nparts))
    parts = torch.zeros(nparts * per * 3, dtype=torch.float32, device="cuda")
    for p in range(nparts):
        f, _ = render(rt, torch, W, O, nx, ny, ns, rt.Partition(p, nparts))
        parts[p * per * 3: p * per * 3 + f.numel()] = f
    full = torch.zeros(nx * ny * 3, dtype=torch.float32, device="cuda")
    rt.assemble(full, parts, nx, ny, nparts)
    torch.cuda.synchronize()
    fb, st = render(rt, torch, W, O, nx, ny, ns)
    assert torch.equal(full.view(torch.int32), fb.view(torch.int32))
    got = fb.cpu().numpy().reshape(ny, nx, 3)
    st_host = st.cpu().numpy().view(np.uint32).reshape(ny, nx, 12)
    bg.join()
    for r in rows:
        assert np.array_equal(bits(got[r]), bits(ref[r][0][0])), "row %d differs" % r
        assert np.array_equal(st_host[r][:, :6], ref[r][1][:, :6]), "RNG state of row %d differs" % r


def test_image_writers_from_a_device_frame(rt, cuda, tmp_path):
    """a frame rendered on the device goes through rt_write_image as P3, P6 and PFM.  The P3 bytes are the
    oracle's output_to_stream bytes (main.cu:321-333) of the oracle's frame, the P6 payload holds the P3 numbers, the PFM
    payload holds the framebuffer's bits (bottom row first)."""
    torch = cuda
    nx, ny, ns, n, spl = 200, 120, 4, 500, 30
    W = rt.World(n, nx, ny)
    O = rt.Octree(W, spl)
    fb, _ = render(rt, torch, W, O, nx, ny, ns)
    host = fb.cpu().numpy()
    ref, _ = OracleScene(n, nx, ny, use_octree=True, spl=spl).render(ns, nthreads=8)
    p3, p6, pfm = tmp_path / "a.ppm", tmp_path / "a6.ppm", tmp_path / "a.pfm"
    rt.write_image(p3, host, nx, ny, fmt=rt.IMAGE_P3)
    rt.write_image(p6, host, nx, ny, fmt=rt.IMAGE_P6)
    rt.write_image(pfm, host, nx, ny, fmt=rt.IMAGE_PFM)
    want = ppm_bytes(ref)
    assert p3.read_bytes() == want
    numbers = np.array(want.split()[4:], dtype=np.int64)              # after "P3", W, H, 255
    raw = p6.read_bytes()
    head = b"P6\n%d %d\n255\n" % (nx, ny)
    assert raw.startswith(head) and len(raw) == len(head) + nx * ny * 3
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).astype(np.int64), numbers)
    raw = pfm.read_bytes()
    head = b"PF\n%d %d\n-1.0\n" % (nx, ny)
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], "<u4"), bits(ref).ravel())
    # binary16 frame: the same three writers on the float images of the halves
    W16 = rt.World(n, nx, ny, precision=rt.FP16)
    O16 = rt.Octree(W16, spl)
    fb16, _ = render(rt, torch, W16, O16, nx, ny, ns, precision=rt.FP16)
    host16 = fb16.cpu().numpy()
    ref16, _ = OracleScene(n, nx, ny, fp16=True, use_octree=True, spl=spl).render(ns, nthreads=8)
    rt.write_image(p3, host16, nx, ny, precision=rt.FP16, fmt=rt.IMAGE_P3)
    assert p3.read_bytes() == ppm_bytes(ref16)


def test_c3_full_size_progressive_64_passes(rt, cuda):
    """render_progressive (main.cu:119-142) at the C3 frame: 64 one-sample passes at 1200x800, N = 10000, octree SPL 32, through the
    kernel the library selects for progressive passes on this tree (the pooled walk, k_render<true,1,4>).  The accumulated frame,
    normalised and gamma-corrected as the viewer does (main.cu:283-284: /n, sqrt), equals the oracle's render(64) rows bit for bit
    (col = ((0+c1)+c2)+... in both); the RNG states equal those render(64) leaves; and the same through the reference traversal."""
    torch = cuda
    nx, ny, ns, n, spl = 1200, 800, 64, 10000, 32
    W = rt.World(n, nx, ny).upload()
    O = rt.Octree(W, spl).upload()
    assert rt.render_kernel_name(W, O, 1) == "k_render<true,1,4>"

    def progressive():
        st = rt.alloc_rand_state(nx, ny)
        fb = rt.alloc_fb(nx, ny)
        rt.render_init(nx, ny, st)
        for k in range(1, ns + 1):
            rt.render_progressive(fb, nx, ny, k, W, st, O)
        torch.cuda.synchronize()
        return fb, st

    fb, st = progressive()
    whole, st_whole = render(rt, torch, W, O, nx, ny, ns)
    assert torch.equal(st.view(torch.uint8), st_whole.view(torch.uint8))              # RNG continuation over 64 launches = one launch of 64
    acc = fb.cpu().numpy().reshape(ny, nx, 3)
    shown = np.sqrt(acc * np.float32(1.0 / 64.0))                                     # exact scale, correctly rounded sqrt
    assert np.array_equal(bits(shown), bits(whole.cpu().numpy().reshape(ny, nx, 3)))
    S = OracleScene(n, nx, ny, use_octree=True, spl=spl)
    rows = (3, 250, 316, 317, 600, 797)
    ref = oracle_rows(S, ns, rows)
    st_host = st.cpu().numpy().view(np.uint32).reshape(-1, 12)
    for r in rows:
        assert np.array_equal(bits(shown[r]), bits(ref[r][0][0])), "row %d differs" % r
        assert np.array_equal(st_host[r * nx:(r + 1) * nx, :6], ref[r][1][:, :6]), "RNG states of row %d differ" % r
    O.set_traversal(rt.TRAVERSAL_REFERENCE)                                           # the literal traverseTree scan, per-lane kernel
    assert rt.render_kernel_name(W, O, 1) == "k_render<true,1,1>"
    fb2, st2 = progressive()
    O.set_traversal(rt.TRAVERSAL_FAST)
    assert torch.equal(fb.view(torch.int32), fb2.view(torch.int32)) and torch.equal(st.view(torch.uint8), st2.view(torch.uint8))
