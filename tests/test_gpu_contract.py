"""rt_world_set_arith(RT_ARITH_CONTRACT): the fp32 render kernels with FMA contraction allowed — the one behaviour of a real nvcc
build of the reference (default -fmad=true, Makefile:9) that can be modelled here.  A TOLERANCE mode, reported separately and never
the parity mode: these tests state the bounds it is held to against the (uncontracted) oracle, and check that selecting it and
deselecting it leaves the default mode's bits alone."""
import numpy as np
import pytest

from oracle_lib import OracleScene

pytestmark = pytest.mark.gpu

# Bounds against the oracle on C3 rows.  MEASURED (profiles/r3/contract_mode.txt): 72.1 % of the colour channels within +-1/255 of
# the oracle, 89.6 % within +-4/255, PSNR 38.6 dB, largest difference 0.166, mean difference -8e-5, 31.5 % of the channels bit-equal.
# The 99.9 %-within-1/255 one might hope for cannot hold for this scene: a contracted dot product differs from the reference's in its
# last bit, a reflection off a sphere multiplies a direction error roughly tenfold, so after ~7 bounces the path is another path —
# every sample that bounces that often (a few per pixel of 64) is a different, equally valid sample.  The image is the same estimate
# with other noise (no bias: the means agree to 1e-4), not the same bits.  The asserts leave a margin below the measured values.
MIN_WITHIN_1_255 = 0.65         # share of colour channels within +-1/255 of the oracle
MIN_WITHIN_4_255 = 0.85
MIN_PSNR_DB = 36.0              # over the compared rows, channels in [0, 1]
MAX_MEAN_DIFF = 1e-3            # |mean(contract) - mean(oracle)| over the compared rows: no bias


def frame(rt, torch, W, O, nx, ny, ns):
    st = rt.alloc_rand_state(nx, ny)
    fb = rt.alloc_fb(nx, ny)
    rt.render_init(nx, ny, st)
    rt.render(fb, nx, ny, ns, W, st, O)
    torch.cuda.synchronize()
    return fb.cpu().numpy().reshape(ny, nx, 3), st.cpu().numpy()


def test_contract_mode_leaves_the_default_mode_untouched(rt, cuda):
    torch = cuda
    nx, ny, ns, n, spl = 400, 232, 16, 10000, 32
    W = rt.World(n, nx, ny).upload()
    O = rt.Octree(W, spl).upload()
    a, sa = frame(rt, torch, W, O, nx, ny, ns)
    W.set_arith(rt.ARITH_CONTRACT)
    c, _ = frame(rt, torch, W, O, nx, ny, ns)
    W.set_arith(rt.ARITH_IEEE)
    b, sb = frame(rt, torch, W, O, nx, ny, ns)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(sa, sb)      # the parity mode: the same bits as before
    S = OracleScene(n, nx, ny, use_octree=True, spl=spl)
    ref, _ = S.render(ns, nthreads=8)
    assert np.array_equal(a.view(np.uint32), ref.view(np.uint32))
    differs = np.mean(a.view(np.uint32) != c.view(np.uint32))
    assert 0.0 < differs                                        # contraction does change last bits ...
    fin = np.isfinite(a) & np.isfinite(c)
    w16 = float(np.mean(np.abs(a[fin] - c[fin]) <= 1.0 / 255.0))
    print("contract vs ieee, 400x232x16: %.3f %% of channels differ in bits, %.3f %% within 1/255" % (100 * differs, 100 * w16))
    assert w16 > 0.75                                           # ... and little else (16 spp: a flipped sample weighs 4x what it does at 64)
    # the list path (no octree) takes the same kernels
    W2 = rt.World(500, nx, ny).upload()
    l0, _ = frame(rt, torch, W2, None, nx, ny, 4)
    W2.set_arith(rt.ARITH_CONTRACT)
    l1, _ = frame(rt, torch, W2, None, nx, ny, 4)
    fin = np.isfinite(l0) & np.isfinite(l1)
    w4 = float(np.mean(np.abs(l0[fin] - l1[fin]) <= 1.0 / 255.0))
    print("contract vs ieee, list path 400x232x4: %.3f %% within 1/255" % (100 * w4))
    assert w4 > 0.98


def test_contract_mode_c3_rows_within_tolerance_of_the_oracle(rt, cuda):
    """C3 (1200x800, 64 spp, N = 10000, octree SPL 32) rendered with contraction allowed, eight rows against the oracle."""
    torch = cuda
    nx, ny, ns, n, spl = 1200, 800, 64, 10000, 32
    W = rt.World(n, nx, ny).upload()
    O = rt.Octree(W, spl).upload()
    W.set_arith(rt.ARITH_CONTRACT)
    got, _ = frame(rt, torch, W, O, nx, ny, ns)
    S = OracleScene(n, nx, ny, use_octree=True, spl=spl)
    rows = (3, 100, 250, 316, 317, 431, 600, 797)
    ref = np.stack([S.render(ns, row0=r, rows=1, nthreads=4)[0][0] for r in rows])
    g = np.stack([got[r] for r in rows])
    fin = np.isfinite(ref) & np.isfinite(g)
    assert fin.mean() > 0.999
    d = np.abs(g[fin].astype(np.float64) - ref[fin].astype(np.float64))
    within = float(np.mean(d <= 1.0 / 255.0))
    psnr = float(10.0 * np.log10(1.0 / max(np.mean(d * d), 1e-30)))
    exact = float(np.mean(g.view(np.uint32) == ref.view(np.uint32)))
    within4 = float(np.mean(d <= 4.0 / 255.0))
    bias = float(np.mean(g[fin].astype(np.float64)) - np.mean(ref[fin].astype(np.float64)))
    print("contract mode, C3 rows %s: %.2f %% of channels within 1/255, %.2f %% within 4/255, PSNR %.1f dB, max |d| %.4f, mean difference %+.2e, bit-equal channels %.2f %%" % (
        rows, 100 * within, 100 * within4, psnr, d.max(), bias, 100 * exact))
    assert within >= MIN_WITHIN_1_255 and within4 >= MIN_WITHIN_4_255 and psnr >= MIN_PSNR_DB and abs(bias) <= MAX_MEAN_DIFF


def test_contract_mode_is_fp32_only(rt, cuda):
    W = rt.World(22, 64, 40, precision=rt.FP16)
    with pytest.raises(rt.RtError):
        W.set_arith(rt.ARITH_CONTRACT)
    with pytest.raises(rt.RtError):
        rt.World(22, 64, 40).set_arith(7)
