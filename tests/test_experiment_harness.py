"""The experiment harness (tools/run_experiment.py: the reference's analysis/run_experiment.sh protocol + the Welch t-test of
evaluations.ipynb:1640-1651): statistics on the CPU, a reduced sweep on the GPU."""
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_welch_t_matches_scipy_and_a_hand_computed_case():
    from scipy import stats
    import run_experiment as rx
    a = [121.8, 122.4, 121.1, 123.0, 122.2]
    b = [58.1, 57.7, 58.9, 58.3, 57.9]
    t, df, p = rx.welch_t(a, b)
    ref = stats.ttest_ind(a, b, equal_var=False)
    assert math.isclose(t, ref.statistic, rel_tol=1e-12) and math.isclose(p, ref.pvalue, rel_tol=1e-9)
    # by hand: means 122.1 / 58.18, sample variances 0.5 / 0.212 -> se^2 = 0.1424, t = 63.92 / sqrt(0.1424)
    assert math.isclose(t, (122.1 - 58.18) / math.sqrt(0.5 / 5 + 0.212 / 5), rel_tol=1e-9)
    assert 4.0 < df < 8.0 and p < 1e-8
    # identical samples: nothing to tell apart
    t0, _, p0 = rx.welch_t([1.0, 2.0, 3.0], [1.0, 2.0, 3.0])
    assert t0 == 0.0 and p0 == 1.0
    c = rx.summarise(dict(list_runs_ms=a, octree_runs_ms=b))
    assert c["significant_5pct"] and math.isclose(c["speedup"], 122.1 / 58.18, rel_tol=1e-3)


def test_image_metrics_restate_the_notebooks_comparison():
    """tools/image_metrics.py (evaluations.ipynb:1021-1027 without cv2 / skimage): identical frames, a known PSNR, SSIM's range"""
    import numpy as np
    import image_metrics as im
    rng = np.random.default_rng(5)
    fb = rng.random((40, 64, 3)).astype(np.float32)
    same = im.compare_frames(fb, fb.copy())
    assert same["identical"] and same["ssim"] == 1.0 and same["psnr_db"] is None           # cv2.PSNR of identical images is infinite
    lv = im.ppm_levels(np.array([[[0.0, 0.5, 1.0]], [[np.nan, 2.0, -1.0]]], np.float32))   # rows flipped: the PPM starts with the top row
    assert lv.tolist() == [[[0, 255, 0]], [[0, 127, 255]]]
    g = im.gray8(np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.int64))
    assert g.tolist() == [[255.0, 76.0, 150.0, 29.0]]                                       # cv2's fixed-point luma
    a = np.full((32, 32), 100.0); b = a + 5.0
    assert math.isclose(im.psnr(a, b), 10.0 * math.log10(255.0 ** 2 / 25.0), rel_tol=1e-12)
    noisy = np.clip(fb + 0.2 * rng.standard_normal(fb.shape).astype(np.float32), 0, 1)
    c = im.compare_frames(fb, noisy)
    assert 0.0 < c["ssim"] < 0.9 and 5.0 < c["psnr_db"] < 30.0 and not c["identical"]


@pytest.mark.gpu
def test_reduced_sweep_keeps_every_run_and_tests_significance(rt, cuda, tmp_path):
    import run_experiment as rx
    cells = rx.run([488, 2000], [0.1], verbose=False, image_dir=str(tmp_path))
    assert [c["n"] for c in cells] == [488, 2000]
    for c in cells:
        assert len(c["list_runs_ms"]) == 5 and len(c["octree_runs_ms"]) == 5 and len(c["list_grid_runs_ms"]) == 5
        assert all(t > 0 for t in c["list_runs_ms"] + c["octree_runs_ms"])
        assert 0.0 <= c["welch_p"] <= 1.0
        # the first run's images are kept, and in fp32 the octree image IS the list image: the notebook's comparison says so
        assert c["ppm_md5"]["list"] == c["ppm_md5"]["octree"] and len(c["ppm_md5"]["list"]) == 32
        assert c["image_list_vs_octree"] == {"ssim": 1.0, "psnr_db": None, "identical": True}
        for v in ("list", "octree"):
            f = tmp_path / ("%s_N%d_r0.1.ppm" % (v, c["n"]))
            assert f.exists() and f.read_bytes()[:15] == b"P6\n1200 800\n255"
        assert c["device_memory_mb"] > 50.0                         # RNG states (46 MB) + frame + scene
    big = cells[1]
    assert big["speedup"] > 2.0 and big["significant_5pct"]        # 2000 spheres: the octree wins clearly (reference: 2.57x, evaluations.ipynb:1151)
    # one counter pass per variant of a cell (rocprofv3 --pmc, the profiled program directly behind `--`)
    p = rx.pmc_pass(488, 0.1, cells[0]["spl"], "octree")
    assert "error" not in p, p
    assert p["SQ_INSTS_VALU"] > 1e6 and p["SQ_WAVES"] > 0 and "k_render" in p["kernel"]
