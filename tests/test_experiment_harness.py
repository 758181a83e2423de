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


@pytest.mark.gpu
def test_reduced_sweep_keeps_every_run_and_tests_significance(rt, cuda):
    import run_experiment as rx
    cells = rx.run([488, 2000], [0.1], verbose=False)
    assert [c["n"] for c in cells] == [488, 2000]
    for c in cells:
        assert len(c["list_runs_ms"]) == 5 and len(c["octree_runs_ms"]) == 5 and len(c["list_grid_runs_ms"]) == 5
        assert all(t > 0 for t in c["list_runs_ms"] + c["octree_runs_ms"])
        assert 0.0 <= c["welch_p"] <= 1.0
    big = cells[1]
    assert big["speedup"] > 2.0 and big["significant_5pct"]        # 2000 spheres: the octree wins clearly (reference: 2.57x, evaluations.ipynb:1151)
