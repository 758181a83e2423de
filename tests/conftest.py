import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """the product binding; building is __graft_entry__.build()'s job, loading failure is an error (no fallback)"""
    import rt_amd
    if not os.path.exists(rt_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    rt_amd.lib()
    return rt_amd


@pytest.fixture(scope="session")
def cuda(rt):
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    rc, n = rt.device_check()
    assert rc == 0 and n >= 1, "rt_device_check failed: %d" % rc
    return torch
