"""The host-callable interfaces of host/rt_scene.hpp — hitable::hit (hitable.h:19), material::scatter (material.h:49),
camera::get_ray (camera.h:45) — driven by a test program that restates render()/color() of main.cu over them; its frames equal
the CPU oracle's bit for bit, in fp32 and in the USE_FP16 arithmetic."""
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import OracleScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("host_iface") / "host_iface_check"
    src = os.path.join(ROOT, "tests", "host", "host_iface_check.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-o", str(out), src])
    return str(out)


@pytest.mark.parametrize("fp16,n,nx,ny,ns", [(0, 22, 48, 28, 3), (0, 500, 40, 24, 2), (1, 22, 48, 28, 3), (1, 500, 32, 20, 2)])
def test_frames_through_the_host_interfaces_equal_the_oracle(exe, fp16, n, nx, ny, ns):
    raw = subprocess.run([exe, str(fp16), str(n), str(nx), str(ny), str(ns)], capture_output=True, check=True, timeout=300).stdout
    got = np.frombuffer(raw, np.float32).reshape(ny, nx, 3)
    ref, _ = OracleScene(n, nx, ny, fp16=bool(fp16), use_octree=False).render(ns, nthreads=8)
    nan = np.isnan(ref)
    assert np.array_equal(got.view(np.uint32)[~nan], ref.view(np.uint32)[~nan])
    assert np.isnan(got[nan]).all()
