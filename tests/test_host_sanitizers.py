"""Host-side code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only — sanitizer flags never touch the gfx950 build).

Three programs are built with `-fsanitize=address,undefined -fno-sanitize-recover=undefined` and must run clean:
  * the product's host layer (host/rt_scene.hpp: create_world, buildOctree; host/rt_image.hpp: the P3 / P6 / PFM writers behind
    rt_format_ppm / rt_write_image) at the BASELINE sizes — N = 10 000 / SPL 32, N = 100 000 / SPL 320, full buckets, both precisions;
  * the host-callable interfaces (hit / scatter / get_ray) rendering small frames (tests/host/host_iface_check.cpp);
  * the CPU oracle's C entry points (oracle/rt_oracle_capi.cpp) — the checker has to be sound itself.
The reference's own host code is the cautionary example: main.cu:410 allocates the octree with `new`, main.cu:473 releases it with free()."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
       "-ffp-contract=off", "-fno-fast-math", "-pthread"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def build(tmp, name, sources):
    out = str(tmp / name)
    p = subprocess.run(["g++"] + SAN + ["-o", out] + [os.path.join(ROOT, s) for s in sources], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return out


def run_clean(cmd, timeout=600):
    p = subprocess.run(cmd, capture_output=True, env=ENV, timeout=timeout)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, "exit %d\n%s" % (p.returncode, err[-4000:])
    assert "runtime error" not in err and "AddressSanitizer" not in err and "LeakSanitizer" not in err, err[-4000:]
    return p.stdout


@pytest.fixture(scope="module")
def tmp(tmp_path_factory):
    return tmp_path_factory.mktemp("sanitize")


def test_product_host_layer_under_asan_ubsan(tmp):
    exe = build(tmp, "host_sanitize_check", ["tests/host/host_sanitize_check.cpp"])
    out = run_clean([exe, "big"]).decode()
    # the counts are the pinned ones of SURVEY.md 8c (tests/test_oracle_pins.py): the sanitized build computes the same worlds and trees
    assert "fp32 N=22 spl=30 created=20 nodes=114 leaves=81 " in out
    assert "fp32 N=10000 spl=32 created=9805 nodes=157 leaves=438 entries=11368 dropped_full=309 " in out
    assert "fp32 N=100000 spl=320 created=99860 nodes=157 " in out and "dropped_full=0 dropped_outside=0" in out.split("N=100000")[1].splitlines()[0]
    line = [l for l in out.splitlines() if l.startswith("fp32 N=2000 spl=3 ")][0]
    assert int(line.split("dropped_full=")[1].split()[0]) > 0            # the "leaf nodes are full" path ran
    assert "images p3=" in out


@pytest.mark.parametrize("args", [("0", "22", "24", "14", "2"), ("1", "22", "16", "10", "1")])
def test_host_interfaces_under_asan_ubsan(tmp, args):
    exe = build(tmp, "host_iface_check_san", ["tests/host/host_iface_check.cpp"])
    raw = run_clean([exe] + list(args))
    assert len(raw) == int(args[2]) * int(args[3]) * 3 * 4


def test_oracle_entry_points_under_asan_ubsan(tmp):
    exe = build(tmp, "oracle_sanitize", ["tests/host/oracle_sanitize_main.cpp", "oracle/rt_oracle_capi.cpp"])
    out = run_clean([exe]).decode()
    lines = out.strip().splitlines()
    assert len(lines) == 7
    # list and octree render the same N = 22 frame (SURVEY fact 6 at this size), also in the sanitized build
    assert lines[0].split("frame=")[1].split()[0] == lines[1].split("frame=")[1].split()[0]
    assert "dropped_full=834" in lines[3] and "real=9805 nodes=157 leaves=438 entries=11368 dropped_full=309" in lines[6]
