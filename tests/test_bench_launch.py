"""bench.py started WITHOUT a launcher (`python3 bench.py --gpus N`): the parent starts its own ranks as a child process group,
relays their outcome and never hangs.  CPU container: the ranks meet over gloo and then refuse to run without a GPU — the
parent must come back non-zero with their message; a rank that never arrives is killed with the whole group at --timeout.
The GPU side (two ranks on one GPU, a real line) is tests/test_gpu_multi.py::test_bench_self_launch_two_ranks_one_gpu."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def _run(extra, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-pmc"] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode(), time.time() - t0


@pytest.mark.skipif(not _no_gpu(), reason="CPU-only check: on a GPU box this would run the real bench (covered by the -m gpu test)")
def test_self_launch_relays_a_failing_job():
    rc, out, err, _ = _run([])
    assert rc != 0
    assert out.strip() == ""                                   # no result line when the ranks failed
    assert "needs a GPU" in err                                # the ranks' own message reaches the caller


@pytest.mark.skipif(not _no_gpu(), reason="CPU-only check")
def test_self_launch_kills_a_hung_job_at_the_timeout():
    rc, out, err, secs = _run(["--timeout", "20"], env={"RT_BENCH_TEST_HANG": "1"})
    assert rc == 124, (rc, err[-1500:])
    assert "stopping it" in err
    assert secs < 90
    # no rank is left behind: the hung rank slept in a session of its own (torch.distributed.run starts its workers that way), out of
    # reach of a signal to the agent's group — the parent records the ranks' pids and kills what survives the agent
    left = []
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                argv = open("/proc/%s/cmdline" % d, "rb").read().split(b"\0")
            except OSError:
                continue
            if any(a.endswith(b"bench.py") for a in argv) and b"--timeout" in argv and b"20" in argv and open("/proc/%s/stat" % d).read().rsplit(")", 1)[1].split()[0] != "Z":
                left.append((int(d), argv))
    assert left == [], left


def test_launcher_form_still_checks_world_size():
    e = dict(os.environ, WORLD_SIZE="3", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=120)
    assert p.returncode != 0 and b"WORLD_SIZE=3" in p.stderr
