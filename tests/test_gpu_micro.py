"""Exhaustive GPU checks of arithmetic shortcuts the kernels rely on (built by __graft_entry__.build())."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_shared_reciprocal_division_equals_ieee_division_for_all_binary16_pairs(cuda):
    """rt_kernels_fp16.hip forms the 27 plane quotients of a ray with div_prepare / div_by: the compiler's IEEE division without its
    v_div_scale steps, the refined reciprocal shared per divisor.  tools/micro/div_shared compares the two forms for all 2^32 pairs
    of binary16 operands, bits for bits (zeros, subnormals, infinities, NaNs included); exit code 0 = no quotient differs."""
    exe = os.path.join(ROOT, "tools", "micro", "div_shared")
    assert os.access(exe, os.X_OK), "tools/micro/div_shared is missing: __graft_entry__.build() builds it"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 float quotients differ" in p.stdout and " 0 after rounding to binary16" in p.stdout, p.stdout
