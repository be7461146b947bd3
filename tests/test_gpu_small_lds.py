"""The LDS-staged few-row GEMM (gemm_f32_small_lds_kernel) must give the bits of the register-fragment kernel it replaces
(same MFMAs, same order): tools/small_lds_check.py runs a shape set (ragged M / N, K not a multiple of 32, row and column
weight operands, every epilogue form) once with VITA_SMALL_NO_LDS=1 and once with the default, in fresh processes (the switch
is read once per process), and compares the outputs bitwise."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_lds_staged_small_gemm_is_bitwise_the_register_kernel(tmp_path):
    ref = str(tmp_path / "small_ref.pt")
    tool = os.path.join(ROOT, "tools", "small_lds_check.py")
    env = dict(os.environ, VITA_SMALL_NO_LDS="1")
    a = subprocess.run([sys.executable, tool, "--save", ref], env=env, capture_output=True, text=True, timeout=140)
    assert a.returncode == 0, a.stderr[-2000:]
    env.pop("VITA_SMALL_NO_LDS")
    b = subprocess.run([sys.executable, tool, "--compare", ref], env=env, capture_output=True, text=True, timeout=140)
    assert b.returncode == 0 and '"bitwise_equal": true' in b.stdout, (b.stdout[-2000:], b.stderr[-2000:])
