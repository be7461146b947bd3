"""Split-K finished inside the few-row GEMM kernels (csrc/vita_splitk_fused.h: partial tiles stored write-through, the last
workgroup to arrive at the tile's counter sums them in split order and applies the epilogue) must give the bits of the two-pass
path it replaces: tools/fused_splitk_check.py runs the B = 128 / 77 training step (losses and every gradient, repeated: the
counters must return to zero) and the per-launch sampler at 65..512 rows once with VITA_NO_FUSED_SPLITK=1 and once with the
default, in fresh processes (the switch is read once per process), and compares bitwise."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_in_kernel_splitk_finish_is_bitwise_the_two_pass_reduction(tmp_path):
    ref = str(tmp_path / "two_pass.pt")
    tool = os.path.join(ROOT, "tools", "fused_splitk_check.py")
    env = dict(os.environ, VITA_NO_FUSED_SPLITK="1")
    a = subprocess.run([sys.executable, tool, "--save", ref], env=env, capture_output=True, text=True, timeout=140)
    assert a.returncode == 0 and '"fused": false' in a.stdout, (a.stdout[-2000:], a.stderr[-2000:])
    env.pop("VITA_NO_FUSED_SPLITK")
    b = subprocess.run([sys.executable, tool, "--compare", ref], env=env, capture_output=True, text=True, timeout=140)
    assert b.returncode == 0 and '"fused": true' in b.stdout and '"bitwise_equal": true' in b.stdout, (b.stdout[-2000:], b.stderr[-2000:])
