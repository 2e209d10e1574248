"""GPU: examples/bal_bench -- the reference's BAL_bench caller (benchmarking/BaAtLargeBench.cpp:44-97)
compiled against THIS library's C++ headers (only the include path and the backend enum differ) --
creates a solver, fills the matrix through accessor().block<9,3>() / diagBlock<3>() views, factors
and solves on the device and checks the residual block by block through the same views."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cams,pts", [(40, 3000), (120, 40000)])
def test_cpp_bal_bench_caller(cams, pts):
    exe = os.path.join(ROOT, "examples", "bal_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(ROOT, "examples", "build.sh")])
    out = subprocess.run([exe, str(cams), str(pts)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "BAL_BENCH_OK" in out.stdout, out.stdout
