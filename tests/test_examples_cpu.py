"""CPU: the C++ callers under examples/ compile against csrc/solver.h (header-level drop-in), and
the host-only one runs: accessor().block() / diagBlock() views (Accessor.h:69-107,165-200) write a
matrix that densify() reads back exactly, flipped blocks included."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_examples_build_and_accessor_views_run():
    subprocess.check_call(["bash", os.path.join(ROOT, "examples", "build.sh")])
    assert os.path.exists(os.path.join(ROOT, "examples", "bal_bench"))
    out = subprocess.run([os.path.join(ROOT, "examples", "accessor_views")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ACCESSOR_VIEWS_OK" in out.stdout
