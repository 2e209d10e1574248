import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "product_defaults: run with the product's own schedule choices")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """the HIP library and the oracle must exist (built in-tree by __graft_entry__.build())"""
    from baspacho_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    from oracle import cref
    cref.build()


@pytest.fixture(autouse=True)
def _lookahead_on_small_problems(request, monkeypatch):
    """The product hands lookahead units to the auxiliary streams only when they are worth the forks
    (HipPlanHost::lookaheadPays): the small matrices of the parity tests would never reach that
    schedule.  Tests therefore run with the threshold at 0 (side streams whenever a plan has
    lookahead units; BSP_NO_LOOKAHEAD=1 in test_schedule_variants covers the in-line order);
    tests/test_full_size_gpu.py and tests marked `product_defaults` keep the product's own choice."""
    if request.node.module.__name__ == "test_full_size_gpu" or request.node.get_closest_marker("product_defaults"):
        return
    monkeypatch.setenv("BSP_LOOKAHEAD_MIN_GF", "0")
