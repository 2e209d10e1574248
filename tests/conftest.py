import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
