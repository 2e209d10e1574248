"""GPU: a seeded 40-case slice of the randomised parity sweep (tools/stress.py runs the long
version: round 2 ran 320 cases clean, outside pytest, where the driver never saw it)."""
import pytest

from stress_cases import run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(1000, 1030))
def test_random_case(seed):
    run_case(seed)


@pytest.mark.parametrize("seed", range(2000, 2010))
def test_random_case_wide_lumps(seed):
    """300-900 parameters, denser: wide lumps, chain steps and lookahead units"""
    run_case(seed, big=True)
