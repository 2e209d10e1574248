"""GPU: the randomised parity sweeps as part of the driver-run suite (round 4: 240 random cases, every
lump width 1 .. 1100 with and without a lump below; tools/stress.py and tools/sweep_widths.py run the
same cases by hand -- round 3 found a wrong factor with them that no pytest had seen)."""
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


@pytest.mark.parametrize("seed", range(3000, 3160))
def test_random_case_more(seed):
    """round 4: 160 further seeds of the same generator (random structure / sizes / elimination set /
    dtype / batch; factor, solves, partial factor + solves, addMvFrom, per-op loops in one of five)"""
    run_case(seed)


@pytest.mark.parametrize("seed", range(4000, 4040))
def test_random_case_wide_lumps_more(seed):
    run_case(seed, big=True)


@pytest.mark.parametrize("first", range(1, 1101, 10))
def test_every_lump_width(first):
    """tools/sweep_widths.py inside the suite: ten consecutive widths of a dense last lump per test,
    alone and followed by a second lump (70 columns), device factor against numpy, fp64"""
    from stress_cases import run_width_case
    for W in range(first, min(first + 10, 1101)):
        for tail in (0, 70):
            err = run_width_case(W, tail)
            assert err < 1e-12, (W, tail, err)


@pytest.mark.parametrize("first", range(1, 1101, 44))
def test_lump_widths_fp32(first):
    """the same sweep in single precision, every fourth width"""
    import numpy as np
    from stress_cases import run_width_case
    for W in range(first, min(first + 44, 1101), 4):
        for tail in (0, 70):
            err = run_width_case(W, tail, dtype=np.float32)
            assert err < 2e-5, (W, tail, err)


def test_seed_that_found_the_single_tile_block_last_step():
    """tools/stress.py seed 9426 (round 3): a last lump 1600 = 6 x 256 + 64 columns wide"""
    run_case(9426)


@pytest.mark.parametrize("tail", [0, 70])
def test_lump_width_slice(tail):
    """every lump width around the panel (64) and outer-block (256) boundaries, with and without a
    second lump below (tools/sweep_widths.py walks 1 .. 1100: round 3 ran it clean)"""
    import numpy as np
    from stress_cases import run_width_case
    widths = list(range(56, 72)) + list(range(250, 262)) + list(range(314, 326)) + list(range(506, 518)) + \
        list(range(570, 582)) + [832, 1088, 1089]
    for W in widths:
        err = run_width_case(W, tail)
        assert err < 1e-12, (W, tail, err)
    for W in (64, 320, 576, 577):
        err = run_width_case(W, tail, dtype=np.float32)
        assert err < 2e-5, (W, tail, err)


@pytest.mark.parametrize("seed", range(0, 24))
def test_structured_family_case(seed):
    """grid / meridians / flat + Schur-set structures at random small sizes (deep elimination trees)"""
    from stress_cases import run_family_case
    run_family_case(seed)
