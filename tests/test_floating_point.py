"""utils/floating_point.py against the reference's own outputs (tests/golden/floating_point.json,
made by tests/golden/make_golden.py from /root/reference) and the known answers its tests hold
(test_simple_mip_solver/test_utils/test_floating_point.py:22-33, :113-145, :190-245)."""
import json
from math import isclose
import os

import numpy as np
import pytest

from simple_mip_solver_amd.lp import CyLPArray
from simple_mip_solver_amd.utils.floating_point import get_fraction, numerically_safe_cut, scale_cut
from simple_mip_solver_amd.utils.tolerance import exact_coefficient_approximation_epsilon as eps

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'floating_point.json')))


def test_get_fraction_matches_reference_vectors():
    assert len(GOLD['get_fraction']) > 1000
    for rec in GOLD['get_fraction']:
        n, d = get_fraction(rec['x'], max_term=rec['max_term'], estimate=rec['estimate'])
        assert (n, d) == (rec['n'], rec['d']), rec


def test_numerically_safe_cut_matches_reference_vectors():
    for rec in GOLD['numerically_safe_cut']:
        pi, pi0 = numerically_safe_cut(CyLPArray(rec['pi']), rec['pi0'], estimate=rec['estimate'],
                                       make_integer=rec['make_integer'])
        assert np.array_equal(np.asarray(pi), np.asarray(rec['safe_pi'], float)), rec
        assert pi0 == rec['safe_pi0'], rec


def test_scale_cut_matches_reference_vectors():
    for rec in GOLD['scale_cut']:
        pi, pi0 = scale_cut(np.asarray(rec['pi'], float), rec['pi0'])
        if rec['out_pi'] is None:
            assert pi is None and pi0 is None
        else:
            assert np.array_equal(pi, np.asarray(rec['out_pi'])) and pi0 == rec['out_pi0']


def test_scale_cut_fails_asserts():
    with pytest.raises(AssertionError, match='pi is an nd.array'):
        scale_cut(pi=(1, 2, 3), pi0=4)
    with pytest.raises(AssertionError, match='pi0 is a number'):
        scale_cut(pi=np.array([1, 2, 3]), pi0='4')
    with pytest.raises(AssertionError, match='max_abs should be positive'):
        scale_cut(pi=np.array([1, 2, 3]), pi0=4, max_abs=-1)


def test_scale_cut():
    pi, pi0 = scale_cut(np.array([1, 0, 2, -4]), 3)
    assert all(pi == [.25, 0, .5, -1]) and pi0 == .75
    pi, pi0 = scale_cut(np.array([0, 0, 0, 0]), 3)
    assert pi is None and pi0 is None


def test_numerically_safe_cut_fails_asserts():
    with pytest.raises(AssertionError, match='pi is a CyLPArray'):
        numerically_safe_cut(pi=np.array([1, 2, 3]), pi0=4)
    with pytest.raises(AssertionError, match='pi0 is a number'):
        numerically_safe_cut(pi=CyLPArray([1, 2, 3]), pi0='4')
    with pytest.raises(AssertionError, match='estimate must be over or under'):
        numerically_safe_cut(pi=CyLPArray([1, 2, 3]), pi0=4, estimate=None)


def test_numerically_safe_cut_really_close():
    pi, pi0 = numerically_safe_cut(pi=CyLPArray([1.25, 2.375, 4]) + eps / 10, pi0=4,
                                   estimate='over', make_integer=True)
    assert all(pi == np.array([10, 19, 32])) and isclose(pi0, 31, abs_tol=eps)
    pi, pi0 = numerically_safe_cut(pi=CyLPArray([1, .333333333333333, .666666666666667]), pi0=1,
                                   estimate='under', make_integer=True)
    assert all(pi == np.array([3, 1, 2])) and isclose(pi0, 3, abs_tol=eps)


def test_numerically_safe_cut_high_dynamism():
    pi, pi0 = numerically_safe_cut(pi=CyLPArray([1, 100, 10000]), pi0=100, estimate='over',
                                   make_integer=True, max_term=1000)
    assert all(pi == np.array([100, 1, 100])) and isclose(pi0, 1, abs_tol=eps)
    pi, pi0 = numerically_safe_cut(pi=CyLPArray([100, 9999, 10000]), pi0=100, estimate='under',
                                   make_integer=True, max_term=1000)
    assert all(pi == np.array([1, 0, 100])) and isclose(pi0, 1, abs_tol=eps)


def test_get_fraction_fails_asserts():
    with pytest.raises(AssertionError, match='should be an int or float'):
        get_fraction('5')
    with pytest.raises(AssertionError, match='should be positive'):
        get_fraction(5, max_term=-1)
    with pytest.raises(AssertionError, match="estimate should be 'over' or 'under'"):
        get_fraction(5, estimate='sideways')


DEAD_ZONES = [  # (x, estimate, n, d) with max_term = 1000, reference test_floating_point.py:190-245
    (0.00000001, None, 0, 1), (0.00000001, 'under', 0, 1), (0.00000001, 'over', 1, 1),
    (0.99999999, None, 1, 1), (0.99999999, 'under', 0, 1), (0.99999999, 'over', 1, 1),
    (3141.59, None, 3142, 1), (3141.59, 'under', 3141, 1), (3141.59, 'over', 3142, 1),
    (-0.00000001, None, 0, 1), (-0.00000001, 'under', -1, 1), (-0.00000001, 'over', 0, 1),
    (-0.99999999, None, -1, 1), (-0.99999999, 'under', -1, 1), (-0.99999999, 'over', 0, 1),
    (-3141.59, None, -3142, 1), (-3141.59, 'under', -3142, 1), (-3141.59, 'over', -3141, 1),
]


@pytest.mark.parametrize('x,estimate,n,d', DEAD_ZONES)
def test_get_fraction_dead_zones(x, estimate, n, d):
    assert get_fraction(x, max_term=1000, estimate=estimate) == (n, d)


def test_get_fraction_directed_property():
    # reference test_floating_point.py:160-188 (1000 random draws; seeded here)
    rng = np.random.default_rng(3)
    for _ in range(1000):
        x = float(rng.uniform(-1000, 1000)) * 10 ** float(rng.integers(-3, 1))
        n, d = get_fraction(x, max_term=1000, estimate='under')
        assert n / d <= x and d > 0
        n, d = get_fraction(x, max_term=1000, estimate='over')
        assert x <= n / d and d > 0


def test_safe_cut_is_outer_approximation():
    # property of reference test_floating_point.py:60-111, checked without an LP: for x >= 0 any
    # point satisfying pi.x >= pi0 also satisfies the 'over' rounded cut (coefficients only grow
    # relative to the scaled cut, the right-hand side only shrinks)
    rng = np.random.default_rng(5)
    for _ in range(100):
        k = int(rng.integers(2, 8))
        pi = CyLPArray(rng.uniform(-5, 5, k))
        pi0 = float(rng.uniform(-5, 5))
        spi, spi0 = numerically_safe_cut(pi, pi0, estimate='over')
        scale = 1 / np.max(np.abs(pi))
        assert np.all(np.asarray(spi) >= np.asarray(pi) * scale - 1e-12)
        assert spi0 <= pi0 * scale + 1e-12
        upi, upi0 = numerically_safe_cut(pi, pi0, estimate='under')
        assert np.all(np.asarray(upi) <= np.asarray(pi) * scale + 1e-12)
        assert upi0 >= pi0 * scale - 1e-12
