"""The continued-fraction rounding of the cut path, pinned in isolation.

Every `get_fraction`, `scale_cut` and `numerically_safe_cut` record the reference's own code
produced (tests/golden/floating_point.json, made by tests/golden/make_golden.py from
/root/reference/simple_mip_solver/utils/floating_point.py:11-167) goes through

  * the CPU oracle (`oracle.get_fraction`, `oracle.safe_cut`, `oracle.safe_cut_ex`)   -- not gpu
  * the device functions K2 rounds with, through the C ABI (`mipx_get_fraction_batch`,
    `mipx_safe_cut_batch`)                                                            -- gpu

Numerators and denominators are integer work: the bar is exact equality of (n, d), and bit-equal
quotients for the rounded cuts."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'floating_point.json')))


def test_golden_file_is_what_the_verdict_counted():
    assert len(GOLD['get_fraction']) == 1710
    assert len(GOLD['numerically_safe_cut']) == 125 and len(GOLD['scale_cut']) == 125


def test_oracle_get_fraction_matches_every_reference_vector(oracle):
    for rec in GOLD['get_fraction']:
        assert oracle.get_fraction(rec['x'], rec['max_term'], rec['estimate']) == (rec['n'], rec['d']), rec


def test_oracle_safe_cut_matches_every_reference_vector(oracle):
    for rec in GOLD['numerically_safe_cut']:
        ex = oracle.safe_cut_ex(rec['pi'], rec['pi0'], rec['estimate'], rec['make_integer'])
        assert np.array_equal(ex['safe_pi'], np.asarray(rec['safe_pi'], float)), rec
        assert ex['safe_pi0'] == rec['safe_pi0'], rec
        if not rec['make_integer']:
            pi, pi0 = oracle.safe_cut(rec['pi'], rec['pi0'], rec['estimate'])   # the form K2's oracle uses
            assert np.array_equal(pi, np.asarray(rec['safe_pi'], float)) and pi0 == rec['safe_pi0'], rec
            # the quotients are the chosen fractions, and they estimate in the safe direction
            assert np.array_equal(ex['num'][:-1] / ex['den'][:-1], ex['safe_pi'])


def directed(rec, scaled_pi, scaled_pi0, safe_pi, safe_pi0):
    """The property that makes the rounding safe (floating_point.py:99-101): an 'over' cut
    pi.x >= pi0 gets every coefficient over- and its right-hand side under-estimated (x >= 0).
    A coefficient may fall short by less than exact_coefficient_approximation_epsilon = 1e-14: the
    reference accepts the undirected convergent where it is that exact (floating_point.py:87-91)."""
    if rec['make_integer']:
        return
    if rec['estimate'] == 'over':
        assert np.all(safe_pi > scaled_pi - 1e-14) and safe_pi0 <= scaled_pi0, rec
    else:
        assert np.all(safe_pi < scaled_pi + 1e-14) and safe_pi0 >= scaled_pi0, rec


def test_oracle_rounding_is_an_outer_approximation(oracle):
    for rec, sc in zip(GOLD['numerically_safe_cut'], GOLD['scale_cut']):
        assert rec['pi'] == sc['pi']
        if sc['out_pi'] is None:
            continue
        ex = oracle.safe_cut_ex(rec['pi'], rec['pi0'], rec['estimate'], rec['make_integer'])
        directed(rec, np.asarray(sc['out_pi']), sc['out_pi0'], ex['safe_pi'], ex['safe_pi0'])


@pytest.mark.gpu
def test_device_get_fraction_matches_every_reference_vector(gpu_ctx, oracle):
    from simple_mip_solver_amd import _ffi
    recs = GOLD['get_fraction']
    num, den = _ffi.get_fraction_batch(gpu_ctx, [r['x'] for r in recs], [r['max_term'] for r in recs],
                                       [r['estimate'] for r in recs])
    for k, rec in enumerate(recs):
        assert (int(num[k]), int(den[k])) == (rec['n'], rec['d']), rec
        assert (int(num[k]), int(den[k])) == oracle.get_fraction(rec['x'], rec['max_term'], rec['estimate'])
    with pytest.raises(_ffi.MipxError, match='MIPX_EINVAL'):
        _ffi.get_fraction_batch(gpu_ctx, [1.5], [-1.0], [None])


@pytest.mark.gpu
def test_device_safe_cut_matches_every_reference_vector(gpu_ctx, oracle):
    """mipx_safe_cut_batch on every numerically_safe_cut / scale_cut record: rounded cut bit-equal
    to the reference's, (n, d) equal to the oracle's, the scaled cut equal to scale_cut's."""
    from simple_mip_solver_amd import _ffi
    groups = {}
    for k, rec in enumerate(GOLD['numerically_safe_cut']):
        groups.setdefault((len(rec['pi']), rec['estimate'], rec['make_integer']), []).append(k)
    seen = 0
    for (n, est, mk), ks in groups.items():
        recs = [GOLD['numerically_safe_cut'][k] for k in ks]
        got = _ffi.safe_cut_batch(gpu_ctx, np.array([r['pi'] for r in recs]), [r['pi0'] for r in recs],
                                  estimate=est, make_integer=mk)
        for i, (k, rec) in enumerate(zip(ks, recs)):
            sc = GOLD['scale_cut'][k]
            assert np.array_equal(got['safe_pi'][i], np.asarray(rec['safe_pi'], float)), rec
            assert got['safe_pi0'][i] == rec['safe_pi0'], rec
            ex = oracle.safe_cut_ex(rec['pi'], rec['pi0'], est, mk)
            assert np.array_equal(got['num'][i].astype(np.int64), ex['num']), rec
            assert np.array_equal(got['den'][i].astype(np.int64), ex['den']), rec
            if sc['out_pi'] is None:
                assert got['nonzero'][i] == 0
            else:
                assert got['nonzero'][i] == 1
                assert np.array_equal(got['scaled_pi'][i], np.asarray(sc['out_pi'])), rec
                assert got['scaled_pi0'][i] == sc['out_pi0'], rec
                directed(rec, got['scaled_pi'][i], got['scaled_pi0'][i], got['safe_pi'][i], got['safe_pi0'][i])
            seen += 1
    assert seen == 125


@pytest.mark.gpu
def test_device_safe_cut_at_benchmark_width(gpu_ctx, oracle):
    """Cuts of 256 and 1000 coefficients (several 256-wide strides of the kernel), random magnitudes
    over twelve decades, zeros and negatives: device == oracle bit for bit, (n, d) included."""
    from simple_mip_solver_amd import _ffi
    rng = np.random.default_rng(7)
    for n in (256, 1000):
        pi = rng.standard_normal((32, n)) * 10.0 ** rng.integers(-6, 6, (32, n))
        pi[rng.random((32, n)) < 0.1] = 0.0
        pi[3] = 0.0                                  # a zero cut is returned unchanged
        pi0 = rng.standard_normal(32) * 100
        for est in ('over', 'under'):
            got = _ffi.safe_cut_batch(gpu_ctx, pi, pi0, estimate=est)
            for k in range(32):
                ex = oracle.safe_cut_ex(pi[k], pi0[k], est)
                assert np.array_equal(got['safe_pi'][k], ex['safe_pi']) and got['safe_pi0'][k] == ex['safe_pi0']
                assert np.array_equal(got['num'][k].astype(np.int64), ex['num'])
                assert np.array_equal(got['den'][k].astype(np.int64), ex['den'])
            assert got['nonzero'][3] == 0 and np.all(got['nonzero'][[0, 1, 2, 4]] == 1)
