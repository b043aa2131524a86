"""K2 (Gomory rows + safe rounding) on the GPU against the oracle (bit-exact: same canonical order)
and against the reference's own outputs (tests/golden/base_node.json; raw cuts to 4 ulp of the
largest coefficient, rounded cuts identical except at the six listed knife-edge coefficients)."""
import json
import os

import numpy as np
import pytest

from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'base_node.json')))
INF = np.inf


def same(g, o):
    assert np.array_equal(g['row_idx'], o['row_idx'])
    for key in ('pi', 'pi0', 'safe_pi', 'safe_pi0'):
        assert np.array_equal(g[key], o[key]), key


# The rounded cuts of K2 differ from the reference's golden ones in exactly these coefficients
# (node, tableau row) -> positions; 6 coefficients in 3 of the 23 golden cuts.  Each is an exact
# small rational after scaling (-121/161, -11/23, -19/23, -751/826 ...) whose double differs by
# 1-2 ulp between the engine's pivoted tableau and numpy's inv(A_B) @ [A | -I] (raw difference
# <= 1.8e-15): a double just below the rational keeps it as its own 'over' estimate, one just
# above must step to the next convergent (-3/4, -10/21, -14/17, -10/11 ...).  Both are valid outer
# approximations.  The rounding itself is identical: the reference's raw cuts, fed through the
# same device functions (mipx_safe_cut_batch), reproduce the reference's rounded cuts bit for bit.
KNIFE_EDGES = {(4, 0): [1, 4, 7], (4, 3): [5], (5, 0): [1, 3]}


@pytest.mark.parametrize('k', range(len(GOLD['nodes'])))
def test_gomory_matches_oracle_and_reference_vectors(k, gpu_ctx, oracle):
    rec = GOLD['nodes'][k]
    A = np.array(rec['A']); u = np.array([INF if v is None else v for v in rec['u']])
    l = np.array(rec['l']); vstat = np.array(rec['vstat'], np.int8); x = np.array(rec['x'])
    p = _ffi.Problem(gpu_ctx, A, rec['b'], rec['c'])
    g = p.gomory_batch(l[None], u[None], vstat[None], x[None], rec['integer_indices'])[0]
    o = oracle.gomory(A, rec['b'], rec['c'], l, u, vstat, x, rec['integer_indices'])
    same(g, o)
    assert sorted(map(str, g['row_idx'])) == sorted(rec['gomory'])
    for c, row in enumerate(g['row_idx']):
        want = rec['gomory'][str(row)]
        raw_ref = np.array(want['pi'])
        assert np.allclose(g['pi'][c], raw_ref, rtol=0, atol=4e-15 * np.max(np.abs(raw_ref)))
        assert abs(g['pi0'][c] - want['pi0']) <= 4e-15 * max(1.0, abs(want['pi0']))
        gen = rec['generated'][f'cut_gomory_0_1_{row}']
        differs = np.where(g['safe_pi'][c] != np.array(gen['pi']))[0].tolist()
        assert differs == KNIFE_EDGES.get((k, int(row)), []), (k, int(row), differs)
        assert g['safe_pi0'][c] == gen['pi0']
        # the rounding is safe whichever side of the knife edge the raw coefficient fell on: an outer
        # approximation of the engine's own scaled cut (over-estimated coefficients, x >= 0, and an
        # under-estimated right-hand side; 1e-14: the reference's exact-convergent rule)
        scale = np.min(np.abs(1.0 / g['pi'][c][g['pi'][c] != 0]))
        assert np.all(g['safe_pi'][c] > g['pi'][c] * scale - 1e-14)
        assert g['safe_pi0'][c] <= g['pi0'][c] * scale
        for j in differs:   # a knife edge moves one rational estimate, never by more than the 1 % band
            assert abs(g['safe_pi'][c][j] / gen['pi'][j] - 1.0) < 1e-2
        # and the device rounding of the REFERENCE's raw cut is the reference's rounded cut
        dev = _ffi.safe_cut_batch(gpu_ctx, raw_ref, [want['pi0']], estimate='over')
        assert np.array_equal(dev['safe_pi'][0], np.array(gen['pi'])) and dev['safe_pi0'][0] == gen['pi0']


def test_knife_edge_list_is_complete():
    assert sum(len(v) for v in KNIFE_EDGES.values()) == 6 and len(KNIFE_EDGES) == 3
    assert sum(len(rec['gomory']) for rec in GOLD['nodes']) == 23


@pytest.mark.parametrize('n,m,seed,boxed', [(64, 32, 0, True), (64, 32, 5, False), (100, 40, 1, True),
                                             (256, 128, 0, True), (256, 128, 1, False)])
def test_gomory_batch_on_random_roots_and_children(n, m, seed, boxed, gpu_ctx, oracle):
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    if not boxed:
        u = np.full(n, INF)   # every nonbasic variable sits at 0: the textbook GMI setting
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = p.solve_batch(l[None], u[None])
    x = root['x'][0]
    frac = np.minimum(x - np.floor(x), np.ceil(x) - x)
    L, U = [l], [u]
    for j in np.argsort(-frac, kind='stable')[:3]:
        if frac[j] > 1e-4:
            u2 = u.copy(); u2[j] = np.floor(x[j]); L.append(l); U.append(u2)
    L, U = np.array(L), np.array(U)
    V = np.repeat(root['vstat'], len(L), axis=0)
    sol = p.solve_batch(L, U, V)
    ok = sol['status'] == 0
    L, U, V2, X = L[ok], U[ok], sol['vstat'][ok], sol['x'][ok]
    cuts = p.gomory_batch(L, U, V2, X, ints)
    assert len(cuts) == len(L) and sum(len(c['row_idx']) for c in cuts) > 0
    for k in range(len(L)):
        o = oracle.gomory(A, b, c, L[k], U[k], V2[k], X[k], ints)
        same(cuts[k], o)
        xk = np.maximum(X[k], 0)
        at_zero = not boxed and np.all(U[k] == INF)
        for c_ in range(len(cuts[k]['row_idx'])):
            if at_zero:
                # with every nonbasic at 0 the GMI cut reads sum(...) >= 1 and the vertex gives 0:
                # pi.x - pi0 = -1 after the slack substitution.  (The reference applies the same
                # formula when nonbasics sit at upper bounds, where this does not hold.)
                assert abs(cuts[k]['pi'][c_] @ xk - cuts[k]['pi0'][c_] + 1.0) < 1e-6


def test_select_cuts_matches_oracle_and_reference_vectors(gpu_ctx, oracle):
    import math
    names = {0: None, 1: 'no cuts', 2: 'no improving cuts', 3: 'no sufficient cuts'}
    for rec in GOLD['select_cuts']:
        kw, pool = rec['kwargs'], rec['pool']
        keys = list(pool)
        pi = np.array([pool[k]['pi'] for k in keys], float)
        pi0 = np.array([pool[k]['pi0'] for k in keys], float)
        args = (pi, pi0, rec['x'], kw.get('max_nonzero_coefs', 1000000), kw.get('min_cut_depth', 1e-8),
                math.cos(math.radians(kw.get('parallel_cut_tolerance', 10))),
                kw.get('max_relative_cut_term_ratio', 1000) * 1.0)
        ga, gt, gd = _ffi.select_cuts(gpu_ctx, *args)
        oa, ot, od = oracle.select_cuts(*args)
        assert np.array_equal(ga, oa) and gt == ot and np.array_equal(gd, od)
        assert [keys[i] for i in ga] == rec['selected'] and names[gt] == rec['terminator']


def test_select_cuts_on_generated_pools(gpu_ctx, oracle):
    import math
    rng = np.random.default_rng(11)
    for trial in range(30):
        n = int(rng.integers(3, 300)); K = int(rng.integers(0, 40))
        x = rng.uniform(0, 5, n)
        pi = rng.uniform(-3, 1, (K, n)) * (rng.random((K, n)) < 0.6)
        if K > 3:
            pi[1] = pi[0] * 1.001          # nearly parallel pair
            pi[2] = 0.0                    # empty support
        pi0 = rng.uniform(-2 * n, 1, K)
        args = (pi, pi0, x, int(rng.integers(1, n + 1)), 1e-8, math.cos(math.radians(10)), 1000.0)
        ga, gt, gd = _ffi.select_cuts(gpu_ctx, *args)
        oa, ot, od = oracle.select_cuts(*args)
        assert np.array_equal(ga, oa) and gt == ot and np.array_equal(gd, od), trial


def test_mfma_substitution_experiment_distance_from_the_canonical_order(gpu_ctx, capsys):
    """MIPX_K2_MFMA=1 (an experiment, never the default): K2's slack substitution pi + A' pi_s as
    v_mfma_f64_16x16x4_f64 tiles -- rows reduced four at a time, fused -- instead of the reference's
    row-by-row multiply-then-add.  Not asserted equal: MEASURED.  On the reference-generated 64 x 32 and
    256 x 128 nodes (tests/golden/base_node_large.npz) the rows are the same, the raw coefficients move by
    a few ulp of the cut's largest coefficient, and the safely rounded cuts -- what the selection sees --
    almost never change (the continued-fraction rounding absorbs last-bit noise except at knife edges)."""
    import os
    Z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'base_node_large.npz'))
    worst, coefs, moved_raw, moved_rounded, cuts = 0.0, 0, 0, 0, 0
    for ci, (n, m, seed, density, boxed) in enumerate(Z['cases']):
        A, b, c, _, _, ints = random_dense_milp_arrays(int(n), int(m), density=float(density), seed=int(seed))
        p = _ffi.Problem(gpu_ctx, A, b, c)
        keys = [f'c{ci}_k{k}_' for k in range(int(Z[f'c{ci}_count'][0]))]
        args = (np.stack([Z[q + 'l'] for q in keys]), np.stack([Z[q + 'u'] for q in keys]),
                np.stack([Z[q + 'vstat'] for q in keys]), np.stack([Z[q + 'x'] for q in keys]), ints)
        canon = p.gomory_batch(*args)
        os.environ['MIPX_K2_MFMA'] = '1'
        try:
            fused = p.gomory_batch(*args)
        finally:
            os.environ.pop('MIPX_K2_MFMA', None)
        for a, f in zip(canon, fused):
            assert np.array_equal(a['row_idx'], f['row_idx']) and np.array_equal(a['pi0'], f['pi0'])
            scale = np.max(np.abs(a['pi']), axis=1, keepdims=True)
            worst = max(worst, float(np.max(np.abs(a['pi'] - f['pi']) / scale)))
            coefs += a['pi'].size
            cuts += len(a['pi'])
            moved_raw += int(np.sum(a['pi'] != f['pi']))
            moved_rounded += int(np.sum(a['safe_pi'] != f['safe_pi']))
        p.close()
    with capsys.disabled():
        print(f'\n[K2 MFMA experiment] {cuts} cuts, {coefs} coefficients: {moved_raw} raw coefficients differ, largest '
              f'difference {worst:.2e} of the cut\'s largest coefficient ({worst / 2.2e-16:.1f} ulp); '
              f'{moved_rounded} rounded coefficients differ')
    assert 0 < moved_raw and worst < 1e-13          # another summation order: last bits only
    assert moved_rounded <= coefs // 500            # ... which the safe rounding almost always absorbs
