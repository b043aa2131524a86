"""GPU parity of K1 (batched dual simplex) against the CPU oracle, through the C ABI.

Bar: bit-exact status / basis / iteration counts / x / objective (integer and f64 alike: the
kernel follows the oracle's canonical operation order).  Comparisons use == so that -0.0 == 0.0.
"""
import numpy as np
import pytest

from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays

pytestmark = pytest.mark.gpu
INF = np.inf


def assert_same(g, o, what=''):
    assert np.array_equal(g['status'], o['status']), f'{what} status'
    assert np.array_equal(g['iters'], o['iters']), f'{what} iters'
    assert np.array_equal(g['npivots'], o['npivots']), f'{what} npivots'
    assert np.array_equal(g['vstat'], o['vstat']), f'{what} basis'
    ok = (g['status'] != 1)
    fin = (g['status'] == 0) | (g['status'] == 3)
    assert np.array_equal(g['x'][fin], o['x'][fin]), f'{what} x'
    assert np.array_equal(g['obj'][ok], o['obj'][ok]), f'{what} obj'
    assert np.all(np.isposinf(g['obj'][~ok]))
    assert np.array_equal(g['y'][fin], o['y'][fin]), f'{what} duals'


SMALL = {
    'no_branch': (-np.eye(3), [-1, -1, -1], [-1, -1, 0], [0, 0, 0], [INF] * 3),
    'small_branch': ([[-1, 0, -1], [0, -1, 0]], [-1.5, -1.25], [-1, -1, -1], [0, 0, 0], [10] * 3),
    'infeasible': ([[-1, -1, 0]], [1], [-1, -1, 0], [0, 0, 0], [INF] * 3),
    'unbounded': ([[-1, 1], [1, -1]], [-.5, -.5], [-1, -1], [0, 0], [INF] * 2),
    'cut2': ([[-4, -1], [-1, -4], [-1, 1]], [-28, -27, -1], [-2, -5], [0, 0], [INF] * 2),
    'cut3': ([[-3, -4], [-5, -10], [-1, -2]], [-10, -8, -1.2], [-8, -12], [0, 0], [INF] * 2),
}


@pytest.mark.parametrize('name', sorted(SMALL))
def test_small_models_match_oracle(name, gpu_ctx, oracle):
    A, b, c, l, u = [np.asarray(a, float) for a in SMALL[name]]
    p = _ffi.Problem(gpu_ctx, A, b, c)
    g = p.solve_batch(l[None], u[None])
    o = oracle.lp_solve_batch(A, b, c, l[None], u[None])
    assert_same(g, o, name)


def test_small_branch_pinned_root(gpu_ctx):
    # test_base_node.py:406-416 (reference): objective -2.75 at x = [0, 1.25, 1.5]
    A, b, c, l, u = [np.asarray(a, float) for a in SMALL['small_branch']]
    g = _ffi.Problem(gpu_ctx, A, b, c).solve_batch(l[None], u[None])
    assert g['status'][0] == 0 and g['obj'][0] == -2.75
    assert np.array_equal(g['x'][0], [0, 1.25, 1.5])


def _children(A, b, c, l, u, root, k):
    """2k child nodes of a solved root: branch on the k most fractional variables."""
    x = root['x'][0]
    frac = np.minimum(x - np.floor(x), np.ceil(x) - x)
    idx = np.argsort(-frac, kind='stable')[:k]
    ls, us = [], []
    for j in idx:
        if frac[j] <= 1e-4:
            continue
        l2, u2 = l.copy(), u.copy()
        u2[j] = np.floor(x[j])
        ls.append(l2); us.append(u2)
        l2, u2 = l.copy(), u.copy()
        l2[j] = np.ceil(x[j])
        ls.append(l2); us.append(u2)
    L, U = np.array(ls), np.array(us)
    V = np.repeat(root['vstat'], len(L), axis=0)
    return L, U, V


@pytest.mark.parametrize('n,m,seeds', [(64, 32, range(16)), (20, 10, range(8)), (100, 40, range(4))])
def test_cold_roots_batch(n, m, seeds, gpu_ctx, oracle):
    # config C2 shape: independent instances, cold start (each instance is its own problem)
    for seed in seeds:
        A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=seed)
        p = _ffi.Problem(gpu_ctx, A, b, c)
        g = p.solve_batch(l[None], u[None])
        o = oracle.lp_solve_batch(A, b, c, l[None], u[None])
        assert_same(g, o, f'{n}x{m} seed {seed}')
        assert g['status'][0] == 0


def test_s3_root_children_and_probes(gpu_ctx, oracle):
    # config C3 shape: 256 x 128, cold root, warm-started children, 5-iteration probes
    A, b, c, l, u, _ = random_dense_milp_arrays(256, 128, seed=0)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = p.solve_batch(l[None], u[None])
    oroot = oracle.lp_solve_batch(A, b, c, l[None], u[None])
    assert_same(root, oroot, 'root')
    L, U, V = _children(A, b, c, l, u, root, 24)
    g = p.solve_batch(L, U, V)
    o = oracle.lp_solve_batch(A, b, c, L, U, V)
    assert_same(g, o, 'children')
    assert np.all(g['obj'][g['status'] == 0] >= root['obj'][0] - 1e-9)
    g5 = p.solve_batch(L, U, V, max_iter=5)
    o5 = oracle.lp_solve_batch(A, b, c, L, U, V, max_iter=5)
    assert_same(g5, o5, 'probes')
    assert set(np.unique(g5['status'])) <= {0, 1, 3}
    # grandchildren from each child's own basis (deeper refactorisations)
    ok = np.where(g['status'] == 0)[0][:8]
    for k in ok:
        one = {key: val[k:k + 1] for key, val in g.items()}
        L2, U2, V2 = _children(A, b, c, L[k], U[k], one, 4)
        if len(L2) == 0:
            continue
        g2 = p.solve_batch(L2, U2, V2)
        o2 = oracle.lp_solve_batch(A, b, c, L2, U2, V2)
        assert_same(g2, o2, f'grandchildren of {k}')


def test_with_extra_rows_uses_tall_kernel(gpu_ctx, oracle):
    # m > 128 (cut rows appended) dispatches to the tall instantiation
    A, b, c, l, u, _ = random_dense_milp_arrays(200, 150, seed=3)
    assert _ffi.kernel_name(150, 200) == 'lp_dual_simplex<7,7,16>'
    p = _ffi.Problem(gpu_ctx, A, b, c)
    g = p.solve_batch(l[None], u[None])
    o = oracle.lp_solve_batch(A, b, c, l[None], u[None])
    assert_same(g, o, 'tall')


def test_too_big_fails_loudly(gpu_ctx):
    A, b, c, l, u, _ = random_dense_milp_arrays(1100, 20, seed=0)
    with pytest.raises(_ffi.MipxError, match='MIPX_ETOOBIG'):
        _ffi.Problem(gpu_ctx, A, b, c)


@pytest.mark.parametrize('n,m,seed', [(300, 150, 0), (512, 256, 1)])
def test_hbm_streaming_kernel_full_solves(n, m, seed, gpu_ctx, oracle):
    """K1b (tableau streamed from HBM) above the register tiles: cold root to optimality, then
    warm-started children and 5-iteration probes, bit-exact against the oracle."""
    assert _ffi.kernel_name(m, n) == 'lp_dual_simplex_big'
    A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=seed)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = p.solve_batch(l[None], u[None])
    assert_same(root, oracle.lp_solve_batch(A, b, c, l[None], u[None]), 'root')
    assert root['status'][0] == 0
    L, U, V = _children(A, b, c, l, u, root, 6)
    assert_same(p.solve_batch(L, U, V), oracle.lp_solve_batch(A, b, c, L, U, V), 'children')
    assert_same(p.solve_batch(L, U, V, max_iter=5), oracle.lp_solve_batch(A, b, c, L, U, V, max_iter=5), 'probes')


@pytest.mark.parametrize('n,m,seed', [(300, 150, 0), (512, 256, 1), (600, 70, 0), (1024, 512, 0), (1000, 1000, 2)])
def test_one_cold_lp_over_the_chip(n, m, seed, gpu_ctx, oracle, monkeypatch):
    """K1c (lp_kernel_root.hip.h): a single cold LP above the register tiles is spread over up to 256
    workgroups, one launch per pivot.  Same arithmetic and selection rules as K1b: the full solve is the
    oracle's bit for bit, whatever the number of rows each workgroup holds (MIPX_ROOT_WG 32: eight rows,
    64: ..., 256: the default), and it is K1b's (MIPX_NO_COOP_ROOT=1) -- states after k pivots included."""
    A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=seed)
    want = oracle.lp_solve_batch(A, b, c, l[None], u[None])
    p = _ffi.Problem(gpu_ctx, A, b, c)
    for wg in ('256', '128', '64', '32'):
        if (m + int(wg) - 1) // int(wg) > 8:
            continue
        monkeypatch.setenv('MIPX_ROOT_WG', wg)
        q = _ffi.Problem(gpu_ctx, A, b, c)       # (its buffers are sized for the number of workgroups)
        assert_same(q.solve_batch(l[None], u[None]), want, f'cold root, at most {wg} workgroups')
        q.close()
    monkeypatch.delenv('MIPX_ROOT_WG')
    for k in (1, 2, 33):
        g = p.solve_batch(l[None], u[None], max_iter=k)
        assert_same(g, oracle.lp_solve_batch(A, b, c, l[None], u[None], max_iter=k), f'{k} pivots')
        monkeypatch.setenv('MIPX_NO_COOP_ROOT', '1')
        assert_same(p.solve_batch(l[None], u[None], max_iter=k), g, f'{k} pivots, one workgroup')
        monkeypatch.delenv('MIPX_NO_COOP_ROOT')
    p.close()


@pytest.mark.parametrize('n,m', [(257, 100), (300, 64), (520, 260), (700, 300)])
def test_one_cold_lp_over_the_chip_statuses(n, m, gpu_ctx, oracle, monkeypatch):
    """... on instances with fixed variables, infinite bounds and an empty row, and on an infeasible and an
    unbounded one: status, basis, x, duals as the oracle's."""
    A, b, c, l, u = _mixed_instance(n, m, seed=n + m)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    assert_same(p.solve_batch(l[None], u[None]), oracle.lp_solve_batch(A, b, c, l[None], u[None]), 'mixed')
    b2 = b.copy(); A2 = A.copy()
    A2[1] = -1.0; b2[1] = 1.0; l2 = np.zeros(n)                    # -sum x >= 1 with x >= 0: infeasible
    p2 = _ffi.Problem(gpu_ctx, A2, b2, c)
    g = p2.solve_batch(l2[None], u[None])
    assert_same(g, oracle.lp_solve_batch(A2, b2, c, l2[None], u[None]), 'infeasible')
    assert g['status'][0] == 1
    c3 = c.copy(); A3 = A.copy(); u3 = u.copy()
    c3[0] = -1.0; A3[:, 0] = 0.0; u3[0] = INF                      # a free ride down column 0: unbounded
    p3 = _ffi.Problem(gpu_ctx, A3, b, c3)
    g = p3.solve_batch(l[None], u3[None])
    assert_same(g, oracle.lp_solve_batch(A3, b, c3, l[None], u3[None]), 'unbounded')
    assert g['status'][0] in (1, 2)
    for q in (p, p2, p3):
        q.close()


@pytest.mark.parametrize('n,m,seed', [(128, 64, 1), (256, 128, 0), (300, 150, 0), (512, 256, 0), (600, 300, 0),
                                      (1024, 512, 0)])
def test_infinite_upper_bounds_against_highs_and_the_oracle(n, m, seed, gpu_ctx, oracle, monkeypatch):
    """u = +inf everywhere (values a + b M carried through thousands of pivots; see
    tests/test_oracle_known_answers.py::test_infinite_upper_bounds_match_highs): every kernel -- K1, K1c (one
    cold LP), K1b (the same LP twice in a batch, and with MIPX_NO_COOP_ROOT) -- ends optimal on HiGHS's
    objective to 1e-9 and on the oracle's bits."""
    from scipy.optimize import linprog
    A, b, c, l, _, _ = random_dense_milp_arrays(n, m, seed=seed)
    u = np.full(n, INF)
    want = oracle.lp_solve_batch(A, b, c, l[None], u[None])
    h = linprog(c, A_ub=-A, b_ub=-b, bounds=[(lo, None) for lo in l], method='highs')
    p = _ffi.Problem(gpu_ctx, A, b, c)
    g = p.solve_batch(l[None], u[None])
    assert_same(g, want, 'one cold LP')
    assert g['status'][0] == 0 and abs(g['obj'][0] - h.fun) <= 1e-9 * abs(h.fun)
    two = p.solve_batch(np.stack([l, l]), np.stack([u, u]))
    for k in range(2):
        assert_same({q: v[k:k + 1] for q, v in two.items()}, want, f'batch of two, node {k}')
    monkeypatch.setenv('MIPX_NO_COOP_ROOT', '1')
    assert_same(p.solve_batch(l[None], u[None]), want, 'one workgroup')
    p.close()


def test_s5_shape_1024x512(gpu_ctx, oracle):
    """BASELINE config C5 shape: 1024 vars x 512 rows.  The cold root needs tens of thousands of
    pivots, so parity is checked on truncated solves (every state after k pivots must agree) and on
    a warm start from the truncated basis."""
    A, b, c, l, u, _ = random_dense_milp_arrays(1024, 512, seed=0)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    for k in (1, 7, 200):
        assert_same(p.solve_batch(l[None], u[None], max_iter=k),
                    oracle.lp_solve_batch(A, b, c, l[None], u[None], max_iter=k), f'{k} pivots')
    g = p.solve_batch(l[None], u[None], max_iter=200)
    assert g['status'][0] == 3 and g['iters'][0] == 200
    warm = p.solve_batch(l[None], u[None], g['vstat'], max_iter=50)   # refactor + 50 more pivots
    assert_same(warm, oracle.lp_solve_batch(A, b, c, l[None], u[None], g['vstat'], max_iter=50), 'warm')
    assert warm['npivots'][0] > 50 and warm['obj'][0] >= g['obj'][0] - 1e-6


def test_warm_start_with_padding_after_lds_pollution(gpu_ctx, oracle):
    """Regression: uninitialised LDS in the padded part of the pivot vectors once leaked NaN into
    the refactored tableau of LPs much smaller than the tile (n=2, m=5 in a 64x32 tile)."""
    A, b, c, l, u, _ = random_dense_milp_arrays(256, 128, seed=1)
    _ffi.Problem(gpu_ctx, A, b, c).solve_batch(l[None], u[None])  # leaves junk in LDS
    A = np.array([[-4., -1.], [-1., -4.], [-1., 1.], [-0.36363636363636365, -1.], [-1., -2 / 3]])
    b = np.array([-28., -27., -1., -7.090909090909091, -9.])
    c = np.array([-2., -5.])
    l, u = np.zeros((1, 2)), np.full((1, 2), INF)
    V = np.array([[1, 1, 3, 3, 1, 1, 1]], np.int8)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    for _ in range(3):
        assert_same(p.solve_batch(l, u, V), oracle.lp_solve_batch(A, b, c, l, u, V), 'padded warm')
    for n, m, seed in [(5, 3, 0), (9, 7, 1), (33, 17, 2), (70, 20, 3), (130, 66, 4)]:
        A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=seed)
        p = _ffi.Problem(gpu_ctx, A, b, c)
        root = p.solve_batch(l[None], u[None])
        L, U, V = _children(A, b, c, l, u, root, 6)
        if len(L):
            assert_same(p.solve_batch(L, U, V), oracle.lp_solve_batch(A, b, c, L, U, V), f'{n}x{m}')


def test_c2_batch_of_1024_independent_roots(gpu_ctx, oracle):
    """BASELINE config C2: 1024 random dense MILPs, 64 vars x 32 rows, seeds 0..1023, root
    relaxation of each (cold start), one launch."""
    probs = [random_dense_milp_arrays(64, 32, seed=s) for s in range(1024)]
    A = np.stack([p[0] for p in probs]); b = np.stack([p[1] for p in probs])
    c = np.stack([p[2] for p in probs]); l = np.stack([p[3] for p in probs])
    u = np.stack([p[4] for p in probs])
    g = _ffi.solve_multi(gpu_ctx, A, b, c, l, u)
    assert np.all(g['status'] == 0)
    for k in range(0, 1024, 8):  # the oracle on every 8th instance: bit-exact
        o = oracle.lp_solve_batch(A[k], b[k], c[k], l[k][None], u[k][None])
        for key in ('status', 'iters', 'npivots', 'vstat', 'x', 'obj'):
            assert np.array_equal(g[key][k:k + 1], o[key]), (k, key)
    # every instance: feasibility and the size-independent LP bound property
    for k in range(1024):
        assert np.all(A[k] @ g['x'][k] >= b[k] - 1e-6)
        assert np.all(g['x'][k] >= -1e-9) and np.all(g['x'][k] <= 10 + 1e-9)
        assert abs(g['obj'][k] - c[k] @ g['x'][k]) <= 1e-9 * max(1, abs(g['obj'][k]))


@pytest.mark.parametrize('n,m,kernel', [(256, 128, 'lp_dual_simplex<7,5,16>'), (300, 200, 'lp_dual_simplex_big')])
def test_anchored_refactorisation(n, m, kernel, gpu_ctx, oracle):
    """Warm starts that refactor from the root's tableau instead of the slack basis: bit-exact
    against the oracle doing the same, far fewer pivots, same optima within rounding (register
    kernel and the HBM-streaming one)."""
    A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=0)
    assert _ffi.kernel_name(m, n) == kernel
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = p.solve_batch(l[None], u[None])
    L, U, V = _children(A, b, c, l, u, root, 16)
    plain = p.solve_batch(L, U, V)
    ok = np.where(plain['status'] == 0)[0][:6]
    L2, U2, V2 = [], [], []
    for k in ok:  # grandchildren: their bases are a few pivots away from the root's
        one = {key: val[k:k + 1] for key, val in plain.items()}
        a_, b_, c_ = _children(A, b, c, L[k], U[k], one, 3)
        L2.append(a_); U2.append(b_); V2.append(c_)
    L2, U2, V2 = np.concatenate(L2), np.concatenate(U2), np.concatenate(V2)
    plain2 = p.solve_batch(L2, U2, V2)
    p.set_anchor(root['vstat'][0])
    anch = oracle.make_anchor(A, b, c, root['vstat'][0])
    for (Lx, Ux, Vx, ref) in ((L, U, V, plain), (L2, U2, V2, plain2)):
        g = p.solve_batch(Lx, Ux, Vx)
        with oracle.anchored(anch):
            o = oracle.lp_solve_batch(A, b, c, Lx, Ux, Vx)
        assert_same(g, o, 'anchored')
        assert np.array_equal(g['status'], ref['status'])
        fin = g['status'] == 0
        assert np.allclose(g['obj'][fin], ref['obj'][fin], rtol=0, atol=1e-7)
        assert g['npivots'].sum() < 0.6 * ref['npivots'].sum()
    # children of the root start AT the anchor basis: no refactorisation pivots at all
    g = p.solve_batch(L, U, V)
    assert np.array_equal(g['npivots'], g['iters'])
    p.set_anchor(None)
    assert_same(p.solve_batch(L, U, V), plain, 'anchor off')


def _mixed_instance(n, m, seed):
    """A random dense instance with the awkward ingredients mixed in: some fixed variables
    (l == u), some infinite upper bounds, a few zero columns/rows."""
    rng = np.random.default_rng(1000 + seed)
    if m > 0:
        A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=seed)
    else:
        A, b = np.zeros((0, n)), np.zeros(0)
        c = -rng.integers(1, 10, n).astype(float)
        l, u = np.zeros(n), np.full(n, 10.0)
    l, u = l.copy(), u.copy()
    fixed = rng.random(n) < 0.1
    u[fixed] = l[fixed] = np.floor(rng.uniform(0, 3, fixed.sum()))
    u[(rng.random(n) < 0.15) & ~fixed] = INF
    if m > 2:
        A = A.copy()
        A[rng.integers(0, m)] *= 0.0      # an empty row (b <= 0 keeps it feasible or not: either is fine)
    return A, b, c, l, u


@pytest.mark.parametrize('n,m', [(1, 1), (2, 1), (5, 0), (3, 40), (63, 31), (64, 32), (65, 33), (40, 33), (128, 64),
                                 (129, 64), (100, 65), (255, 127), (256, 128), (256, 129), (200, 192), (256, 193),
                                 (257, 100)])
def test_tile_boundaries_and_edge_cases(n, m, gpu_ctx, oracle):
    """Shapes on both sides of every tile limit (and the hand-over to the HBM-streaming kernel),
    with fixed variables, infinite bounds and an empty row: cold roots, warm-started children,
    truncated probes -- all bit-exact against the oracle."""
    L, U, V = [], [], []
    A, b, c, l, u = _mixed_instance(n, m, seed=n + m)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = p.solve_batch(l[None], u[None])
    o = oracle.lp_solve_batch(A, b, c, l[None], u[None])
    assert_same(root, o, f'root {n}x{m} [{_ffi.kernel_name(m, n)}]')
    if root['status'][0] not in (0, 2):
        return
    x = np.minimum(root['x'][0], 1e6)
    frac = np.minimum(x - np.floor(x), np.ceil(x) - x)
    for j in np.argsort(-frac, kind='stable')[:4]:
        for side in (0, 1):
            l2, u2 = l.copy(), u.copy()
            if side == 0:
                u2[j] = np.floor(x[j])
            else:
                l2[j] = np.ceil(x[j]) if frac[j] > 1e-9 else x[j] + 1
            L.append(l2); U.append(u2); V.append(root['vstat'][0])
    L, U, V = np.array(L), np.array(U), np.array(V)
    for max_iter in (0, 3):
        g = p.solve_batch(L, U, V, max_iter)
        o = oracle.lp_solve_batch(A, b, c, L, U, V, max_iter)
        assert_same(g, o, f'children {n}x{m} max_iter={max_iter}')
    # a second generation from the first child that solved
    ok = np.where(g['status'] == 0)[0]
    if len(ok):
        k = ok[0]
        g0 = p.solve_batch(L[k:k + 1], U[k:k + 1], V[k:k + 1])
        l3 = L[k].copy(); l3[np.argmax(frac)] = L[k][np.argmax(frac)]
        gg = p.solve_batch(L, U, np.repeat(g0['vstat'], len(L), 0))
        oo = oracle.lp_solve_batch(A, b, c, L, U, np.repeat(g0['vstat'], len(L), 0))
        assert_same(gg, oo, f'cousins {n}x{m}')


def _assert_same_dive(g, o, what=''):
    for key in ('status', 'iters', 'npivots', 'dive_var', 'dive_dir'):
        assert np.array_equal(g[key], o[key]), f'{what} {key}'
    assert np.array_equal(g['dive_val'][g['dive_var'] >= 0], o['dive_val'][o['dive_var'] >= 0]), f'{what} dive_val'
    done = g['status'] >= 0
    assert np.array_equal(g['vstat'][done], o['vstat'][done]), f'{what} basis'
    fin = (g['status'] == 0) | (g['status'] == 3)
    assert np.array_equal(g['x'][fin], o['x'][fin]), f'{what} x'
    ok = done & (g['status'] != 1)
    assert np.array_equal(g['obj'][ok], o['obj'][ok]), f'{what} obj'
    assert np.all(np.isposinf(g['obj'][g['status'] == 1]))


@pytest.mark.parametrize('n,m,seed', [(24, 10, 1), (64, 32, 0), (100, 40, 2), (256, 128, 0), (300, 150, 1)])
@pytest.mark.parametrize('rule', [0, 1])
def test_in_place_dive_matches_oracle(n, m, seed, rule, gpu_ctx, oracle):
    """The dive (child LP continued on the register tableau after a bound of the branching
    variable moved) is bit-identical to the oracle's restatement of it: same branching decision,
    same pivots, same child solution; and the child is the LP an ordinary warm-started solve of
    the moved bounds reaches (same optimum within rounding)."""
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = p.solve_batch(l[None], u[None])
    L, U, V = _children(A, b, c, l, u, root, 8)
    rng = np.random.default_rng(seed)
    cost_l, cost_r = rng.uniform(0.5, 4.0, n), rng.uniform(0.5, 4.0, n)
    has = np.ones(n, np.uint8)
    for anchored in (False, True):
        if anchored:
            p.set_anchor(root['vstat'][0])
            cm = oracle.anchored(oracle.make_anchor(A, b, c, root['vstat'][0]))
        else:
            import contextlib
            cm = contextlib.nullcontext()
        g = p.dive_batch(L, U, V, rule, ints, cost_l, cost_r, has)
        with cm:
            o = oracle.lp_solve_dive_batch(A, b, c, L, U, V, rule, ints, cost_l, cost_r, has, np.inf)
        _assert_same_dive(g, o, f'dive rule {rule} anchored {anchored}')
        B = len(L)
        dived = np.where(g['dive_var'] >= 0)[0]
        assert len(dived) >= 1 and np.all(g['status'][B:][g['dive_var'] < 0] == -1)
        # the parents are what the plain kernel computes
        plain = p.solve_batch(L, U, V)
        for key in ('status', 'obj', 'x', 'vstat', 'iters', 'npivots'):
            assert np.array_equal(g[key][:B], plain[key]), key
        # each child = a warm-started solve of the parent's bounds with the dive's bound moved
        L2, U2 = L[dived].copy(), U[dived].copy()
        for r, k in enumerate(dived):
            v = g['dive_var'][k]
            assert v in ints and abs(g['dive_val'][k] - plain['x'][k][v]) == 0
            if g['dive_dir'][k] == 0:
                U2[r, v] = np.floor(g['dive_val'][k])
            else:
                L2[r, v] = np.ceil(g['dive_val'][k])
        again = p.solve_batch(L2, U2, plain['vstat'][dived])
        assert np.array_equal(again['status'], g['status'][B:][dived])
        fin = again['status'] == 0
        assert np.allclose(again['obj'][fin], g['obj'][B:][dived][fin], rtol=0, atol=1e-7)
        # and it costs its simplex iterations only: no refactorisation pivots
        assert np.array_equal(g['npivots'][B:][dived], g['iters'][B:][dived])
    p.set_anchor(None)
    # a cutoff below every objective, or a missing pseudo-cost entry, switches the dive off
    g = p.dive_batch(L, U, V, rule, ints, cost_l, cost_r, has, cutoff=-1e30)
    assert np.all(g['dive_var'] == -1) and np.all(g['status'][len(L):] == -1)
    if rule == 1:
        g = p.dive_batch(L, U, V, 1, ints, cost_l, cost_r, np.zeros(n, np.uint8))
        assert np.all(g['dive_var'] == -1)


@pytest.mark.parametrize('n,m,seed', [(24, 10, 1), (64, 32, 0), (256, 128, 0), (300, 150, 1)])
@pytest.mark.parametrize('rule,depth', [(0, 2), (1, 4), (0, 8)])
def test_plunge_matches_oracle(n, m, seed, rule, depth, gpu_ctx, oracle):
    """mipx_lp_plunge_batch: up to `depth` dive children in a row on one tableau (register tiles and
    the HBM-streaming kernel), bit-identical to the oracle's restatement level by level; a level is
    the LP an ordinary warm-started solve reaches from the level before with the plunge's bound moved."""
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = p.solve_batch(l[None], u[None])
    L, U, V = _children(A, b, c, l, u, root, 8)
    B = len(L)
    rng = np.random.default_rng(seed)
    cost_l, cost_r = rng.uniform(0.5, 4.0, n), rng.uniform(0.5, 4.0, n)
    has = np.ones(n, np.uint8)
    g = p.dive_batch(L, U, V, rule, ints, cost_l, cost_r, has, depth=depth)
    o = oracle.lp_solve_dive_batch(A, b, c, L, U, V, rule, ints, cost_l, cost_r, has, np.inf, depth=depth)
    _assert_same_dive(g, o, f'plunge rule {rule} depth {depth}')
    assert len(g['status']) == (depth + 1) * B and len(g['dive_var']) == depth * B
    # depth 1 of the same call is the one-level dive
    g1 = p.dive_batch(L, U, V, rule, ints, cost_l, cost_r, has)
    for key in ('status', 'obj', 'iters', 'npivots'):
        assert np.array_equal(g[key][:2 * B], g1[key]), key
    assert np.array_equal(g['dive_var'][:B], g1['dive_var'])
    # a level exists exactly where the level before decided to dive; chains really go deep
    reached = np.zeros(depth + 1, int)
    for lvl in range(depth):
        went = g['dive_var'][lvl * B:(lvl + 1) * B] >= 0
        assert np.array_equal(g['status'][(lvl + 1) * B:(lvl + 2) * B] >= 0, went)
        if lvl > 0:   # no decision without a solved level
            assert not np.any(went & (g['status'][lvl * B:(lvl + 1) * B] != 0))
        reached[lvl + 1] = went.sum()
    assert reached[min(depth, 2)] >= 1, reached
    # level p + 1 = a warm-started solve from level p's basis with the accumulated bounds
    Lc, Uc = L.copy(), U.copy()
    for lvl in range(depth):
        sel = np.where(g['dive_var'][lvl * B:(lvl + 1) * B] >= 0)[0]
        if len(sel) == 0:
            break
        for k in sel:
            v = g['dive_var'][lvl * B + k]
            val = g['dive_val'][lvl * B + k]
            assert val == g['x'][lvl * B + k][v]
            if g['dive_dir'][lvl * B + k] == 0:
                Uc[k, v] = np.floor(val)
            else:
                Lc[k, v] = np.ceil(val)
        again = p.solve_batch(Lc[sel], Uc[sel], g['vstat'][lvl * B + sel])
        child = (lvl + 1) * B + sel
        assert np.array_equal(again['status'], g['status'][child])
        fin = again['status'] == 0
        assert np.allclose(again['obj'][fin], g['obj'][child][fin], rtol=0, atol=1e-6)
        assert np.array_equal(g['npivots'][child], g['iters'][child])   # no refactorisation below the node
