"""The frontier engine against an independent MILP solver (scipy.optimize.milp = HiGHS branch and cut) on
families the oracle-parity tests do not cover by construction -- infinite upper bounds (symbolic values
a + b M in the node LPs), mixed bounds (fixed variables, some infinite) -- with both branching rules, per-node
steps and batched steps with anchors and the plunge: same verdict, same optimum (1e-6 relative)."""
import numpy as np
import pytest
from scipy.optimize import Bounds, LinearConstraint, milp

from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays

pytestmark = pytest.mark.gpu
INF = np.inf
VERDICT = {1: 'optimal', 2: 'infeasible', 3: 'unbounded'}


def instance(n, m, seed, family):
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    rng = np.random.default_rng(77 + seed)
    if family == 'unboxed':
        u = np.full(n, INF)
    elif family == 'mixed':
        l, u = l.copy(), u.copy()
        fixed = rng.random(n) < 0.1
        u[fixed] = l[fixed] = np.floor(rng.uniform(0, 3, fixed.sum()))
        u[(rng.random(n) < 0.3) & ~fixed] = INF
    return A, b, c, l, u, ints


@pytest.mark.parametrize('family', ['boxed', 'unboxed', 'mixed'])
@pytest.mark.parametrize('n,m', [(12, 6), (20, 10), (30, 15), (40, 20), (50, 25)])
def test_optimum_is_highs_optimum(n, m, family, gpu_ctx):
    for seed in range(4):
        A, b, c, l, u, ints = instance(n, m, seed, family)
        integrality = np.zeros(n)
        integrality[ints] = 1
        h = milp(c, constraints=LinearConstraint(A, lb=b, ub=np.inf), bounds=Bounds(l, u), integrality=integrality,
                 options={'mip_rel_gap': 0.0, 'time_limit': 120})
        assert h.status in (0, 2, 3), h.message
        for rule, batch, dive in (('most fractional', 1, 0), ('pseudo cost', 64, 4)):
            p = _ffi.Problem(gpu_ctx, A, b, c)
            t = _ffi.Tree(p, ints, l, u, branch_rule=rule, max_batch=batch, pool_capacity=1 << 20)
            if batch > 1:
                t.set_anchor_mode(True)
                t.set_dive(dive)
            st = t.solve(mip_gap=1e-9, max_seconds=120.0)
            what = f'{n}x{m} seed {seed} {family} {rule} batch {batch}'
            assert VERDICT.get(st['status']) == {0: 'optimal', 2: 'infeasible', 3: 'unbounded'}[h.status], what
            if h.status == 0:
                assert abs(st['primal_bound'] - h.fun) <= 1e-6 * max(1.0, abs(h.fun)), what
                x = t.solution()
                assert np.all(A @ x >= b - 1e-6) and np.all(x >= l - 1e-9) and np.all(x <= u + 1e-9), what
                assert np.all(np.abs(x[ints] - np.round(x[ints])) <= 1e-4), what
            t.close()
            p.close()


@pytest.mark.parametrize('n,m,seeds', [(60, 30, (0, 1, 2)), (80, 40, (0, 1, 2)), (100, 50, (1, 2))])
def test_fast_path_proves_the_optimum_highs_proves(n, m, seeds, gpu_ctx):
    """The configuration the bench measures -- 8 192 nodes per step, device finish, anchors, plunge of depth 8, the
    two phases of bench.two_phase -- on instances that close: the proven optimum is HiGHS's (scripts/
    milp_vs_highs_large.py prints the times: 0.01-1.7 s against 0.4-100 s of HiGHS on one core, up to 120 x 60)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    for seed in seeds:
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
        integrality = np.zeros(n)
        integrality[ints] = 1
        h = milp(c, constraints=LinearConstraint(A, lb=b, ub=np.inf), bounds=Bounds(l, u), integrality=integrality,
                 options={'mip_rel_gap': 0.0, 'time_limit': 120})
        assert h.status == 0, h.message
        out = bench.two_phase(gpu_ctx, A, b, c, l, u, ints, 8, dfs_seconds=0.5, limit=60.0, pool_log2=22, mip_gap=1e-9)
        assert out['status'] == 'optimal', (n, m, seed, out['status'], out['gap'])
        assert abs(out['primal_bound'] - h.fun) <= 1e-6 * max(1.0, abs(h.fun)), (n, m, seed)
