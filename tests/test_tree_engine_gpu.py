"""Native frontier engine (mipx_tree_*) against the Python driver that mirrors the reference's
BranchAndBound.solve loop.  frontier_batch=1 must reproduce it node for node: evaluation order,
LP status, branching variable, objective, node ids, pseudo-cost table, incumbent."""
import json
from math import isclose
import os

import numpy as np
import pytest

from simple_mip_solver_amd import (BaseNode, BranchAndBound, DepthFirstSearchNode, MILPInstance,
                                   PseudoCostBranchNode, PseudoCostBranchDepthFirstSearchNode)
from simple_mip_solver_amd import lp as lpmod
from simple_mip_solver_amd.generators import random_dense_milp_arrays
from tests.support.example_models import model, std_model

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
TABLE = json.load(open(os.path.join(HERE, 'golden', 'example_models_optima.json')))['models']
NODES = [BaseNode, PseudoCostBranchNode, DepthFirstSearchNode, PseudoCostBranchDepthFirstSearchNode]


@pytest.fixture(autouse=True)
def hip_backend():
    from tests.support.compare_backend import CompareBackend
    lpmod.set_backend(CompareBackend())
    yield
    lpmod.set_backend(None)


def python_run(make_model, Node, **kw):
    """The per-node Python loop, with a trace of every evaluated node."""
    bb = BranchAndBound(make_model(), Node, pseudo_costs={}, gomory_cuts=False, **kw)
    trace = []
    inner = bb._evaluate_node

    def spy(node):
        before = bb.evaluated_nodes
        inner(node)
        if bb.evaluated_nodes > before:
            kids = bb.tree.get_children(node.idx)
            bvar = bb.tree.get_node_instances(kids[0])._b_idx if kids else -1
            trace.append((node.idx, node.lp.getStatusCode(), bvar,
                          node.lp.objectiveValue))
    bb._evaluate_node = spy
    bb.solve()
    return bb, trace


def assert_same_search(make_model, Node, **kw):
    py, ptrace = python_run(make_model, Node, **kw)
    nb = BranchAndBound(make_model(), Node, pseudo_costs={}, gomory_cuts=False, frontier_batch=1, **kw)
    # the engine is created on the first solve(): make it come up with tracing switched on
    from simple_mip_solver_amd import _ffi
    real_tree = _ffi.Tree

    class TracedTree(real_tree):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.set_trace(True)
    _ffi.Tree = TracedTree
    try:
        nb.solve()
    finally:
        _ffi.Tree = real_tree
    assert nb.status == py.status
    assert nb.evaluated_nodes == py.evaluated_nodes
    assert nb.objective_value == py.objective_value
    assert nb._kwargs['next_node_idx'] == py._kwargs['next_node_idx']
    if py.solution is None:
        assert nb.solution is None
    else:
        assert np.array_equal(nb.solution, py.solution)
    tr = nb._native.trace()
    assert len(tr['node_id']) == len(ptrace)
    for k, (idx, st, bvar, obj) in enumerate(ptrace):
        assert tr['node_id'][k] == idx and tr['status'][k] == st, (k, ptrace[k])
        assert tr['branch_var'][k] == bvar, (k, ptrace[k], tr['branch_var'][k])
        if st in (0, 3):
            assert tr['objective'][k] == obj
    if issubclass(Node, PseudoCostBranchNode):
        assert nb._kwargs['pseudo_costs'] == py._kwargs['pseudo_costs']
    assert nb.dual_bound == py.dual_bound
    assert nb.current_gap == py.current_gap
    return nb, py


@pytest.mark.parametrize('Node', NODES)
def test_small_branch_exact_mode(Node):
    nb, py = assert_same_search(lambda: std_model('small_branch'), Node)
    assert nb.status == 'optimal' and nb.objective_value == -2
    if Node is BaseNode:
        assert nb.evaluated_nodes == 13 and nb._kwargs['next_node_idx'] == 13


@pytest.mark.parametrize('Node', [BaseNode, PseudoCostBranchNode])
@pytest.mark.parametrize('name', ['no_branch', 'infeasible2', 'unbounded', 'small_branch_max', 'cut2'])
def test_inline_models_exact_mode(Node, name):
    assert_same_search(lambda: model(name), Node)


@pytest.mark.parametrize('Node', NODES)
def test_example_models_exact_mode(Node):
    for f, rec in sorted(TABLE.items()):
        path = os.path.join(HERE, 'golden', 'example_models', f)
        nb, py = assert_same_search(lambda: MILPInstance(file_name=path), Node)
        assert nb.status == 'optimal' and isclose(nb.objective_value, rec['milp_opt'], abs_tol=1e-6), f


def random_model(n, m, seed, density=1.0):
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
    return MILPInstance(A=A, b=b, c=c, l=l, u=u, sense=['Min', '>='], integerIndices=ints, numVars=n)


@pytest.mark.parametrize('Node', [BaseNode, PseudoCostBranchNode, PseudoCostBranchDepthFirstSearchNode])
@pytest.mark.parametrize('n,m,seed', [(12, 6, 0), (20, 10, 1), (30, 15, 2)])
def test_random_models_exact_mode_with_node_limit(Node, n, m, seed):
    assert_same_search(lambda: random_model(n, m, seed), Node, node_limit=120)


def test_reentrant_solve_and_limits():
    bb = BranchAndBound(std_model('small_branch'), BaseNode, gomory_cuts=False, frontier_batch=1,
                        node_limit=1)
    bb.solve()
    assert bb.status == 'stopped on iterations or time' and bb.evaluated_nodes == 1
    assert bb.current_gap is None and bb.dual_bound == -2.75
    bb.node_limit = 10
    bb.solve()
    assert bb.evaluated_nodes == 10 and bb.current_gap == .125   # test_branch_and_bound.py:267-276
    bb.node_limit = float('inf')
    bb.solve()
    assert bb.status == 'optimal' and bb.current_gap == 0 and bb.evaluated_nodes == 13


def test_frontier_batches_reach_the_same_optimum():
    for seed in range(3):
        make = lambda: random_model(24, 10, 10 + seed)
        ref = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False,
                             frontier_batch=1)
        ref.solve()
        assert ref.status == 'optimal'
        for Node in (BaseNode, PseudoCostBranchNode):
            for batch in (4, 64, 1024):
                bb = BranchAndBound(make(), Node, pseudo_costs={}, gomory_cuts=False,
                                    frontier_batch=batch, pool_capacity=1 << 15)
                bb.solve()
                assert bb.status == 'optimal' and isclose(bb.objective_value, ref.objective_value,
                                                          abs_tol=1e-6)
                x = bb.solution
                assert np.max(np.abs(x - np.round(x))) <= 1e-4


def test_native_mode_argument_checks():
    m = std_model('small_branch')
    with pytest.raises(AssertionError, match='frontier_batch must be a positive integer'):
        BranchAndBound(m, gomory_cuts=False, frontier_batch=0)
    with pytest.raises(AssertionError, match='gomory_cuts is boolean'):
        BranchAndBound(m, frontier_batch=4, gomory_cuts=1)

    class Mine(BaseNode):
        pass
    with pytest.raises(AssertionError, match='only available for the stock node classes'):
        BranchAndBound(m, Node=Mine, gomory_cuts=False, frontier_batch=4)


def test_keep_shard_partitions_the_open_nodes(gpu_ctx=None):
    """Multi-GPU sharding, simulated in one process: two engines run the same deterministic
    ramp-up, each keeps its share; the shares are disjoint, cover the frontier, and the MIN of the
    shard dual bounds is the dual bound of the whole tree."""
    from simple_mip_solver_amd import _ffi
    ctx = _ffi.default_context()
    A, b, c, l, u, ints = random_dense_milp_arrays(40, 16, seed=3)
    prob = _ffi.Problem(ctx, A, b, c)

    def ramp():
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=8, pool_capacity=4096)
        st = t.stats()
        while st['open_nodes'] < 24:
            st = t.solve(mip_gap=0.0, frontier_batch=8, max_steps=1)
            assert st['status'] == 4
        return t, st
    whole, st_all = ramp()
    L_all, U_all, V_all, db_all = whole.peek_open(10 ** 6)
    shards = []
    for rank in range(2):
        t, st = ramp()
        assert st['open_nodes'] == st_all['open_nodes'] and st['dual_bound'] == st_all['dual_bound']
        t.keep_shard(rank, 2)
        shards.append(t)
    parts = [s.peek_open(10 ** 6) for s in shards]
    assert len(parts[0][0]) + len(parts[1][0]) == len(L_all)
    assert abs(len(parts[0][0]) - len(parts[1][0])) <= 1
    keys = lambda P: sorted(map(lambda k: (P[0][k].tobytes(), P[1][k].tobytes()), range(len(P[0]))))
    assert sorted(keys(parts[0]) + keys(parts[1])) == keys((L_all, U_all))
    assert min(s.stats()['dual_bound'] for s in shards) == st_all['dual_bound']
    # every shard can keep searching on its own and never beats the whole tree's bound
    for s in shards:
        st = s.solve(mip_gap=0.0, frontier_batch=8, max_steps=3)
        assert st['dual_bound'] >= st_all['dual_bound'] - 1e-9


def test_step_hook_runs_inside_the_step_loop():
    """mipx_tree_set_step_hook: called every k steps of one solve() with the pipeline running; it
    may read the tree and install an incumbent bound / pseudo-cost table; a truthy return or an
    exception stops the solve."""
    from simple_mip_solver_amd import _ffi
    ctx = _ffi.default_context()
    A, b, c, l, u, ints = random_dense_milp_arrays(40, 16, seed=3)
    prob = _ffi.Problem(ctx, A, b, c)
    t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=8, pool_capacity=1 << 14)
    seen = []

    def hook():
        st = t.stats()
        seen.append(st['steps'])
        t.set_pseudo_cost_arrays(*t.pseudo_cost_arrays())   # what a rank does with the merged table
        return False
    t.set_step_hook(hook, 3)
    st = t.solve(mip_gap=0.0, frontier_batch=8, max_steps=10)
    assert st['status'] == 4 and st['steps'] == 10
    assert len(seen) == 3 and seen == sorted(seen)          # after 3, 6 and 9 launched steps

    # the hook's search goes the same way as a search without one (the table it installs is its own)
    t2 = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=8, pool_capacity=1 << 14)
    st2 = t2.solve(mip_gap=0.0, frontier_batch=8, max_steps=10)
    assert st2['dual_bound'] == st['dual_bound'] and st2['evaluated_nodes'] == st['evaluated_nodes']

    # an incumbent bound from another rank prunes here: everything is worse than -inf
    t.set_step_hook(lambda: t.set_primal_bound(-1e30), 1)
    st = t.solve(mip_gap=0.0, frontier_batch=8, max_steps=50)
    assert st['open_nodes'] == 0

    # stopping and failing hooks
    t3 = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=8, pool_capacity=1 << 14)
    t3.set_step_hook(lambda: True, 2)
    with pytest.raises(_ffi.MipxError, match='MIPX_EHOOK'):
        t3.solve(mip_gap=0.0, frontier_batch=8, max_steps=10)
    assert 2 <= t3.stats()['steps'] <= 3

    def boom():
        raise RuntimeError('exchange failed')
    t3.set_step_hook(boom, 1)
    with pytest.raises(RuntimeError, match='exchange failed'):
        t3.solve(mip_gap=0.0, frontier_batch=8, max_steps=10)
    t3.set_step_hook(None)
    assert t3.solve(mip_gap=0.0, frontier_batch=8, max_steps=2)['status'] == 4
    with pytest.raises(_ffi.MipxError, match='MIPX_EINVAL'):
        t3.set_step_hook(lambda: False, 0)


@pytest.mark.parametrize('rule,depth', [('most fractional', 3), ('pseudo cost', 4), ('pseudo cost', 8)])
def test_plunge_reaches_the_same_optimum(rule, depth):
    """mipx_tree_set_dive(depth > 1): up to `depth` dive children in a row per node.  Another node
    order, the same optimum; every dive child is a real evaluated node (one LP each); the Python
    driver passes the depth on."""
    from simple_mip_solver_amd import _ffi
    ctx = _ffi.default_context()
    for n, m, seed in ((24, 10, 11), (30, 12, 3), (40, 16, 3), (48, 20, 0)):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
        prob = _ffi.Problem(ctx, A, b, c)
        out = {}
        for d in (0, 1, depth):
            t = _ffi.Tree(prob, ints, l, u, branch_rule=rule, max_batch=64, pool_capacity=1 << 17)
            t.set_anchor_mode(True)
            t.set_dive(d)
            st = t.solve(mip_gap=0.0, frontier_batch=64, node_limit=200000)
            out[d] = (st, t.solution() if st['has_solution'] else None)
            t.close()
        (s0, x0), (s1, x1), (sd, xd) = out[0], out[1], out[depth]
        assert s0['status'] == s1['status'] == sd['status'] == 1, (s0, s1, sd)
        assert isclose(s0['primal_bound'], sd['primal_bound'], abs_tol=1e-6)
        assert sd['lp_solved'] == sd['evaluated_nodes'] and 0 < sd['dives'] < sd['evaluated_nodes']
        assert sd['steps'] <= s1['steps'] + 2                                # no more steps than the one-level dive
        assert np.max(np.abs(xd[ints] - np.round(xd[ints]))) <= 1e-4
        assert np.all(A @ xd >= b - 1e-6) and isclose(float(c @ xd), sd['primal_bound'], abs_tol=1e-6)
        prob.close()
    with pytest.raises(_ffi.MipxError, match='MIPX_EINVAL'):
        A, b, c, l, u, ints = random_dense_milp_arrays(24, 10, seed=10)
        _ffi.Tree(_ffi.Problem(ctx, A, b, c), ints, l, u, max_batch=8).set_dive(9)
    ref = BranchAndBound(random_model(30, 12, 3), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=1)
    ref.solve()
    bb = BranchAndBound(random_model(30, 12, 3), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=64,
                        pool_capacity=1 << 15, dive=depth)
    bb.solve()
    assert bb.status == 'optimal' and isclose(bb.objective_value, ref.objective_value, abs_tol=1e-6)
    assert bb._native_stats['dives'] > 0


@pytest.mark.parametrize('rule', ['most fractional', 'pseudo cost'])
def test_dive_reaches_the_same_optimum(rule):
    """mipx_tree_set_dive: the workgroup that solved a node also solves one child on the tableau it
    holds.  A different node order, the same optimum; every dive child is a real evaluated node."""
    from simple_mip_solver_amd import _ffi
    ctx = _ffi.default_context()
    for n, m, seed in ((20, 8, 10), (24, 10, 11), (30, 12, 3), (40, 16, 3), (48, 20, 0)):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
        prob = _ffi.Problem(ctx, A, b, c)
        out = {}
        for dive in (False, True):
            t = _ffi.Tree(prob, ints, l, u, branch_rule=rule, max_batch=64, pool_capacity=1 << 17)
            t.set_anchor_mode(True)
            t.set_dive(dive)
            st = t.solve(mip_gap=0.0, frontier_batch=64, node_limit=100000)
            out[dive] = (st, t.solution() if st['has_solution'] else None)
            t.close()
        (s0, x0), (s1, x1) = out[False], out[True]
        assert s0['status'] == s1['status'] == 1, (s0, s1)
        assert isclose(s0['primal_bound'], s1['primal_bound'], abs_tol=1e-6)
        assert s0['dives'] == 0 and s1['dives'] > 0
        assert s1['lp_solved'] == s1['evaluated_nodes'] and s1['dives'] < s1['evaluated_nodes']
        assert np.max(np.abs(x1[ints] - np.round(x1[ints]))) <= 1e-4
        assert np.all(A @ x1 >= b - 1e-6) and isclose(float(c @ x1), s1['primal_bound'], abs_tol=1e-6)
        prob.close()
    # the exact mode reproduces the reference's node order: no dive there
    A, b, c, l, u, ints = random_dense_milp_arrays(24, 10, seed=10)
    prob = _ffi.Problem(ctx, A, b, c)
    t = _ffi.Tree(prob, ints, l, u, max_batch=1)
    with pytest.raises(_ffi.MipxError, match='MIPX_EINVAL'):
        t.set_dive(True)


def test_full_node_pool_stops_the_search_cleanly():
    """A pool too small for the tree: the engine shrinks its batches to what fits and, when not even
    one node's children fit, stops with status 4 and valid bounds instead of failing."""
    from simple_mip_solver_amd import _ffi
    ctx = _ffi.default_context()
    A, b, c, l, u, ints = random_dense_milp_arrays(40, 16, seed=3)   # needs about 4000 nodes
    prob = _ffi.Problem(ctx, A, b, c)
    big = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=64, pool_capacity=1 << 16)
    ref = big.solve(mip_gap=0.0, frontier_batch=64)
    assert ref['status'] == 1 and not ref['pool_exhausted']
    for dive in (False, True):
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=64, pool_capacity=256)
        t.set_dive(dive)
        st = t.solve(mip_gap=0.0, frontier_batch=64)
        assert st['status'] == 4 and st['pool_exhausted'] == 1
        assert st['open_nodes'] > 0 and st['dual_bound'] <= ref['primal_bound'] + 1e-9
        assert st['primal_bound'] >= ref['primal_bound'] - 1e-9        # any incumbent found is feasible
        assert t.solve(mip_gap=0.0, frontier_batch=64)['status'] == 4   # and it stays stopped
    # the Python driver passes it on as a warning
    bb = BranchAndBound(random_model(40, 16, 3), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False,
                        frontier_batch=64, pool_capacity=256)
    with pytest.warns(RuntimeWarning, match='node pool is full'):
        bb.solve()
    assert bb.status == 'stopped on iterations or time'


@pytest.mark.parametrize('depth', [1, 4])
@pytest.mark.parametrize('n,m,MB,K,target', [(64, 32, 1024, 200, 300), (300, 150, 256, 40, 60),
                                             (256, 128, 1024, 300, 600),    # (bench.py's tile + anchor table + slots)
                                             (1024, 512, 64, 16, 32)])      # (C5 at size: plunge + re-anchor on the HBM-streaming kernel)
def test_reanchored_step_matches_the_oracle(n, m, MB, K, target, depth, oracle):
    """mipx_tree_reanchor: open nodes get the tableau of their warm-start basis as their own anchor.
    The table is what the oracle builds for the same bases (refactor-only from the root's tableau), a
    warm start from one's own anchor needs no refactorisation pivot, and one engine step over all
    open nodes does exactly the pivots the oracle does on them with the same table."""
    from simple_mip_solver_amd import _ffi
    ctx = _ffi.default_context()
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
    prob = _ffi.Problem(ctx, A, b, c)

    def ramp():
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=MB, pool_capacity=1 << 15)
        t.set_anchor_mode(True)
        t.set_dive(depth)
        st = t.stats()
        while st['open_nodes'] < target:
            st = t.solve(mip_gap=0.0, frontier_batch=32, max_steps=1)
        return t, st
    t, st0 = ramp()
    N = st0['open_nodes']
    assert N < MB and st0['primal_bound'] == float('inf')
    with pytest.raises(_ffi.MipxError, match='MIPX_EINVAL'):
        t.reanchor(0)
    assert t.anchor_table() is None and np.all(t.peek_anchors(10) == -1)
    t.reanchor(K)                           # the first K open nodes; the others keep the root's anchor
    L, U, V, _ = t.peek_open(10 ** 6)
    sel = t.peek_anchors(10 ** 6)
    T, vec, idx = t.anchor_table()
    assert len(L) == N and len(T) == K and list(sel[:K]) == list(range(K)) and np.all(sel[K:] == -1)
    root = oracle.lp_solve(A, b, c, l, u)
    root_anchor = oracle.make_anchor(A, b, c, root['vstat'])
    with oracle.anchored(root_anchor):
        for k in (0, 1, K // 2, K - 1):     # a table entry is the oracle's tableau of that basis, bit for bit
            mine = oracle.make_anchor(A, b, c, V[k])
            assert np.array_equal(mine['T'], T[k]) and np.array_equal(mine['idx'][:n + m], idx[k][:n + m])
            assert np.array_equal(mine['vec'][:n + m], vec[k][:n + m])
    # one step over all N open nodes; the oracle on the same nodes, table and pseudo costs
    cl, cr, tl, tr = t.pseudo_cost_arrays()
    has = ((tl > 0) | (tr > 0)).astype(np.uint8)
    with oracle.anchored(root_anchor):
        o = oracle.lp_solve_dive_batch(A, b, c, L, U, V, 1, ints, cl, cr, has, float('inf'),
                                       anchor_table=(T, vec, idx), anchor_sel=sel, depth=depth)
    assert np.array_equal(o['npivots'][:K], o['iters'][:K])       # own anchor: no refactorisation at all
    assert np.any(o['npivots'][K:N] > o['iters'][K:N])            # from the root's anchor: some
    before = t.stats()
    after = t.solve(mip_gap=0.0, frontier_batch=MB, max_steps=1)
    dives = after['dives'] - before['dives']
    # The engine keeps a dive child unless it needs strong-branching probes of its own (a fractional
    # integer variable without a pseudo-cost entry: it is then queued like any other child) -- which
    # children those are follows from the oracle's child solutions and the table, so the step's LP and
    # pivot counts are matched exactly: the nodes' pivots plus those of the kept children.
    # With a plunge (depth > 1) the chain of a node ends at the first child that is not kept.
    kept = np.zeros((depth, N), bool)
    alive = np.ones(N, bool)
    for lvl in range(depth):
        for k in np.where(alive & (o['dive_var'][lvl * N:(lvl + 1) * N] >= 0))[0]:
            child = (lvl + 1) * N + k
            xk = o['x'][child][ints]
            frac = np.minimum(xk - np.floor(xk), np.ceil(xk) - xk) > 1e-4
            kept[lvl, k] = o['status'][child] >= 0 and not np.any(frac & (has[ints] == 0))
        alive = kept[lvl]
        # (a kept child goes on only if it is itself optimal, fractional and below the cutoff: then
        # the oracle took a decision for it -- the same condition the engine applies)
    assert dives == int(kept.sum()) and after['lp_solved'] - before['lp_solved'] == N + dives
    assert after['pivots'] - before['pivots'] == int(o['npivots'][:N].sum()) + int(o['npivots'][N:][kept.reshape(-1)].sum())
    # descendants inherit their ancestor's entry; the search itself is unchanged by the anchors
    assert (t.peek_anchors(10 ** 6) >= 0).sum() >= K
    t2, _ = ramp()
    t2.solve(mip_gap=0.0, frontier_batch=MB, max_steps=1)
    a, b2 = (x.solve(mip_gap=0.0, frontier_batch=min(256, MB), max_steps=3) for x in (t, t2))
    assert a['evaluated_nodes'] == b2['evaluated_nodes'] and isclose(a['dual_bound'], b2['dual_bound'], abs_tol=1e-7)
    if target >= 300:   # (deep enough: near the root the root's own tableau is the closer anchor)
        assert a['pivots'] < b2['pivots']


def test_driver_options_for_the_native_mode():
    """BranchAndBound(frontier_batch > 1) turns the anchor and the dive on by default; both can be
    switched off; the exact mode has neither."""
    make = lambda: random_model(30, 12, 3)
    ref = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=1)
    ref.solve()
    assert ref._native_stats['dives'] == 0
    for kw in ({}, {'dive': False}, {'anchor': False}, {'anchor': False, 'dive': False}):
        bb = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=64,
                            pool_capacity=1 << 15, **kw)
        bb.solve()
        assert bb.status == 'optimal' and isclose(bb.objective_value, ref.objective_value, abs_tol=1e-6)
        assert (bb._native_stats['dives'] > 0) == kw.get('dive', True)
    with pytest.raises(AssertionError, match='dive needs frontier_batch > 1'):
        BranchAndBound(make(), gomory_cuts=False, frontier_batch=1, dive=True)


@pytest.mark.parametrize('Node', [DepthFirstSearchNode, PseudoCostBranchDepthFirstSearchNode])
def test_depth_first_batches_with_the_default_dive(Node):
    """Depth-first node classes in the batched native mode (anchor and dive on by default)."""
    for seed in (3, 4):
        make = lambda: random_model(30, 12, seed)
        ref = BranchAndBound(make(), Node, pseudo_costs={}, gomory_cuts=False, frontier_batch=1)
        ref.solve()
        assert ref.status == 'optimal'
        for batch in (8, 128):
            bb = BranchAndBound(make(), Node, pseudo_costs={}, gomory_cuts=False, frontier_batch=batch,
                                pool_capacity=1 << 15)
            bb.solve()
            assert bb.status == 'optimal' and isclose(bb.objective_value, ref.objective_value, abs_tol=1e-6)
            assert bb._native_stats['dives'] > 0
            x = bb.solution
            assert np.max(np.abs(x - np.round(x))) <= 1e-4


@pytest.mark.parametrize('Node', NODES)
def test_example_models_batched_mode_with_defaults(Node):
    """The 64 example models in the batched native mode (anchor + dive on by default): the optimum of
    the committed HiGHS table within 1e-6, an integral feasible solution."""
    for f, rec in sorted(TABLE.items()):
        path = os.path.join(HERE, 'golden', 'example_models', f)
        bb = BranchAndBound(MILPInstance(file_name=path), Node, pseudo_costs={}, gomory_cuts=False,
                            frontier_batch=16, pool_capacity=1 << 13)
        bb.solve()
        assert bb.status == 'optimal' and isclose(bb.objective_value, rec['milp_opt'], abs_tol=1e-6), f
        x = bb.solution
        idx = bb.model.integerIndices
        assert np.max(np.abs(x[idx] - np.round(x[idx]))) <= 1e-4, f


@pytest.mark.parametrize('n,m,B,dive,rule,search,steps', [
    (64, 32, 256, 1, 'pseudo cost', 'best first', 25),
    (64, 32, 256, 8, 'pseudo cost', 'best first', 25),
    (64, 32, 64, 4, 'most fractional', 'best first', 25),
    (40, 16, 16, 3, 'pseudo cost', 'depth first', 60),
    (256, 128, 512, 8, 'pseudo cost', 'best first', 40),
    (300, 150, 64, 4, 'pseudo cost', 'best first', 40)])    # (the HBM-streaming kernel)
def test_device_finish_matches_the_host_loop(n, m, B, dive, rule, search, steps):
    """The device finish (csrc/finish_kernels.hip.h: decisions, child records, compact open list, free
    rows, pseudo-cost recurrence as kernels) against the host loop it replaces (MIPX_HOST_FINISH=1: the
    restatement of branch_and_bound.py:243-266 / pseudo_cost.py:68-100 this engine was pinned with), one
    step per call so that both see the same table at every launch: after every step the same counters,
    bounds, open-node count and -- bit for bit -- the same pseudo-cost table; at the end the same open
    nodes (bounds, bases, inherited bounds)."""
    from simple_mip_solver_amd import _ffi
    ctx = _ffi.default_context()
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=1)
    probs = []

    def make(host):
        old = os.environ.pop('MIPX_HOST_FINISH', None)
        if host:
            os.environ['MIPX_HOST_FINISH'] = '1'
        probs.append(_ffi.Problem(ctx, A, b, c))     # (a problem of its own: the root's anchor belongs to the problem)
        try:
            t = _ffi.Tree(probs[-1], ints, l, u, branch_rule=rule, search_rule=search, max_batch=B, pool_capacity=1 << 15)
        finally:
            os.environ.pop('MIPX_HOST_FINISH', None)
            if old is not None:
                os.environ['MIPX_HOST_FINISH'] = old
        t.set_anchor_mode(True)
        t.set_dive(dive)
        return t
    dev, host = make(False), make(True)
    keys = ('evaluated_nodes', 'lp_solved', 'probes_solved', 'pivots', 'open_nodes', 'steps', 'dives', 'dual_bound',
            'primal_bound', 'status')
    fast_steps, prev = 0, 0
    for step in range(steps):
        a = dev.solve(mip_gap=1e-4, frontier_batch=B, max_steps=1)
        h = host.solve(mip_gap=1e-4, frontier_batch=B, max_steps=1)
        assert {k: a[k] for k in keys} == {k: h[k] for k in keys}, (step, a, h)
        pa, ph = dev.pseudo_cost_arrays(), host.pseudo_cost_arrays()
        for x, y in zip(pa, ph):
            assert np.array_equal(x, y), step
        fast_steps += a['probes_solved'] == prev    # (no probe request in the step: the device finished it)
        prev = a['probes_solved']
        if a['status'] != 4:
            break
    assert fast_steps > 0      # steps without probe requests did go through the device finish
    N = a['open_nodes']
    if N:
        key = lambda r: sorted((r[3][k], r[0][k].tobytes(), r[1][k].tobytes(), r[2][k].tobytes()) for k in range(len(r[3])))
        assert key(dev.peek_open(N)) == key(host.peek_open(N))
    if a['primal_bound'] < float('inf'):
        assert np.array_equal(dev.solution(), host.solution())
    # and pipelined (steps overlap, the table a launch sees is fresher on the device): the same optimum
    if (n, m) == (40, 16):
        ends = [t.solve(mip_gap=1e-4, frontier_batch=B) for t in (dev, host)]
        assert ends[0]['status'] == ends[1]['status'] == 1
        assert isclose(ends[0]['primal_bound'], ends[1]['primal_bound'], abs_tol=1e-9)
    dev.close(); host.close()
    for p in probs:
        p.close()
