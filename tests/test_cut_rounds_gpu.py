"""Cut rounds inside the native frontier engine (BASELINE config C4).

(1) K1's cut-row variant -- a node's LP over the m shared rows plus the cut rows its list names --
    against the CPU oracle on the problem with those rows materialised: bit-exact.
(2) BranchAndBound(frontier_batch=1, gomory_cuts=True): the engine reproduces the per-node Python
    path (itself cross-checked against the oracle solve by solve) node for node -- ids, LP status,
    branching variable, objective, incumbent, pseudo costs and the running GMIC totals of
    BaseNode._base_bound -- on the reference's inline cut models, the 64 example models and random
    instances, and at 256 x 128 (config C4).
(3) Larger frontier batches with cuts reach the same optimum.
Reference: simple_mip_solver/nodes/base_node.py:137-230, :292-466; test_base_node.py:316-337 pins
the cut2 counters the totals are compared on."""
import json
from math import isclose
import os

import numpy as np
import pytest

from simple_mip_solver_amd import (BaseNode, BranchAndBound, DepthFirstSearchNode, MILPInstance,
                                   PseudoCostBranchNode, PseudoCostBranchDepthFirstSearchNode, _ffi)
from simple_mip_solver_amd import lp as lpmod
from simple_mip_solver_amd.generators import random_dense_milp_arrays
from tests.support.example_models import model, std_model

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(__file__)
TABLE = json.load(open(os.path.join(HERE, 'golden', 'example_models_optima.json')))['models']
NODES = [BaseNode, PseudoCostBranchNode, DepthFirstSearchNode, PseudoCostBranchDepthFirstSearchNode]
INF = np.inf


@pytest.fixture(autouse=True)
def hip_backend():
    from tests.support.compare_backend import CompareBackend
    lpmod.set_backend(CompareBackend())
    yield
    lpmod.set_backend(None)


# ---- (1) the kernel variant -----------------------------------------------------------------------
@pytest.mark.parametrize('n,m,seed', [(12, 6, 0), (64, 32, 1), (100, 40, 2), (256, 128, 0), (200, 150, 3),
                                       (300, 150, 0), (600, 300, 2)])   # (the last two: K1b, tableau streamed)
def test_lp_with_cut_rows_matches_the_oracle_on_materialised_rows(n, m, seed, gpu_ctx, oracle):
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    root = oracle.lp_solve(A, b, c, l, u)
    assert root['status'] == 0
    # a store of real cuts: the root's Gomory cuts, raw (deep: they bind) and rounded
    g = oracle.gomory(A, b, c, l, u, root['vstat'], root['x'], ints)
    assert len(g['row_idx']) >= 2
    store_pi = np.vstack([g['pi'], g['safe_pi']])
    store_pi0 = np.concatenate([g['pi0'], g['safe_pi0']])
    K = len(store_pi0)
    kc = 64 if n > 256 else (min(64, 192 - m) if m + 64 > 128 else 64)   # (above the register tiles: K1b takes 64)
    rng = np.random.default_rng(seed)
    sizes = [0, 1, 2, min(5, K), min(kc, K), min(kc // 2, K), 0, 3]
    lists = [sorted(rng.choice(K, size=s, replace=False).tolist()) if s else [] for s in sizes]
    B = len(lists)
    L = np.repeat(l[None], B, axis=0); U = np.repeat(u[None], B, axis=0)
    for k in range(B):     # a branching bound on some of them
        if k % 2:
            j = int(np.argmax(np.minimum(root['x'] - np.floor(root['x']), np.ceil(root['x']) - root['x'])))
            U[k, j] = np.floor(root['x'][j])
    # warm start: the root's basis, every cut row's slack basic (a new row enters that way)
    V = [np.concatenate([root['vstat'], np.ones(len(c_), np.int8)]) for c_ in lists]
    for warm in (True, False):
        got = p.solve_batch_cuts(L, U, V if warm else None, store_pi, store_pi0, lists, kc=kc)
        for k, ids in enumerate(lists):
            Ak = np.vstack([A, store_pi[ids]]) if ids else A
            bk = np.concatenate([b, store_pi0[ids]]) if ids else b
            want = oracle.lp_solve(Ak, bk, c, L[k], U[k], V[k] if warm else None)
            assert got['status'][k] == want['status'] and got['iters'][k] == want['iters'], (k, ids)
            assert got['npivots'][k] == want['npivots']
            assert np.array_equal(got['vstat'][k], want['vstat'])
            if want['status'] != 1:
                assert np.array_equal(got['x'][k], want['x']) and got['obj'][k] == want['obj']
                assert np.array_equal(got['y'][k], want['y'])
    # truncated solves (strong-branching probes of a node that carries cuts)
    got = p.solve_batch_cuts(L, U, V, store_pi, store_pi0, lists, max_iter=5, kc=kc)
    for k, ids in enumerate(lists):
        Ak = np.vstack([A, store_pi[ids]]) if ids else A
        bk = np.concatenate([b, store_pi0[ids]]) if ids else b
        want = oracle.lp_solve(Ak, bk, c, L[k], U[k], V[k], 5)
        assert got['status'][k] == want['status'] and got['iters'][k] == want['iters']
        if want['status'] != 1:
            assert got['obj'][k] == want['obj']


def test_cut_row_entry_point_checks_its_arguments(gpu_ctx):
    A, b, c, l, u, ints = random_dense_milp_arrays(12, 6, seed=0)
    p = _ffi.Problem(gpu_ctx, A, b, c)
    pi = np.ones((2, 12)); pi0 = np.zeros(2)
    with pytest.raises(_ffi.MipxError, match='cut id out of range'):
        p.solve_batch_cuts(l[None], u[None], None, pi, pi0, [[0, 5]])
    A2, b2, c2, l2, u2, _ = random_dense_milp_arrays(256, 150, seed=0)   # a register-tile shape whose cut rows leave the tiles
    p2 = _ffi.Problem(gpu_ctx, A2, b2, c2)
    with pytest.raises(_ffi.MipxError, match='MIPX_ETOOBIG'):
        p2.solve_batch_cuts(l2[None], u2[None], None, np.ones((64, 256)), np.zeros(64), [list(range(64))])
    A3, b3, c3, l3, u3, _ = random_dense_milp_arrays(600, 1000, seed=0)  # 1000 + 64 rows: above every kernel
    p3 = _ffi.Problem(gpu_ctx, A3, b3, c3)
    with pytest.raises(_ffi.MipxError, match='MIPX_ETOOBIG'):
        p3.solve_batch_cuts(l3[None], u3[None], None, np.ones((64, 600)), np.zeros(64), [list(range(64))])


# ---- (2) the engine against the per-node Python path -------------------------------------------------
def python_run(make_model, Node, **kw):
    bb = BranchAndBound(make_model(), Node, pseudo_costs={}, gomory_cuts=True, **kw)
    trace = []
    inner = bb._evaluate_node

    def spy(node):
        before = bb.evaluated_nodes
        inner(node)
        if bb.evaluated_nodes > before:
            kids = bb.tree.get_children(node.idx)
            bvar = bb.tree.get_node_instances(kids[0])._b_idx if kids else -1
            trace.append((node.idx, node.lp.getStatusCode(), bvar, node.lp.objectiveValue,
                          node.lp.nConstraints))
    bb._evaluate_node = spy
    bb.solve()
    return bb, trace


def native_run(make_model, Node, frontier_batch=1, **kw):
    nb = BranchAndBound(make_model(), Node, pseudo_costs={}, gomory_cuts=True, frontier_batch=frontier_batch, **kw)
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
    return nb


def assert_same_search(make_model, Node, **kw):
    py, ptrace = python_run(make_model, Node, **kw)
    nb = native_run(make_model, Node, **kw)
    assert nb._native.cuts and nb._native_cuts_dropped == 0
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
    for k, (idx, st, bvar, obj, rows) in enumerate(ptrace):
        assert tr['node_id'][k] == idx and tr['status'][k] == st, (k, ptrace[k])
        assert tr['branch_var'][k] == bvar, (k, ptrace[k], tr['branch_var'][k])
        if st in (0, 3):
            assert tr['objective'][k] == obj, (k, ptrace[k], tr['objective'][k])
    for key in _ffi.CUT_TOTAL_KEYS:     # the totals BaseNode._base_bound threads through the kwargs
        assert nb._kwargs[key] == py._kwargs[key], (key, nb._kwargs[key], py._kwargs[key])
    if issubclass(Node, PseudoCostBranchNode):
        assert nb._kwargs['pseudo_costs'] == py._kwargs['pseudo_costs']
    assert nb.dual_bound == py.dual_bound
    assert nb.current_gap == py.current_gap
    return nb, py


def test_cut2_root_reproduces_the_reference_counters():
    """test_base_node.py:316-337: cut2's root goes -38 -> -36.48 -> -36 in three rounds, six GMICs
    created and added, two removed in one round; one node, mip-feasible after the cuts."""
    nb, py = assert_same_search(lambda: model('cut2'), BaseNode)
    assert nb.status == 'optimal' and nb.evaluated_nodes == 1 and isclose(nb.objective_value, -36, abs_tol=1e-9)
    want = dict(total_cut_generation_iterations=3, total_iterations_gmic_created=3, total_number_gmic_created=6,
                total_iterations_gmic_added=3, total_number_gmic_added=6, total_iterations_gmic_removed=1,
                total_number_gmic_removed=2)
    assert {k: nb._kwargs[k] for k in want} == want


@pytest.mark.parametrize('Node', NODES)
@pytest.mark.parametrize('name', ['cut1', 'cut2', 'cut3', 'small_branch', 'no_branch', 'infeasible2', 'unbounded',
                                  'square', 'lift_project', 'h3p1'])
def test_inline_models_exact_mode_with_cuts(Node, name):
    assert_same_search(lambda: model(name), Node)


@pytest.mark.parametrize('Node', NODES)
def test_std_small_branch_exact_mode_with_cuts(Node):
    assert_same_search(lambda: std_model('small_branch'), Node)


@pytest.mark.parametrize('Node', NODES)
def test_example_models_exact_mode_with_cuts(Node):
    added = 0
    for f, rec in sorted(TABLE.items()):
        path = os.path.join(HERE, 'golden', 'example_models', f)
        nb, py = assert_same_search(lambda: MILPInstance(file_name=path), Node)
        # (equal to the per-node path exactly; against the independent optimum the reference's own bar
        # for runs with cuts, helpers.py:45: near-integral incumbents move the objective by ~1e-4)
        assert nb.status == 'optimal' and isclose(nb.objective_value, rec['milp_opt'], abs_tol=.01), f
        added += nb._kwargs['total_number_gmic_added']
    assert added > 0     # the family does exercise added cut rows, inherited by children


def random_model(n, m, seed, density=1.0):
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
    return MILPInstance(A=A, b=b, c=c, l=l, u=u, sense=['Min', '>='], integerIndices=ints, numVars=n)


@pytest.mark.parametrize('Node', [BaseNode, PseudoCostBranchNode])
@pytest.mark.parametrize('n,m,seed,density', [(12, 6, 0, 1.0), (20, 10, 1, 1.0), (30, 15, 2, 0.3), (40, 30, 3, 0.2)])
def test_random_models_exact_mode_with_cuts(Node, n, m, seed, density):
    assert_same_search(lambda: random_model(n, m, seed, density), Node, node_limit=60)


def test_keyword_overrides_reach_the_engine():
    """max_cut_generation_iterations / min_cut_depth / parallel_cut_tolerance / ... travel through
    **kwargs to the node methods in the reference; the engine takes them from the same place."""
    for kw in (dict(max_cut_generation_iterations=1), dict(min_cut_depth=.05), dict(parallel_cut_tolerance=45),
               dict(max_nonzero_coefs=1), dict(cutting_plane_progress_tolerance=.05),
               dict(max_relative_cut_term_ratio=.05)):
        assert_same_search(lambda: model('cut2'), BaseNode, **kw)
        assert_same_search(lambda: random_model(20, 10, 1), PseudoCostBranchNode, node_limit=25, **kw)


def test_c4_at_256_x_128_exact_mode():
    """BASELINE config C4's instance with the reference's default gomory_cuts=True: the first nodes
    of the engine's search are those of the per-node path (every LP, Gomory round and selection of
    which is checked against the oracle by the comparing backend), cut rounds included."""
    nb, py = assert_same_search(lambda: random_model(256, 128, 0), PseudoCostBranchNode, node_limit=8)
    assert nb._kwargs['total_cut_generation_iterations'] >= 1
    assert nb._kwargs['total_number_gmic_created'] > 50


# ---- (3) batches ------------------------------------------------------------------------------------
@pytest.mark.parametrize('Node', [BaseNode, PseudoCostBranchNode])
def test_frontier_batches_with_cuts_find_feasible_incumbents(Node):
    """Batches with cut rounds.  NOT asserted: the optimum of the run without cuts.  The reference
    applies the textbook GMI formula also where a nonbasic variable sits at an upper bound
    (base_node.py:496-503 -- every child of a left branch has one), where the cut is not valid in
    general; it is reproduced as is, so a run with cuts can lose the optimum, and which cuts a run
    meets depends on its node order: the per-node Python path itself ends at -165 on the first
    instance below, whose optimum is -166 (exact mode reproduces that path node for node, above).
    Asserted: the search ends, with an integral solution that satisfies every original row and
    bound and has the objective reported, never better than the true optimum."""
    for n, m, seed, density in ((20, 10, 1, 1.0), (24, 10, 12, 1.0), (30, 15, 2, 0.3)):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
        make = lambda: random_model(n, m, seed, density)
        ref = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=1)
        ref.solve()
        assert ref.status == 'optimal'
        for batch in (4, 64):
            for anchor in (True, False):
                bb = BranchAndBound(make(), Node, pseudo_costs={}, gomory_cuts=True, frontier_batch=batch,
                                    pool_capacity=1 << 15, anchor=anchor)
                bb.solve()
                assert bb.status == 'optimal'
                x = bb.solution
                assert np.max(np.abs(x - np.round(x))) <= 1e-4
                assert np.all(A @ x >= b - 1e-6) and np.all(x >= l - 1e-9) and np.all(x <= u + 1e-9)
                assert isclose(float(c @ x), bb.objective_value, abs_tol=1e-6)
                assert ref.objective_value - 1e-6 <= bb.objective_value <= ref.objective_value + 2.0
                assert bb._native_stats['dives'] == 0 and bb._kwargs['total_cut_generation_iterations'] > 0
                assert bb._native_cuts_dropped == 0


@pytest.mark.parametrize('n,m,seed', [(20, 10, 1), (40, 16, 3), (64, 32, 0), (256, 128, 0)])
def test_the_tableau_a_solve_ends_with_serves_the_cut_rounds(n, m, seed):
    """Without exact_tableau K2 reads the tableau the LP launch itself ends with (dumped by that launch)
    instead of one refactored from the slack basis by a launch of its own: two tableaus of the same basis
    that differ in their last bits.  At the root the cut rounds must come out the same: rounds, GMICs
    created / added / removed, and the bound after the last round."""
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    ctx = _ffi.default_context()
    prob = _ffi.Problem(ctx, A, b, c)
    out = []
    for exact in (1, 0):
        t = _ffi.Tree(prob, ints, l, u, branch_rule='most fractional', max_batch=4, pool_capacity=64,
                      cut_params=dict(max_abs_coef=1000.0 * float(np.max(np.abs(A))), exact_tableau=exact))
        if not exact:
            t.set_anchor_mode(True)
        st = t.solve(mip_gap=0.0, frontier_batch=4, node_limit=1)
        out.append((st, t.cut_stats()))
        t.close()
    (se, ce), (sf, cf) = out
    assert se['evaluated_nodes'] == sf['evaluated_nodes'] == 1
    assert ce['total_cut_generation_iterations'] >= 1 and ce['total_number_gmic_created'] > 0
    for key in _ffi.CUT_TOTAL_KEYS:
        assert ce[key] == cf[key], (key, ce, cf)
    assert isclose(se['dual_bound'], sf['dual_bound'], rel_tol=1e-9, abs_tol=1e-9)
    prob.close()


def test_reanchoring_with_cut_rounds():
    """mipx_tree_reanchor on a tree with cut rounds: the open nodes that carry no cut row get the tableau of
    their own basis as anchor, the others keep refactoring from the slack basis.  Same end as without:
    a feasible integral solution, never better than the optimum of the run without cuts."""
    for n, m, seed in ((20, 10, 1), (24, 10, 12)):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
        ref = BranchAndBound(random_model(n, m, seed, 1.0), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False,
                             frontier_batch=1)
        ref.solve()
        ctx = _ffi.default_context()
        prob = _ffi.Problem(ctx, A, b, c)
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=8, pool_capacity=1 << 14,
                      cut_params=dict(max_abs_coef=1000.0 * float(np.max(np.abs(A))), exact_tableau=0))
        t.set_anchor_mode(True)
        st = t.solve(mip_gap=1e-4, frontier_batch=8, max_steps=6)
        rounds = 0
        while st['status'] == 4 and rounds < 200:
            t.reanchor(st['open_nodes'])
            st = t.solve(mip_gap=1e-4, frontier_batch=8, max_steps=6)
            rounds += 1
        assert st['status'] == 1 and rounds >= 1
        cs = t.cut_stats()
        assert cs['total_cut_generation_iterations'] > 0
        x = t.solution()
        assert np.max(np.abs(x[ints] - np.round(x[ints]))) <= 1e-4
        assert np.all(A @ x >= b - 1e-6) and np.all(x >= l - 1e-9) and np.all(x <= u + 1e-9)
        assert isclose(float(c @ x), st['primal_bound'], abs_tol=1e-6)
        assert ref.objective_value - 1e-6 <= st['primal_bound'] <= ref.objective_value + 2.0
        t.close()
        prob.close()


def test_example_models_batched_with_cuts():
    for f, rec in sorted(TABLE.items()):
        path = os.path.join(HERE, 'golden', 'example_models', f)
        bb = BranchAndBound(MILPInstance(file_name=path), PseudoCostBranchNode, pseudo_costs={}, frontier_batch=16,
                            pool_capacity=1 << 13)
        bb.solve()
        assert bb.status == 'optimal' and isclose(bb.objective_value, rec['milp_opt'], abs_tol=.01), f


def test_cut_mode_limits():
    with pytest.raises(AssertionError, match='dive is not available'):
        BranchAndBound(random_model(20, 10, 1), frontier_batch=8, dive=True)
    # shapes beyond the register tiles run their cut rounds on the HBM-streaming kernel (reference default
    # gomory_cuts=True at BASELINE config C5's kind of shape)
    bb = BranchAndBound(random_model(300, 150, 0), PseudoCostBranchNode, pseudo_costs={}, frontier_batch=4, node_limit=24)
    bb.solve()
    assert bb._native.cuts and bb._kwargs['total_cut_generation_iterations'] > 0 and bb._kwargs['total_number_gmic_created'] > 0


# ---- (4) one step of the BATCHED cut mode, node by node through the oracle ---------------------------
def replay_cut_loop(oracle, A, b, c, l, u, ints, first, rows_pi, rows_pi0, vstat, params, first_dump=None,
                    root_anchor=None):
    """BaseNode._base_bound's loop (base_node.py:196-203, :292-341) for one node on oracle pieces only:
    `first` is the node's LP result, rows_pi / rows_pi0 the cut rows it carries, vstat its basis over
    n + m + cuts.  first_dump = None: every round's tableau is refactored from the basis (the per-node
    path, exact_tableau = 1).  first_dump given (the tableau state the first solve ended with): the
    batched mode -- a round reads the tableau the last solve ENDED WITH, and a tableau of its own only
    where it starts by removing rows (from the root's tableau if no cut row is left and the root has
    one, else from the slack basis).  Returns (8 counters like mipx_tree_trace_cuts, status, objective)."""
    import contextlib
    m, n = A.shape
    rows_pi, rows_pi0 = [np.asarray(p) for p in rows_pi], list(rows_pi0)
    cnt = dict(rounds=0, it_created=0, n_created=0, it_added=0, n_added=0, it_removed=0, n_removed=0)
    pool_pi, pool_pi0 = [], []
    r, dump, stalled = first, first_dump, False
    fused = first_dump is not None
    vstat = np.asarray(vstat, np.int8).copy()

    def materialised():
        Ak = np.vstack([A] + [p[None] for p in rows_pi]) if rows_pi else A
        bk = np.concatenate([b, rows_pi0]) if rows_pi else b
        return Ak, bk
    while True:
        feas = r['status'] in (0, 2)
        obj = r['obj'] if feas else INF
        if not (feas and not oracle.mip_feasible(ints, r['x']) and not stalled and
                cnt['rounds'] < params['max_rounds'] and obj < INF):
            break
        cnt['rounds'] += 1
        before = obj
        changed = False
        if rows_pi:     # _remove_slack_cuts: rows whose dual is exactly 0 leave, the basis is compacted
            y = r['y']
            keep = [k for k in range(len(rows_pi)) if y[m + k] != 0.0]
            if len(keep) < len(rows_pi):
                cnt['it_removed'] += 1; cnt['n_removed'] += len(rows_pi) - len(keep)
                vstat = np.concatenate([vstat[:n + m], vstat[n + m:][keep]])
                rows_pi = [rows_pi[k] for k in keep]; rows_pi0 = [rows_pi0[k] for k in keep]
                changed = True
                if fused:   # the dumped tableau has the removed rows in it: one of its own
                    Ak, bk = materialised()
                    cm = oracle.anchored(root_anchor) if (root_anchor is not None and not rows_pi) else contextlib.nullcontext()
                    with cm:
                        _, dump = oracle.debug_dump(Ak, bk, c, l, u, vstat, 0)
        Ak, bk = materialised()
        x = np.maximum(r['x'], 0)
        g = oracle.gomory_from_dump(Ak, bk, dump, x, ints, params['max_term']) if fused else \
            oracle.gomory(Ak, bk, c, l, u, vstat, x, ints, params['max_term'])
        if len(g['row_idx']):
            cnt['it_created'] += 1; cnt['n_created'] += len(g['row_idx'])
        pool_pi += list(g['safe_pi']); pool_pi0 += list(g['safe_pi0'])
        added, _, _ = oracle.select_cuts(np.array(pool_pi).reshape(len(pool_pi0), n), np.array(pool_pi0), x, 1000000,
                                         params['min_cut_depth'], params['cos_parallel'], params['max_abs_coef'])
        if len(added):
            cnt['it_added'] += 1; cnt['n_added'] += len(added)
            for i in added:
                rows_pi.append(pool_pi[i]); rows_pi0.append(pool_pi0[i])
            vstat = np.concatenate([vstat, np.ones(len(added), np.int8)])   # a new row enters with its slack basic
            gone = set(int(i) for i in added)
            pool_pi = [p for i, p in enumerate(pool_pi) if i not in gone]
            pool_pi0 = [p for i, p in enumerate(pool_pi0) if i not in gone]
            changed = True
        if changed:     # (an unchanged LP is not re-solved by the engine: same objective, it stalls either way)
            Ak, bk = materialised()
            if fused:
                rb, dump = oracle.debug_dump(Ak, bk, c, l, u, vstat, 0)
                r = {k: v[0] for k, v in rb.items()}
            else:
                r = oracle.lp_solve(Ak, bk, c, l, u, vstat)
            vstat = r['vstat']
        new = r['obj'] if r['status'] in (0, 2) else INF
        with np.errstate(invalid='ignore', divide='ignore'):
            if abs(before - new) / abs(before) < params['progress_tol']:
                stalled = True
    return ([cnt['rounds'], cnt['it_created'], cnt['n_created'], cnt['it_added'], cnt['n_added'], cnt['it_removed'],
             cnt['n_removed'], len(rows_pi)], r['status'], r['obj'] if r['status'] in (0, 2) else INF)


@pytest.mark.parametrize('exact', [0, 1])
@pytest.mark.parametrize('n,m,seed,density,boxed,target,MB', [
    (256, 128, 0, 1.0, True, 400, 1024),     # BASELINE C4: the bench's family (cuts created, none added)
    (256, 128, 1, 1.0, False, 200, 512),     # 256 x 128 where the reference's rules DO add cuts: nodes carry rows
    (64, 32, 5, 1.0, False, 200, 512),       # nodes that carry, gain and lose cut rows
    (64, 32, 2, 0.25, True, 200, 512),
    (300, 30, 5, 1.0, False, 60, 128)])      # above the register tiles (n > 256): K1b with cut rows carried, gained and lost
def test_batched_cut_mode_step_replays_through_the_oracle(n, m, seed, density, boxed, target, MB, exact, gpu_ctx, oracle):
    """The configuration bench.py measures C4 on -- exact_tableau=0 (K2 reads the tableau a solve ends
    with), anchored, re-anchored, a whole frontier per step -- checked at step level: every node of one
    step against the oracle's replay of BaseNode._base_bound on the same record (bounds, basis, the cut
    rows it carries): cut rounds, GMICs created / added / removed, cut rows kept, final LP status, final
    objective to 1e-9.  exact = 1: the same batch with the per-node path's tableau (refactored per round)."""
    import contextlib
    import math
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
    if not boxed:
        u = np.full(n, INF)
    prob = _ffi.Problem(gpu_ctx, A, b, c)
    params = dict(max_rounds=10, progress_tol=1e-4, min_cut_depth=1e-8, cos_parallel=math.cos(math.radians(10)),
                  max_abs_coef=1000.0 * float(np.max(np.abs(A))), max_term=1e3)
    t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=MB, pool_capacity=1 << 15,
                  cut_params=dict(max_abs_coef=params['max_abs_coef'], exact_tableau=exact))
    t.set_anchor_mode(True)
    st = t.stats()
    while st['open_nodes'] < target:
        st = t.solve(mip_gap=0.0, frontier_batch=64, max_steps=1)
        assert st['status'] == 4
    t.reanchor(st['open_nodes'])
    N = st['open_nodes']
    assert N <= MB
    L, U, V, _ = t.peek_open(N)
    ids, ncut, lists, codes = t.peek_cuts(N)
    sel = t.peek_anchors(N)
    atab = t.anchor_table()
    store_pi, store_pi0 = t.cut_store()
    assert len(ids) == N and np.all(sel[ncut > 0] == -1)
    t.set_trace(True)
    t.solve(mip_gap=0.0, frontier_batch=MB, max_steps=1)
    tr, tc = t.trace(), t.trace_cuts()
    assert len(tr['node_id']) == len(tc) > 0 and set(tr['node_id']) <= set(ids.tolist())
    where = {int(i): k for k, i in enumerate(ids)}
    # the root's tableau is an anchor only if the root kept no cut row (its final basis is then its LP optimum's:
    # rounds that add no row leave the basis alone)
    root = oracle.lp_solve(A, b, c, l, u)
    root_kept_rows = replay_cut_loop(oracle, A, b, c, l, u, ints, root, [], [], root['vstat'], params)[0][7] > 0
    root_anchor = None if root_kept_rows else oracle.make_anchor(A, b, c, root['vstat'])
    seen_rows = seen_added = seen_removed = 0
    for pos, nid in enumerate(tr['node_id']):
        k = where[int(nid)]
        rp = [store_pi[i] for i in lists[k, :ncut[k]]]
        rp0 = [store_pi0[i] for i in lists[k, :ncut[k]]]
        vfull = np.concatenate([V[k], codes[k, :ncut[k]]])
        if ncut[k] == 0:
            # the first LP refactors from the anchor the record names: its own table entry, else the root's
            # tableau, else the slack basis
            anchor = root_anchor if sel[k] < 0 else dict(T=atab[0][sel[k]], vec=atab[1][sel[k]], idx=atab[2][sel[k]])
            with (oracle.anchored(anchor) if anchor is not None else contextlib.nullcontext()):
                fb, dump = oracle.debug_dump(A, b, c, L[k], U[k], vfull, 0)
        else:       # a node that carries cut rows refactors from the slack basis of its own rows
            fb, dump = oracle.debug_dump(np.vstack([A] + [p[None] for p in rp]), np.concatenate([b, rp0]), c, L[k], U[k], vfull, 0)
        f = {key: v[0] for key, v in fb.items()}
        if f['status'] in (0, 2):
            vfull = f['vstat']
        counters, status, obj = replay_cut_loop(oracle, A, b, c, L[k], U[k], ints, f, rp, rp0, vfull, params,
                                                first_dump=None if exact else dump, root_anchor=root_anchor)
        assert tr['status'][pos] == status, (pos, int(nid), tr['status'][pos], status)
        assert tc[pos].tolist() == counters, (pos, int(nid), ncut[k], tc[pos].tolist(), counters)
        if status in (0, 2):
            assert isclose(tr['objective'][pos], obj, rel_tol=1e-9, abs_tol=1e-9), (pos, tr['objective'][pos], obj)
        seen_rows += ncut[k] > 0; seen_added += counters[4]; seen_removed += counters[6]
    assert tc[:, 2].sum() > 0                      # GMICs were created
    if (n, boxed) != (256, True):                  # ... and on these shapes nodes carry cut rows into the step
        assert seen_rows > 0
    if (n, m, seed) == (64, 32, 5):                # ... gain and lose them
        assert seen_added > 0 and seen_removed > 0
    t.close()
    prob.close()
