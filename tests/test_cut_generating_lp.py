"""CutGeneratingLP against the reference's structural checks and known answers
(test_simple_mip_solver/test_utils/test_cut_generating_lp.py).  CPU oracle backend and, marked
gpu, the HIP engine."""
import glob
import os
from math import isclose
from unittest.mock import patch

import numpy as np
import pytest
from numpy.testing import assert_allclose

from simple_mip_solver_amd import BaseNode, BranchAndBound, CyLPArray, MILPInstance
from simple_mip_solver_amd.lp import COIN_INFINITY as inf
from simple_mip_solver_amd.utils.cut_generating_lp import CutGeneratingLP
from tests.support.example_models import model, std_model

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'example_models')


def test_init_fails_asserts(engine):
    bb = BranchAndBound(std_model('small_branch'), BaseNode, gomory_cuts=False)
    bb.solve()
    with pytest.raises(AssertionError, match='bb must be a BranchAndBound'):
        CutGeneratingLP(bb=5, root_id=0)
    with pytest.raises(AssertionError, match='root node of the disjunction must be present'):
        CutGeneratingLP(bb=bb, root_id=100)
    with pytest.raises(AssertionError, match='depth is postive integer'):
        CutGeneratingLP(bb=bb, root_id=0, depth=0)


def test_init(engine):
    bb = BranchAndBound(std_model('small_branch'), BaseNode, gomory_cuts=False)
    bb.solve()
    with patch.object(CutGeneratingLP, '_create_cglp') as cp:
        cglp = CutGeneratingLP(bb, bb.root_node.idx)
        assert cglp.bb is bb and cglp.root_id == 0 and cp.called


def test_create_cglp_fails_asserts(engine):
    m = std_model('small_branch')
    bb = BranchAndBound(m, gomory_cuts=False)
    bb.solve()
    n = [k for k in bb.tree.get_leaves(0) if k.lp_feasible is not False][0]
    n.lp.addVariable('d', 3)
    with patch.object(CutGeneratingLP, '_create_cglp'):
        cglp = CutGeneratingLP(bb, bb.root_node.idx)
    with pytest.raises(AssertionError, match='Each disjunctive term should have the same variables'):
        cglp._create_cglp()

    m = std_model('small_branch')
    bb = BranchAndBound(m, gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, bb.root_node.idx)
    A = np.matrix(np.append(m.A.copy(), [[-1, -1, -1]], axis=0))
    A_prime = np.matrix(np.append(m.A.copy(), [[-1], [-1]], axis=1))
    b = CyLPArray(np.append(m.b.copy(), [-3]))
    for kwargs, msg in [(dict(A=A), 'A and b must both'), (dict(b=b), 'A and b must both'),
                        (dict(A=np.array(A), b=b), 'A must be a numpy'),
                        (dict(A=A_prime, b=b), 'A must have same number of columns'),
                        (dict(A=A, b=np.append(m.b.copy(), [-3])), 'b must be a CyLPArray'),
                        (dict(A=A, b=CyLPArray(m.b.copy())), 'A must have the same number of rows'),
                        (dict(var_lb=[1, 0, 0]), 'var_lb must be a CyLPArray'),
                        (dict(var_lb=CyLPArray([1, 0])), 'Must have same number of lower bounds as variables'),
                        (dict(var_ub=[1, 0, 0]), 'var_ub must be a CyLPArray'),
                        (dict(var_ub=CyLPArray([1, 0])), 'Must have same number of upper bounds as variables')]:
        with pytest.raises(AssertionError, match=msg):
            cglp._create_cglp(**kwargs)


def _blocks(lp, *names):
    return [lp.getVarByName(k) for k in names]


def _check_term(lp, k, pi, pi0, u, w, v, At, b_row, lb_row, ub_row, nvar):
    c0, c1 = lp.constraints[2 * k], lp.constraints[2 * k + 1]
    assert len(c0.varCoefs) == 4 and len(c1.varCoefs) == 4
    assert (c0.varCoefs[pi] == -np.eye(nvar)).all() and (c0.varCoefs[u] == At).all()
    assert (c0.varCoefs[w] == np.eye(nvar)).all() and (c0.varCoefs[v] == -np.eye(nvar)).all()
    assert (c1.varCoefs[pi0] == -1).all() and (c1.varCoefs[u] == b_row).all()
    assert (c1.varCoefs[w] == lb_row).all() and (c1.varCoefs[v] == ub_row).all()
    assert all(c0.upper == 0) and all(c0.lower <= -1e300) and all(c1.lower == 0) and all(c1.upper >= 1e300)


def test_create_cglp_standard(engine):
    bb = BranchAndBound(std_model('small_branch'), gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, bb.root_node.idx)
    lp = cglp._create_cglp()
    dn = {n.idx: n for n in bb.tree.get_leaves(0) if n.lp_feasible is not False}
    assert sorted(dn) == [5, 11]
    pi, pi0, u_5, w_5, v_5, u_11, w_11, v_11 = _blocks(lp, 'pi', 'pi0', 'u_5', 'w_5', 'v_5', 'u_11', 'w_11', 'v_11')
    assert len(lp.variables) == 8
    for var in lp.variables:
        if var.name in ['pi', 'pi0']:
            assert_allclose(var.lower / -inf, 1)
        else:
            assert (var.lower == 0).all()
        assert_allclose(var.upper / inf, 1)
    assert (np.concatenate((bb.root_node.solution, [-1], np.zeros(16)), axis=None) == lp.objective).all()
    assert len(lp.constraints) == 5
    _check_term(lp, 0, pi, pi0, u_5, w_5, v_5, dn[5].lp.dense_rows().T, [-1.5, -1.25], np.zeros(3), [0, -1, -1], 3)
    _check_term(lp, 1, pi, pi0, u_11, w_11, v_11, dn[11].lp.dense_rows().T, [-1.5, -1.25], [1, 0, 0], [-1, -1, 0], 3)
    norm = lp.constraints[4]
    assert len(norm.varCoefs) == 6 and all((a == 1).all() for a in norm.varCoefs.values())
    assert norm.lower == 1 and norm.upper == 1 and norm.name == 'normalize'
    assert lp.constraints[0].name == 'Au_5 + Iw_5 - Iv5 <= pi' and lp.constraints[1].name == 'bu_5 + lbw_5 - ubv5 >= pi0'


def test_create_cglp_depth_1(engine):
    bb = BranchAndBound(std_model('small_branch'), gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, root_id=1, depth=1)
    lp = cglp._create_cglp()
    dn = {n.idx: n for n in bb.tree.get_leaves(subtree_root_id=1, depth=1, keep='feasible')}
    pi, pi0, u_3, w_3, v_3, u_4, w_4, v_4 = _blocks(lp, 'pi', 'pi0', 'u_3', 'w_3', 'v_3', 'u_4', 'w_4', 'v_4')
    assert len(lp.variables) == 8
    for var in lp.variables:
        if var.name in ['pi', 'pi0']:
            assert_allclose(var.lower / -inf, 1)
        else:
            assert (var.lower == 0).all()
        expect = {'v_3': [1, 0, 1], 'v_4': [0, 0, 1]}.get(var.name, 1)
        assert_allclose(var.upper / inf, np.array(expect))
    obj = np.concatenate((bb.tree.get_node_instances(1).solution, [-1], np.zeros(16)), axis=None)
    assert (obj == lp.objective).all() and len(lp.constraints) == 5
    _check_term(lp, 0, pi, pi0, u_3, w_3, v_3, dn[3].lp.dense_rows().T, [-1.5, -1.25], np.zeros(3), [0, 0, -1], 3)
    _check_term(lp, 1, pi, pi0, u_4, w_4, v_4, dn[4].lp.dense_rows().T, [-1.5, -1.25], [1, 0, 0], [0, 0, -1], 3)


def test_create_cglp_infinite_bounds(engine):
    bb = BranchAndBound(model('square'), node_limit=1, gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, bb.root_node.idx)
    lp = cglp._create_cglp()
    dn = {n.idx: n for n in bb.tree.get_leaves(0) if n.lp_feasible is not False}
    pi, pi0, u_1, w_1, v_1, u_2, w_2, v_2 = _blocks(lp, 'pi', 'pi0', 'u_1', 'w_1', 'v_1', 'u_2', 'w_2', 'v_2')
    assert_allclose(v_1.upper, np.array([inf, 0]))
    assert (v_2.upper == np.zeros(2)).all()
    assert (np.concatenate((bb.root_node.solution, [-1], np.zeros(12)), axis=None) == lp.objective).all()
    assert len(lp.constraints) == 5
    _check_term(lp, 0, pi, pi0, u_1, w_1, v_1, dn[1].lp.dense_rows().T, [-1.5, -1.5], np.zeros(2), [-1, 0], 2)
    _check_term(lp, 1, pi, pi0, u_2, w_2, v_2, dn[2].lp.dense_rows().T, [-1.5, -1.5], [2, 0], np.zeros(2), 2)


def test_create_cglp_new_coef_matrix_and_var_bounds(engine):
    m = model('small_branch_max')
    bb = BranchAndBound(m, gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, bb.root_node.idx)
    A = np.matrix(np.append(-np.array(m.A), [[-1, -1, -1]], axis=0))
    b = CyLPArray(np.append(-np.array(m.b), [-3]))
    lp = cglp._create_cglp(A=A, b=b, var_lb=CyLPArray([1, 0, 0]), var_ub=CyLPArray([1, 1, 0]))
    # every other term's bounds cross the new ones: a single term is left
    pi, pi0, u_11, w_11, v_11 = _blocks(lp, 'pi', 'pi0', 'u_11', 'w_11', 'v_11')
    assert len(lp.variables) == 5 and len(lp.constraints) == 3
    assert (np.concatenate((bb.root_node.solution, [-1], np.zeros(9)), axis=None) == lp.objective).all()
    _check_term(lp, 0, pi, pi0, u_11, w_11, v_11, np.asarray(A).T, [-1.5, -1.25, -3], [1, 0, 0], [-1, -1, 0], 3)
    assert len(lp.constraints[2].varCoefs) == 3


def test_solve_fails_asserts(engine):
    bb = BranchAndBound(model('square'), gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, bb.root_node.idx)
    with pytest.raises(AssertionError, match='x_star must be a CyLPArray'):
        cglp.solve(x_star=[1.5, 2])
    with pytest.raises(AssertionError, match='x_star must have the same number of variables'):
        cglp.solve(x_star=CyLPArray([1.5, 2, 5]))
    cglp.solve()
    basis = cglp.lp.getBasisStatus()
    with pytest.raises(AssertionError, match='first starting_basis element'):
        cglp.solve(starting_basis=(np.append(basis[0], [1]), basis[1]))
    with pytest.raises(AssertionError, match='second starting_basis element'):
        cglp.solve(starting_basis=(basis[0], np.append(basis[1], [1])))


def test_solve(engine):
    bb = BranchAndBound(model('square'), gomory_cuts=False)
    bb.solve()
    pi, pi0 = CutGeneratingLP(bb, bb.root_node.idx).solve()
    assert isinstance(pi, CyLPArray)  # the cut is x1 <= 1 or x2 <= 1
    assert_allclose(pi / pi0, [0, 1] if abs(pi[1]) > abs(pi[0]) else [1, 0], atol=.01)
    assert (pi - .01 < 0).all() and pi0 - .01 < 0

    bb = BranchAndBound(std_model('small_branch'), node_limit=10, gomory_cuts=False)
    bb.solve()
    pi, pi0 = CutGeneratingLP(bb, bb.root_node.idx).solve()
    assert_allclose(pi / pi0, np.array([0, 0, 1]), atol=.01)  # x3 <= 1
    assert (pi - .01 < 0).all() and pi0 - .01 < 0


def test_solve_doesnt_separate(engine):
    bb = BranchAndBound(model('square'), gomory_cuts=False)
    bb.solve()
    pi, pi0 = CutGeneratingLP(bb, bb.root_node.idx).solve(x_star=CyLPArray([.5, .5]))
    assert pi is not None and pi0 is not None


def test_solve_different_x_star(engine):
    bb = BranchAndBound(model('square'), node_limit=1, gomory_cuts=False)
    bb.solve()
    pi, pi0 = CutGeneratingLP(bb, bb.root_node.idx).solve(x_star=CyLPArray([1.5, 2]))
    assert isclose(pi0, -.75, abs_tol=.01)
    assert_allclose(pi, np.array([0, -.5]), atol=.01)


def test_solve_starting_basis(engine):
    bb = BranchAndBound(std_model('small_branch'), node_limit=10, gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, bb.root_node.idx)
    cglp.solve()
    basis = cglp.lp.getBasisStatus()
    bb = BranchAndBound(std_model('small_branch'), node_limit=10, gomory_cuts=False)
    bb.solve()
    cglp = CutGeneratingLP(bb, bb.root_node.idx)
    cglp.solve(starting_basis=basis)
    assert cglp.lp.iteration == 0


def test_solve_many_times(engine):
    """The first ten example models: the cut separates the root solution and keeps every
    feasible leaf of the tree."""
    for f in sorted(glob.glob(os.path.join(GOLD, '*.mps')))[:10]:
        bb = BranchAndBound(MILPInstance(file_name=f), gomory_cuts=False)
        bb.solve()
        cglp = CutGeneratingLP(bb=bb, root_id=bb.root_node.idx)
        pi, pi0 = cglp.solve()
        assert pi is not None, f
        assert sum(pi * bb.root_node.solution) <= pi0 + 1e-9, f
        for n in bb.tree.get_leaves(0):
            if n.lp_feasible:
                assert sum(pi * n.solution) >= pi0 - .01, f
