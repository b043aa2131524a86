"""PseudoCostBranchNode against the reference's known answers
(test_simple_mip_solver/test_nodes/test_branch/test_pseudo_cost.py:28-220) and the golden vectors
made from its own _calculate_costs / _best_pseudo_costs_index."""
from itertools import product
import json
import os
from unittest.mock import patch

import numpy as np
import pytest

from simple_mip_solver_amd import BranchAndBound, PseudoCostBranchNode, BaseNode
from simple_mip_solver_amd.lp import DenseLP
from tests.support.example_models import model

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'base_node.json')))


def make_node(**kw):
    m = model('small_branch')
    return PseudoCostBranchNode(m.lp, m.integerIndices, **kw)


def expect_root_table(table):
    for idx, direction in product([1, 2], ['right', 'left']):
        assert table[idx][direction]['times'] == 1
        assert table[idx][direction]['cost'] == (1 if (idx, direction) == (1, 'left') else 0)


def test_init(engine):
    node = make_node()
    assert node.branch_method == 'pseudo cost' and node.pseudo_costs is None
    assert node.strong_branch_iters is None and isinstance(node, BaseNode)


def test_bound(engine):
    with pytest.raises(AssertionError, match='pseudo cost dict has following errors:'):
        make_node().bound({1: 'hi'})
    node = make_node()
    rtn = node.bound({}, gomory_cuts=False)
    expect_root_table(node.pseudo_costs)
    expect_root_table(rtn['pseudo_costs'])
    assert node.strong_branch_iters == 5 and rtn['pseudo_costs'] is node.pseudo_costs
    for feasible, calls in [(False, 0), (True, 1)]:
        node = make_node()
        node.lp_feasible = feasible
        with patch.object(node, '_check_pseudo_costs') as cpc, patch.object(node, '_base_bound') as bb, \
                patch.object(node, '_update_pseudo_costs') as upc:
            cpc.return_value = []
            node.bound({}, gomory_cuts=False)
            assert cpc.call_count == 1 and bb.call_count == 1 and upc.call_count == calls


def test_update_pseudo_costs_call_graph(engine):
    node = make_node()
    node.pseudo_costs, node.strong_branch_iters = {}, 5
    node._base_bound(gomory_cuts=False)
    dummy = {'right': make_node(), 'left': make_node()}
    with patch.object(node, '_strong_branch') as sb, patch.object(node, '_calculate_costs') as cc:
        sb.return_value = dummy
        node._update_pseudo_costs()
        assert sb.call_count == 2 and cc.call_count == 4
    left = node._base_branch(2)['left']
    left.pseudo_costs = {1: {'right': {'cost': 0, 'times': 1}, 'left': {'cost': 1, 'times': 1}},
                         2: {'right': {'cost': 0, 'times': 1}, 'left': {'cost': 0, 'times': 1}}}
    left.strong_branch_iters = 5
    left._base_bound(gomory_cuts=False)
    with patch.object(left, '_strong_branch') as sb, patch.object(left, '_calculate_costs') as cc:
        sb.return_value = dummy
        left._update_pseudo_costs()
        assert sb.call_count == 1 and cc.call_count == 3


def test_batched_probes_equal_one_at_a_time(engine):
    """The single-batch strong branching gives the table the per-index reference loop gives."""
    a, b = make_node(), make_node()
    a.pseudo_costs, a.strong_branch_iters = {}, 5
    a._base_bound(gomory_cuts=False)
    a._update_pseudo_costs()
    b.pseudo_costs = {}
    b._base_bound(gomory_cuts=False)
    for idx in [1, 2]:
        for probe in b._strong_branch(idx, 5).values():
            b._calculate_costs(probe)
    assert a.pseudo_costs == b.pseudo_costs
    expect_root_table(a.pseudo_costs)


def test_calculate_costs(engine):
    node = make_node()
    node.pseudo_costs = {}
    node._base_bound(gomory_cuts=False)
    for idx in [1, 2]:
        for probe in node._strong_branch(idx).values():
            node._calculate_costs(probe)
    expect_root_table(node.pseudo_costs)
    kids = {k: v for k, v in node._base_branch(1).items() if k in ['left', 'right']}
    for child in kids.values():
        child.pseudo_costs = node.pseudo_costs
        child._base_bound(gomory_cuts=False)
        child._calculate_costs(child)
    for direction in ['right', 'left']:
        assert node.pseudo_costs[1][direction]['times'] == 2
        assert node.pseudo_costs[1][direction]['cost'] == (1 if direction == 'left' else 0)


def test_calculate_costs_matches_reference_vectors(engine):
    for rec in GOLD['pseudo_costs']:
        up = rec['update']
        node = make_node()
        node._integer_indices = rec['integer_indices']
        node.pseudo_costs = {int(i): {d: dict(v) for d, v in e.items()} for i, e in rec['table'].items()}

        class Child:
            pass
        child = Child()
        child._b_idx, child._b_dir, child._b_val = up['b_idx'], up['b_dir'], up['b_val']
        child.dual_bound = up['dual_bound']
        child.lp = DenseLP()
        child.lp.addVariable('x', len(up['l']))
        child.lp.variablesLower, child.lp.variablesUpper = np.array(up['l']), np.array(up['u'])
        child.lp._status, child.lp._obj_value = up['status'], up['objective']
        node._calculate_costs(child)
        want = {int(i): e for i, e in up['table_after'].items()}
        assert node.pseudo_costs == want, rec
        if 'best_index' in rec:
            node.solution = np.array(rec['x'])
            table = {int(i): e for i, e in rec['table'].items()}
            assert node._best_pseudo_costs_index(table) == rec['best_index']


def test_branch(engine):
    node = make_node()
    with pytest.raises(AssertionError, match='pseudo cost dict has following errors:'):
        node.branch({1: 'hi'})
    node.mip_feasible = True
    with pytest.raises(AssertionError, match='must have fractional value to branch'):
        node.branch({1: 'hi'})
    node = make_node()
    rtn = node.bound({}, gomory_cuts=False)
    with patch.object(node, '_check_pseudo_costs') as cpc, \
            patch.object(node, '_best_pseudo_costs_index') as bpci, patch.object(node, '_base_branch') as bb:
        cpc.return_value, bpci.return_value = [], 2
        node.branch(rtn['pseudo_costs'])
        assert cpc.called and bpci.called and bb.call_args.args == (2,)
    node = make_node()
    rtn = node.bound({}, gomory_cuts=False)
    kids = node.branch(rtn['pseudo_costs'])
    assert all(isinstance(kids[d], PseudoCostBranchNode) for d in ['right', 'left'])
    assert kids['left']._b_idx == 1  # scores: idx 1 -> min(0, .25) = 0, idx 2 -> 0: earliest wins


def test_best_pseudo_cost_index(engine):
    pc = {1: {'right': {'cost': 1, 'times': 1}, 'left': {'cost': 1, 'times': 1}},
          2: {'right': {'cost': 1, 'times': 1}, 'left': {'cost': 1, 'times': 1}}}
    node = make_node()
    node.solution = [0, 1.25, 2.5]
    assert node._best_pseudo_costs_index(pc) == 2
    pc[1] = {'right': {'cost': 10, 'times': 1}, 'left': {'cost': 1, 'times': 1}}
    assert node._best_pseudo_costs_index(pc) == 2
    pc[1] = {'right': {'cost': 10, 'times': 1}, 'left': {'cost': 10, 'times': 1}}
    assert node._best_pseudo_costs_index(pc) == 1


def test_check_pseudo_costs(engine):
    node = make_node()
    node._integer_indices = [0, 1]
    good = {'cost': 1, 'times': 1}
    assert node._check_pseudo_costs({2: {}}) == ['index 2 not integer index']
    assert node._check_pseudo_costs({1: {'left': good}}) == ['index 1 missing direction right']
    assert node._check_pseudo_costs({1: {'left': good, 'right': {'times': 1}}}) == \
        ['index 1 direction right missing cost']
    assert node._check_pseudo_costs({1: {'left': good, 'right': {'cost': -1, 'times': 1}}}) == \
        ['index 1 direction right cost must be nonnegative number']
    assert node._check_pseudo_costs({1: {'left': good, 'right': {'cost': 1}}}) == \
        ['index 1 direction right missing times']
    assert node._check_pseudo_costs({1: {'left': good, 'right': {'cost': 1, 'times': 1.5}}}) == \
        ['index 1 direction right times must be nonnegative int']
    assert node._check_pseudo_costs({1: {'left': good, 'right': good}}) == []


def test_pseudo_costs_flow_through_branch_and_bound(engine):
    bb = BranchAndBound(model('small_branch'), Node=PseudoCostBranchNode, pseudo_costs={},
                        gomory_cuts=False)
    bb.solve()
    assert bb.status == 'optimal' and bb.objective_value == -2
    p = bb._kwargs['pseudo_costs']
    assert set(p) <= {0, 1, 2} and all(set(e) == {'left', 'right'} for e in p.values())
    assert sum(r['times'] for e in p.values() for r in e.values()) <= 2 * (bb.evaluated_nodes + 3)
