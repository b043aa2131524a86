"""DisjunctiveCutBoundNode against the reference's unit tests
(test_simple_mip_solver/test_nodes/test_bound/test_disjunctive_cut.py)."""
from unittest.mock import patch

import numpy as np
import pytest
from scipy.sparse import csc_matrix

from simple_mip_solver_amd import BaseNode, BranchAndBound, CyLPArray, DisjunctiveCutBoundNode
from simple_mip_solver_amd.utils.cut_generating_lp import CutGeneratingLP
from tests.support.example_models import std_model


def _setup(gomory_cuts=True):
    m = std_model('cut1')
    bb = BranchAndBound(m, gomory_cuts=gomory_cuts)
    bb.solve()
    return m, bb, CutGeneratingLP(bb, bb.root_node.idx)


def test_init_fails_asserts(engine):
    m, bb, cglp = _setup()
    with pytest.raises(AssertionError, match='cglp must be CutGeneratingLP instance'):
        DisjunctiveCutBoundNode(cglp=cglp.lp, lp=bb.root_node.lp, integer_indices=m.integerIndices)
    with pytest.raises(AssertionError, match='is bool'):
        DisjunctiveCutBoundNode(force_create_cglp=1, lp=bb.root_node.lp, integer_indices=m.integerIndices)
    with pytest.raises(AssertionError, match='cannot force'):
        DisjunctiveCutBoundNode(force_create_cglp=True, lp=bb.root_node.lp, integer_indices=m.integerIndices)


def test_init(engine):
    m, bb, cglp = _setup()
    n = DisjunctiveCutBoundNode(lp=bb.root_node.lp, integer_indices=m.integerIndices, cglp=cglp)
    assert cglp is n.cglp and n.prev_cglp_basis is None and not n.current_node_added_cglp
    assert n.previous_cglp_added and n.cglp_name_pattern and not n.sharable_cuts
    assert not n.number_cglp_created and not n.number_cglp_added and not n.number_cglp_removed
    assert not n.force_create_cglp
    n = DisjunctiveCutBoundNode(lp=bb.root_node.lp, integer_indices=m.integerIndices, cglp=cglp,
                                force_create_cglp=True)
    assert n.current_node_added_cglp and n.force_create_cglp


def test_bound(engine):
    m, bb, cglp = _setup()
    node = DisjunctiveCutBoundNode(lp=m.lp, integer_indices=m.integerIndices, cglp=cglp)
    for key in ('created', 'added', 'removed'):
        with pytest.raises(AssertionError, match='is nonnegative integer'):
            node.bound(**{f'total_number_cglp_{key}': -1})
    with patch.object(BaseNode, 'bound') as b:
        b.return_value = {}
        rtn = node.bound()
        assert b.called and not rtn['total_number_cglp_created'] and not rtn['total_number_cglp_added'] \
            and not rtn['total_number_cglp_removed']
        for key in rtn:
            rtn[key] += 1
        node.sharable_cuts = {'cut_cglp_1_1': 'some cut'}
        rtn = node.bound(**rtn)
        assert rtn['cuts'] == node.sharable_cuts
        assert rtn['total_number_cglp_created'] and rtn['total_number_cglp_added'] and rtn['total_number_cglp_removed']


def test_remove_slack_cuts(engine):
    m, bb, cglp = _setup()
    node = DisjunctiveCutBoundNode(lp=m.lp, integer_indices=m.integerIndices, cglp=cglp)
    with patch.object(BaseNode, '_remove_slack_cuts') as rsc:
        rsc.return_value = ['cut_gomory_0_2_0', 'cut_cglp_0_0']
        assert node._remove_slack_cuts() == ['cut_gomory_0_2_0', 'cut_cglp_0_0']
        assert rsc.called and node.number_cglp_removed == 1


def test_generate_cuts(engine):
    m, bb, cglp = _setup(gomory_cuts=False)
    node = DisjunctiveCutBoundNode(lp=m.lp, integer_indices=m.integerIndices, cglp=cglp, idx=0)
    with pytest.raises(AssertionError, match='max_cglp_calls is a nonnegative integer'):
        node._generate_cuts(max_cglp_calls=-1)
    target = 'simple_mip_solver_amd.nodes.bound.disjunctive_cut.numerically_safe_cut'
    with patch.object(BaseNode, '_generate_cuts') as gc, patch.object(node.cglp, 'solve') as s, \
            patch(target) as nsc, patch.object(node, '_get_cglp_starting_basis') as gcsb:
        gc.return_value = {}
        s.side_effect = [(None, None), (CyLPArray([1e-12, 1e-8]), 1e-10), (CyLPArray([0, 1]), 1)]
        nsc.return_value = (CyLPArray([0, 1]), 1)
        gcsb.return_value = None
        for calls, made in [(1, 0), (2, 0), (3, 1)]:   # no cut, a negligible cut, a good cut
            pool = node._generate_cuts()
            assert gc.call_count == calls and s.call_count == calls and gcsb.call_count == calls
            assert s.call_args.kwargs['starting_basis'] is None
            assert nsc.call_count == made and node.number_cglp_created == made
            assert set(pool) == ({'cut_cglp_0_0'} if made else set())
    with patch.object(BaseNode, '_generate_cuts') as gc, patch.object(node.cglp, 'solve') as s, \
            patch(target) as nsc, patch.object(node, '_get_cglp_starting_basis') as gcsb:
        gc.return_value = {}
        node.previous_cglp_added = False          # the previous round got nothing out of the CGLP
        assert not node._generate_cuts() and gc.call_count == 1
        assert not s.called and not gcsb.called and not nsc.called and node.number_cglp_created == 1
        node.cut_generation_iterations += 1       # past the allowed number of CGLP rounds
        node.previous_cglp_added = True
        assert not node._generate_cuts(max_cglp_calls=0)
        assert not s.called and not gcsb.called and not nsc.called and node.number_cglp_created == 1


def test_generate_cuts_gets_warm_start_right(engine):
    m, bb, cglp = _setup(gomory_cuts=False)
    node = DisjunctiveCutBoundNode(lp=m.lp, cglp=cglp, idx=0, integer_indices=m.integerIndices)
    with patch.object(BaseNode, '_generate_cuts') as gc, patch.object(node.cglp, 'solve') as s, \
            patch('simple_mip_solver_amd.nodes.bound.disjunctive_cut.numerically_safe_cut') as nsc, \
            patch.object(node, '_get_cglp_starting_basis') as gcsb:
        gc.return_value = {}
        s.return_value = (CyLPArray([0, 1]), 1)
        nsc.return_value = (CyLPArray([0, 1]), 1)
        gcsb.return_value = None
        node._generate_cuts(cut_generating_lp=True, warm_start_cglp=True)
        assert gcsb.call_args.kwargs['warm_start_cglp']
        node._generate_cuts(cut_generating_lp=True, warm_start_cglp=False)
        assert not gcsb.call_args.kwargs['warm_start_cglp']


def test_get_cglp_starting_basis(engine):
    m, bb, cglp = _setup()
    node = DisjunctiveCutBoundNode(lp=m.lp, integer_indices=m.integerIndices, cglp=cglp,
                                   prev_cglp_basis=(np.array([5]), np.array([5])))
    with pytest.raises(AssertionError, match='warm_start_cglp is boolean'):
        node._get_cglp_starting_basis(warm_start_cglp=None)
    basis = node._get_cglp_starting_basis(warm_start_cglp=False)
    assert (basis[0] == [3] * cglp.lp.nVariables).all() and (basis[1] == [1] * cglp.lp.nConstraints).all()
    assert not node._get_cglp_starting_basis(warm_start_cglp=True)   # first round: cold
    node.cut_generation_iterations = 1
    basis = node._get_cglp_starting_basis(warm_start_cglp=True)
    assert basis[0] == 5 and basis[1] == 5


def test_select_cuts(engine):
    m, bb, cglp = _setup(gomory_cuts=False)
    node = DisjunctiveCutBoundNode(lp=m.lp, integer_indices=m.integerIndices, cglp=cglp, idx=0)
    with pytest.raises(AssertionError, match='cglp_cumulative_constraints is bool'):
        node._select_cuts(cglp_cumulative_constraints=0)
    with pytest.raises(AssertionError, match='cglp_cumulative_bounds is bool'):
        node._select_cuts(cglp_cumulative_bounds=0)
    with patch.object(BaseNode, '_select_cuts') as sc:
        sc.return_value = {'cut_gomory_0_0_0': (CyLPArray([1, 0]), 1), 'cut_cglp_1_0': (CyLPArray([0, 1]), 1)}
        added = node._select_cuts(cglp_cumulative_constraints=False, cglp_cumulative_bounds=False)
        assert sc.call_count == 1 and not node.current_node_added_cglp and not node.previous_cglp_added
        assert not node.sharable_cuts and node.number_cglp_added == 1
        assert set(added) == {'cut_gomory_0_0_0', 'cut_cglp_1_0'}
        sc.return_value = {'cut_cglp_0_0': (CyLPArray([0, 1]), 1), 'cut_gomory_0_0_0': (CyLPArray([1, 0]), 1)}
        for count, (cc, cb) in enumerate([(True, True), (False, True), (True, False), (False, False)], 2):
            added = node._select_cuts(cglp_cumulative_constraints=cc, cglp_cumulative_bounds=cb)
            assert sc.call_count == count and node.current_node_added_cglp and node.previous_cglp_added
            assert node.number_cglp_added == count and set(added) == {'cut_cglp_0_0', 'cut_gomory_0_0_0'}
            assert bool(node.sharable_cuts) == (not cc and not cb)   # shared only from the original rows/bounds
    node = DisjunctiveCutBoundNode(lp=m.lp, integer_indices=m.integerIndices, cglp=cglp, idx=0,
                                   force_create_cglp=True)
    with patch.object(BaseNode, '_select_cuts') as sc:
        sc.return_value = {'cut_gomory_0_0_0': (CyLPArray([1, 0]), 1), 'cut_cglp_1_0': (CyLPArray([0, 1]), 1)}
        node._select_cuts(cglp_cumulative_constraints=False, cglp_cumulative_bounds=False)
        assert node.current_node_added_cglp and node.previous_cglp_added and not node.sharable_cuts


def test_branch(engine):
    m, bb, cglp = _setup(gomory_cuts=False)
    n = DisjunctiveCutBoundNode(lp=std_model('cut1').lp, integer_indices=m.integerIndices)
    n.bound()
    with pytest.raises(AssertionError, match='cglp_cumulative_constraints is bool'):
        n.branch(cglp_cumulative_constraints=0)
    with pytest.raises(AssertionError, match='cglp_cumulative_bounds is bool'):
        n.branch(cglp_cumulative_bounds=0)
    for node in (n, DisjunctiveCutBoundNode(lp=std_model('cut1').lp, integer_indices=m.integerIndices, cglp=cglp)):
        with patch.object(BaseNode, 'branch') as bm:   # no CGLP / CGLP cut never added: plain branch
            bm.return_value = 'rtn'
            assert node.branch() == 'rtn' and bm.called and not bm.call_args.args
            assert not bm.call_args.kwargs['force_create_cglp'] and 'cglp' not in bm.call_args.kwargs

    # cumulative modes: the children get a CGLP rebuilt on this node's rows and bounds
    m, bb, cglp = _setup()
    n = DisjunctiveCutBoundNode(lp=std_model('cut1').lp, integer_indices=m.integerIndices, cglp=cglp,
                                force_create_cglp=True)
    n._bound_lp()
    n.current_node_added_cglp = True
    with patch('simple_mip_solver_amd.nodes.bound.disjunctive_cut.CutGeneratingLP', spec=CutGeneratingLP) as cm, \
            patch.object(BaseNode, 'branch') as bm:
        bm.return_value = 'rtn'
        assert n.branch(cglp_cumulative_constraints=True, cglp_cumulative_bounds=True) == 'rtn'
        kwargs = cm.call_args.kwargs
        assert isinstance(kwargs['A'], csc_matrix) and (kwargs['A'] != n.lp.coefMatrix).nnz == 0
        assert isinstance(kwargs['b'], CyLPArray) and (kwargs['b'] == n.lp.constraintsLower).all()
        assert isinstance(kwargs['var_lb'], CyLPArray) and (kwargs['var_lb'] == n.lp.variablesLower).all()
        assert isinstance(kwargs['var_ub'], CyLPArray) and (kwargs['var_ub'] == n.lp.variablesUpper).all()
        assert isinstance(bm.call_args.kwargs['cglp'], CutGeneratingLP) and bm.call_args.kwargs['force_create_cglp']

    # static mode: the same CGLP and its basis travel to the children
    n = DisjunctiveCutBoundNode(lp=std_model('cut1').lp, integer_indices=m.integerIndices, cglp=cglp)
    n._bound_lp()
    n.current_node_added_cglp = True
    with patch.object(BaseNode, 'branch') as bm:
        bm.return_value = 'rtn'
        n.branch()
        kwargs = bm.call_args.kwargs
        assert kwargs['cglp'] is cglp
        assert (kwargs['prev_cglp_basis'][0] == cglp.lp.getBasisStatus()[0]).all()
        assert (kwargs['prev_cglp_basis'][1] == cglp.lp.getBasisStatus()[1]).all()
