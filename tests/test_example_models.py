"""The reference's integration suite (helpers.TestModels.base_test_models, helpers.py:30-73): every
one of its 64 random example .mps instances is solved by BranchAndBound and compared with an
independent optimum -- committed HiGHS values (tests/golden/example_models_optima.json) instead of
Gurobi at test time.  The reference's bar is abs_tol=.01; BASELINE asks 1e-6 (all optima are
integers).  Parity of these objectives is not pinned by the reference itself ("parity unpinned")."""
import json
from math import isclose
import os

import pytest

from simple_mip_solver_amd import (BaseNode, BranchAndBound, DepthFirstSearchNode, MILPInstance,
                                   PseudoCostBranchNode, PseudoCostBranchDepthFirstSearchNode)

HERE = os.path.dirname(__file__)
TABLE = json.load(open(os.path.join(HERE, 'golden', 'example_models_optima.json')))['models']


def check_pseudo_costs(bb):  # helpers.py:54-60
    if bb.evaluated_nodes >= 4 and isinstance(bb.root_node, PseudoCostBranchNode):
        p = bb._kwargs['pseudo_costs']
        assert len(p) <= bb.root_node.lp.nVariables
        assert sum(sum(b['times'] for b in e.values()) for e in p.values()) <= \
            2 * (bb.evaluated_nodes + bb.root_node.lp.nVariables)


def check_gmics(bb):  # helpers.py:62-73
    if bb.evaluated_nodes >= 2 and bb._kwargs.get('gomory_cuts', True) and \
            bb._kwargs['total_cut_generation_iterations']:
        k = bb._kwargs
        assert k['total_number_gmic_added'] <= k['total_number_gmic_created']
        # (the reference also compares the *iteration* counts, "just rough values": a round can
        # add cuts left in the pool by an earlier round while creating none, so that one is not
        # an invariant of the algorithm and is not asserted here)


@pytest.mark.parametrize('Node,kwargs', [
    (BaseNode, {'gomory_cuts': False}), (PseudoCostBranchNode, {'gomory_cuts': False}),
    (DepthFirstSearchNode, {'gomory_cuts': False}),
    (PseudoCostBranchDepthFirstSearchNode, {'gomory_cuts': False})])
def test_models_without_cuts(engine, Node, kwargs):
    for f, rec in sorted(TABLE.items()):
        m = MILPInstance(file_name=os.path.join(HERE, 'golden', 'example_models', f))
        bb = BranchAndBound(m, Node, pseudo_costs={}, **kwargs)
        bb.solve()
        assert bb.status == 'optimal', f
        assert isclose(bb.objective_value, rec['milp_opt'], abs_tol=1e-6), \
            f'{f}: {bb.objective_value} vs HiGHS {rec["milp_opt"]}'
        assert isclose(bb.root_node.objective_value, rec['lp_opt'], rel_tol=1e-6, abs_tol=1e-6), f
        check_pseudo_costs(bb)


@pytest.mark.parametrize('Node', [BaseNode, PseudoCostBranchNode])
def test_models_with_gomory_cuts(engine, Node):
    """Reference defaults (gomory_cuts=True, tolerance.py values).  Rounded cuts move optima by
    up to ~1e-5 (e.g. cut2: -36.00001), hence the reference's own abs_tol=.01 (helpers.py:45)."""
    for f, rec in sorted(TABLE.items()):
        m = MILPInstance(file_name=os.path.join(HERE, 'golden', 'example_models', f))
        bb = BranchAndBound(m, Node, pseudo_costs={})
        bb.solve()
        assert bb.status == 'optimal', f
        assert isclose(bb.objective_value, rec['milp_opt'], abs_tol=.01), \
            f'{f}: {bb.objective_value} vs HiGHS {rec["milp_opt"]}'
        check_pseudo_costs(bb)
        check_gmics(bb)


@pytest.mark.parametrize('kwargs', [
    dict(cglp_cumulative_constraints=cc, cglp_cumulative_bounds=cb, gomory_cuts=gc, warm_start_cglp=ws,
         max_cglp_calls=mc)
    for cc, cb, gc, ws, mc in [(True, True, True, True, 1), (False, False, False, True, None),
                               (True, False, False, False, 1), (False, True, True, True, None),
                               (False, False, True, False, None)]])
def test_models_with_disjunctive_cuts(engine, kwargs):
    """helpers.TestModels.disjunctive_cut_test_models (helpers.py:75-131) on a fixed third of the
    models and five of its keyword combinations: a CGLP from an 8-node tree, then the full solve
    with DisjunctiveCutBoundPseudoCostBranchNode; optimum within the reference's abs_tol=.01."""
    from simple_mip_solver_amd import DisjunctiveCutBoundPseudoCostBranchNode
    from simple_mip_solver_amd.utils.cut_generating_lp import CutGeneratingLP
    for k, (f, rec) in enumerate(sorted(TABLE.items())):
        if k % 3 or k == 3:   # (the reference skips its model 3 too: a bad GMIC)
            continue
        m = MILPInstance(file_name=os.path.join(HERE, 'golden', 'example_models', f))
        cglp_bb = BranchAndBound(m, node_limit=8, gomory_cuts=kwargs['gomory_cuts'])
        cglp_bb.solve()
        cglp = CutGeneratingLP(cglp_bb, cglp_bb.root_node.idx)
        bb = BranchAndBound(m, DisjunctiveCutBoundPseudoCostBranchNode, cglp=cglp, pseudo_costs={}, **kwargs)
        bb.solve()
        assert bb.status == 'optimal', f
        assert isclose(bb.objective_value, rec['milp_opt'], abs_tol=.01), \
            f'{f} {kwargs}: {bb.objective_value} vs HiGHS {rec["milp_opt"]}'
        check_pseudo_costs(bb)
        check_gmics(bb)
