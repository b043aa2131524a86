"""DisjunctiveCutBoundNode: BaseNode plus cuts from a cut-generating LP (mirror of the reference's
simple_mip_solver/nodes/bound/disjunctive_cut.py: same keyword arguments, counters, cut names
`cut_cglp_<node>_<round>` and assert messages).  SURVEY.md 8f rank 4; outside the node hot path.

Keyword arguments it adds to BranchAndBound(...):
    max_cglp_calls (int)                cut rounds in which the CGLP may contribute (default: all)
    warm_start_cglp (bool)              start the CGLP from its previous optimal basis (default True)
    cglp_cumulative_constraints (bool)  children rebuild the CGLP with this node's rows
    cglp_cumulative_bounds (bool)       ... and / or with this node's variable bounds
"""
import re

import numpy as np

from simple_mip_solver_amd.lp import CyLPArray
from simple_mip_solver_amd.nodes.base_node import BaseNode
from simple_mip_solver_amd.utils import tolerance as tol
from simple_mip_solver_amd.utils.cut_generating_lp import CutGeneratingLP
from simple_mip_solver_amd.utils.floating_point import numerically_safe_cut


class DisjunctiveCutBoundNode(BaseNode):

    def __init__(self, cglp=None, prev_cglp_basis=None, force_create_cglp=False, *args, **kwargs):
        """cglp: the CutGeneratingLP to cut with (children inherit or rebuild it);
        prev_cglp_basis: its starting basis; force_create_cglp: keep cutting with it even when an
        earlier round or node got nothing out of it."""
        super().__init__(*args, **kwargs)
        assert isinstance(force_create_cglp, bool), 'force_create_cglp is bool'
        if cglp is not None:
            assert isinstance(cglp, CutGeneratingLP), 'cglp must be CutGeneratingLP instance'
        else:
            assert not force_create_cglp, 'cannot force creation of CGLP that does not exist'
        self.cglp = cglp
        self.prev_cglp_basis = prev_cglp_basis
        self.current_node_added_cglp = force_create_cglp
        # did this node's parent / the previous round get a cut out of the CGLP
        self.previous_cglp_added = self.cglp is not None
        self.cglp_name_pattern = re.compile('^cut_cglp_')
        self.current_cglp_name_pattern = re.compile(f'^cut_cglp_{self.idx}_')
        self.sharable_cuts = {}
        self.number_cglp_created = 0
        self.number_cglp_added = 0
        self.number_cglp_removed = 0
        self.force_create_cglp = force_create_cglp

    def bound(self, total_number_cglp_created=0, total_number_cglp_added=0,
              total_number_cglp_removed=0, **kwargs):
        """BaseNode.bound plus the running totals of disjunctive cuts and, under 'cuts', the ones
        that are valid for every other node."""
        assert isinstance(total_number_cglp_added, int) and total_number_cglp_added >= 0, \
            "total_number_cglp_added is nonnegative integer"
        assert isinstance(total_number_cglp_created, int) and total_number_cglp_created >= 0, \
            "total_number_gmic_created is nonnegative integer"
        assert isinstance(total_number_cglp_removed, int) and total_number_cglp_removed >= 0, \
            "total_number_cglp_removed is nonnegative integer"
        rtn = super().bound(**kwargs)
        rtn['total_number_cglp_created'] = total_number_cglp_created + self.number_cglp_created
        rtn['total_number_cglp_added'] = total_number_cglp_added + self.number_cglp_added
        rtn['total_number_cglp_removed'] = total_number_cglp_removed + self.number_cglp_removed
        if self.sharable_cuts:
            rtn['cuts'] = self.sharable_cuts
        return rtn

    def _remove_slack_cuts(self, **kwargs):
        removed = super()._remove_slack_cuts(**kwargs)
        self.number_cglp_removed += sum(bool(self.cglp_name_pattern.match(name)) for name in removed)
        return removed

    def _generate_cuts(self, max_cglp_calls=None, min_cglp_norm=tol.min_cglp_norm, **kwargs):
        """BaseNode's candidates plus, while the CGLP keeps paying off, the deepest disjunctive
        cut at the current LP solution."""
        if max_cglp_calls is not None:
            assert isinstance(max_cglp_calls, int) and max_cglp_calls >= 0, \
                'max_cglp_calls is a nonnegative integer'
        assert isinstance(min_cglp_norm, (float, int)) and min_cglp_norm > 0, \
            'min_cglp_norm is a positive number'
        limit = float('inf') if max_cglp_calls is None else max_cglp_calls
        pool = super()._generate_cuts(**kwargs)
        if self.previous_cglp_added and self.cut_generation_iterations <= limit:
            pi, pi0 = self.cglp.solve(x_star=CyLPArray(self.solution),
                                      starting_basis=self._get_cglp_starting_basis(**kwargs))
            if pi is not None and pi0 is not None and np.linalg.norm(pi) > min_cglp_norm:
                name = f'cut_cglp_{self.idx}_{self.cut_generation_iterations}'
                pool[name] = numerically_safe_cut(pi=pi, pi0=pi0, estimate='over')
                self.number_cglp_created += 1
        return pool

    def _get_cglp_starting_basis(self, warm_start_cglp=True, **kwargs):
        """None keeps the CGLP at the basis its last solve ended in."""
        assert isinstance(warm_start_cglp, bool), 'warm_start_cglp is boolean'
        if not warm_start_cglp:
            return (np.array([3] * self.cglp.lp.nVariables, dtype=np.int32),
                    np.array([1] * self.cglp.lp.nConstraints, dtype=np.int32))
        if self.cut_generation_iterations == 1:
            return self.prev_cglp_basis
        return None

    def _select_cuts(self, cglp_cumulative_constraints=True, cglp_cumulative_bounds=True, **kwargs):
        """BaseNode's selection; notes whether this round's disjunctive cut made it in, and shares
        it with the other nodes when it was built from the original rows and bounds."""
        assert isinstance(cglp_cumulative_constraints, bool), 'cglp_cumulative_constraints is bool'
        assert isinstance(cglp_cumulative_bounds, bool), 'cglp_cumulative_bounds is bool'
        self.previous_cglp_added = self.force_create_cglp
        added = super()._select_cuts(**kwargs)
        for name, (pi, pi0) in added.items():
            if self.cglp_name_pattern.match(name):
                self.number_cglp_added += 1
                if self.current_cglp_name_pattern.match(name):
                    self.current_node_added_cglp = True
                    self.previous_cglp_added = True
                    if not cglp_cumulative_bounds and not cglp_cumulative_constraints:
                        self.sharable_cuts[name] = (pi, pi0)
        return added

    def branch(self, cglp_cumulative_constraints=False, cglp_cumulative_bounds=False, cglp=None,
               **kwargs):
        """Children get no CGLP if this node never used its cut, a CGLP rebuilt on this node's
        rows / bounds in the cumulative modes, else the same CGLP and its basis."""
        assert isinstance(cglp_cumulative_constraints, bool), 'cglp_cumulative_constraints is bool'
        assert isinstance(cglp_cumulative_bounds, bool), 'cglp_cumulative_bounds is bool'
        if self.cglp is None or not self.current_node_added_cglp:
            return super().branch(force_create_cglp=self.force_create_cglp, **kwargs)
        if cglp_cumulative_constraints or cglp_cumulative_bounds:
            A = b = var_lb = var_ub = None
            if cglp_cumulative_constraints:
                A = self.lp.coefMatrix.copy()
                b = CyLPArray(np.array(self.lp.constraintsLower).copy())
            if cglp_cumulative_bounds:
                var_lb = CyLPArray(np.array(self.lp.variablesLower).copy())
                var_ub = CyLPArray(np.array(self.lp.variablesUpper).copy())
            child_cglp = CutGeneratingLP(bb=self.cglp.bb, root_id=self.cglp.root_id, A=A, b=b,
                                         var_lb=var_lb, var_ub=var_ub)
            return super().branch(cglp=child_cglp, force_create_cglp=self.force_create_cglp, **kwargs)
        return super().branch(cglp=self.cglp, prev_cglp_basis=self.cglp.lp.getBasisStatus(),
                              force_create_cglp=self.force_create_cglp, **kwargs)
