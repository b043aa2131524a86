"""PseudoCostBranchNode: pseudo-cost branching with strong-branching initialisation.

Mirror of simple_mip_solver/nodes/branch/pseudo_cost.py:12-163 (same method names, keyword
protocol, table layout `{idx: {'left'|'right': {'cost': float, 'times': int}}}` and messages).
All strong-branching probes of one node -- two truncated dual simplex solves per index that has no
table entry yet (reference :57-62 runs them one LP at a time) -- go to the MI355X engine as ONE
batch; the table is then updated in the reference's order, so the result is identical.
"""
from math import ceil, floor

from simple_mip_solver_amd.nodes.base_node import BaseNode
from simple_mip_solver_amd.utils.tolerance import variable_epsilon


class PseudoCostBranchNode(BaseNode):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.branch_method = 'pseudo cost'
        self.pseudo_costs = None
        self.strong_branch_iters = None

    def bound(self, pseudo_costs, strong_branch_iters=5, **kwargs):
        """BaseNode.bound, then refresh the shared pseudo-cost table if the LP is feasible; the
        table travels back to BranchAndBound under the key 'pseudo_costs' (reference :22-44)."""
        problems = self._check_pseudo_costs(pseudo_costs)
        assert not problems, f'pseudo cost dict has following errors: {problems}'
        self.pseudo_costs = pseudo_costs
        self.strong_branch_iters = strong_branch_iters
        rtn = super().bound(**kwargs)
        if self.lp_feasible:
            self._update_pseudo_costs()
        rtn['pseudo_costs'] = self.pseudo_costs
        return rtn

    def _update_pseudo_costs(self):
        """Strong-branch every fractional integer index without an entry, then account for the
        branch that created this node unless it was just initialised (reference :46-66)."""
        fresh = [i for i in self._fractional_indices() if i not in self.pseudo_costs]
        stock = type(self)._strong_branch is BaseNode._strong_branch and \
            '_strong_branch' not in self.__dict__
        if fresh and stock:
            # one engine batch for all 2 * len(fresh) truncated solves
            assert isinstance(self.strong_branch_iters, int) and self.strong_branch_iters > 0, \
                'iterations must be positive integer'
            probes = []
            for i in fresh:
                pair = self._base_branch(i)
                probes.append((pair['left'], pair['right']))
            self._solve_probes([n for pair in probes for n in pair], self.strong_branch_iters)
            for left, right in probes:
                self._calculate_costs(left)
                self._calculate_costs(right)
        else:
            # a subclass (or a test double) supplies its own _strong_branch: call it per index,
            # exactly as the reference does
            for i in fresh:
                for probe in self._strong_branch(i, self.strong_branch_iters).values():
                    self._calculate_costs(probe)
        if self._b_idx is not None and self._b_idx not in fresh:
            self._calculate_costs(self)

    def _calculate_costs(self, node):
        """Fold node's objective gain per unit of variable change into the running mean of
        (node._b_idx, node._b_dir); an infeasible probe only counts a visit (reference :68-100)."""
        entry = self.pseudo_costs.setdefault(node._b_idx, {}) \
            .setdefault(node._b_dir, {'cost': 0, 'times': 0})
        if node.lp.getStatusCode() in (0, 3):  # optimal or stopped on the iteration limit
            gain = max(node.lp.objectiveValue - node.dual_bound, 0)
            if node._b_dir == 'left':
                moved = node._b_val - node.lp.variablesUpper[node._b_idx]
            else:
                moved = node.lp.variablesLower[node._b_idx] - node._b_val
            entry['cost'] = (entry['cost'] * entry['times'] + gain / moved) / (entry['times'] + 1)
        entry['times'] += 1

    def branch(self, pseudo_costs, **kwargs):
        """Branch on the index with the best pseudo-cost score (reference :102-116)."""
        assert not self.mip_feasible, 'must have fractional value to branch'
        problems = self._check_pseudo_costs(pseudo_costs)
        assert not problems, f'pseudo cost dict has following errors: {problems}'
        return self._base_branch(self._best_pseudo_costs_index(pseudo_costs), **kwargs)

    def _best_pseudo_costs_index(self, pseudo_costs):
        """argmax over fractional integer i of min(cost_up * (ceil - x), cost_down * (x - floor));
        ties go to the earliest index in integer_indices order (reference :118-133)."""
        best, best_score = None, None
        for i in self._fractional_indices():
            x = self.solution[i]
            score = min(pseudo_costs[i]['right']['cost'] * (ceil(x) - x),
                        pseudo_costs[i]['left']['cost'] * (x - floor(x)))
            if best is None or score > best_score:
                best, best_score = i, score
        if best is None:
            raise IndexError('list index out of range')  # what the reference's sorted(...)[0] raises
        return best

    def _check_pseudo_costs(self, pseudo_costs):
        """List of structural problems with a pseudo-cost table (reference :135-163)."""
        problems = []
        for idx in pseudo_costs:
            if idx not in self._integer_indices:
                problems.append(f'index {idx} not integer index')
                continue
            for direction in ('right', 'left'):
                if direction not in pseudo_costs[idx]:
                    problems.append(f'index {idx} missing direction {direction}')
                    continue
                record = pseudo_costs[idx][direction]
                if 'cost' not in record:
                    problems.append(f'index {idx} direction {direction} missing cost')
                elif not (isinstance(record['cost'], (int, float)) and
                          record['cost'] + variable_epsilon >= 0):
                    problems.append(f'index {idx} direction {direction} cost must'
                                    ' be nonnegative number')
                if 'times' not in record:
                    problems.append(f'index {idx} direction {direction} missing times')
                elif not (isinstance(record['times'], int) and record['times'] >= 0):
                    problems.append(f'index {idx} direction {direction} times must'
                                    ' be nonnegative int')
        return problems
