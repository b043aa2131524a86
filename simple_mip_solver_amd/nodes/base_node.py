"""BaseNode: one branch-and-bound subproblem, bounded on the MI355X engine.

Mirror of the reference's plugin class simple_mip_solver/nodes/base_node.py:23-710 -- same
constructor, attributes, method names, keyword protocol, returned dicts and assertion messages
(those are what BranchAndBound and user subclasses program against) -- with the Clp calls replaced
by the engine behind `DenseLP` (simple_mip_solver_amd/lp.py) and child creation done by sharing
the parent's rows instead of rebuilding a model row by row (reference :592-608).

Best-first search (`__lt__` on dual_bound) and most-fractional branching, as in the reference.
"""
from math import acos, ceil, degrees, floor
import re
import time

import numpy as np

from simple_mip_solver_amd.lp import CyLPArray, DenseLP
from simple_mip_solver_amd.utils import tolerance as tol
from simple_mip_solver_amd.utils.floating_point import numerically_safe_cut

_COUNTER_KEYS = ('total_cut_generation_iterations', 'total_iterations_gmic_created',
                 'total_number_gmic_created', 'total_iterations_gmic_added',
                 'total_number_gmic_added', 'total_iterations_gmic_removed',
                 'total_number_gmic_removed')


def _is_int_type(t):
    return t is int or issubclass(t, int)


def _is_nonneg_int(v):
    return isinstance(v, int) and v >= 0


class BaseNode:
    """Node with LP-relaxation bounding (+ Gomory rounds), most-fractional branching and
    best-first ordering.  Subclasses override bound / branch / the comparators."""

    def __init__(self, lp, integer_indices, idx=None, dual_bound=-float('inf'), b_idx=None,
                 b_dir=None, b_val=None, depth=0, ancestors=None, *args, **kwargs):
        # argument checks; messages as at reference base_node.py:49-71
        assert isinstance(lp, DenseLP), 'lp must be CyClpSimplex instance'
        # (same test as `all(isinstance(i, int) and 0 <= i < nVariables ...)`, at C speed: it runs
        # for every node and n is in the hundreds)
        assert all(map(_is_int_type, set(map(type, integer_indices)))) and \
            (len(integer_indices) == 0 or 0 <= min(integer_indices) <= max(integer_indices) < lp.nVariables), \
            'indices must match variables'
        assert idx is None or isinstance(idx, int), 'node idx must be integer if provided'
        assert len(set(integer_indices)) == len(integer_indices), 'indices must be distinct'
        assert isinstance(dual_bound, (float, int)), 'dual bound must be a float or an int'
        assert (b_dir is None) == (b_idx is None) == (b_val is None), \
            'none are none or all are none'
        assert b_idx is None or b_idx in integer_indices, \
            'branch index corresponds to integer variable if it exists'
        assert b_dir is None or b_dir in ('right', 'left'), 'we can only branch right or left'
        if b_val is not None:
            gap = b_val - lp.variablesUpper[b_idx] if b_dir == 'left' else \
                lp.variablesLower[b_idx] - b_val
            assert 0 < gap < 1, 'branch val should be within 1 of both bounds'
        assert isinstance(depth, int) and depth >= 0, 'depth is a positive integer'
        if ancestors is not None:
            assert isinstance(ancestors, tuple), 'ancestors must be a tuple if provided'
            assert idx not in ancestors, 'idx cannot be an ancestor of itself'

        lp.logLevel = 0
        self.lp = lp
        self._integer_indices = integer_indices
        self._int_idx = np.asarray(integer_indices, dtype=np.int64)
        self.idx = idx
        self.dual_bound = dual_bound
        self.objective_value = None
        self.solution = None
        self.lp_feasible = None
        self.unbounded = None
        self.mip_feasible = None
        self._b_idx, self._b_dir, self._b_val = b_idx, b_dir, b_val
        self.depth = depth
        self.search_method = 'best first'
        self.branch_method = 'most fractional'
        self.is_leaf = True
        own = (idx,) if idx is not None else ()
        self.lineage = ((ancestors or ()) + own) or None
        self.children = None

        # cut bookkeeping (reference :93-108)
        self.cut_generation_iterations = 0
        self.cut_generation_stalled = False
        self.cut_generation_terminator = None
        self.cut_generation_dual_bound = {}
        self.tracked_cut_generation_iterations = 0
        self.cut_name_pattern = re.compile('^cut_')
        self.gmic_name_pattern = re.compile('^cut_gomory_')
        for op in ('created', 'added', 'removed'):
            setattr(self, f'iterations_gmic_{op}', 0)
            setattr(self, f'number_gmic_{op}', 0)
        self._cut_pool = {}
        self._engine_rounded = {}
        first_block = lp.constraints[0]
        self.max_term = np.max(np.abs(first_block.varCoefs[lp.getVarByName('x')]))

        assert self._sense == '>=', 'must have Ax >= b'
        assert self._variables_nonnegative, 'must have x >= 0 for all variables'

    # ---- cut pool ----------------------------------------------------------------------------
    @property
    def cut_pool(self):
        return self._cut_pool

    @cut_pool.setter
    def cut_pool(self, cuts):
        for name, (pi, pi0) in cuts.items():
            assert self.cut_name_pattern.match(name), 'idx should start with "cut_"'
            assert isinstance(pi, CyLPArray), 'pi should be CyLPArray'
            assert isinstance(pi0, (int, float)), 'pi0 should be number'
        self._cut_pool = cuts

    # ---- bounding ----------------------------------------------------------------------------
    def bound(self, **kwargs):
        """Entry point BranchAndBound calls (reference :127-135)."""
        return self._base_bound(**kwargs)

    def _base_bound(self, max_cut_generation_iterations=tol.max_cut_generation_iterations,
                    total_cut_generation_iterations=0, total_iterations_gmic_created=0,
                    total_number_gmic_created=0, total_iterations_gmic_added=0,
                    total_number_gmic_added=0, total_iterations_gmic_removed=0,
                    total_number_gmic_removed=0, cut_generation_dual_bound_dict=None,
                    max_cut_generation_run_time=None, max_dual_bound=float('inf'), **kwargs):
        """Solve the LP relaxation, then run cut rounds while they make progress
        (reference :137-230).  Returns the running GMIC totals BranchAndBound threads through
        its kwargs."""
        totals = dict(zip(_COUNTER_KEYS, (
            total_cut_generation_iterations, total_iterations_gmic_created,
            total_number_gmic_created, total_iterations_gmic_added, total_number_gmic_added,
            total_iterations_gmic_removed, total_number_gmic_removed)))
        cut_generation_dual_bound_dict = cut_generation_dual_bound_dict or {}

        assert isinstance(max_cut_generation_iterations, (int, float)) and \
            max_cut_generation_iterations > 0, \
            'max_cut_generation_iterations must be a positive number'
        for key, value in totals.items():
            assert _is_nonneg_int(value), f'{key} is nonnegative integer'
        if cut_generation_dual_bound_dict:
            good, msg = self._good_cut_generation_dual_bound_dict(cut_generation_dual_bound_dict)
            assert good, msg
        if max_cut_generation_run_time is None:
            max_cut_generation_run_time = float('inf')
        assert isinstance(max_cut_generation_run_time, (float, int)) and \
            max_cut_generation_run_time >= 0, 'max_cut_generation_run_time is nonnegative'
        assert isinstance(max_dual_bound, (float, int)), 'max_dual_bound is a number'

        self._bound_lp()
        # the reference budgets CPU time (time.process_time); GPU time does not advance that
        # clock, so wall time is used here (DESIGN.md "deviations")
        start = time.perf_counter()

        def out_of_time():
            return time.perf_counter() - start >= max_cut_generation_run_time

        while self.lp_feasible and not self.mip_feasible and not self.cut_generation_stalled \
                and self.cut_generation_iterations < max_cut_generation_iterations \
                and not out_of_time() and self.objective_value < max_dual_bound:
            self._cut_generation_iteration(**kwargs)

        if self.cut_generation_iterations == max_cut_generation_iterations:
            self.cut_generation_terminator = 'max iterations'
        elif out_of_time():
            self.cut_generation_terminator = 'time'
        elif self.objective_value > max_dual_bound:
            self.cut_generation_terminator = 'dual bound'

        mine = (self.cut_generation_iterations, self.iterations_gmic_created,
                self.number_gmic_created, self.iterations_gmic_added, self.number_gmic_added,
                self.iterations_gmic_removed, self.number_gmic_removed)
        rtn = {key: totals[key] + inc for key, inc in zip(_COUNTER_KEYS, mine)}
        if self.idx is not None and self.cut_generation_dual_bound:
            cut_generation_dual_bound_dict[self.idx] = self.cut_generation_dual_bound
            rtn['cut_generation_dual_bound_dict'] = cut_generation_dual_bound_dict
        return rtn

    def _good_cut_generation_dual_bound_dict(self, d):
        """(ok, message) for a {node idx: {cut round: dual bound}} dict (reference :232-257)."""
        if not isinstance(d, dict):
            return False, 'cut_generation_dual_bound_dict should be a dictionary'
        for idx, rounds in d.items():
            if not isinstance(idx, int):
                return False, f'index {idx} should be integer'
            if idx == self.idx:
                return False, f'index {idx} has already been processed'
            if not isinstance(rounds, dict):
                return False, f'index {idx} should have dictionary value'
            for cut_idx, bound in rounds.items():
                if not isinstance(cut_idx, int):
                    return False, f'cut index {cut_idx} for node {idx} should be integer'
                if not isinstance(bound, (int, float)):
                    return False, \
                        f'dual bound for node {idx} cut index {cut_idx} should be a number'
            if set(rounds) != set(range(max(rounds) + 1)):
                return False, f'index {idx} should have dictionary keyed by range of ints'
        return True, None

    def _bound_lp(self, track_dual_bound=False):
        """One LP relaxation on the engine; fills the status flags, objective and solution
        (reference :259-286)."""
        assert self._x_only_variable, 'x must be our only variable'
        assert isinstance(track_dual_bound, bool), 'track_dual_bound is boolean'
        if track_dual_bound:
            assert self.tracked_cut_generation_iterations not in self.cut_generation_dual_bound, \
                'lp is only bound once per cut generation iteration'

        self.lp.dual()
        code = self.lp.getStatusCode()
        self.lp_feasible = code in (0, 2)      # optimal, or unbounded (dual infeasible)
        self.unbounded = code == 2
        self.objective_value = self.lp.objectiveValue if self.lp_feasible else float('inf')
        sol = self.lp.primalVariableSolution
        self.solution = None if not self.lp_feasible else \
            sol['x'] if isinstance(sol, dict) else sol
        if self.lp_feasible:
            ints = self.solution[self._int_idx]
            self.mip_feasible = bool(np.max(np.abs(np.round(ints) - ints)) <= tol.variable_epsilon) \
                if ints.size else True
        else:
            self.mip_feasible = False
        if track_dual_bound:
            self.cut_generation_dual_bound[self.tracked_cut_generation_iterations] = \
                self.objective_value

    # ---- cutting planes ----------------------------------------------------------------------
    def _cut_generation_iteration(self, cutting_plane_progress_tolerance=
                                  tol.cutting_plane_progress_tolerance,
                                  track_dual_bound=False, **kwargs):
        """One cut round: drop slack cuts, generate, select/add, re-solve, stall test
        (reference :292-324)."""
        assert all(self.solution > -tol.variable_epsilon), 'we must have x >= 0'
        assert isinstance(cutting_plane_progress_tolerance, float) and \
            cutting_plane_progress_tolerance > 0, \
            'cutting_plane_progress_tolerance must be positive'
        assert isinstance(track_dual_bound, bool), 'track_dual_bound is boolean'

        self.solution = np.maximum(self.solution, 0)  # clip tiny negatives
        self.cut_generation_iterations += 1
        if track_dual_bound:
            self.tracked_cut_generation_iterations += 1
        before = self.objective_value

        self._remove_slack_cuts(**kwargs)
        self.cut_pool = {**self.cut_pool, **self._generate_cuts(**kwargs)}
        self._select_cuts(**kwargs)
        self._bound_lp(track_dual_bound=track_dual_bound)
        # NB: like the reference (:320) this divides by |previous objective|
        if abs(before - self.objective_value) / abs(before) < cutting_plane_progress_tolerance:
            self.cut_generation_stalled = True
            self.cut_generation_terminator = self.cut_generation_terminator or \
                'cuts not deep enough'

    def _remove_slack_cuts(self, **kwargs):
        """Drop previously added cuts whose dual is exactly 0 (reference :326-341)."""
        names = [name for name, duals in self.lp.dualConstraintSolution.items()
                 if self.cut_name_pattern.match(name) and all(duals == 0)]
        for name in names:
            self.lp.removeConstraint(name)
        self._update_gmic_counts(cut_idxs=names, operation='removed')
        return names

    def _update_gmic_counts(self, cut_idxs, operation):
        """Count GMICs among cut_idxs for 'added' / 'created' / 'removed' (reference :343-363)."""
        assert isinstance(cut_idxs, (set, list, dict)), \
            'cut_idxs should be an iterable of strings, but not a single string itself'
        for name in cut_idxs:
            assert isinstance(name, str), 'each item in cut_idx should be str type'
        assert operation in ['added', 'created', 'removed'], \
            'operation must be "added", "created", or "removed"'
        hits = sum(1 for name in cut_idxs if self.gmic_name_pattern.match(name))
        self.__dict__[f'iterations_gmic_{operation}'] += 1 if hits else 0
        self.__dict__[f'number_gmic_{operation}'] += hits

    def _generate_cuts(self, gomory_cuts=True, **kwargs):
        """One round of candidate cuts, each rounded to a safe outer approximation
        (reference :365-385).  Names: cut_gomory_<node>_<round>_<tableau row>."""
        assert isinstance(gomory_cuts, bool), 'gomory_cuts is boolean'
        pool = {}
        if gomory_cuts:
            self._engine_rounded = {}
            for row, (pi, pi0) in self._find_gomory_cuts().items():
                name = f'cut_gomory_{self.idx}_{self.cut_generation_iterations}_{row}'
                ready = self._engine_rounded.get(row)
                if ready is not None and ready[0] is pi:
                    pool[name] = ready[1]   # rounded by the cut kernel together with the row
                else:
                    pool[name] = numerically_safe_cut(pi=pi, pi0=pi0, estimate='over')
            self._update_gmic_counts(cut_idxs=pool, operation='created')
        return pool

    def _select_cuts(self, max_nonzero_coefs=tol.max_nonzero_coefs,
                     min_cut_depth=tol.min_cut_depth,
                     parallel_cut_tolerance=tol.parallel_cut_tolerance,
                     max_relative_cut_term_ratio=tol.max_relative_cut_term_ratio, **kwargs):
        """Add the deepest, mutually non-parallel cuts of the pool to the LP
        (reference :387-466)."""
        assert isinstance(max_nonzero_coefs, int) and 0 < max_nonzero_coefs, \
            'max_nonzero_coefs must be positive int'
        assert isinstance(min_cut_depth, (float, int)) and 0 < min_cut_depth, \
            'min_cut_depth must be > 0'
        assert 0 < parallel_cut_tolerance <= 90, \
            'parallel_cut_tolerance must be number in (0, 90]'
        assert isinstance(max_relative_cut_term_ratio, (int, float)) and \
            0 < max_relative_cut_term_ratio, 'max_relative_cut_term_ratio must be positive'

        from math import cos, radians
        from simple_mip_solver_amd.lp import get_backend
        names = list(self.cut_pool)
        if not names:  # (what K3 answers for an empty pool, without the GPU round trip)
            self.cut_generation_terminator = 'no cuts'
            self._update_gmic_counts(cut_idxs={}, operation='added')
            return {}
        x = self.lp.getVarByName('x')
        engine = None
        if len(self.solution) <= 1024:
            engine = get_backend().select_cuts(
                np.array([self.cut_pool[k][0] for k in names],
                         dtype=np.float64).reshape(len(names), len(self.solution)),
                np.array([self.cut_pool[k][1] for k in names], dtype=np.float64), self.solution,
                max_nonzero_coefs, min_cut_depth, cos(radians(parallel_cut_tolerance)),
                max_relative_cut_term_ratio * self.max_term)
        added = {}
        if engine is not None:
            # K3 on the MI355X did the arithmetic (depths, filters, greedy pass); apply its verdict
            picked, terminator, _ = engine
            if terminator:
                self.cut_generation_terminator = ('no cuts', 'no improving cuts',
                                                  'no sufficient cuts')[terminator - 1]
            for pos in picked:
                name = names[int(pos)]
                pi, pi0 = self.cut_pool[name]
                self.lp.addConstraint(pi * x >= pi0, name)
                added[name] = (pi, pi0)
                del self.cut_pool[name]
            self._update_gmic_counts(cut_idxs=added, operation='added')
            return added

        eps = tol.good_coefficient_approximation_epsilon
        depth = {}
        for name, (pi, pi0) in self.cut_pool.items():
            support = int(np.sum(pi > eps) + np.sum(pi < -eps))
            if 0 < support <= max_nonzero_coefs:
                depth[name] = (np.dot(pi, self.solution) - pi0) / np.linalg.norm(pi)

        if not depth:
            self.cut_generation_terminator = 'no cuts'
        else:
            deepest = min(depth.values())
            if deepest >= 0:
                self.cut_generation_terminator = 'no improving cuts'
            elif deepest >= -min_cut_depth:
                self.cut_generation_terminator = 'no sufficient cuts'

        for name in sorted(depth, key=depth.get):  # most violated first (stable)
            if depth[name] >= -min_cut_depth:
                break
            pi, pi0 = self.cut_pool[name]
            if np.max(np.abs(pi)) > max_relative_cut_term_ratio * self.max_term:
                continue
            norm = np.linalg.norm(pi)
            too_parallel = False
            for other, _ in added.values():
                cosine = np.dot(pi, other) / (norm * np.linalg.norm(other))
                cosine = min(1, max(-1, cosine))  # median([-1, cos, 1]) of the reference
                if degrees(acos(cosine)) < parallel_cut_tolerance:
                    too_parallel = True
                    break
            if too_parallel:
                continue
            self.lp.addConstraint(pi * x >= pi0, name)
            added[name] = (pi, pi0)
            del self.cut_pool[name]

        self._update_gmic_counts(cut_idxs=added, operation='added')
        return added

    def _find_gomory_cuts(self):
        """Gomory mixed-integer cuts from the optimal tableau, slack variables substituted out
        (reference :468-511).  Returns {tableau row: (pi, pi0)} meaning pi.x >= pi0."""
        cuts = {}
        if len(self.basic_variable_indices) != self.lp.nConstraints:
            return cuts  # not a square basis: the reference's tableau is None here (:518-519)
        from_engine = self.lp.gomory_rows(self.solution, self._integer_indices, tol.max_term)
        if from_engine is not None:
            # K2 on the MI355X: tableau rows, GMI coefficients, slack substitution and the safe
            # rounding in one launch (the rounded form is handed to _generate_cuts)
            self._engine_rounded = {}
            for k, row in enumerate(from_engine['row_idx']):
                pi = CyLPArray(from_engine['pi'][k])
                cuts[int(row)] = (pi, float(from_engine['pi0'][k]))
                self._engine_rounded[int(row)] = (pi, (CyLPArray(from_engine['safe_pi'][k]),
                                                       float(from_engine['safe_pi0'][k])))
            return cuts
        tableau = self.tableau
        if tableau is None:
            return cuts
        n = self.lp.nVariables
        basic = self.basic_variable_indices
        nonbasic_col = np.ones(n, dtype=bool)
        nonbasic_col[[j for j in basic if j < n]] = False
        is_int = np.zeros(n, dtype=bool)
        is_int[self._int_idx] = True
        A = self.lp.dense_rows()
        rhs = self.lp.constraintsLower
        eps = tol.good_coefficient_approximation_epsilon
        for row, var in enumerate(basic):
            if var >= n or not is_int[var] or not self._is_fractional(self.solution[var]):
                continue
            f0 = self._get_fraction(self.solution[var])
            if f0 < eps or f0 + eps > 1:
                continue  # nearly integral: dividing by f0 or 1 - f0 would blow up
            a = np.where(nonbasic_col, tableau[row, :n], 0.0)   # basic columns count as 0
            f = a - np.floor(a)
            int_coef = np.where(f <= f0, f / f0, (1 - f) / (1 - f0))
            cont_coef = np.where(a > 0, a / f0, -a / (1 - f0))
            pi = np.where(is_int, int_coef, cont_coef)
            s = tableau[row, n:]
            pi_slack = np.where(s > 0, s / f0, -s / (1 - f0))
            # s = A x - b  =>  (pi + A' pi_s) x >= 1 + pi_s . b.  A' pi_s is accumulated row by
            # row, the order of the reference's sparse product (coefMatrix.T * pi_slacks): the
            # continued-fraction rounding that follows is sensitive to the last bit
            back = np.zeros(n)
            for i in range(A.shape[0]):
                back += A[i] * pi_slack[i]
            coefs = CyLPArray(pi + back)
            cuts[row] = (coefs, 1 + np.dot(pi_slack, rhs))
        return cuts

    @property
    def tableau(self):
        """Dense simplex tableau inv([A | -I]_B) [A | -I] for the current basis, or None when the
        basis is not square / singular (reference :513-526)."""
        basic = self.basic_variable_indices
        m = self.lp.nConstraints
        if len(basic) != m:
            return None
        full = np.concatenate((self.lp.dense_rows(), -np.identity(m)), axis=1)
        try:
            return np.linalg.inv(full[:, basic]) @ full
        except np.linalg.LinAlgError:
            return None

    @property
    def basic_variable_indices(self):
        return np.where(np.concatenate(self.lp.getBasisStatus()) == 1)[0]

    # ---- branching ---------------------------------------------------------------------------
    def branch(self, **kwargs):
        """Two children split on the most fractional integer variable (reference :532-541)."""
        return self._base_branch(self._most_fractional_index, **kwargs)

    @property
    def _most_fractional_index(self):
        """Integer index furthest from integrality (> variable_epsilon), lowest index on ties;
        None when all are integral or nothing is solved (reference :544-562)."""
        if not self.lp_feasible or not self._int_idx.size:
            return None
        x = self.solution[self._int_idx]
        dist = np.minimum(x - np.floor(x), np.ceil(x) - x)
        k = int(np.argmax(dist))  # first maximum == the reference's strict '>' scan
        return self._integer_indices[k] if dist[k] > tol.variable_epsilon else None

    def _base_branch(self, branch_idx, next_node_idx=None, **kwargs):
        """Children with x[branch_idx] <= floor / >= ceil, warm-started from this node's basis
        (reference :564-627).  Extra kwargs go to the child constructors, as there."""
        assert self._x_only_variable, 'x must be our only variable'
        assert next_node_idx is None or isinstance(next_node_idx, int), \
            'next node index should be integer if provided'
        assert self.lp_feasible, 'must solve before branching'
        assert branch_idx in self._integer_indices, 'must branch on integer index'
        b_val = self.solution[branch_idx]
        assert self._is_fractional(b_val), "index branched on must be fractional"

        self.is_leaf = False
        lower, upper = self.lp.variablesLower, self.lp.variablesUpper
        lp_left = self.lp.copy_with_bounds(lower, upper)
        lp_left.variablesUpper[branch_idx] = floor(b_val)
        lp_right = self.lp.copy_with_bounds(lower, upper)
        lp_right.variablesLower[branch_idx] = ceil(b_val)

        have_ids = next_node_idx is not None
        self.children = (next_node_idx, next_node_idx + 1) if have_ids else None
        common = dict(integer_indices=self._integer_indices, dual_bound=self.objective_value,
                      b_idx=branch_idx, b_val=b_val, depth=self.depth + 1,
                      ancestors=self.lineage)
        return {
            'left': type(self)(lp=lp_left, idx=next_node_idx, b_dir='left', **common, **kwargs),
            'right': type(self)(lp=lp_right, idx=next_node_idx + 1 if have_ids else None,
                                b_dir='right', **common, **kwargs),
            'next_node_idx': next_node_idx + 2 if have_ids else None,
        }

    def _strong_branch(self, idx, iterations=5):
        """Both children of a branch on idx, each run for at most `iterations` dual simplex
        iterations from this node's optimal basis (reference :629-647).  The two truncated
        solves go to the engine as one batch."""
        assert isinstance(iterations, int) and iterations > 0, \
            'iterations must be positive integer'
        nodes = {k: v for k, v in self._base_branch(idx).items() if k in ('left', 'right')}
        self._solve_probes(list(nodes.values()), iterations)
        return nodes

    @staticmethod
    def _solve_probes(nodes, iterations):
        """Run `lp.maxNumIteration = iterations; lp.dual()` for every probe node; nodes sharing
        one row set are sent to the engine together."""
        from simple_mip_solver_amd.lp import get_backend
        groups = {}
        for node in nodes:
            node.lp.maxNumIteration = iterations
            groups.setdefault(node.lp._engine_form().key, []).append(node)
        for group in groups.values():
            rs = group[0].lp._engine_form()
            bounds = [node.lp._bounds() for node in group]
            warm = [node.lp._warm_start(rs) for node in group]
            vstat = None if warm[0] is None else np.concatenate(warm)
            res = get_backend().solve(rs.A, rs.b, rs.c, np.stack([b[0] for b in bounds]),
                                      np.stack([b[1] for b in bounds]), vstat, iterations, rs.key)
            for k, node in enumerate(group):
                node.lp._store(res, k)

    def _is_fractional(self, value):
        assert isinstance(value, (int, float)), 'value should be a number'
        return min(value - floor(value), ceil(value) - value) > tol.variable_epsilon

    def _fractional_indices(self):
        """The integer indices whose value in `solution` is fractional, in integer_indices order:
        `[i for i in integer_indices if self._is_fractional(solution[i])]`, vectorised unless a
        subclass brings its own `_is_fractional`."""
        if type(self)._is_fractional is not BaseNode._is_fractional or '_is_fractional' in self.__dict__:
            return [i for i in self._integer_indices if self._is_fractional(self.solution[i])]
        idx = np.asarray(self._integer_indices, dtype=np.intp)
        if idx.size == 0:
            return []
        x = np.asarray(self.solution, dtype=np.float64)[idx]
        frac = np.minimum(x - np.floor(x), np.ceil(x) - x) > tol.variable_epsilon
        return [int(i) for i in idx[frac]]

    @staticmethod
    def _get_fraction(value):
        assert isinstance(value, (int, float)), 'value should be a number'
        return value - floor(value)

    # ---- best-first ordering (reference :669-681) --------------------------------------------
    def __eq__(self, other):
        if not isinstance(other, BaseNode):
            raise TypeError('A Node can only be compared with another Node')
        return self.dual_bound == other.dual_bound

    def __lt__(self, other):
        if not isinstance(other, BaseNode):
            raise TypeError('A Node can only be compared with another Node')
        return self.dual_bound < other.dual_bound

    __hash__ = object.__hash__

    def __repr__(self):
        return f'node {self.idx}'

    # ---- format checks (reference :687-710) --------------------------------------------------
    @property
    def _sense(self):
        inf = self.lp.getCoinInfinity()
        has_lower = self.lp.constraintsLower.max() > -inf
        has_upper = self.lp.constraintsUpper.min() < inf
        assert not (has_lower and has_upper), "all constraints should be bounded same way"
        return '<=' if has_upper else '>='

    @property
    def _variables_nonnegative(self):
        return bool((self.lp.variablesLower >= 0).all())

    @property
    def _x_only_variable(self):
        return len(self.lp.variables) == 1 and self.lp.variables[0].name == 'x'
