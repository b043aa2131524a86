"""Cut-generating LP of a branch-and-bound disjunction (mirror of the reference's
simple_mip_solver/utils/cut_generating_lp.py: same constructor, `_create_cglp`, `solve`, the same
variable / constraint names and assert messages).  Outside the node hot path:
the LP is assembled on the host and solved by the same GPU dual simplex engine through `DenseLP`;
its free columns (pi, pi0) go through DenseLP's boxed substitution.

For the leaves t of the subtree (the disjunctive terms, each  A_t x >= b_t, l_t <= x <= u_t):

    min  x*.pi - pi0
    s.t. pi  >= A_t' u_t + w_t - v_t          for every t       (names 'Au_t + Iw_t - Ivt <= pi')
         pi0 <= b_t.u_t + l_t.w_t - u_t.v_t   for every t       ('bu_t + lbw_t - ubvt >= pi0')
         sum(u, w, v) = 1                                       ('normalize')
         u, w, v >= 0;  w_t[j] (v_t[j]) fixed at 0 where l_t[j] (u_t[j]) is infinite

so that pi.x >= pi0 holds on every term, and the objective is most violated at x*.
"""
from collections.abc import Iterable

import numpy as np
from scipy.sparse import csc_matrix

from simple_mip_solver_amd.lp import Constraint, CyLPArray, DenseLP


class CutGeneratingLP:

    def __init__(self, bb, root_id, A=None, b=None, var_lb=None, var_ub=None, depth=None):
        """bb: a solved BranchAndBound; root_id: node whose subtree gives the disjunction; A, b:
        rows to use for every term instead of its own; var_lb / var_ub: bounds intersected with
        every term's; depth: cut the subtree below this many levels."""
        from simple_mip_solver_amd.algorithms.branch_and_bound import BranchAndBound
        assert isinstance(bb, BranchAndBound), 'bb must be a BranchAndBound instance'
        assert root_id in bb.tree, 'root node of the disjunction must be present in B & B tree'
        if depth is not None:
            assert isinstance(depth, int) and depth > 0, 'depth is postive integer'
        self.bb = bb
        self.root_id = root_id
        self.depth = depth
        self.lp = self._create_cglp(A, b, var_lb, var_ub)
        self.cylp_failure = False

    def _create_cglp(self, A=None, b=None, var_lb=None, var_ub=None):
        terms = {n.idx: n for n in self.bb.tree.get_leaves(self.root_id, depth=self.depth,
                                                           keep='not infeasible')}
        shapes = [{v.name: v.dim for v in n.lp.variables} for n in terms.values()]
        assert all(shapes[0] == d for d in shapes), \
            'Each disjunctive term should have the same variables. The feature allowing' \
            ' otherwise remains to be developed.'
        num_vars = sum(shapes[0].values())
        root = self.bb.tree.get_node_instances(self.root_id)
        assert root.solution is not None, 'root must be solved to create CGLP'
        inf = root.lp.getCoinInfinity()

        assert (A is None and b is None) or (A is not None and b is not None), \
            "A and b must both have values or must both be None"
        if A is not None:
            assert isinstance(A, np.matrix) or isinstance(A, csc_matrix), \
                "A must be a numpy or sparse csc matrix"
            assert A.shape[1] == num_vars, \
                "A must have same number of columns as each disjunctive term has variables"
        if b is not None:
            assert isinstance(b, CyLPArray), "b must be a CyLPArray"
            assert b.shape == (A.shape[0],), "A must have the same number of rows " \
                                             "as b has entries"
        if var_lb is not None:
            assert isinstance(var_lb, CyLPArray), "var_lb must be a CyLPArray"
            assert var_lb.shape == (num_vars,), "Must have same number of lower bounds as variables"
        else:
            var_lb = CyLPArray([-float('inf')] * num_vars)
        if var_ub is not None:
            assert isinstance(var_ub, CyLPArray), "var_ub must be a CyLPArray"
            assert var_ub.shape == (num_vars,), "Must have same number of upper bounds as variables"
        else:
            var_ub = CyLPArray([float('inf')] * num_vars)

        # per term: bound coefficients (0 where the bound is infinite) and the caps that switch the
        # matching multiplier off; a term whose intersected bounds cross is dropped
        lb, ub, w_cap, v_cap = {}, {}, {}, {}
        for idx, node in list(terms.items()):
            lo = np.maximum(np.asarray(node.lp.variablesLower), np.asarray(var_lb))
            up = np.minimum(np.asarray(node.lp.variablesUpper), np.asarray(var_ub))
            if np.any(lo > up):
                del terms[idx]
                continue
            lb[idx] = np.where(lo > -inf, lo, 0.0)
            ub[idx] = np.where(up < inf, up, 0.0)
            w_cap[idx] = np.where(lo > -inf, inf, 0.0)
            v_cap[idx] = np.where(up < inf, inf, 0.0)
        rows = {}
        for idx, node in terms.items():
            if A is not None:
                At = A.toarray() if isinstance(A, csc_matrix) else np.asarray(A, dtype=np.float64)
                bt = np.asarray(b, dtype=np.float64)
            else:
                At, bt = node.lp.dense_rows(), np.asarray(node.lp.constraintsLower, dtype=np.float64)
            rows[idx] = (np.ascontiguousarray(At), bt)

        lp = DenseLP()
        pi = lp.addVariable('pi', num_vars)
        pi0 = lp.addVariable('pi0', 1)
        u = {idx: lp.addVariable(f'u_{idx}', bt.size) for idx, (_, bt) in rows.items()}
        w = {idx: lp.addVariable(f'w_{idx}', node.lp.nVariables) for idx, node in terms.items()}
        v = {idx: lp.addVariable(f'v_{idx}', node.lp.nVariables) for idx, node in terms.items()}
        for var in (pi, pi0):
            lp.variablesLower[var.indices] = -inf
        for idx in terms:
            lp.variablesUpper[w[idx].indices] = w_cap[idx]
            lp.variablesUpper[v[idx].indices] = v_cap[idx]

        eye = np.eye(num_vars)
        for idx, (At, bt) in rows.items():
            # (pi, pi0) is valid for the term's LP relaxation
            lp.addConstraint(Constraint(pi, -eye, upper=np.zeros(num_vars),
                                        extra={u[idx]: At.T, w[idx]: eye, v[idx]: -eye}),
                             name=f'Au_{idx} + Iw_{idx} - Iv{idx} <= pi')
            lp.addConstraint(Constraint(pi0, np.array([[-1.0]]), lower=np.zeros(1),
                                        extra={u[idx]: bt.reshape(1, -1), w[idx]: lb[idx].reshape(1, -1),
                                               v[idx]: -ub[idx].reshape(1, -1)}),
                             name=f'bu_{idx} + lbw_{idx} - ubv{idx} >= pi0')
        # the multipliers are normalised so that the cut cannot be scaled at will
        mult = [var for group in (u, w, v) for var in group.values()]
        lp.addConstraint(Constraint(mult[0], np.ones((1, mult[0].dim)), lower=np.ones(1), upper=np.ones(1),
                                    extra={var: np.ones((1, var.dim)) for var in mult[1:]}),
                         name='normalize')
        # deepest cut at the root solution: min pi.x* - pi0
        lp.objective = np.concatenate([np.asarray(root.solution, dtype=np.float64), [-1.0],
                                       np.zeros(lp.nVariables - num_vars - 1)])
        return lp

    def solve(self, x_star=None, starting_basis=None):
        """The inequality pi.x >= pi0, valid for every disjunctive term, that x_star (default:
        the LP solution of the subtree's root) violates most; (None, None) if the engine fails."""
        if x_star is not None:
            pi = self.lp.getVarByName('pi')
            assert isinstance(x_star, CyLPArray), 'x_star must be a CyLPArray'
            assert x_star.shape == (pi.dim,), \
                'x_star must have the same number of variables as the LP relaxations ' \
                'in the branch and bound tree this instance was created with'
            obj = np.asarray(self.lp.objective).copy()
            obj[:pi.dim] = np.asarray(x_star)
            basis = (self.lp._var_status, self.lp._row_status)
            self.lp.objective = obj
            self.lp._var_status, self.lp._row_status = basis  # a new objective keeps the basis
        if starting_basis is not None:
            assert isinstance(starting_basis, Iterable) and not isinstance(starting_basis, str) \
                and len(starting_basis) == 2, 'starting basis must be an iterable with two elements'
            for status_array in starting_basis:
                assert isinstance(status_array, np.ndarray), \
                    'elements of starting basis must be np.ndarrays'
            assert starting_basis[0].shape == (self.lp.nVariables,), \
                'first starting_basis element should give status for exactly each decision variable in CGLP'
            assert starting_basis[1].shape == (self.lp.nConstraints,), \
                'second starting_basis element should give status for exactly each slack variable in CGLP'
            self.lp.setBasisStatus(*starting_basis)

        self.lp.primal()

        if self.lp.getStatusCode() in [0, 2]:
            sol = self.lp.primalVariableSolution
            return CyLPArray(sol['pi']), float(sol['pi0'][0])
        self.cylp_failure = True
        return None, None
