"""BranchAndBound driver and its tree.

Mirror of simple_mip_solver/algorithms/branch_and_bound.py:19-306: same constructor keywords,
validation messages, `solve()` semantics (re-entrant: a second call continues from the live
queue), status strings, `_kwargs` protocol and tree bookkeeping.  Nodes are duck-typed plugin
objects exactly as in the reference; the stock node classes of this package do their bounding on
the MI355X engine.

Deviations: wall-clock instead of CPU-clock time limits (GPU work does not advance
time.process_time); the root dual bound used by the gap test is tracked incrementally instead of
re-scanning every tree vertex up to four times per iteration (reference :199-213, :226-229) --
same value, O(log N) instead of O(N).
"""
import heapq
from queue import PriorityQueue
import time

import numpy as np

from simple_mip_solver_amd.algorithms.base_algorithm import BaseAlgorithm
from simple_mip_solver_amd.lp import Constraint, CyLPArray, DenseLP
from simple_mip_solver_amd.nodes.base_node import BaseNode
from simple_mip_solver_amd.utils.binary_tree import BinaryTree

from simple_mip_solver_amd.nodes.branch.pseudo_cost import PseudoCostBranchNode
from simple_mip_solver_amd.nodes.search.depth_first import DepthFirstSearchNode
from simple_mip_solver_amd.nodes.nodes import PseudoCostBranchDepthFirstSearchNode

INF = float('inf')
_NATIVE_NODES = (BaseNode, PseudoCostBranchNode, DepthFirstSearchNode,
                 PseudoCostBranchDepthFirstSearchNode)


def _leaf_value(node):
    return node.objective_value if node.objective_value is not None else node.dual_bound


class BranchAndBoundTree(BinaryTree):
    """Search tree; every vertex carries its Node under attr['node'] (reference :19-108)."""

    def get_leaves(self, subtree_root_id, depth=None, keep='all'):
        """Leaves of the subtree under subtree_root_id, optionally after cutting everything more
        than `depth` edges below it.  keep: 'all' | 'feasible' | 'not infeasible'."""
        assert subtree_root_id in self, 'subtree_root_id must belong to the tree'
        assert keep in ['all', 'feasible', 'not infeasible'], \
            "keep is one of 'all', 'feasible', or 'not infeasible'"
        everyone = [v.attr['node'] for v in self.nodes.values()]
        if depth is None:
            found = [n for n in everyone if n.is_leaf and subtree_root_id in n.lineage]
        else:
            assert isinstance(depth, int) and depth >= 0, 'depth is a nonnegative integer'
            if depth == 0:
                found = self.get_node_instances([subtree_root_id])
            elif depth == 1:
                found = self.get_node_instances(self.get_children(subtree_root_id))
            else:
                shallow = [n for n in everyone
                           if n.is_leaf and subtree_root_id in n.lineage[-depth:]]
                at_depth = [n for n in everyone if len(n.lineage) >= depth + 1 and
                            n.lineage[-(depth + 1)] == subtree_root_id]
                found = shallow + at_depth
        if keep == 'feasible':
            return [n for n in found if n.lp_feasible]
        if keep == 'not infeasible':
            return [n for n in found if n.lp_feasible is not False]
        return found

    def get_disjunction(self, subtree_root_id):
        """{leaf idx: (lower bounds, upper bounds)} over the not-infeasible leaves."""
        return {n.idx: (n.lp.variablesLower.copy(), n.lp.variablesUpper.copy())
                for n in self.get_leaves(subtree_root_id, keep='not infeasible')}

    def get_node_instances(self, node_ids):
        single = isinstance(node_ids, int)
        if single:
            node_ids = [node_ids]
        else:
            assert hasattr(node_ids, '__iter__') and not isinstance(node_ids, str), \
                'node_ids must be an integer or iterable (that is not a string)'
            node_ids = list(node_ids)
        missing = set(node_ids) - set(self.nodes)
        assert not missing, f'the following node_ids are not in the tree: {missing}'
        found = [self.nodes[i].attr.get('node') for i in node_ids]
        assert all(n is not None for n in found), \
            'each vertex in the branch and bound tree must have an attribute for a node instance'
        return found[0] if single else found

    def subtree_dual_bound(self, subtree_root_id, depth=None):
        """min over the subtree's leaves of their LP objective (or inherited bound if unsolved)."""
        assert subtree_root_id in self, 'subtree_root_id must belong to the tree'
        return min(_leaf_value(n) for n in self.get_leaves(subtree_root_id, depth=depth))


class _LeafBounds:
    """Lazy min-heap over leaf values: same answer as scanning all leaves of the root."""

    def __init__(self):
        self._heap = []
        self._count = 0

    def push(self, node):
        self._count += 1
        heapq.heappush(self._heap, (_leaf_value(node), self._count, node))

    def minimum(self):
        while self._heap:
            value, _, node = self._heap[0]
            if node.is_leaf and _leaf_value(node) == value:
                return value
            heapq.heappop(self._heap)  # stale: the node was branched on or re-valued
        return INF


class BranchAndBound(BaseAlgorithm):
    """Solve a MILP by branch and bound with the bound / branch / search rules of a Node class."""

    _node_attributes = ['dual_bound', 'objective_value', 'solution', 'lp_feasible',
                        'mip_feasible', 'search_method', 'branch_method', 'idx', 'lp',
                        'is_leaf', 'lineage']
    _node_funcs = ['bound', 'branch', '__lt__', '__eq__']
    _queue_funcs = ['put', 'get', 'empty']

    def __init__(self, model, Node=BaseNode, node_queue=None, node_limit=INF, mip_gap=.0001,
                 logging=False, max_run_time=INF, initial_primal_bound=INF, frontier_batch=None,
                 pool_capacity=1 << 16, anchor=None, dive=None, comm=None, exchange_every=5, **kwargs):
        """All problems are converted to minimisation with A x >= b on the way in.  **kwargs are
        handed to every bound()/branch() call and refreshed from what those calls return
        (e.g. pseudo_costs={}, strong_branch_iters=5, gomory_cuts=False).

        frontier_batch (extension, default None = the reference's one-node-at-a-time Python
        loop): run the whole search in the native frontier engine (mipx_tree_*), evaluating that
        many open nodes per GPU step with node records resident in HBM.  Only for the stock node
        classes and the default queue; frontier_batch=1 keeps the reference's exact node order.
        With gomory_cuts=True (the reference's default, base_node.py:365) every node runs the cut
        rounds of BaseNode._base_bound inside the engine (register-tile shapes: m + 64 <= 192 rows,
        n <= 256; no dive then).  In this mode `tree` holds only the root (nodes live on the device).
        comm (extension; needs frontier_batch): an _ffi.Comm shared by one process per GPU
        (simple_mip_solver_amd.parallel.init_comm).  Every rank builds the same BranchAndBound and
        calls solve(): after a replicated ramp-up the open nodes are sharded over the ranks, which
        exchange incumbent (value and solution), bounds, pseudo costs and node records every
        `exchange_every` steps over RCCL; every rank ends with the same status, objective_value and
        solution; evaluated_nodes is the total over all ranks.
        anchor / dive (default: on for frontier_batch > 1, register-tile shapes): warm starts
        refactor from the root's optimal tableau instead of the slack basis; the workgroup that
        solved a node also solves one child on the tableau it holds (same optimum, another node
        order -- see DESIGN.md section 4)."""
        self._native = None
        self._native_stats = None
        assert comm is None or frontier_batch is not None, 'comm needs frontier_batch'
        self._comm, self._exchange_every, self._sharded = comm, exchange_every, False
        if frontier_batch is not None:
            assert isinstance(frontier_batch, int) and frontier_batch > 0, \
                'frontier_batch must be a positive integer'
            assert node_queue is None, 'frontier_batch needs the default node queue'
            assert Node in _NATIVE_NODES, \
                'frontier_batch is only available for the stock node classes'
            assert isinstance(kwargs.get('gomory_cuts', True), bool), 'gomory_cuts is boolean'
        self.frontier_batch = frontier_batch
        batched = frontier_batch is not None and frontier_batch > 1
        native_cuts = frontier_batch is not None and kwargs.get('gomory_cuts', True)
        self._anchor = batched if anchor is None else bool(anchor)
        # dive: False / 0 off, True / 1 one dive child per node, an int up to 8 that many in a row
        self._dive = int(batched and not native_cuts) if dive is None else int(dive)
        assert 0 <= self._dive <= 8, 'dive is a depth between 0 and 8'
        assert not (self._dive and native_cuts), 'dive is not available with gomory_cuts=True'
        assert not (self._dive and not batched), 'dive needs frontier_batch > 1'
        self._pool_capacity = pool_capacity
        node_queue = node_queue or PriorityQueue()
        super().__init__(model=model, Node=Node, node_attributes=self._node_attributes,
                         node_funcs=self._node_funcs, **kwargs)

        for func in self._queue_funcs:
            assert callable(getattr(node_queue, func, None)), f'node_queue needs a {func} function'
        assert node_limit == INF or (isinstance(node_limit, int) and node_limit > 0), \
            "node limit must be positive integer or infinity"
        assert 0 <= mip_gap < 1, 'mip_gap is a ratio between 0 and 1'
        assert isinstance(logging, bool), 'logging is boolean'
        assert max_run_time > 0, 'max_run_time is positive value'
        assert initial_primal_bound > -INF, 'initial_primal_bound is real or infinite'
        special_keys = {'right', 'left', 'cuts'}
        assert set(kwargs.keys()).isdisjoint(special_keys), \
            f'keys {special_keys} are saved for later use'
        assert all(isinstance(k, str) for k in kwargs), 'kwargs keys must be strings'

        self._node_queue = node_queue
        self._unbounded = None
        self._best_solution = None
        self.solution = None
        self.status = 'unsolved'
        self.objective_value = None
        self.primal_bound = initial_primal_bound
        self.node_limit = node_limit
        self.tree = BranchAndBoundTree()
        self.tree.add_root(self.root_node.idx, node=self.root_node)
        self._leaf_bounds = _LeafBounds()
        self._leaf_bounds.push(self.root_node)
        self.solve_time = 0
        self.mip_gap = mip_gap
        self.logging = logging
        self.max_run_time = max_run_time

    @property
    def dual_bound(self):
        if self._native_stats is not None:
            return self._native_stats['dual_bound']
        return self._leaf_bounds.minimum()

    @property
    def current_gap(self):
        """|primal - dual| / |primal|; None until an incumbent exists (reference :203-213)."""
        primal, dual = self.primal_bound, self.dual_bound
        if primal == dual == 0:
            return 0
        if primal == 0:
            return INF
        if primal == INF:
            return None
        return abs(primal - dual) / abs(primal)

    def _gap_closed(self):
        gap = self.current_gap
        return gap is not None and gap <= self.mip_gap

    def solve(self):
        """Run (or continue) the search until the queue empties, the problem proves unbounded, or
        the node / gap / time limit is hit (reference :215-241)."""
        if self.frontier_batch is not None:
            return self._solve_native()
        start = time.perf_counter()
        if self.status == 'unsolved':
            self._node_queue.put(self.root_node)

        while not (self._node_queue.empty() or self._unbounded or
                   self.evaluated_nodes >= self.node_limit or self._gap_closed() or
                   time.perf_counter() - start > self.max_run_time):
            if self.logging and self.evaluated_nodes % 100 == 0:
                print(f'{self.evaluated_nodes} nodes evaluated gap: {self.current_gap}')
            self._evaluate_node(self._node_queue.get())

        self.solve_time += time.perf_counter() - start
        if self._unbounded:
            self.status = 'unbounded'
        elif self._node_queue.empty() and self.primal_bound == INF:
            self.status = 'infeasible'
        elif self.primal_bound < INF and self.current_gap <= self.mip_gap:
            self.status = 'optimal'
        else:
            self.status = 'stopped on iterations or time'
        self.solution = self._best_solution
        self.objective_value = self.primal_bound

    def _solve_native(self):
        """The same search, run by the native frontier engine (include/mipx.h mipx_tree_*)."""
        from simple_mip_solver_amd import _ffi
        from simple_mip_solver_amd.lp import get_backend, HipBackend
        if self._native is None:
            self._native_totals0 = {k: self._kwargs.get(k, 0) for k in _ffi.CUT_TOTAL_KEYS}
            backend = get_backend()
            assert isinstance(backend, HipBackend), 'frontier_batch needs the HIP backend'
            lp = self.root_node.lp
            rs = lp._engine_form()
            problem = backend._problem(rs.A, rs.b, rs.c, rs.key)
            l, u = lp._bounds()
            pseudo = issubclass(self._Node, PseudoCostBranchNode)
            cut_params = None
            if self._kwargs.get('gomory_cuts', True):
                # the keyword defaults of BaseNode._base_bound / _cut_generation_iteration / _select_cuts
                # (base_node.py:137, :292, :387), overridable through **kwargs as there
                from math import cos, radians
                from simple_mip_solver_amd.utils import tolerance as tol
                kw = self._kwargs
                rounds = kw.get('max_cut_generation_iterations', tol.max_cut_generation_iterations)
                cut_params = dict(
                    max_cut_generation_iterations=int(min(rounds, 2 ** 31 - 1)),
                    max_nonzero_coefs=int(min(kw.get('max_nonzero_coefs', tol.max_nonzero_coefs), 2 ** 31 - 1)),
                    cutting_plane_progress_tolerance=kw.get('cutting_plane_progress_tolerance',
                                                            tol.cutting_plane_progress_tolerance),
                    min_cut_depth=kw.get('min_cut_depth', tol.min_cut_depth),
                    cos_parallel=cos(radians(kw.get('parallel_cut_tolerance', tol.parallel_cut_tolerance))),
                    max_abs_coef=kw.get('max_relative_cut_term_ratio', tol.max_relative_cut_term_ratio) *
                    float(self.root_node.max_term),
                    max_term=tol.max_term, max_dual_bound=kw.get('max_dual_bound', INF),
                    exact_tableau=0 if self._anchor else 1)
            self._native = _ffi.Tree(
                problem, self.model.integerIndices, l, u,
                branch_rule='pseudo cost' if pseudo else 'most fractional',
                search_rule=self.root_node.search_method,
                strong_branch_iters=self._kwargs.get('strong_branch_iters', 5),
                max_batch=self.frontier_batch, pool_capacity=self._pool_capacity, cut_params=cut_params)
            if self.primal_bound < INF:
                self._native.set_primal_bound(self.primal_bound)
            table = self._kwargs.get('pseudo_costs')
            if pseudo and table:
                # a table handed in by the caller (pseudo_cost.py:22: `pseudo_costs` is an input of
                # bound) seeds the engine's; an entry counts as present once either side was visited
                n = problem.n
                cl, cr = np.zeros(n), np.zeros(n)
                tl, tr = np.zeros(n, np.int32), np.zeros(n, np.int32)
                for i, rec in table.items():
                    cl[i], tl[i] = rec['left']['cost'], rec['left']['times']
                    cr[i], tr[i] = rec['right']['cost'], rec['right']['times']
                self._native.set_pseudo_cost_arrays(cl, cr, tl, tr)
            if self._anchor:
                self._native.set_anchor_mode(True)
            if self._dive:
                self._native.set_dive(self._dive)
        st = None
        if self._comm is not None and not self._sharded:
            from simple_mip_solver_amd.parallel import shard_and_attach
            ramp = shard_and_attach(self._native, self._comm, self.frontier_batch, self._exchange_every)
            self._sharded = True
            if ramp['status'] not in (0, 4) or ramp['open_nodes'] == 0:
                st = ramp      # finished inside the replicated ramp-up: every rank holds the result
                self._comm = None
        if st is None:
            st = self._native.solve(node_limit=0 if self.node_limit == INF else self.node_limit,
                                    mip_gap=self.mip_gap,
                                    max_seconds=0.0 if self.max_run_time == INF else self.max_run_time,
                                    frontier_batch=self.frontier_batch)
        if self._comm is not None:
            g = self._native.global_stats()
            st = dict(st, evaluated_nodes=g['evaluated_nodes'])
            self._native_global = g
        self._native_stats = st
        if st['pool_exhausted']:
            import warnings
            warnings.warn('the GPU node pool is full (pool_capacity=%d): the search stopped with the bounds '
                          'found so far; pass a larger pool_capacity' % self._pool_capacity, RuntimeWarning)
        self.solve_time = st['solve_seconds']
        self.evaluated_nodes = st['evaluated_nodes']
        self.primal_bound = st['primal_bound']
        self.status = _ffi.TREE_STATUS[st['status']]
        self._unbounded = True if st['status'] == 3 else self._unbounded
        self._best_solution = self._native.solution() if st['has_solution'] else None
        self.solution = self._best_solution
        self.objective_value = self.primal_bound
        self._kwargs['next_node_idx'] = st['created_nodes']
        if issubclass(self._Node, PseudoCostBranchNode):
            self._kwargs['pseudo_costs'] = self._native.pseudo_costs()
        if self._native.cuts:   # the running GMIC totals bound() threads through the kwargs
            totals = self._native.cut_stats()
            self._native_cuts_dropped = totals.pop('dropped')
            for key, value in totals.items():
                self._kwargs[key] = self._native_totals0.get(key, 0) + value

    def _evaluate_node(self, node):
        """Bound the node unless its inherited bound already prunes it; record an incumbent or
        branch (reference :243-266)."""
        if not node.dual_bound < self.primal_bound:
            return
        self.evaluated_nodes += 1
        self._process_bound_rtn(node.bound(**self._kwargs))
        self._leaf_bounds.push(node)  # the node now carries its own LP objective

        # like the reference, an unbounded relaxation is taken to mean an unbounded MILP
        if node.unbounded:
            self._unbounded = True

        if node.lp_feasible and node.objective_value < self.primal_bound:
            if node.mip_feasible:
                self._best_solution = node.solution
                self.primal_bound = node.objective_value
            else:
                self._process_branch_rtn(node.idx, node.branch(**self._kwargs))

    def _process_branch_rtn(self, parent_id, rtn):
        """Queue the two children ('left' = down, 'right' = up), hang them in the tree, merge the
        remaining keys into the kwargs (reference :268-289)."""
        assert isinstance(rtn, dict), 'rtn must be a dictionary'
        assert isinstance(parent_id, int), 'parent_id must be integer'
        assert parent_id in self.tree, 'parent must already exist in tree'
        for direction in ['left', 'right']:
            assert direction in rtn, f'{direction} must be in the returned dict'
            child = rtn.pop(direction)
            assert isinstance(child, self._Node), \
                f'{direction} value must be type {type(self._Node)}'
            assert child.idx not in self.tree, 'please give unique node ID'
            self._node_queue.put(child)
            getattr(self.tree, f'add_{direction}_child')(child.idx, parent_id, node=child)
            self._leaf_bounds.push(child)
        self._process_rtn(rtn)

    def _process_bound_rtn(self, rtn):
        """Share returned 'cuts' with every queued node's cut pool, merge the rest into the
        kwargs (reference :291-306)."""
        assert isinstance(rtn, dict), 'rtn must be a dictionary'
        cuts = rtn.get('cuts')
        if cuts:
            for name, (pi, pi0) in cuts.items():
                for queued in self._node_queue.queue:
                    queued.cut_pool[name] = (pi, pi0)
            del rtn['cuts']
        self._process_rtn(rtn)

    def find_parameterized_dual_bound(self, b):
        """Lower bound on the optimal value of the MILP at a new right-hand side b, from the dual
        solutions stored along every leaf's lineage (reference :314-360): per leaf the best of
        `y.b + max(d, 0).l + min(d, 0).u` over its solved ancestors and itself, then the worst
        leaf.  Infeasible leaves are first re-solved with penalised slacks so that they carry a
        finite dual solution (`_bound_parameterized_dual`).  Nodes that were never solved (pruned
        by their inherited bound) contribute through their ancestors only."""
        assert isinstance(b, CyLPArray), 'this function only works with CyLP arrays'
        assert self.status != 'unsolved', 'must solve this instance before using this method'
        assert self.frontier_batch is None, \
            'the native frontier engine keeps no per-node duals; solve with frontier_batch=None'
        terminal_nodes = self.tree.get_leaves(self.root_node.idx)
        multi_const_nodes = [n.idx for n in terminal_nodes if len(n.lp.constraints) != 1]
        assert not multi_const_nodes, \
            f'This feature expects the root node to have a single constraint object and ' \
            f'all nodes to branch by bounding variables instead of by adding constraints. ' \
            f'It does not currently handle cuts being added after bounding. The following ' \
            f'IDs belong to nodes that do not conform to these rules: {multi_const_nodes}'
        assert all(b.shape == n.lp.constraints[0].lower.shape for n in terminal_nodes), \
            'the shape of the RHS being added should match that of each node'
        if self._swapped_constraint_direction:
            b = -b
            print('WARNING: your rhs was made negative to reflect constraints'
                  ' flipping direction at instantiation')
        for n in terminal_nodes:
            if n.lp._status == 1:  # primal infeasible: bound its dual ray (once)
                n.lp = self._bound_parameterized_dual(n.lp)
        assert all(n.lp._status in [None, 0] for n in terminal_nodes)

        def evaluate(lp):
            d = np.concatenate(list(lp.dualVariableSolution.values()))
            return float(np.inner(lp.dualConstraintSolution[lp.constraints[0].name], b) +
                         np.inner(np.maximum(d, 0), lp.variablesLower) +
                         np.inner(np.minimum(d, 0), lp.variablesUpper))

        bounds = {}
        for leaf in terminal_nodes:
            solved = [n.lp for n in self.tree.get_node_instances(leaf.lineage) if n.lp._status == 0]
            bounds[leaf.idx] = max(evaluate(lp) for lp in solved)
        return min(bounds.values())

    def _bound_parameterized_dual(self, cur_lp):
        """The same LP with a slack block `s_i >= 0` on every constraint block i, priced at a
        large M: its dual is the original dual with the multipliers capped at M, so a node with
        an infeasible relaxation gets a finite (very large) dual solution to evaluate at other
        right-hand sides (reference :362-417).  Solved before it is returned."""
        assert isinstance(cur_lp, DenseLP), 'must give CyClpSimplex instance'
        for i, constr in enumerate(cur_lp.constraints):
            assert f's_{i}' not in [v.name for v in cur_lp.variables], \
                f"variable 's_{i}' is a reserved name. please name your variable something else"
        new_lp = DenseLP()
        var_map = {v: new_lp.addVariable(v.name, v.dim) for v in cur_lp.variables}
        n0 = cur_lp.nVariables
        new_lp.variablesLower[:n0] = cur_lp.variablesLower
        new_lp.variablesUpper[:n0] = cur_lp.variablesUpper
        slacks = [new_lp.addVariable(f's_{i}', constr.rows) for i, constr in enumerate(cur_lp.constraints)]
        for constr, s_i in zip(cur_lp.constraints, slacks):
            extra = {var_map[v]: a for v, a in constr.varCoefs.items() if v is not constr.variables[0]}
            extra[s_i] = np.identity(constr.rows)
            new_lp.addConstraint(Constraint(var_map[constr.variables[0]], constr._coefs, constr.lower,
                                            constr.upper, constr.name, extra))
        new_lp.objective = np.concatenate([cur_lp.objective, np.full(new_lp.nVariables - n0, float(self._M))])
        var_status, row_status = cur_lp.getBasisStatus()
        # every s_i enters at its lower bound of 0 (status 3)
        new_lp.setBasisStatus(np.concatenate([var_status, np.full(new_lp.nVariables - n0, 3)]), row_status)
        new_lp.dual()
        return new_lp
