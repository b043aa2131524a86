"""LP container behind the Node plugin surface.

The reference gives every node a `cylp.cy.CyClpSimplex` and reaches COIN-OR Clp through it
(simple_mip_solver/nodes/base_node.py:74, :273).  `DenseLP` offers the members of that class the
hot path actually touches (SURVEY.md section 8b lists them with call sites) and sends every solve
to the MI355X engine (libmipx.so, include/mipx.h) through an `LPBackend`.  There is no CPU solver
in this package: the default backend raises if the HIP library or the GPU is missing.

Also here: a few lines of modelling sugar (`CyLPArray`, `x = lp.addVariable('x', n)`,
`lp += l <= x <= u`, `lp.addConstraint(pi * x >= pi0, name)`) so that code and tests written
against the reference's cylp idioms (base_node.py:459-460, :594-606) read the same.
"""
from collections import OrderedDict

import numpy as np

COIN_INFINITY = 1.7976931348623157e308  # what CyClpSimplex.getCoinInfinity() reports


class CyLPArray(np.ndarray):
    """Float vector; stands in for cylp.py.modeling.CyLPModel.CyLPArray."""

    def __new__(cls, values):
        return np.asarray(values, dtype=np.float64).view(cls)

    # comparisons with modelling objects must build constraints, not boolean arrays
    def __le__(self, other):
        if isinstance(other, (LinearExpression, Variable)):
            return other.__ge__(self)
        return np.ndarray.__le__(self, other)

    def __ge__(self, other):
        if isinstance(other, (LinearExpression, Variable)):
            return other.__le__(self)
        return np.ndarray.__ge__(self, other)

    def __mul__(self, other):
        if isinstance(other, Variable):
            return LinearExpression(other, np.asarray(self, dtype=np.float64))
        return np.ndarray.__mul__(self, other)


class _Bounded:
    """Something that chained comparisons `lo <= thing <= hi` accumulate bounds on."""

    def _fresh(self):
        raise NotImplementedError

    def _target(self):
        if getattr(self, '_pending', None) is None:
            self._pending = self._fresh()
        return self._pending

    def __ge__(self, other):
        t = self._target()
        t.lower = _as_bound(other, t.rows)
        return t

    def __le__(self, other):
        t = self._target()
        t.upper = _as_bound(other, t.rows)
        return t


def _as_bound(value, rows):
    arr = np.asarray(value, dtype=np.float64).reshape(-1)
    if arr.size == 1 and rows != 1:
        arr = np.full(rows, arr[0])
    assert arr.size == rows, 'bound has the wrong length'
    return arr.copy()


class Variable(_Bounded):
    """A named block of `dim` columns (the reference only ever uses one, called 'x')."""

    def __init__(self, name, dim):
        self.name = name
        self.dim = int(dim)
        self._pending = None
        self._lp = None      # the DenseLP the block belongs to, and its first column there
        self._offset = 0

    def _fresh(self):
        return BoundSpec(self)

    @property
    def indices(self):
        return np.arange(self._offset, self._offset + self.dim)

    @property
    def lower(self):
        return self._lp.variablesLower[self._offset:self._offset + self.dim]

    @property
    def upper(self):
        return self._lp.variablesUpper[self._offset:self._offset + self.dim]

    def __rmul__(self, coefs):
        return LinearExpression(self, coefs)

    # make numpy operators defer to the reflected methods above
    __array_ufunc__ = None
    __array_priority__ = 1000
    __hash__ = object.__hash__


class BoundSpec:
    """Result of `l <= x <= u`."""

    def __init__(self, var):
        self.var = var
        self.rows = var.dim
        self.lower = np.full(var.dim, -np.inf)
        self.upper = np.full(var.dim, np.inf)

    def __bool__(self):
        return True


class LinearExpression(_Bounded):
    """`coefs * x` with coefs a vector (one row) or a matrix (a block of rows)."""

    __array_ufunc__ = None
    __array_priority__ = 1000
    __hash__ = object.__hash__

    def __init__(self, var, coefs):
        m = np.asarray(coefs, dtype=np.float64)
        if m.ndim == 1:
            m = m.reshape(1, -1)
        assert m.ndim == 2 and m.shape[1] == var.dim, 'coefficient shape must match the variable'
        self.var = var
        self.coefs = np.ascontiguousarray(m)
        self._pending = None

    def _fresh(self):
        return Constraint(self.var, self.coefs)


class Constraint:
    """A block of rows `lower <= coefs x <= upper`; mirrors the members read at
    base_node.py:104, :602-606 (`lower`, `upper`, `varCoefs`, `variables`, `name`)."""

    def __init__(self, var, coefs, lower=None, upper=None, name=None, extra=None):
        # `extra`: coefficient blocks of further variable blocks {Variable: rows x dim}; the node
        # hot path never has any (one block 'x'), the parametric dual bound adds slack blocks
        self._extra = {} if not extra else {v: np.ascontiguousarray(a, dtype=np.float64)
                                            for v, a in extra.items()}
        self.variables = [var] + list(self._extra)
        self._coefs = np.ascontiguousarray(coefs, dtype=np.float64)
        self.rows = self._coefs.shape[0]
        self.lower = np.full(self.rows, -np.inf) if lower is None else _as_bound(lower, self.rows)
        self.upper = np.full(self.rows, np.inf) if upper is None else _as_bound(upper, self.rows)
        self.name = name

    @property
    def varCoefs(self):
        return {self.variables[0]: self._coefs, **self._extra}

    def dense(self, n):
        """The rows over all n columns of the LP."""
        if not self._extra and self._coefs.shape[1] == n:
            return self._coefs
        out = np.zeros((self.rows, n))
        for v, a in self.varCoefs.items():
            out[:, v._offset:v._offset + v.dim] = a
        return out

    def __bool__(self):
        return True


class LPBackend:
    """What DenseLP needs from an engine.  The only implementation shipped is HipBackend."""

    def solve(self, A, b, c, l, u, vstat, max_iter, cache_key):
        raise NotImplementedError

    def gomory(self, A, b, c, l, u, vstat, x, integer_indices, max_term, cache_key):
        """Raw and safely rounded GMI cuts of one solved node, or None if the engine has no
        cut kernel (the node then uses its host arithmetic)."""
        return None

    def select_cuts(self, pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef):
        """(added pool positions in order, terminator code, depths) or None if the engine has no
        selection kernel."""
        return None


class HipBackend(LPBackend):
    """Sends the solve to libmipx.so on the MI355X (no fallback)."""

    MAX_RESIDENT_ROWSETS = 64

    def __init__(self, device=None):
        self._device = device
        self._ctx = None
        self._problems = OrderedDict()   # row-set key -> resident Problem, least recently used first

    def _context(self):
        if self._ctx is None:
            from simple_mip_solver_amd import _ffi
            self._ctx = _ffi.default_context() if self._device is None else \
                _ffi.Context(self._device)
        return self._ctx

    def _problem(self, A, b, c, cache_key):
        from simple_mip_solver_amd import _ffi
        p = self._problems.get(cache_key)
        if p is not None:
            self._problems.move_to_end(cache_key)
            return p
        # A bounded number of row sets stays resident.  Eviction only drops the cache's reference:
        # a live _ffi.Tree (BranchAndBound._native) or an LP in flight may still hold the Problem's
        # raw handle, so the device buffers are freed by Problem.__del__ once nobody refers to it,
        # never by an explicit close() here.  Least recently used goes first (the root row set of
        # a tree is hit by every cut-free node and must outlive the per-round row sets).
        while len(self._problems) >= self.MAX_RESIDENT_ROWSETS:
            self._problems.popitem(last=False)
        p = _ffi.Problem(self._context(), A, b, c)
        self._problems[cache_key] = p
        return p

    def solve(self, A, b, c, l, u, vstat, max_iter, cache_key):
        """l, u: (batch, n); vstat (batch, n+m) or None.  Returns dict of batch arrays."""
        return self._problem(A, b, c, cache_key).solve_batch(l, u, vstat, max_iter)

    def gomory(self, A, b, c, l, u, vstat, x, integer_indices, max_term, cache_key):
        return self._problem(A, b, c, cache_key).gomory_batch(
            l[None], u[None], vstat[None], x[None], integer_indices, max_term)[0]

    def select_cuts(self, pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef):
        from simple_mip_solver_amd import _ffi
        return _ffi.select_cuts(self._context(), pi, pi0, x, max_nonzero_coefs, min_cut_depth,
                                cos_parallel, max_abs_coef)


_backend = None


def get_backend():
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(backend):
    """Install an LPBackend (tests use this to run host logic where no GPU exists)."""
    global _backend
    assert backend is None or isinstance(backend, LPBackend), 'backend must be an LPBackend'
    _backend = backend


class _RowSet:
    """Immutable snapshot (A, b, c) shared by every node LP that has the same rows."""

    _next_key = 0

    def __init__(self, A, b, c):
        self.A = np.ascontiguousarray(A, dtype=np.float64)
        self.b = np.ascontiguousarray(b, dtype=np.float64)
        self.c = np.ascontiguousarray(c, dtype=np.float64)
        _RowSet._next_key += 1
        self.key = _RowSet._next_key


class DenseLP:
    """Stand-in for CyClpSimplex restricted to what the node hot path uses (plus multi-block
    models and free columns for the cut-generating LP)."""

    def __init__(self):
        self.logLevel = 0
        self.maxNumIteration = None
        self.iteration = 0
        self.variables = []
        self.constraints = []
        self.variablesLower = np.zeros(0)
        self.variablesUpper = np.zeros(0)
        self._objective = np.zeros(0)
        self._status = None
        self._obj_value = None
        self._x = None
        self._row_duals = None
        self._col_duals = None
        self._var_status = None   # Clp codes per column
        self._row_status = None   # Clp codes per row
        self._rowset = None       # cached engine form of the rows
        self._rowmap = None       # (constraint index, row in block, sign) per engine row
        self._solved_sig = None   # what the stored optimal solution belongs to (see dual())

    # ---- model building ------------------------------------------------------------------
    def addVariable(self, name, dim):
        v = Variable(name, dim)
        v._lp, v._offset = self, self.nVariables
        self.variables.append(v)
        # Clp's defaults for a new column: 0 <= x <= +inf, zero cost
        self.variablesLower = np.concatenate([self.variablesLower, np.zeros(v.dim)])
        self.variablesUpper = np.concatenate([self.variablesUpper, np.full(v.dim, COIN_INFINITY)])
        self._objective = np.concatenate([self._objective, np.zeros(v.dim)])
        self._invalidate()
        return v

    def getVarByName(self, name):
        for v in self.variables:
            if v.name == name:
                return v
        raise KeyError(name)

    def __iadd__(self, spec):
        if isinstance(spec, BoundSpec):
            assert spec.var in self.variables and len(self.variables) == 1
            self.variablesLower = np.where(np.isneginf(spec.lower), -COIN_INFINITY, spec.lower)
            self.variablesUpper = np.where(np.isposinf(spec.upper), COIN_INFINITY, spec.upper)
            spec.var._pending = None
        elif isinstance(spec, Constraint):
            self.addConstraint(spec)
        else:
            raise TypeError('can only add bounds or constraints to an LP')
        return self

    def addConstraint(self, constraint, name=None):
        assert isinstance(constraint, Constraint), 'expected lower <= coefs * x <= upper'
        for v in constraint.variables:
            v._pending = None
        c = Constraint(constraint.variables[0], constraint._coefs, constraint.lower,
                       constraint.upper, name if name is not None else
                       (constraint.name or f'R_{len(self.constraints)}'), constraint._extra)
        self.constraints.append(c)
        if self._row_status is not None:
            # a new row enters with its slack basic, as Clp does
            self._row_status = np.concatenate([self._row_status, np.ones(c.rows, np.int8)])
        self._invalidate()

    def removeConstraint(self, name):
        for k, c in enumerate(self.constraints):
            if c.name == name:
                start = sum(cc.rows for cc in self.constraints[:k])
                del self.constraints[k]
                if self._row_status is not None:
                    self._row_status = np.delete(self._row_status, np.s_[start:start + c.rows])
                self._invalidate()
                return
        raise Exception(f'Constraint "{name}" does not exist')

    def _invalidate(self):
        self._rowset = None
        self._rowmap = None

    # ---- dimensions / data views ------------------------------------------------------------
    @property
    def nVariables(self):
        return sum(v.dim for v in self.variables)

    nCols = nVariables

    @property
    def nConstraints(self):
        return sum(c.rows for c in self.constraints)

    @property
    def objective(self):
        return self._objective

    @objective.setter
    def objective(self, values):
        vals = np.asarray(values, dtype=np.float64).reshape(-1)
        assert vals.size == self.nVariables, 'objective length must match the variables'
        self._objective = vals.copy()
        self._invalidate()

    @property
    def constraintsLower(self):
        if not self.constraints:
            return np.zeros(0)
        lo = np.concatenate([c.lower for c in self.constraints])
        return np.where(np.isneginf(lo), -COIN_INFINITY, lo)

    @property
    def constraintsUpper(self):
        if not self.constraints:
            return np.zeros(0)
        up = np.concatenate([c.upper for c in self.constraints])
        return np.where(np.isposinf(up), COIN_INFINITY, up)

    @property
    def coefMatrix(self):
        from scipy.sparse import csc_matrix
        return csc_matrix(self.dense_rows())

    def dense_rows(self):
        n = self.nVariables
        if not self.constraints:
            return np.zeros((0, n))
        return np.vstack([c.dense(n) for c in self.constraints])

    @staticmethod
    def getCoinInfinity():
        return COIN_INFINITY

    # ---- basis -------------------------------------------------------------------------------
    def getBasisStatus(self):
        n, m = self.nVariables, self.nConstraints
        vs = np.full(n, 3, np.int8) if self._var_status is None else self._var_status.copy()
        rs = np.full(m, 1, np.int8) if self._row_status is None else self._row_status.copy()
        return vs, rs

    def setBasisStatus(self, var_status, row_status):
        vs = np.asarray(var_status, dtype=np.int8).reshape(-1)
        rs = np.asarray(row_status, dtype=np.int8).reshape(-1)
        assert vs.size == self.nVariables and rs.size == self.nConstraints, \
            'basis status must match the LP dimensions'
        self._var_status, self._row_status = vs.copy(), rs.copy()

    # ---- solve -------------------------------------------------------------------------------
    def _engine_form(self):
        """Rows rewritten as A x >= b (what the engine solves): a row with only an upper bound
        is negated, a ranged/equality row is entered twice."""
        if self._rowset is None:
            n = self.nVariables
            rows, rhs, rowmap = [], [], []
            for ci, c in enumerate(self.constraints):
                full = c.dense(n)
                for r in range(c.rows):
                    lo, up = c.lower[r], c.upper[r]
                    has_lo = lo > -COIN_INFINITY / 2 and not np.isneginf(lo)
                    has_up = up < COIN_INFINITY / 2 and not np.isposinf(up)
                    if has_lo or not has_up:
                        rows.append(full[r]); rhs.append(lo if has_lo else -np.inf)
                        rowmap.append((ci, r, 1.0))
                    if has_up:
                        rows.append(-full[r]); rhs.append(-up)
                        rowmap.append((ci, r, -1.0))
            A = np.array(rows, dtype=np.float64).reshape(len(rows), n)
            keep = [k for k, v in enumerate(rhs) if not np.isneginf(v)]  # free rows never bind
            self._rowmap = [rowmap[k] for k in keep]
            self._rowset = _RowSet(A[keep], np.array(rhs, dtype=np.float64)[keep], self._objective)
        return self._rowset

    def _bounds(self):
        l = np.where(self.variablesLower <= -COIN_INFINITY / 2, -np.inf, self.variablesLower)
        u = np.where(self.variablesUpper >= COIN_INFINITY / 2, np.inf, self.variablesUpper)
        return l.astype(np.float64), u.astype(np.float64)

    def _warm_start(self, rowset):
        """Engine status vector (n structural + one per engine row) or None for a cold start."""
        if self._var_status is None:
            return None
        n = self.nVariables
        offsets = np.cumsum([0] + [c.rows for c in self.constraints])
        rstat = np.ones(len(self._rowmap), np.int8)
        for k, (ci, r, sign) in enumerate(self._rowmap):
            code = self._row_status[offsets[ci] + r] if self._row_status is not None else 1
            # Clp's row status refers to the row activity: "at upper" of a <= row is its slack at 0;
            # of the two engine rows of a ranged / equality row only the tight side is nonbasic
            tight = (code == 3 and sign > 0) or (code == 2 and sign < 0) or \
                    (code not in (1, 2, 3))
            rstat[k] = 3 if tight else 1
        return np.concatenate([self._var_status.astype(np.int8), rstat])[None]

    def dual(self):
        """Solve with the dual simplex engine (the reference's `lp.dual()`, base_node.py:273)."""
        rs = self._engine_form()
        l, u = self._bounds()
        max_iter = int(self.maxNumIteration) if self.maxNumIteration else 0
        if not np.all(np.isfinite(l)):
            return self._dual_with_free_columns(rs, l, u, max_iter)
        # Re-solving an LP that has not changed since it was solved to optimality (same rows,
        # bounds and basis) is a no-op: the reference does exactly that once per fractional node
        # when a cut round adds nothing (base_node.py:317-319).  Keep the solution, skip the GPU.
        sig = self._solved_sig
        if sig is not None and self._status == 0 and sig[0] == rs.key and sig[1] == max_iter and \
                sig[2] is self._var_status and np.array_equal(sig[3], l) and np.array_equal(sig[4], u):
            return self._status
        res = get_backend().solve(rs.A, rs.b, rs.c, l[None], u[None], self._warm_start(rs),
                                  max_iter, rs.key)
        self._store(res, 0)
        self._solved_sig = (rs.key, max_iter, self._var_status, l, u)
        return self._status

    primal = dual  # the engine has one algorithm; results (status/objective/solution) are the same

    FREE_BOX = (1e6, 1024.0)  # half-widths of the boxes put around columns without a bound

    def _dual_with_free_columns(self, rs, l, u, max_iter):
        """Columns without a lower bound (the cut-generating LP's pi, pi0) are outside the node
        hot path, and the engine wants every l finite.  They are shifted into a finite box:
        x_j = x'_j + centre_j - K (free) or x_j = u_j - x'_j (upper bound only), 0 <= x' <= 2K.
        First a wide box around 0; then, warm-started from that basis, a narrow one around the
        solution found, which removes the rounding the wide shift costs.  A box bound that is
        active at the optimum with a nonzero reduced cost means the LP is unbounded in that
        direction: status 2."""
        m_e, n = rs.A.shape
        kinds = np.where(np.isfinite(l), 0, np.where(np.isfinite(u), 1, 2))
        k1, k2 = kinds == 1, kinds == 2

        def solve_boxed(centre, K, ws):
            A_e, c_e = rs.A.copy(), rs.c.copy()
            lo, up = l.copy(), u.copy()
            # x = shift + sgn * x'
            shift = np.where(k1, u, np.where(k2, centre - K, 0.0))
            A_e[:, k1] *= -1.0
            c_e[k1] *= -1.0
            b_e = rs.b - rs.A[:, kinds != 0] @ shift[kinds != 0]
            offset = float(rs.c[kinds != 0] @ shift[kinds != 0])
            lo[kinds != 0], up[kinds != 0] = 0.0, 2.0 * K
            gen_rs = _RowSet(A_e, b_e, c_e)
            res = get_backend().solve(gen_rs.A, gen_rs.b, gen_rs.c, lo[None], up[None], ws, max_iter,
                                      gen_rs.key)
            xe = np.asarray(res['x'][0], dtype=np.float64)
            ve = np.asarray(res['vstat'][0], np.int8)
            x = np.where(k1, u - xe, np.where(k2, xe + centre - K, xe))
            y = np.asarray(res['y'][0], dtype=np.float64)
            d = rs.c - rs.A.T @ y if m_e else rs.c.copy()
            status = int(res['status'][0])
            stuck = (kinds != 0) & (ve[:n] != 1) & ((xe >= 2.0 * K) | (k2 & (xe <= 0.0))) & (np.abs(d) > 1e-7)
            return dict(status=status, stuck=bool(np.any(stuck)), x=x, y=y, ve=ve,
                        obj=float(res['obj'][0]) + offset, iters=int(res['iters'][0]))

        ws = None
        if self._var_status is not None:
            ws = self._warm_start(rs)[0].copy()
            ws[:n][k1] = np.where(ws[:n][k1] == 1, 1, np.where(ws[:n][k1] == 2, 3, 2))
            ws = ws[None]
        first = solve_boxed(np.zeros(n), float(self.FREE_BOX[0]), ws)
        out = first
        if first['status'] == 0 and not first['stuck'] and max_iter == 0:
            second = solve_boxed(np.round(first['x']), float(self.FREE_BOX[1]), first['ve'][None])
            if second['status'] == 0 and not second['stuck']:
                second['iters'] += first['iters']
                out = second
        status = 2 if (out['status'] == 0 and out['stuck']) else out['status']
        ve = out['ve']
        vs = ve[:n].copy()
        vs[k1] = np.where(ve[:n][k1] == 1, 1, np.where(ve[:n][k1] == 2, 3, 2))
        mapped = dict(status=np.array([status]), obj=np.array([out['obj']]), x=out['x'][None],
                      y=out['y'][None], iters=np.array([out['iters']]),
                      vstat=np.concatenate([vs, ve[n:]])[None])
        self._store(mapped, 0)
        self._solved_sig = None
        return self._status

    def gomory_rows(self, x, integer_indices, max_term):
        """GMI cuts of the current (optimal) basis from the engine's cut kernel, in the LP's own
        row numbering; None if the backend has none or the rows are not all plain `>=` rows."""
        rs = self._engine_form()
        if self._var_status is None or any(sign < 0 for _, _, sign in self._rowmap) or \
                len(self._rowmap) != self.nConstraints:
            return None
        l, u = self._bounds()
        return get_backend().gomory(rs.A, rs.b, rs.c, l, u, self._warm_start(rs)[0],
                                    np.asarray(x, dtype=np.float64), integer_indices, max_term, rs.key)

    def _store(self, res, k):
        n = self.nVariables
        self._status = int(res['status'][k])
        self._obj_value = float(res['obj'][k])
        self._x = np.array(res['x'][k], dtype=np.float64)
        self.iteration = int(res['iters'][k])
        vs = np.asarray(res['vstat'][k], dtype=np.int8)
        self._var_status = vs[:n].copy()
        offsets = np.cumsum([0] + [c.rows for c in self.constraints])
        row_status = np.ones(self.nConstraints, np.int8)
        duals = np.zeros(self.nConstraints)
        y = np.asarray(res['y'][k], dtype=np.float64)
        for e, (ci, r, sign) in enumerate(self._rowmap):
            pos = offsets[ci] + r
            if vs[n + e] != 1:
                # slack at zero: the row is tight at its lower (>= row) or upper (<= row) side
                row_status[pos] = 3 if sign > 0 else 2
            duals[pos] += sign * y[e]
        self._row_status = row_status
        self._row_duals = duals
        # reduced costs of the columns: d = c - A'y over the engine's rows
        rs = self._rowset
        self._col_duals = rs.c - rs.A.T @ y if rs.A.shape[0] else rs.c.copy()

    def getStatusCode(self):
        assert self._status is not None, 'LP has not been solved'
        return self._status

    @property
    def objectiveValue(self):
        return self._obj_value

    @property
    def primalVariableSolution(self):
        if len(self.variables) == 1:
            return {self.variables[0].name: self._x}
        return {v.name: self._x[v._offset:v._offset + v.dim] for v in self.variables}

    @property
    def dualVariableSolution(self):
        """Reduced costs by variable block (CyClpSimplex.dualVariableSolution)."""
        return {v.name: self._col_duals[v._offset:v._offset + v.dim] for v in self.variables}

    @property
    def dualConstraintSolution(self):
        out, pos = {}, 0
        for c in self.constraints:
            out[c.name] = self._row_duals[pos:pos + c.rows]
            pos += c.rows
        return out

    # ---- cheap child creation (base_node.py:592-608 rebuilds the model row by row) -----------
    def copy_with_bounds(self, lower, upper):
        """New LP sharing this one's rows and objective, with its own bounds and basis."""
        child = DenseLP()
        assert len(self.variables) == 1, 'children are made of single-block LPs'
        child.variables = [Variable(v.name, v.dim) for v in self.variables]
        child.variables[0]._lp = child
        vmap = dict(zip(self.variables, child.variables))
        child.constraints = [Constraint(vmap[c.variables[0]], c._coefs, c.lower, c.upper, c.name)
                             for c in self.constraints]
        child.variablesLower = np.array(lower, dtype=np.float64)
        child.variablesUpper = np.array(upper, dtype=np.float64)
        child._objective = self._objective.copy()
        child._rowset, child._rowmap = self._engine_form(), self._rowmap
        if self._var_status is not None:
            child._var_status = self._var_status.copy()
            child._row_status = self._row_status.copy()
        return child


# the name the reference's code asserts on (base_node.py:49)
CyClpSimplex = DenseLP
