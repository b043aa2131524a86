"""Problem container handed to BranchAndBound.

Stands in for `coinor.cuppy.milpInstance.MILPInstance`, of which the reference uses the attributes
`A, b, c, l, u, sense, integerIndices, lp` (simple_mip_solver/algorithms/base_algorithm.py:53-59)
and the two constructor forms `MILPInstance(A=, b=, c=, l=, u=, sense=[..], integerIndices=,
numVars=)` (test_simple_mip_solver/example_models.py:24-25) and `MILPInstance(file_name=...)`
(test_simple_mip_solver/helpers.py:42).  The MPS reader covers the subset those files use
(N/L/G/E rows, COLUMNS with integer MARKERs, RHS, RANGES ignored, UP/LO/FX/MI/PL/BV/UI/LI bounds).
"""
import numpy as np

from simple_mip_solver_amd.lp import COIN_INFINITY, CyLPArray, DenseLP


class MILPInstance:
    def __init__(self, A=None, b=None, c=None, l=None, u=None, sense=None, integerIndices=None,
                 numVars=None, file_name=None):
        if file_name is not None:
            A, b, c, l, u, sense, integerIndices = read_mps(file_name)
            numVars = len(c)
        assert A is not None and b is not None and c is not None, 'need A, b and c (or file_name)'
        assert sense is not None and len(sense) == 2 and sense[0] in ('Min', 'Max') and \
            sense[1] in ('<=', '>='), "sense is ['Min'|'Max', '<='|'>=']"
        self.A = np.asarray(A, dtype=np.float64)
        if self.A.ndim == 1:
            self.A = self.A.reshape(1, -1)
        self.numCons, n = self.A.shape
        self.numVars = int(numVars) if numVars is not None else n
        assert self.numVars == n, 'numVars must match the columns of A'
        self.b = CyLPArray(np.asarray(b, dtype=np.float64).reshape(-1))
        self.c = CyLPArray(np.asarray(c, dtype=np.float64).reshape(-1))
        assert self.b.size == self.numCons and self.c.size == n, 'b and c must match A'
        self.l = CyLPArray(np.zeros(n) if l is None else np.asarray(l, dtype=np.float64).reshape(-1))
        self.u = CyLPArray(np.full(n, COIN_INFINITY) if u is None else
                           np.asarray(u, dtype=np.float64).reshape(-1))
        self.sense = sense[1]
        self.integerIndices = [int(i) for i in (integerIndices or [])]

        lp = DenseLP()
        x = lp.addVariable('x', n)
        lp += self.l <= x <= self.u
        if self.sense == '>=':
            lp += np.asarray(self.A) * x >= self.b
        else:
            lp += np.asarray(self.A) * x <= self.b
        # everything is solved as a minimisation (cuppy flips the objective of a Max problem)
        lp.objective = self.c if sense[0] == 'Min' else -self.c
        self.lp = lp


def read_mps(path):
    """Parse a (free or fixed format, names without blanks) MPS file into dense arrays."""
    rows, row_type, obj_row = [], {}, None
    cols, col_index, entries = [], {}, []
    rhs, bounds, integer = {}, [], set()
    section, in_int = None, False
    with open(path) as f:
        for raw in f:
            line = raw.rstrip('\n')
            if not line.strip() or line.lstrip().startswith('*'):
                continue
            if not line[0].isspace():
                section = line.split()[0].upper()
                continue
            tok = line.split()
            if section == 'ROWS':
                kind, name = tok[0].upper(), tok[1]
                if kind == 'N':
                    obj_row = obj_row or name
                else:
                    row_type[name] = kind
                    rows.append(name)
            elif section == 'COLUMNS':
                if len(tok) >= 3 and tok[1].strip("'").upper() == 'MARKER':
                    in_int = 'INTORG' in tok[2].upper()
                    continue
                name = tok[0]
                if name not in col_index:
                    col_index[name] = len(cols)
                    cols.append(name)
                if in_int:
                    integer.add(col_index[name])
                for k in range(1, len(tok) - 1, 2):
                    entries.append((tok[k], col_index[name], float(tok[k + 1])))
            elif section == 'RHS':
                for k in range(1, len(tok) - 1, 2):
                    rhs[tok[k]] = float(tok[k + 1])
            elif section == 'BOUNDS':
                kind = tok[0].upper()
                name = tok[2]
                val = float(tok[3]) if len(tok) > 3 else None
                bounds.append((kind, name, val))
    n, m = len(cols), len(rows)
    row_index = {r: i for i, r in enumerate(rows)}
    A = np.zeros((m, n))
    c = np.zeros(n)
    for rname, j, val in entries:
        if rname == obj_row:
            c[j] = val
        elif rname in row_index:
            A[row_index[rname], j] = val
    b = np.array([rhs.get(r, 0.0) for r in rows])
    l = np.zeros(n)
    u = np.full(n, COIN_INFINITY)
    for kind, name, val in bounds:
        j = col_index[name]
        if kind in ('UP', 'UI'):
            u[j] = val
            if kind == 'UI':
                integer.add(j)
        elif kind in ('LO', 'LI'):
            l[j] = val
            if kind == 'LI':
                integer.add(j)
        elif kind == 'FX':
            l[j] = u[j] = val
        elif kind == 'BV':
            l[j], u[j] = 0.0, 1.0
            integer.add(j)
        elif kind == 'MI':
            l[j] = -COIN_INFINITY
        elif kind == 'PL':
            u[j] = COIN_INFINITY
    kinds = {row_type[r] for r in rows}
    assert kinds <= {'L'} or kinds <= {'G'}, 'rows must all be <= or all be >= (one sense per model)'
    sense = ['Min', '<=' if kinds <= {'L'} and kinds else '>=']
    return A, b, c, l, u, sense, sorted(integer)


def write_mps(path, A, b, c, l, u, sense, integer_indices, name='mipx'):
    """Write min/max c'x, Ax (<=|>=) b, l <= x <= u, x_I integer in the MPS subset the reference's
    fixtures use (N/L/G rows, RHS, UP/UI/LO/FX/MI bounds; one sense per model), such that
    `read_mps` returns the same arrays.  A 'Max' objective is written negated (MPS minimises)."""
    A = np.asarray(A, dtype=np.float64)
    A = A.reshape(-1, len(c)) if A.ndim != 2 else A
    m, n = A.shape
    b = np.asarray(b, dtype=np.float64).reshape(m)
    c = np.asarray(c, dtype=np.float64).reshape(n)
    c = c if sense[0] == 'Min' else -c
    l = np.zeros(n) if l is None else np.asarray(l, dtype=np.float64).reshape(n)
    u = np.full(n, COIN_INFINITY) if u is None else np.asarray(u, dtype=np.float64).reshape(n)
    ints = set(int(i) for i in integer_indices)
    kind = 'L' if sense[1] == '<=' else 'G'
    num = lambda v: repr(float(v))
    with open(path, 'w') as f:
        f.write(f'NAME          {name}\nROWS\n N  OBJROW\n')
        for i in range(m):
            f.write(f' {kind}  R_{i}\n')
        f.write('COLUMNS\n')
        for j in range(n):
            wrote = False
            if c[j] != 0:
                f.write(f'    x_{j}  OBJROW  {num(c[j])}\n')
                wrote = True
            for i in np.nonzero(A[:, j])[0]:
                f.write(f'    x_{j}  R_{i}  {num(A[i, j])}\n')
                wrote = True
            if not wrote:  # a column must appear to exist
                f.write(f'    x_{j}  OBJROW  0.0\n')
        f.write('RHS\n')
        for i in range(m):
            if b[i] != 0:
                f.write(f'    RHS  R_{i}  {num(b[i])}\n')
        f.write('BOUNDS\n')
        for j in range(n):
            lo_inf, up_inf = l[j] <= -COIN_INFINITY / 2, u[j] >= COIN_INFINITY / 2
            if j in ints:
                # integrality travels on the bound type, as in the fixtures CyLP wrote (UI)
                f.write(f' UI BOUND  x_{j}  {num(u[j] if not up_inf else COIN_INFINITY)}\n')
                if l[j] != 0:
                    f.write(f' {"MI" if lo_inf else "LO"} BOUND  x_{j}  {"" if lo_inf else num(l[j])}\n')
                continue
            if not lo_inf and not up_inf and l[j] == u[j]:
                f.write(f' FX BOUND  x_{j}  {num(l[j])}\n')
                continue
            if lo_inf:
                f.write(f' MI BOUND  x_{j}\n')
            elif l[j] != 0:
                f.write(f' LO BOUND  x_{j}  {num(l[j])}\n')
            if not up_inf:
                f.write(f' UP BOUND  x_{j}  {num(u[j])}\n')
        f.write('ENDATA\n')
