"""Tiny LPs with infinite upper bounds, engine vs oracle (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from oracle import oracle
INF = np.inf
ctx = _ffi.default_context()
A = np.array([[-1., 0, 0], [0, -1, 0], [0, 0, -1]]); b = np.array([-1., -1, -1]); c = np.array([-1., -1, 0])
l = np.zeros(3); u = np.full(3, INF)
p = _ffi.Problem(ctx, A, b, c)
g = p.solve_batch(l[None], u[None])
o = oracle.lp_solve(A, b, c, l, u)
print('engine', {k: v.tolist() for k, v in g.items()})
print('oracle', o)
res, dump = _ffi.debug_dump(p, l, u)
for k, v in dump.items():
    print(k, v.tolist())
