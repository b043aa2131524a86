import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from simple_mip_solver_amd import _ffi
ctx = _ffi.default_context()
A = np.array([[-4., -1.], [-1., -4.], [-1., 1.], [-0.36363636363636365, -1.], [-1., -0.6666666666666666]])
b = np.array([-28., -27., -1., -7.090909090909091, -9.])
c = np.array([-2., -5.]); l = np.zeros(2); u = np.full(2, np.inf)
vstat = np.array([1, 1, 3, 3, 1, 1, 1], np.int8)
p = _ffi.Problem(ctx, A, b, c)
np.set_printoptions(linewidth=200)
for mi in (0,):
    g, gd = _ffi.debug_dump(p, l, u, vstat, max_iter=mi)
    o, od = O.debug_dump(A, b, c, l, u, vstat, max_iter=mi)
    print('gpu', {k: g[k] for k in ('status', 'iters', 'npivots', 'x', 'obj', 'vstat')})
    print('ora', {k: o[k] for k in ('status', 'iters', 'npivots', 'x', 'obj', 'vstat')})
    for key in gd:
        print(key, 'GPU', gd[key].ravel(), '\n    ORA', od[key].ravel())
