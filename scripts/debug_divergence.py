"""Find the first dual-simplex iteration where the HIP kernel and the oracle diverge."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays

ctx = _ffi.default_context()
for (n, m, seed) in [(64, 33, 0), (128, 64, 0), (100, 40, 0), (256, 128, 0)]:
    A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=seed)
    p = _ffi.Problem(ctx, A, b, c)
    print('case', n, m, _ffi.kernel_name(m, n))
    g = p.solve_batch(l[None], u[None]); o = O.lp_solve_batch(A, b, c, l[None], u[None])
    print('  full: gpu iters', g['iters'], 'oracle', o['iters'], 'obj', g['obj'], o['obj'])
    lo, hi = 1, int(max(g['iters'][0], o['iters'][0]))
    first = None
    for k in range(1, hi + 1):
        g = p.solve_batch(l[None], u[None], max_iter=k)
        o = O.lp_solve_batch(A, b, c, l[None], u[None], max_iter=k)
        if not (np.array_equal(g['vstat'], o['vstat']) and np.array_equal(g['x'], o['x'])):
            first = k
            break
    print('  first divergent iteration', first)
    if first:
        dv = np.where(g['vstat'][0] != o['vstat'][0])[0]
        print('  vstat diff idx', dv, g['vstat'][0][dv], o['vstat'][0][dv])
        dx = np.where(g['x'][0] != o['x'][0])[0]
        print('  x diff idx', dx[:10], g['x'][0][dx][:10], o['x'][0][dx][:10])
