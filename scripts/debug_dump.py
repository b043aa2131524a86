import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
ctx = _ffi.default_context()
n, m, seed = 256, 128, 0
A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=seed)
p = _ffi.Problem(ctx, A, b, c)
for k in (1, 2):
    g, gd = _ffi.debug_dump(p, l, u, max_iter=k)
    o, od = O.debug_dump(A, b, c, l, u, max_iter=k)
    print('k', k)
    for key in gd:
        eq = np.array_equal(gd[key], od[key])
        print('  ', key, 'equal' if eq else 'DIFF')
        if not eq:
            d = np.argwhere(gd[key] != od[key])
            print('     ndiff', len(d), 'first', d[:8].tolist())
            for idx in d[:5]:
                idx = tuple(idx)
                print('      ', idx, gd[key][idx], od[key][idx])
    print('  basic swapped: gpu nvar vs oracle', np.where(gd['nvar'] != od['nvar'])[0], np.where(gd['bvar'] != od['bvar'])[0])
