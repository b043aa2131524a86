"""K1c (one cold LP over the chip) against K1b (one workgroup) on cold roots: results bit for bit, and the time.
usage: root_coop.py [n m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(300, 150), (512, 256), (600, 70), (1024, 512)]
ctx = _ffi.default_context()
for n, m in shapes:
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
    out = {}
    for mode in ('1', '0'):
        os.environ['MIPX_NO_COOP_ROOT'] = mode
        p = _ffi.Problem(ctx, A, b, c)
        p.solve_batch(l[None], u[None])      # (first call: allocations)
        ctx.sync(); t0 = time.perf_counter()
        r = p.solve_batch(l[None], u[None])
        ctx.sync(); out[mode] = (r, time.perf_counter() - t0)
        p.close()
    os.environ.pop('MIPX_NO_COOP_ROOT', None)
    (rb, tb), (rc, tc) = out['1'], out['0']
    same = all(np.array_equal(rb[k], rc[k]) for k in ('status', 'iters', 'npivots', 'vstat', 'x', 'obj', 'y'))
    print(f'{n}x{m}: status {rc["status"][0]} iters {rc["iters"][0]} | K1b {tb * 1e3:.1f} ms, K1c {tc * 1e3:.1f} ms | identical: {same}', flush=True)
