"""Config C2: root relaxations (cold start) of 1024 independent 64 vars x 32 rows instances in one
launch (mipx_lp_solve_multi, host buffers: PCIe-inclusive), against the oracle on one thread."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
from oracle import oracle as O
B, n, m = 1024, 64, 32
ctx = _ffi.default_context()
P = [random_dense_milp_arrays(n, m, seed=k) for k in range(B)]
A = np.stack([p[0] for p in P]); b = np.stack([p[1] for p in P]); c = np.stack([p[2] for p in P])
l = np.stack([p[3] for p in P]); u = np.stack([p[4] for p in P])
for rep in range(3):
    t0 = time.perf_counter()
    g = _ffi.solve_multi(ctx, A, b, c, l, u)
    dt = time.perf_counter() - t0
print('GPU %s: %d root LPs in %.2f ms (host buffers, PCIe inclusive) -> %.0f LP/s; mean pivots %.1f; status %s' % (
    _ffi.kernel_name(m, n), B, dt * 1e3, B / dt, g['npivots'].mean(), np.bincount(g['status'], minlength=4)))
t0 = time.perf_counter()
for k in range(128):
    o = O.lp_solve(A[k], b[k], c[k], l[k], u[k])
dt = time.perf_counter() - t0
print('oracle, 1 thread: %.0f LP/s' % (128 / dt))
