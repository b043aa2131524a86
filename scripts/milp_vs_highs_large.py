"""The engine's FAST path (8 192 nodes per step, device finish, anchors, plunge of depth 8, two phases as bench.py)
against HiGHS branch and cut on instances that close: same proven optimum.  usage: milp_vs_highs_large.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds
import bench
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
ctx = _ffi.default_context()
bad = 0
for n, m in [(60, 30), (80, 40), (100, 50), (120, 60)]:
    for seed in range(3):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
        integrality = np.zeros(n); integrality[ints] = 1
        t0 = time.time()
        h = milp(c, constraints=LinearConstraint(A, lb=b, ub=np.inf), bounds=Bounds(l, u), integrality=integrality,
                 options={'mip_rel_gap': 0.0, 'time_limit': 120})
        th = time.time() - t0
        out = bench.two_phase(ctx, A, b, c, l, u, ints, 8, dfs_seconds=0.5, limit=60.0, pool_log2=24, mip_gap=1e-9)
        ok = None
        if h.status == 0 and out['status'] == 'optimal':
            ok = abs(out['primal_bound'] - h.fun) <= 1e-6 * max(1, abs(h.fun))
        if ok is False: bad += 1
        print(f'{n}x{m} seed {seed}: engine {out["status"]} {out["primal_bound"]} in {out["seconds"]:.2f} s ({out["nodes"]} nodes) | '
              f'HiGHS status {h.status} {h.fun} ({th:.1f} s) {"OK" if ok else "MISMATCH" if ok is False else "?"}', flush=True)
print('mismatches:', bad)
