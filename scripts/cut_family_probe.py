"""Which generator families add cuts below the root at a shape (engine with cut rounds): prints the GMIC totals
after a short ramp.  usage: cut_family_probe.py n m seed density unboxed [nodes]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m, seed, density, unboxed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
target = int(sys.argv[6]) if len(sys.argv) > 6 else 60
ctx = _ffi.default_context()
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
if unboxed:
    u = np.full(n, np.inf)
p = _ffi.Problem(ctx, A, b, c)
t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', max_batch=64, pool_capacity=1 << 13,
              cut_params=dict(max_abs_coef=1000.0 * float(np.max(np.abs(A))), exact_tableau=0))
t.set_anchor_mode(True)
st = t.stats()
while st['open_nodes'] < target and st['status'] in (0, 4):
    st = t.solve(mip_gap=0.0, frontier_batch=32, max_steps=1)
ids, ncut, _, _ = t.peek_cuts(st['open_nodes'])
print(sys.argv[1:], 'kernel', _ffi.kernel_name(m, n), 'open', st['open_nodes'], 'nodes with cut rows', int((ncut > 0).sum()), t.cut_stats())
