"""Config C5 (1024 vars x 512 rows) on the frontier engine: K1b (tableau streamed from HBM)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m = 1024, 512
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dive = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # depth of the in-place dive (0: off)
ctx = _ffi.default_context()
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)
t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=128 * B)
t.set_anchor_mode(True)
t.set_dive(dive)
st = t.stats()
t0 = time.perf_counter()
while st['open_nodes'] < B or st['evaluated_nodes'] == 0:
    st = t.solve(mip_gap=0.0, frontier_batch=min(B, 256), max_steps=1)
if len(sys.argv) > 3 and sys.argv[3] != '0':
    t.reanchor(st['open_nodes'])
print('ramp-up: %d nodes, %d LPs, %.2f s, kernel %s' % (st['evaluated_nodes'], st['lp_solved'], time.perf_counter() - t0, _ffi.kernel_name(m, n)))
b0 = t.stats(); t0 = time.perf_counter()
st = t.solve(mip_gap=0.0, frontier_batch=B, max_steps=5)
dt = time.perf_counter() - t0
lps = st['lp_solved'] - b0['lp_solved']; piv = st['pivots'] - b0['pivots']; kms = st['kernel_ms'] - b0['kernel_ms']
bytes_per_pivot = 2 * 8 * (m + 1) * (n + m + 1)
print('5 steps: %d LPs in %.3f s -> %.0f LP/s end to end; kernel %.1f ms -> %.0f LP/s; %.1f pivots/LP; algorithmic %.2f TB/s (%.0f%% of 8 TB/s)' % (
    lps, dt, lps / dt, kms, lps / kms * 1e3, piv / lps, piv * bytes_per_pivot / (kms * 1e-3) / 1e12, piv * bytes_per_pivot / (kms * 1e-3) / 8e12 * 100))
print('dual bound', st['dual_bound'], 'open', st['open_nodes'], 'dives', st['dives'] - b0['dives'])
