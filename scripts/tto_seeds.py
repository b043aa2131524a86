"""The time-to-optimal leg of bench.py (two phases) on other seeds of the metric's 256 x 128 config: does any close?
usage: tto_seeds.py [first_seed last_seed limit_seconds]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 6
limit = float(sys.argv[3]) if len(sys.argv) > 3 else 12.0
ctx = _ffi.default_context()
for seed in range(lo, hi + 1):
    A, b, c, l, u, ints = random_dense_milp_arrays(256, 128, seed=seed)
    out = bench.two_phase(ctx, A, b, c, l, u, ints, 8, dfs_seconds=2.0, limit=limit, pool_log2=24, marks=bench.GAP_MARKS)
    print('seed', seed, out['status'], 'time_to_optimal', out['time_to_optimal'], 'gap', out['gap'], 'nodes', out['nodes'],
          'primal', out['primal_bound'], 'dual', out['dual_bound'], {k: (v and round(v['seconds'], 2)) for k, v in out['time_to_gap'].items()}, flush=True)
