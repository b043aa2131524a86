import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
ctx = _ffi.default_context()
for (n, m, seed) in [(100, 50, 0), (128, 64, 0), (160, 80, 0)]:
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    p = _ffi.Problem(ctx, A, b, c)
    t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', max_batch=8192, pool_capacity=1 << 23)
    t0 = time.perf_counter()
    s = t.solve(mip_gap=1e-4, frontier_batch=8192, max_seconds=40.0)
    print(n, m, seed, 'sec %.3f' % (time.perf_counter() - t0), _ffi.TREE_STATUS[s['status']], s['primal_bound'], s['dual_bound'], 'nodes', s['evaluated_nodes'], 'open', s['open_nodes'], flush=True)
    t.close(); p.close()
