"""Experiment: time to a proven optimum, two phases (depth first -> best first with the incumbent), a big pool.
usage: tto_experiment.py n m dfs_seconds limit_seconds pool_log2 [dive]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m = int(sys.argv[1]), int(sys.argv[2])
dfs_s, limit, plog = float(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
dive = int(sys.argv[6]) if len(sys.argv) > 6 else 8
ctx = _ffi.default_context()
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)
t0 = time.perf_counter()
t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', search_rule='depth first', max_batch=1024, pool_capacity=1 << 21)
t.set_anchor_mode(True); t.set_dive(dive)
s = None
last = None
while time.perf_counter() - t0 < dfs_s:
    s = t.solve(mip_gap=1e-4, frontier_batch=1024, max_steps=10)
    if s['primal_bound'] != last:
        last = s['primal_bound']
        print('  dfs %.3f s incumbent %s nodes %d' % (time.perf_counter() - t0, last, s['evaluated_nodes']), flush=True)
    if s['status'] != 4:
        break
t1 = time.perf_counter()
pb, n1 = s['primal_bound'], s['evaluated_nodes']
print('phase 1 (depth first): %.3f s, %d nodes, status %s, incumbent %s dual %s' % (t1 - t0, n1, _ffi.TREE_STATUS[s['status']], pb, s['dual_bound']), flush=True)
done = s['status'] == 1
t.close()
if not done:
    t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', max_batch=8192, pool_capacity=1 << plog)
    t.set_anchor_mode(True); t.set_dive(dive)
    if pb < float('inf'):
        t.set_primal_bound(pb)
    while True:
        s = t.solve(mip_gap=1e-4, frontier_batch=8192, max_seconds=2.0)
        t2 = time.perf_counter()
        print('  bfs %.3f s, %d nodes, status %s, primal %s dual %s gap %s open %d' % (
            t2 - t1, s['evaluated_nodes'], _ffi.TREE_STATUS[s['status']], s['primal_bound'], s['dual_bound'], s['gap'], s['open_nodes']), flush=True)
        if s['status'] != 4 or t2 - t1 > limit or s['pool_exhausted']:
            break
    print('total %.3f s, %d nodes' % (t2 - t0, n1 + s['evaluated_nodes']))
