"""Time to a proven optimum in two phases on the engine: depth first (the reference's DepthFirstSearchNode
semantics) for a short while to get an incumbent, then best first with that incumbent installed as the
initial primal bound.  usage: tto_two_phase.py n m [dfs_seconds] [limit_seconds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m = int(sys.argv[1]), int(sys.argv[2])
dfs_s = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
limit = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
ctx = _ffi.default_context()
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)
t0 = time.perf_counter()
t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', search_rule='depth first', max_batch=1024, pool_capacity=1 << 21)
t.set_anchor_mode(True); t.set_dive(4)
s = None
while time.perf_counter() - t0 < dfs_s:
    s = t.solve(mip_gap=1e-4, frontier_batch=1024, max_steps=10)
    if s['status'] != 4:
        break
t1 = time.perf_counter()
pb, x, n1 = s['primal_bound'], (t.solution() if s['primal_bound'] < float('inf') else None), s['evaluated_nodes']
pc = t.pseudo_cost_arrays()
print('phase 1 (depth first): %.3f s, %d nodes, status %s, incumbent %s' % (t1 - t0, n1, _ffi.TREE_STATUS[s['status']], pb), flush=True)
done = s['status'] == 1
t.close()
if not done:
    t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', max_batch=8192, pool_capacity=1 << 23)
    t.set_anchor_mode(True); t.set_dive(4)
    if pb < float('inf'):
        t.set_primal_bound(pb)
    s = t.solve(mip_gap=1e-4, frontier_batch=8192, max_seconds=limit)
    t2 = time.perf_counter()
    print('phase 2 (best first): %.3f s, %d nodes, status %s, primal %s dual %s gap %s open %d' % (
        t2 - t1, s['evaluated_nodes'], _ffi.TREE_STATUS[s['status']], s['primal_bound'], s['dual_bound'], s['gap'], s['open_nodes']), flush=True)
    print('total %.3f s, %d nodes' % (t2 - t0, n1 + s['evaluated_nodes']))
