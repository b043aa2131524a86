"""Time to first incumbent on the metric's 256 x 128 instance: search rule x frontier batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
ctx = _ffi.default_context()
n, m = 256, 128
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)
for search, B, rule in (('depth first', 1, 'pseudo cost'), ('depth first', 64, 'pseudo cost'), ('depth first', 1024, 'pseudo cost'),
                        ('depth first', 64, 'most fractional'), ('best first', 1024, 'pseudo cost')):
    t = _ffi.Tree(p, ints, l, u, branch_rule=rule, search_rule=search, max_batch=B, pool_capacity=1 << 21)
    if B > 1:
        t.set_anchor_mode(True); t.set_dive(True)
    t0 = time.perf_counter()
    first = None
    while time.perf_counter() - t0 < 15.0:
        s = t.solve(mip_gap=1e-4, frontier_batch=B, max_steps=20 if B > 1 else 200)
        if first is None and s['primal_bound'] < float('inf'):
            first = (time.perf_counter() - t0, s['evaluated_nodes'], s['primal_bound'])
        if s['status'] != 4:
            break
    print(search, B, rule, 'first incumbent', first, '| after %.1f s: primal %s dual %.3f gap %s nodes %d status %s' % (
        time.perf_counter() - t0, s['primal_bound'], s['dual_bound'], s['gap'], s['evaluated_nodes'], _ffi.TREE_STATUS[s['status']]), flush=True)
    t.close()
