"""How far does the search get on the metric's own 256 x 128 instance in minutes rather than seconds?
(a) depth-first engine (the bench's time-to-optimal leg) for a long time; (b) best-first with the incumbent
of (a) installed: how fast the dual bound moves.  Prints a trace; not part of the bench."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 128
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
ctx = _ffi.default_context()
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)


def run(rule, B, primal=None, dive=4):
    t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', search_rule=rule, max_batch=B, pool_capacity=1 << 23)
    t.set_anchor_mode(True); t.set_dive(dive)
    if primal is not None:
        t.set_primal_bound(primal)
    t0 = time.perf_counter(); nxt = 5.0
    while True:
        s = t.solve(mip_gap=1e-4, frontier_batch=B, max_steps=50)
        el = time.perf_counter() - t0
        if el > nxt or s['status'] != 4 or el > secs:
            print(f'{rule} B={B} t={el:7.1f}s status={_ffi.TREE_STATUS[s["status"]]} primal={s["primal_bound"]:.6f} '
                  f'dual={s["dual_bound"]:.6f} gap={s["gap"]:.3e} nodes={s["evaluated_nodes"]} open={s["open_nodes"]}', flush=True)
            nxt += 5.0
        if s['status'] != 4 or el > secs:
            break
    pb = s['primal_bound']
    t.close()
    return pb


pb = run('depth first', 1024)
run('best first', 8192, primal=pb)
