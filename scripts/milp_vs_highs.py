"""The frontier engine against an independent MILP solver (scipy.optimize.milp = HiGHS) on families the parity tests
do not cover by construction: infinite upper bounds, mixed bounds (fixed variables, some infinite), both branching
rules, batched steps with the plunge.  usage: milp_vs_highs.py [max_n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
ctx = _ffi.default_context()
INF = np.inf
bad = 0
cases = []
for n, m in [(12, 6), (20, 10), (30, 15), (40, 20), (50, 25)]:
    for seed in range(4):
        for fam in ('boxed', 'unboxed', 'mixed'):
            cases.append((n, m, seed, fam))
for n, m, seed, fam in cases:
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
    rng = np.random.default_rng(77 + seed)
    if fam == 'unboxed':
        u = np.full(n, INF)
    elif fam == 'mixed':
        l, u = l.copy(), u.copy()
        fixed = rng.random(n) < 0.1
        u[fixed] = l[fixed] = np.floor(rng.uniform(0, 3, fixed.sum()))
        u[(rng.random(n) < 0.3) & ~fixed] = INF
    integrality = np.zeros(n); integrality[ints] = 1
    t0 = time.time()
    h = milp(c, constraints=LinearConstraint(A, lb=b, ub=np.inf), bounds=Bounds(l, u), integrality=integrality,
             options={'mip_rel_gap': 0.0, 'time_limit': 60})
    th = time.time() - t0
    for rule, mb, dive in (('most fractional', 1, 0), ('pseudo cost', 64, 4)):
        p = _ffi.Problem(ctx, A, b, c)
        t = _ffi.Tree(p, ints, l, u, branch_rule=rule, max_batch=mb, pool_capacity=1 << 20)
        if mb > 1:
            t.set_anchor_mode(True); t.set_dive(dive)
        t0 = time.time()
        st = t.solve(mip_gap=1e-9, max_seconds=60.0)
        tg = time.time() - t0
        status = {1: 'optimal', 2: 'infeasible', 3: 'unbounded'}.get(st['status'], st['status'])
        if h.status == 0:
            ok = status == 'optimal' and abs(st['primal_bound'] - h.fun) <= 1e-6 * max(1, abs(h.fun))
        elif h.status == 2:
            ok = status == 'infeasible'
        elif h.status == 3:
            ok = status == 'unbounded'
        else:
            ok = None
        if ok is False:
            bad += 1
        print(f'{n}x{m} seed {seed} {fam:8s} {rule:16s} batch {mb:3d}: engine {status} {st["primal_bound"]:.6f} '
              f'({st["evaluated_nodes"]} nodes, {tg:.2f} s) | HiGHS status {h.status} {h.fun if h.status == 0 else None} ({th:.2f} s) '
              f'{"OK" if ok else "MISMATCH" if ok is False else "?"}', flush=True)
        t.close(); p.close()
print('mismatches:', bad)
