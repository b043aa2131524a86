"""Workgroup turnaround: a batch of LPs that are optimal at their (anchored) warm start, so each
workgroup only runs setup, value initialisation, one selection and the outputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m, B = 256, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = _ffi.default_context()
A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)
root = p.solve_batch(l[None], u[None])
p.set_anchor(root['vstat'][0])
L = np.repeat(l[None], B, 0); U = np.repeat(u[None], B, 0); V = np.repeat(root['vstat'], B, 0)
d_l = ctx.to_device(L); d_u = ctx.to_device(U); d_v = ctx.to_device(V)
d_st = ctx.alloc(B * 4); d_obj = ctx.alloc(B * 8); d_x = ctx.alloc(B * n * 8); d_y = ctx.alloc(B * m * 8)
d_vo = ctx.alloc(B * (n + m)); d_it = ctx.alloc(B * 4); d_np = ctx.alloc(B * 4)
for rep in range(3):
    ctx.timer_start()
    p.solve_batch_dev(B, d_l, d_u, d_v, 0, d_st, d_obj, d_x, d_y, d_vo, d_it, d_np)
    ms = ctx.timer_stop()
    npv = np.zeros(B, np.int32); ctx.d2h(npv, d_np)
    print('rep %d: %.3f ms for %d trivial LPs (pivots %d) -> %.2f us per LP slot (256 CUs)' % (rep, ms, B, npv.sum(), ms * 1e3 * 256 / B))
