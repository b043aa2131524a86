"""One cold 256x128 root LP (a single workgroup, ~3.6k dual simplex iterations): the per-iteration
cost of the kernel in isolation.  Used under rocprofv3 --pmc."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m = 256, 128
ctx = _ffi.Context(0)
A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)
d_l = ctx.to_device(l[None]); d_u = ctx.to_device(u[None])
d_st = ctx.alloc(4); d_obj = ctx.alloc(8); d_x = ctx.alloc(n * 8); d_y = ctx.alloc(m * 8)
d_vo = ctx.alloc(n + m); d_it = ctx.alloc(4); d_np = ctx.alloc(4)
for rep in range(3):
    ctx.timer_start()
    p.solve_batch_dev(1, d_l, d_u, None, 0, d_st, d_obj, d_x, d_y, d_vo, d_it, d_np)
    ms = ctx.timer_stop()
    it = np.zeros(1, np.int32); ctx.d2h(it, d_it)
    print('rep', rep, 'ms %.3f' % ms, 'iters', it[0], 'us/iter %.3f' % (ms * 1e3 / it[0]))
