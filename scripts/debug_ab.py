import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays
n, m, B, dive = 64, 32, 256, 1
ctx = _ffi.default_context()
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=1)
def make(host):
    if host: os.environ['MIPX_HOST_FINISH'] = '1'
    else: os.environ.pop('MIPX_HOST_FINISH', None)
    p = _ffi.Problem(ctx, A, b, c)
    t = _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=1 << 15)
    os.environ.pop('MIPX_HOST_FINISH', None)
    t.set_anchor_mode(True); t.set_dive(dive)
    return t, p
dev, p1 = make(False); host, p2 = make(True)
for step in range(8):
    a = dev.solve(mip_gap=1e-4, frontier_batch=B, max_steps=1)
    h = host.solve(mip_gap=1e-4, frontier_batch=B, max_steps=1)
    pa, ph = dev.pseudo_cost_arrays(), host.pseudo_cost_arrays()
    same = all(np.array_equal(x, y) for x, y in zip(pa, ph))
    key = lambda r: sorted((r[3][k], r[0][k].tobytes(), r[1][k].tobytes(), r[2][k].tobytes()) for k in range(len(r[3])))
    ka, kh = key(dev.peek_open(a['open_nodes'])), key(host.peek_open(h['open_nodes']))
    print('   open sets equal', ka == kh, 'bounds', [x[0] for x in ka][:6], [x[0] for x in kh][:6])
    if ka != kh and len(ka) == len(kh):
        for x, y in zip(ka, kh):
            if x != y:
                print('   first diff: key', x[0], y[0], 'l', np.flatnonzero(np.frombuffer(x[1]) != np.frombuffer(y[1])), 'u', np.flatnonzero(np.frombuffer(x[2]) != np.frombuffer(y[2])), 'v', np.flatnonzero(np.frombuffer(x[3], np.int8) != np.frombuffer(y[3], np.int8)))
                break
    print(step, 'dev', {k: a[k] for k in ('evaluated_nodes','probes_solved','dives','open_nodes','pivots')}, 'host', {k: h[k] for k in ('evaluated_nodes','probes_solved','dives','open_nodes','pivots')}, 'tables equal', same, flush=True)
