"""Quick kernel timing: batch of warm-started child LPs of one 256x128 tree (device resident)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.generators import random_dense_milp_arrays

def children(l, u, res, k):
    x = res['x']; frac = np.minimum(x - np.floor(x), np.ceil(x) - x)
    idx = np.argsort(-frac, kind='stable')[:k]
    ls, us = [], []
    for j in idx:
        if frac[j] <= 1e-4: continue
        l2, u2 = l.copy(), u.copy(); u2[j] = np.floor(x[j]); ls.append(l2); us.append(u2)
        l2, u2 = l.copy(), u.copy(); l2[j] = np.ceil(x[j]); ls.append(l2); us.append(u2)
    return np.array(ls), np.array(us)

ctx = _ffi.default_context()
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
A, b, c, l, u, _ = random_dense_milp_arrays(n, m, seed=0)
p = _ffi.Problem(ctx, A, b, c)
t = time.time(); root = p.solve_batch(l[None], u[None]); t = time.time() - t
print('root: status', root['status'], 'iters', root['iters'], 'obj', root['obj'], 'wall %.3f ms' % (t * 1e3))
r0 = {k: v[0] for k, v in root.items()}
# two levels of children to get a diverse frontier
L, U = children(l, u, r0, 32)
V = np.repeat(root['vstat'], len(L), axis=0)
g = p.solve_batch(L, U, V)
Ls, Us, Vs = [], [], []
for k in range(len(L)):
    if g['status'][k] != 0: continue
    rk = {key: val[k] for key, val in g.items()}
    L2, U2 = children(L[k], U[k], rk, 32)
    if len(L2) == 0: continue
    Ls.append(L2); Us.append(U2); Vs.append(np.repeat(g['vstat'][k:k+1], len(L2), axis=0))
L2 = np.concatenate(Ls)[:B]; U2 = np.concatenate(Us)[:B]; V2 = np.concatenate(Vs)[:B]
B = len(L2)
print('frontier batch', B)
d_l = ctx.to_device(L2); d_u = ctx.to_device(U2); d_v = ctx.to_device(V2)
d_st = ctx.alloc(B * 4); d_obj = ctx.alloc(B * 8); d_x = ctx.alloc(B * n * 8); d_y = ctx.alloc(B * m * 8)
d_vo = ctx.alloc(B * (n + m)); d_it = ctx.alloc(B * 4); d_np = ctx.alloc(B * 4)
for rep in range(3):
    ctx.timer_start()
    p.solve_batch_dev(B, d_l, d_u, d_v, 0, d_st, d_obj, d_x, d_y, d_vo, d_it, d_np)
    ms = ctx.timer_stop()
    it = np.zeros(B, np.int32); npv = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
    ctx.d2h(it, d_it); ctx.d2h(npv, d_np); ctx.d2h(st, d_st)
    piv = npv.sum()
    print('rep %d: %.3f ms  %.0f LP/s  mean iters %.1f mean pivots %.1f  us/pivot/CU %.3f  status counts %s' % (
        rep, ms, B / ms * 1e3, it.mean(), npv.mean(), ms * 1e3 * 256 / piv, np.bincount(st, minlength=4)))
bytes_per_pivot = 2 * 8 * (m + 1) * (n + m + 1)
print('algorithmic GB/s %.1f' % (piv * bytes_per_pivot / (ms * 1e-3) / 1e9))
# the same batch through the host-buffer entry point (mipx_lp_solve_batch: PCIe copies + read-back inside)
for rep in range(3):
    t0 = time.perf_counter()
    h = p.solve_batch(L2, U2, V2)
    dt = time.perf_counter() - t0
    print('host buffers rep %d: %.3f ms wall -> %.0f LP/s (PCIe inclusive; %.1f KB in + %.1f KB out per LP)' % (
        rep, dt * 1e3, B / dt, (2 * n * 8 + n + m) / 1e3, (n * 8 + m * 8 + n + m + 20) / 1e3))
