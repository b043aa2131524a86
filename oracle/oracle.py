"""ctypes wrapper around oracle/libmipx_oracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
See the header of mipx_oracle.c for what it restates (reference file:line) and its parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_i8p = C.POINTER(C.c_int8)
_i32p = C.POINTER(C.c_int32)


def build():
    subprocess.check_call(['make', '-C', _HERE, '-s'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libmipx_oracle.so')
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.mipx_oracle_lp_solve_batch.restype = C.c_int
        _LIB.mipx_oracle_lp_solve_dive_batch.restype = C.c_int
        _LIB.mipx_oracle_most_fractional.restype = C.c_int
        _LIB.mipx_oracle_mip_feasible.restype = C.c_int
        _LIB.mipx_oracle_best_pseudo_cost.restype = C.c_int
        _LIB.mipx_oracle_pseudo_cost_update.restype = None
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def lp_solve_batch(A, b, c, l, u, vstat=None, max_iter=0):
    """Solve a batch of node LPs  min c'x, Ax >= b, l_k <= x <= u_k  sharing (A, b, c).

    l, u: (batch, n); vstat: (batch, n+m) int8 Clp status codes or None (cold start).
    Returns dict(status, obj, x, y, vstat, iters, npivots).
    """
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(m)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(n)
    l = np.ascontiguousarray(l, dtype=np.float64).reshape(-1, n)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, n)
    batch = l.shape[0]
    if vstat is not None:
        vstat = np.ascontiguousarray(vstat, dtype=np.int8).reshape(batch, n + m)
    status = np.zeros(batch, np.int32)
    obj = np.zeros(batch, np.float64)
    x = np.zeros((batch, n), np.float64)
    y = np.zeros((batch, m), np.float64)
    vout = np.zeros((batch, n + m), np.int8)
    iters = np.zeros(batch, np.int32)
    npiv = np.zeros(batch, np.int32)
    rc = lib().mipx_oracle_lp_solve_batch(
        C.c_int(m), C.c_int(n), _p(A, _dp), _p(b, _dp), _p(c, _dp), C.c_int(batch), _p(l, _dp),
        _p(u, _dp), _p(vstat, _i8p), C.c_int(int(max_iter)), _p(status, _i32p), _p(obj, _dp),
        _p(x, _dp), _p(y, _dp), _p(vout, _i8p), _p(iters, _i32p), _p(npiv, _i32p))
    assert rc == 0, f'oracle lp_solve_batch failed rc={rc}'
    return dict(status=status, obj=obj, x=x, y=y, vstat=vout, iters=iters, npivots=npiv)


def lp_solve_dive_batch(A, b, c, l, u, vstat, rule, int_idx, cost_l, cost_r, has_entry, cutoff,
                        max_iter=0, anchor_table=None, anchor_sel=None, depth=1):
    """Node LPs with the in-place dive (LpArgs::dive of the GPU kernel), up to `depth` children in a
    row on one tableau: every array of the result has (depth + 1) * batch rows -- the nodes, then
    their first dive children, then the second ... (status -1 where none was solved) -- plus
    dive_var (-1: no dive), dive_dir, dive_val with depth * batch entries (the decision taken after
    level p's LP at p * batch + k)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(m)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(n)
    l = np.ascontiguousarray(l, dtype=np.float64).reshape(-1, n)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, n)
    B = l.shape[0]
    if vstat is not None:
        vstat = np.ascontiguousarray(vstat, dtype=np.int8).reshape(B, n + m)
    ii = np.ascontiguousarray(int_idx, np.int32)
    cl = np.ascontiguousarray(cost_l, np.float64); cr = np.ascontiguousarray(cost_r, np.float64)
    he = np.ascontiguousarray(has_entry, np.uint8)
    D = int(depth)
    status = np.full((D + 1) * B, -1, np.int32); obj = np.zeros((D + 1) * B); x = np.zeros(((D + 1) * B, n))
    vout = np.zeros(((D + 1) * B, n + m), np.int8); iters = np.zeros((D + 1) * B, np.int32)
    npiv = np.zeros((D + 1) * B, np.int32)
    dvar = np.full(D * B, -1, np.int32); ddir = np.zeros(D * B, np.int32); dval = np.zeros(D * B)
    rc = lib().mipx_oracle_lp_solve_dive_batch(
        C.c_int(m), C.c_int(n), _p(A, _dp), _p(b, _dp), _p(c, _dp), C.c_int(B), _p(l, _dp), _p(u, _dp),
        _p(vstat, _i8p), C.c_int(int(max_iter)), C.c_int(int(rule)), C.c_int(len(ii)), _p(ii, _i32p),
        _p(cl, _dp), _p(cr, _dp), he.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_double(float(cutoff)),
        _p(status, _i32p), _p(obj, _dp), _p(x, _dp), _p(vout, _i8p), _p(iters, _i32p), _p(npiv, _i32p),
        _p(dvar, _i32p), _p(ddir, _i32p), _p(dval, _dp),
        None if anchor_table is None else _p(anchor_table[0], _dp),
        None if anchor_table is None else _p(anchor_table[1], _dp),
        None if anchor_table is None else _p(anchor_table[2], _i32p),
        None if anchor_sel is None else _p(np.ascontiguousarray(anchor_sel, np.int32), _i32p), C.c_int(D))
    assert rc == 0, f'oracle lp_solve_dive_batch failed rc={rc}'
    return dict(status=status, obj=obj, x=x, vstat=vout, iters=iters, npivots=npiv, dive_var=dvar,
                dive_dir=ddir, dive_val=dval)


class pricing:
    """Context manager: force the pricing rule of the oracle's node LPs (0 largest violation, 1 dual
    steepest edge -- the register-tile kernel --, 2 dual Devex -- the HBM-streaming kernel); the
    default (-1) follows the GPU path's choice by shape."""

    def __init__(self, rule):
        self.rule = int(rule)

    def __enter__(self):
        lib().mipx_oracle_set_pricing(C.c_int(self.rule))

    def __exit__(self, *exc):
        lib().mipx_oracle_set_pricing(C.c_int(-1))


def lp_solve(A, b, c, l, u, vstat=None, max_iter=0):
    r = lp_solve_batch(A, b, c, np.asarray(l, float)[None], np.asarray(u, float)[None],
                       None if vstat is None else np.asarray(vstat, np.int8)[None], max_iter)
    return {k: v[0] for k, v in r.items()}


def most_fractional(int_idx, x):
    ii = np.ascontiguousarray(int_idx, np.int32)
    x = np.ascontiguousarray(x, np.float64)
    r = lib().mipx_oracle_most_fractional(C.c_int(len(ii)), _p(ii, _i32p), _p(x, _dp))
    return None if r < 0 else int(r)


def mip_feasible(int_idx, x):
    ii = np.ascontiguousarray(int_idx, np.int32)
    x = np.ascontiguousarray(x, np.float64)
    return bool(lib().mipx_oracle_mip_feasible(C.c_int(len(ii)), _p(ii, _i32p), _p(x, _dp)))


def best_pseudo_cost(int_idx, x, cost_left, cost_right):
    ii = np.ascontiguousarray(int_idx, np.int32)
    x = np.ascontiguousarray(x, np.float64)
    cl = np.ascontiguousarray(cost_left, np.float64)
    cr = np.ascontiguousarray(cost_right, np.float64)
    r = lib().mipx_oracle_best_pseudo_cost(C.c_int(len(ii)), _p(ii, _i32p), _p(x, _dp),
                                           _p(cl, _dp), _p(cr, _dp))
    return None if r < 0 else int(r)


def pseudo_cost_update(cost, times, status, objective, dual_bound, variable_change):
    cc = C.c_double(cost)
    tt = C.c_int32(times)
    lib().mipx_oracle_pseudo_cost_update(C.byref(cc), C.byref(tt), C.c_int(status),
                                         C.c_double(objective), C.c_double(dual_bound),
                                         C.c_double(variable_change))
    return cc.value, tt.value


def debug_dump(A, b, c, l, u, vstat=None, max_iter=0):
    """Solve one LP and return the final tableau state (mirror of _ffi.debug_dump)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    T = np.zeros((m, n)); vec = np.zeros(n + 3 * m); idx = np.zeros(2 * n + m, np.int32)
    lib().mipx_oracle_set_dump(_p(T, _dp), _p(vec, _dp), _p(idx, _i32p))
    try:
        res = lp_solve_batch(A, b, c, np.asarray(l, float)[None], np.asarray(u, float)[None],
                             None if vstat is None else np.asarray(vstat, np.int8)[None], max_iter)
    finally:
        lib().mipx_oracle_set_dump(None, None, None)
    return res, dict(T=T, d=vec[:n], beta0=vec[n:n + m], ba=vec[n + m:n + 2 * m],
                     bb=vec[n + 2 * m:], nvar=idx[:n], bvar=idx[n:n + m], side=idx[n + m:])


_EST = {None: 0, 'over': 1, 'under': 2}


def get_fraction(x, max_term=1e3, estimate=None):
    n = C.c_double(); d = C.c_double()
    lib().mipx_oracle_get_fraction(C.c_double(x), C.c_double(max_term), C.c_int(_EST[estimate]),
                                   C.byref(n), C.byref(d))
    return int(n.value), int(d.value)


def safe_cut(pi, pi0, estimate='over', max_term=1e3):
    pi = np.ascontiguousarray(pi, np.float64)
    out = np.zeros_like(pi); out0 = C.c_double()
    lib().mipx_oracle_safe_cut(C.c_int(len(pi)), _p(pi, _dp), C.c_double(pi0),
                               C.c_int(_EST[estimate]), C.c_double(max_term), _p(out, _dp),
                               C.byref(out0))
    return out, out0.value


def safe_cut_ex(pi, pi0, estimate='over', make_integer=False, max_term=1e3):
    """safe_cut with the chosen (numerator, denominator) per coefficient and for the right-hand
    side (last entry), and the make_integer form: dict(safe_pi, safe_pi0, num, den)."""
    pi = np.ascontiguousarray(pi, np.float64)
    n = len(pi)
    out = np.zeros(n); out0 = C.c_double(); num = np.zeros(n + 1); den = np.zeros(n + 1)
    lib().mipx_oracle_safe_cut_ex(C.c_int(n), _p(pi, _dp), C.c_double(pi0), C.c_int(_EST[estimate]),
                                  C.c_double(max_term), C.c_int(int(bool(make_integer))), _p(out, _dp),
                                  C.byref(out0), _p(num, _dp), _p(den, _dp))
    return dict(safe_pi=out, safe_pi0=out0.value, num=num.astype(np.int64), den=den.astype(np.int64))


def gomory(A, b, c, l, u, vstat, x, int_idx, max_term=1e3):
    """GMI cuts (+ safe rounding) for the basis `vstat` of one node LP; x is the node's solution.

    Returns dict(row_idx, pi, pi0, safe_pi, safe_pi0) with one row per cut.
    """
    _, dump = debug_dump(A, b, c, l, u, vstat, 0)
    return gomory_from_dump(A, b, dump, x, int_idx, max_term)


def gomory_from_dump(A, b, dump, x, int_idx, max_term=1e3):
    """The same from a tableau state a solve ended with (debug_dump): what the engine's batched cut rounds
    read (the LP launch dumps the tableau it ends with; no launch of its own refactors one)."""
    A = np.ascontiguousarray(A, np.float64)
    m, n = A.shape
    b = np.ascontiguousarray(b, np.float64)
    T = np.ascontiguousarray(dump['T']); bvar = np.ascontiguousarray(dump['bvar'], np.int32)
    nvar = np.ascontiguousarray(dump['nvar'], np.int32)
    is_int = np.zeros(n, np.uint8); is_int[np.asarray(int_idx, int)] = 1
    xs = np.ascontiguousarray(np.maximum(np.asarray(x, np.float64), 0))
    row_idx = np.zeros(m, np.int32); pi = np.zeros((m, n)); pi0 = np.zeros(m)
    spi = np.zeros((m, n)); spi0 = np.zeros(m)
    lib().mipx_oracle_gomory.restype = C.c_int
    k = lib().mipx_oracle_gomory(C.c_int(m), C.c_int(n), _p(A, _dp), _p(b, _dp), _p(T, _dp),
                                 _p(bvar, _i32p), _p(nvar, _i32p), _p(xs, _dp),
                                 is_int.ctypes.data_as(C.c_void_p), C.c_double(max_term),
                                 _p(row_idx, _i32p), _p(pi, _dp), _p(pi0, _dp), _p(spi, _dp),
                                 _p(spi0, _dp))
    return dict(row_idx=row_idx[:k], pi=pi[:k], pi0=pi0[:k], safe_pi=spi[:k], safe_pi0=spi0[:k])


def select_cuts(pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef):
    """Greedy cut selection; returns (added positions in order, terminator code, depths)."""
    pi = np.ascontiguousarray(pi, np.float64).reshape(len(pi0), -1) if len(pi0) else np.zeros((0, len(x)))
    K, n = pi.shape
    pi0 = np.ascontiguousarray(pi0, np.float64); x = np.ascontiguousarray(x, np.float64)
    added = np.zeros(max(K, 1), np.int32); term = C.c_int32(); depth = np.zeros(max(K, 1))
    lib().mipx_oracle_select_cuts.restype = C.c_int
    k = lib().mipx_oracle_select_cuts(C.c_int(n), C.c_int(K), _p(pi, _dp), _p(pi0, _dp), _p(x, _dp),
                                      C.c_int(int(min(max_nonzero_coefs, 2**31 - 1))),
                                      C.c_double(min_cut_depth), C.c_double(cos_parallel),
                                      C.c_double(max_abs_coef), _p(added, _i32p), C.byref(term),
                                      _p(depth, _dp))
    return added[:k].copy(), int(term.value), depth[:K].copy()


def make_anchor(A, b, c, vstat):
    """Tableau state of basis `vstat` (refactorisation only), in the dump layout."""
    A = np.ascontiguousarray(A, np.float64)
    m, n = A.shape
    lib().mipx_oracle_set_refactor_only(C.c_int(1))
    try:
        _, d = debug_dump(A, b, c, np.zeros(n), np.zeros(n), vstat, 0)
    finally:
        lib().mipx_oracle_set_refactor_only(C.c_int(0))
    vec = np.ascontiguousarray(np.concatenate([d['d'], d['beta0'], d['ba'], d['bb']]))
    idx = np.ascontiguousarray(np.concatenate([d['nvar'], d['bvar'], d['side']]), np.int32)
    return dict(T=np.ascontiguousarray(d['T']), vec=vec, idx=idx)


class anchored:
    """Context manager: oracle solves inside refactor from `anchor` (see make_anchor)."""

    def __init__(self, anchor):
        self.a = anchor

    def __enter__(self):
        lib().mipx_oracle_set_anchor(_p(self.a['T'], _dp), _p(self.a['vec'], _dp), _p(self.a['idx'], _i32p))

    def __exit__(self, *exc):
        lib().mipx_oracle_set_anchor(None, None, None)
