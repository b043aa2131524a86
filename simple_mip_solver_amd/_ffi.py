"""ctypes binding of libmipx.so (include/mipx.h).  Fails loudly: there is no CPU fallback.

The library is built in-tree by `make -C simple_mip_solver_amd/csrc` (see __graft_entry__.build).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MIPX_LIB') or os.path.join(_HERE, 'csrc', 'libmipx.so')  # MIPX_LIB: a profiling build

MIPX_OK = 0
ERRORS = {-1: 'MIPX_EINVAL', -2: 'MIPX_ENODEV', -3: 'MIPX_EHIP', -4: 'MIPX_ETOOBIG',
          -5: 'MIPX_ENOMEM', -6: 'MIPX_EHOOK', -7: 'MIPX_EPEER'}

# every symbol include/mipx.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    'mipx_abi_version', 'mipx_device_count', 'mipx_ctx_create', 'mipx_ctx_destroy',
    'mipx_last_error', 'mipx_ctx_sync', 'mipx_problem_create', 'mipx_problem_destroy', 'mipx_problem_set_anchor',
    'mipx_tree_set_anchor_mode',
    'mipx_lp_solve_batch', 'mipx_lp_solve_batch_dev', 'mipx_lp_solve_multi', 'mipx_gomory_batch', 'mipx_cut_select_batch',
    'mipx_safe_cut_batch', 'mipx_get_fraction_batch', 'mipx_lp_solve_batch_cuts', 'mipx_tree_create_ex',
    'mipx_tree_cut_stats', 'mipx_comm_unique_id', 'mipx_comm_create_rccl', 'mipx_comm_create_custom',
    'mipx_comm_destroy', 'mipx_comm_rank', 'mipx_comm_size', 'mipx_comm_allgather', 'mipx_comm_barrier',
    'mipx_tree_set_comm', 'mipx_tree_global_stats', 'mipx_exchange_record_len', 'mipx_exchange_decide',
    'mipx_tree_exchange_record', 'mipx_tree_trace_cuts', 'mipx_tree_peek_cuts', 'mipx_tree_cut_store',
    'mipx_tree_cut_rows_per_node', 'mipx_tree_migrate_self', 'mipx_tree_kernel_ms', 'mipx_last_kernel_ms',
    'mipx_dev_alloc', 'mipx_dev_free',
    'mipx_memcpy_h2d', 'mipx_memcpy_d2h', 'mipx_timer_start', 'mipx_timer_stop',
    'mipx_kernel_name', 'mipx_debug_enable', 'mipx_debug_read',
    'mipx_tree_create', 'mipx_tree_destroy', 'mipx_tree_solve', 'mipx_tree_get_stats',
    'mipx_tree_solution', 'mipx_tree_set_primal_bound', 'mipx_tree_pseudo_costs', 'mipx_tree_set_pseudo_costs',
    'mipx_tree_set_trace', 'mipx_tree_trace', 'mipx_tree_peek_open', 'mipx_tree_keep_shard',
    'mipx_tree_set_step_hook', 'mipx_lp_dive_batch', 'mipx_lp_plunge_batch', 'mipx_tree_set_dive', 'mipx_tree_reanchor',
    'mipx_tree_peek_anchors', 'mipx_tree_anchor_table',
]

_dp = C.POINTER(C.c_double)
_i8p = C.POINTER(C.c_int8)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p

_lib = None


class MipxError(RuntimeError):
    pass


class TreeStats(C.Structure):
    """mipx_tree_stats (include/mipx.h)."""
    _fields_ = [('evaluated_nodes', C.c_int64), ('lp_solved', C.c_int64),
                ('probes_solved', C.c_int64), ('pivots', C.c_int64), ('open_nodes', C.c_int64),
                ('created_nodes', C.c_int64), ('steps', C.c_int64), ('primal_bound', C.c_double),
                ('dual_bound', C.c_double), ('gap', C.c_double), ('solve_seconds', C.c_double),
                ('kernel_ms', C.c_double), ('status', C.c_int32), ('has_solution', C.c_int32),
                ('dives', C.c_int64), ('pool_exhausted', C.c_int32), ('reserved', C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class CutParams(C.Structure):
    """mipx_cut_params (include/mipx.h)."""
    _fields_ = [('max_cut_generation_iterations', C.c_int32), ('max_nonzero_coefs', C.c_int32),
                ('max_cuts_per_node', C.c_int32), ('exact_tableau', C.c_int32),
                ('cutting_plane_progress_tolerance', C.c_double), ('min_cut_depth', C.c_double),
                ('cos_parallel', C.c_double), ('max_abs_coef', C.c_double), ('max_term', C.c_double),
                ('max_dual_bound', C.c_double), ('store_capacity', C.c_int64)]


CUT_TOTAL_KEYS = ('total_cut_generation_iterations', 'total_iterations_gmic_created',
                  'total_number_gmic_created', 'total_iterations_gmic_added', 'total_number_gmic_added',
                  'total_iterations_gmic_removed', 'total_number_gmic_removed')

class GlobalStats(C.Structure):
    """mipx_tree_global_stats_t (include/mipx.h)."""
    _fields_ = [('primal_bound', C.c_double), ('dual_bound', C.c_double), ('gap', C.c_double),
                ('evaluated_nodes', C.c_int64), ('lp_solved', C.c_int64), ('probes_solved', C.c_int64),
                ('pivots', C.c_int64), ('open_nodes', C.c_int64), ('exchanges', C.c_int64),
                ('nodes_sent', C.c_int64), ('nodes_received', C.c_int64), ('world', C.c_int32),
                ('incumbent_rank', C.c_int32)]


class ExchangeDecision(C.Structure):
    """mipx_exchange_decision (include/mipx.h)."""
    _fields_ = [('primal', C.c_double), ('dual', C.c_double), ('gap', C.c_double), ('sums', C.c_int64 * 4),
                ('open_nodes', C.c_int64), ('incumbent_rank', C.c_int32), ('done', C.c_int32),
                ('reason', C.c_int32), ('n_moves', C.c_int32), ('moves', C.c_int32 * 192)]


def source_hash():
    """sha256 over the kernel / engine sources next to libmipx.so (csrc/*.hip, *.h, in name order).  The
    profiling scripts store it beside the counters they collect; bench.py flags a counter-based roofline
    figure as stale when the sources have changed since (profiles/pmc_latest.json)."""
    import hashlib
    d = os.path.dirname(LIB_PATH)
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()


def exchange_record_len(n):
    return lib().mipx_exchange_record_len(int(n))


def exchange_decide(records, n, mip_gap=1e-4, allow_migration=True):
    """What every rank concludes from the gathered records ((world, record_len) array): dict."""
    records = np.ascontiguousarray(records, dtype=np.float64)
    world = records.shape[0]
    assert records.shape[1] == exchange_record_len(n)
    d = ExchangeDecision()
    L = lib()
    L.mipx_exchange_decide.argtypes = [C.c_int, C.c_int, _vp, C.c_double, C.c_int, C.POINTER(ExchangeDecision)]
    rc = L.mipx_exchange_decide(world, int(n), _ptr(records), float(mip_gap), int(bool(allow_migration)), C.byref(d))
    if rc != MIPX_OK:
        raise MipxError(f'mipx_exchange_decide failed: {ERRORS.get(rc, rc)}')
    return dict(primal=d.primal, dual=d.dual, gap=None if d.gap < 0 else d.gap, sums=list(d.sums),
                open_nodes=d.open_nodes, incumbent_rank=d.incumbent_rank, done=bool(d.done), reason=d.reason,
                moves=[tuple(d.moves[3 * k:3 * k + 3]) for k in range(d.n_moves)])


class CommOps(C.Structure):
    """mipx_comm_ops: host-buffer primitives of a custom communicator."""
    ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
    RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)
    _fields_ = [('allgather', ALLGATHER), ('send', SEND), ('recv', RECV)]


TREE_STATUS = {0: 'unsolved', 1: 'optimal', 2: 'infeasible', 3: 'unbounded',
               4: 'stopped on iterations or time'}


def lib():
    """Load libmipx.so; raise MipxError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MipxError(
            f'{LIB_PATH} is missing: the HIP extension has not been built '
            f'(run `make -C {os.path.dirname(LIB_PATH)}` or __graft_entry__.build()). '
            'simple_mip_solver_amd has no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    L.mipx_abi_version.restype = C.c_int
    L.mipx_device_count.restype = C.c_int
    L.mipx_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
    L.mipx_ctx_destroy.argtypes = [_vp]
    L.mipx_ctx_destroy.restype = None
    L.mipx_last_error.argtypes = [_vp]
    L.mipx_last_error.restype = C.c_char_p
    L.mipx_ctx_sync.argtypes = [_vp]
    L.mipx_problem_create.argtypes = [_vp, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(_vp)]
    L.mipx_problem_destroy.argtypes = [_vp]
    L.mipx_problem_set_anchor.argtypes = [_vp, _vp]
    L.mipx_tree_set_anchor_mode.argtypes = [_vp, C.c_int]
    L.mipx_problem_destroy.restype = None
    solve_args = [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.mipx_lp_solve_batch.argtypes = solve_args
    L.mipx_lp_solve_batch_dev.argtypes = solve_args
    L.mipx_lp_solve_multi.argtypes = [_vp, C.c_int, C.c_int, C.c_int] + [_vp] * 5 + [C.c_int] + [_vp] * 6
    L.mipx_gomory_batch.argtypes = [_vp, C.c_int] + [_vp] * 5 + [C.c_double] + [_vp] * 6
    L.mipx_cut_select_batch.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int,
                                        C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp]
    L.mipx_dev_alloc.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
    L.mipx_dev_free.argtypes = [_vp, _vp]
    L.mipx_memcpy_h2d.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.mipx_memcpy_d2h.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.mipx_timer_start.argtypes = [_vp]
    L.mipx_timer_stop.argtypes = [_vp, C.POINTER(C.c_float)]
    L.mipx_kernel_name.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.mipx_tree_create.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int64, C.POINTER(_vp)]
    L.mipx_tree_destroy.argtypes = [_vp]
    L.mipx_tree_destroy.restype = None
    L.mipx_tree_solve.argtypes = [_vp, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int64,
                                  C.POINTER(TreeStats)]
    L.mipx_tree_get_stats.argtypes = [_vp, C.POINTER(TreeStats)]
    L.mipx_tree_solution.argtypes = [_vp, _vp]
    L.mipx_tree_set_primal_bound.argtypes = [_vp, C.c_double]
    L.mipx_tree_pseudo_costs.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.mipx_tree_set_trace.argtypes = [_vp, C.c_int]
    L.mipx_tree_trace.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp]
    L.mipx_tree_trace.restype = C.c_int64
    L.mipx_tree_peek_open.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp]
    L.mipx_tree_peek_open.restype = C.c_int64
    L.mipx_tree_keep_shard.argtypes = [_vp, C.c_int, C.c_int]
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def comm_unique_id():
    """The 128 bytes rank 0 makes (ncclGetUniqueId) and the launcher hands to every rank."""
    buf = C.create_string_buffer(128)
    L = lib()
    L.mipx_comm_unique_id.argtypes = [C.c_char_p]
    rc = L.mipx_comm_unique_id(buf)
    if rc != MIPX_OK:
        raise MipxError(f'mipx_comm_unique_id failed: {ERRORS.get(rc, rc)} (librccl.so is needed for more than one GPU)')
    return buf.raw


class Comm:
    """The communicator of a multi-GPU search (mipx_comm): RCCL over xGMI, bound in libmipx.so.

    Comm(ctx, rank, world, unique_id=...) is the product; Comm(ctx, rank, world, allgather=, send=,
    recv=) runs the same protocol over caller-supplied host-buffer primitives (tests / rehearsals:
    allgather(send_bytes) -> list of world bytes objects, send(peer, bytes), recv(peer, nbytes) ->
    bytes)."""

    def __init__(self, ctx, rank, world, unique_id=None, allgather=None, send=None, recv=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        L = lib()
        h = _vp()
        if unique_id is not None:
            assert len(unique_id) == 128, 'the RCCL unique id has 128 bytes'
            L.mipx_comm_create_rccl.argtypes = [_vp, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp)]
            rc = L.mipx_comm_create_rccl(ctx._h, unique_id, self.rank, self.world, C.byref(h))
            ctx.check(rc, 'mipx_comm_create_rccl')
            self.transport = 'rccl'
        else:
            self._err = None

            def guard(fn):
                def inner(*a):
                    try:
                        fn(*a)
                        return 0
                    except BaseException as e:   # never unwind through the C frames
                        self._err = e
                        return 1
                return inner

            def c_allgather(_user, sendp, recvp, nbytes):
                parts = allgather(C.string_at(sendp, nbytes))
                assert len(parts) == self.world and all(len(p) == nbytes for p in parts)
                C.memmove(recvp, b''.join(parts), nbytes * self.world)

            def c_send(_user, peer, bufp, nbytes):
                send(int(peer), C.string_at(bufp, nbytes))

            def c_recv(_user, peer, bufp, nbytes):
                data = recv(int(peer), int(nbytes))
                assert len(data) == nbytes
                C.memmove(bufp, data, nbytes)
            self._ops = CommOps(CommOps.ALLGATHER(guard(c_allgather)), CommOps.SEND(guard(c_send)),
                                CommOps.RECV(guard(c_recv)))
            L.mipx_comm_create_custom.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(CommOps), _vp, C.POINTER(_vp)]
            rc = L.mipx_comm_create_custom(None if ctx is None else ctx._h, self.rank, self.world,
                                           C.byref(self._ops), None, C.byref(h))
            if rc != MIPX_OK:
                raise MipxError(f'mipx_comm_create_custom failed: {ERRORS.get(rc, rc)}')
            self.transport = 'custom'
        self._h = h

    def check(self, rc, what):
        err, self._err = getattr(self, '_err', None), None
        if err is not None:
            raise err
        if self.ctx is not None:
            self.ctx.check(rc, what)
        elif rc != MIPX_OK:
            raise MipxError(f'{what} failed: {ERRORS.get(rc, rc)}')

    def allgather(self, arr):
        """All-gather of one equally sized array per rank (host, blocking): (world, ...) array."""
        arr = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + arr.shape, arr.dtype)
        L = lib()
        L.mipx_comm_allgather.argtypes = [_vp, _vp, _vp, C.c_size_t]
        self.check(L.mipx_comm_allgather(self._h, _ptr(arr), _ptr(out), arr.nbytes), 'mipx_comm_allgather')
        return out

    def barrier(self):
        L = lib()
        L.mipx_comm_barrier.argtypes = [_vp]
        self.check(L.mipx_comm_barrier(self._h), 'mipx_comm_barrier')

    def close(self):
        if getattr(self, '_h', None) and (self.ctx is None or getattr(self.ctx, '_h', None)):
            L = lib()
            L.mipx_comm_destroy.argtypes = [_vp]
            L.mipx_comm_destroy.restype = None
            L.mipx_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream (mipx_ctx)."""

    def __init__(self, device=0):
        L = lib()
        h = _vp()
        rc = L.mipx_ctx_create(int(device), C.byref(h))
        if rc != MIPX_OK:
            raise MipxError(
                f'mipx_ctx_create(device={device}) failed: {ERRORS.get(rc, rc)} '
                f'({L.mipx_device_count()} HIP devices visible). '
                'simple_mip_solver_amd needs an MI355X (gfx950); there is no CPU fallback.')
        self._h = h
        self.device = device

    def check(self, rc, what):
        if rc != MIPX_OK:
            msg = lib().mipx_last_error(self._h)
            raise MipxError(f'{what} failed: {ERRORS.get(rc, rc)}: '
                            f'{msg.decode() if msg else ""}')

    def sync(self):
        self.check(lib().mipx_ctx_sync(self._h), 'mipx_ctx_sync')

    def last_kernel_ms(self):
        """Device time of the LP launch inside the last solve_multi call (mipx_last_kernel_ms)."""
        ms = C.c_float()
        L = lib()
        L.mipx_last_kernel_ms.argtypes = [_vp, C.POINTER(C.c_float)]
        self.check(L.mipx_last_kernel_ms(self._h, C.byref(ms)), 'mipx_last_kernel_ms')
        return float(ms.value)

    def timer_start(self):
        self.check(lib().mipx_timer_start(self._h), 'mipx_timer_start')

    def timer_stop(self):
        ms = C.c_float()
        self.check(lib().mipx_timer_stop(self._h, C.byref(ms)), 'mipx_timer_stop')
        return ms.value

    def alloc(self, nbytes):
        d = _vp()
        self.check(lib().mipx_dev_alloc(self._h, nbytes, C.byref(d)), 'mipx_dev_alloc')
        return d

    def free(self, d):
        self.check(lib().mipx_dev_free(self._h, d), 'mipx_dev_free')

    def h2d(self, d, arr):
        arr = np.ascontiguousarray(arr)
        self.check(lib().mipx_memcpy_h2d(self._h, d, _ptr(arr), arr.nbytes), 'mipx_memcpy_h2d')

    def d2h(self, arr, d):
        assert arr.flags['C_CONTIGUOUS']
        self.check(lib().mipx_memcpy_d2h(self._h, _ptr(arr), d, arr.nbytes), 'mipx_memcpy_d2h')

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        d = self.alloc(arr.nbytes)
        self.h2d(d, arr)
        return d

    def close(self):
        if getattr(self, '_h', None):
            lib().mipx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def select_cuts(ctx, pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef):
    """K3 for ONE node: (added pool positions in order, terminator code, depths)."""
    pi0 = np.ascontiguousarray(pi0, dtype=np.float64).reshape(-1)
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    K, n = len(pi0), len(x)
    kmax = max(K, 1)
    P = np.zeros((kmax, n)); P0 = np.zeros(kmax)
    if K:
        P[:K] = np.asarray(pi, dtype=np.float64).reshape(K, n); P0[:K] = pi0
    npool = np.array([K], np.int32); nadded = np.zeros(1, np.int32); added = np.zeros(kmax, np.int32)
    term = np.zeros(1, np.int32); depth = np.zeros(kmax)
    rc = lib().mipx_cut_select_batch(ctx._h, n, 1, kmax, _ptr(npool), _ptr(P), _ptr(P0), _ptr(x),
                                     int(min(max_nonzero_coefs, 2 ** 31 - 1)), float(min_cut_depth),
                                     float(cos_parallel), float(max_abs_coef), _ptr(nadded),
                                     _ptr(added), _ptr(term), _ptr(depth))
    ctx.check(rc, 'mipx_cut_select_batch')
    return added[:nadded[0]].copy(), int(term[0]), depth[:K].copy()


_EST = {None: 0, 'over': 1, 'under': 2}


def safe_cut_batch(ctx, pi, pi0, estimate='over', make_integer=False, max_term=1e3):
    """numerically_safe_cut for a batch of cuts on the device (mipx_safe_cut_batch): dict with
    safe_pi, safe_pi0, num, den (batch x (n+1): the coefficients, then the right-hand side),
    scaled_pi, scaled_pi0, nonzero."""
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    if pi.ndim == 1:
        pi = pi[None]
    B, n = pi.shape
    pi0 = np.ascontiguousarray(pi0, dtype=np.float64).reshape(B)
    spi = np.zeros((B, n)); spi0 = np.zeros(B)
    num = np.zeros((B, n + 1)); den = np.zeros((B, n + 1))
    cpi = np.zeros((B, n)); cpi0 = np.zeros(B); nz = np.zeros(B, np.int32)
    L = lib()
    L.mipx_safe_cut_batch.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_double] + [_vp] * 7
    rc = L.mipx_safe_cut_batch(ctx._h, n, B, _ptr(pi), _ptr(pi0), _EST[estimate], int(bool(make_integer)),
                               float(max_term), _ptr(spi), _ptr(spi0), _ptr(num), _ptr(den), _ptr(cpi),
                               _ptr(cpi0), _ptr(nz))
    ctx.check(rc, 'mipx_safe_cut_batch')
    return dict(safe_pi=spi, safe_pi0=spi0, num=num, den=den, scaled_pi=cpi, scaled_pi0=cpi0, nonzero=nz)


def get_fraction_batch(ctx, x, max_term, estimate):
    """get_fraction on the device for arrays x, max_term and a list of estimates (None/'over'/'under'):
    (numerators, denominators) as int64 arrays."""
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    mt = np.ascontiguousarray(np.broadcast_to(np.asarray(max_term, dtype=np.float64), x.shape))
    est = np.ascontiguousarray([_EST[e] for e in estimate], dtype=np.int32)
    num = np.zeros(len(x)); den = np.zeros(len(x))
    L = lib()
    L.mipx_get_fraction_batch.argtypes = [_vp, C.c_int] + [_vp] * 5
    rc = L.mipx_get_fraction_batch(ctx._h, len(x), _ptr(x), _ptr(mt), _ptr(est), _ptr(num), _ptr(den))
    ctx.check(rc, 'mipx_get_fraction_batch')
    return num.astype(np.int64), den.astype(np.int64)


def solve_multi(ctx, A, b, c, l, u, max_iter=0):
    """Root relaxations of `batch` independent problems of one shape (cold start)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    B, m, n = A.shape
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(B, m)
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(B, n)
    l = np.ascontiguousarray(l, dtype=np.float64).reshape(B, n)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(B, n)
    status = np.zeros(B, np.int32); obj = np.zeros(B); x = np.zeros((B, n))
    vout = np.zeros((B, n + m), np.int8); iters = np.zeros(B, np.int32); npiv = np.zeros(B, np.int32)
    rc = lib().mipx_lp_solve_multi(ctx._h, m, n, B, _ptr(A), _ptr(b), _ptr(c), _ptr(l), _ptr(u),
                                   int(max_iter), _ptr(status), _ptr(obj), _ptr(x), _ptr(vout),
                                   _ptr(iters), _ptr(npiv))
    ctx.check(rc, 'mipx_lp_solve_multi')
    return dict(status=status, obj=obj, x=x, vstat=vout, iters=iters, npivots=npiv)


class Problem:
    """(A, b, c) of one tree resident in HBM (mipx_problem)."""

    def __init__(self, ctx, A, b, c):
        self.ctx = ctx
        A = np.ascontiguousarray(A, dtype=np.float64)
        if A.ndim != 2:
            A = A.reshape(-1, len(c))
        self.m, self.n = A.shape
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(self.m)
        c = np.ascontiguousarray(c, dtype=np.float64).reshape(self.n)
        h = _vp()
        rc = lib().mipx_problem_create(ctx._h, self.m, self.n, A.ctypes.data_as(_dp),
                                       b.ctypes.data_as(_dp), c.ctypes.data_as(_dp), C.byref(h))
        ctx.check(rc, f'mipx_problem_create(m={self.m}, n={self.n})')
        self._h = h

    def set_anchor(self, vstat):
        """Anchor warm starts at the tableau of basis `vstat` (None switches it off)."""
        v = None if vstat is None else np.ascontiguousarray(vstat, dtype=np.int8).reshape(self.n + self.m)
        self.ctx.check(lib().mipx_problem_set_anchor(self._h, _ptr(v)), 'mipx_problem_set_anchor')

    def solve_batch(self, l, u, vstat=None, max_iter=0):
        """Host-buffer batched LP relaxation; returns dict like the oracle's."""
        n, m = self.n, self.m
        l = np.ascontiguousarray(l, dtype=np.float64).reshape(-1, n)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, n)
        B = l.shape[0]
        assert u.shape[0] == B
        if vstat is not None:
            vstat = np.ascontiguousarray(vstat, dtype=np.int8).reshape(B, n + m)
        status = np.zeros(B, np.int32)
        obj = np.zeros(B, np.float64)
        x = np.zeros((B, n), np.float64)
        y = np.zeros((B, m), np.float64)
        vout = np.zeros((B, n + m), np.int8)
        iters = np.zeros(B, np.int32)
        npiv = np.zeros(B, np.int32)
        rc = lib().mipx_lp_solve_batch(self._h, B, _ptr(l), _ptr(u), _ptr(vstat), int(max_iter),
                                       _ptr(status), _ptr(obj), _ptr(x), _ptr(y), _ptr(vout),
                                       _ptr(iters), _ptr(npiv))
        self.ctx.check(rc, 'mipx_lp_solve_batch')
        return dict(status=status, obj=obj, x=x, y=y, vstat=vout, iters=iters, npivots=npiv)

    def solve_batch_cuts(self, l, u, vstat, cut_pi, cut_pi0, cut_lists, max_iter=0, kc=64):
        """Node LPs with per-node cut rows (mipx_lp_solve_batch_cuts): cut_lists[k] = ids (rows of
        cut_pi / cut_pi0) node k carries after the m shared rows.  vstat rows: n + m + len(cut_lists[k])
        codes each (a list of arrays) or None.  Returns dict like solve_batch; y and vstat are lists of
        per-node arrays over the node's own rows."""
        n, m = self.n, self.m
        l = np.ascontiguousarray(l, dtype=np.float64).reshape(-1, n)
        B = l.shape[0]
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(B, n)
        cut_pi = np.ascontiguousarray(cut_pi, dtype=np.float64).reshape(-1, n)
        cut_pi0 = np.ascontiguousarray(cut_pi0, dtype=np.float64).reshape(-1)
        ncut = np.array([len(c) for c in cut_lists], np.int32)
        ids = np.zeros((B, kc), np.int32)
        for k, c in enumerate(cut_lists):
            ids[k, :len(c)] = c
        M = m + kc
        vin = None
        if vstat is not None:
            vin = np.zeros((B, n + M), np.int8)
            for k in range(B):
                vin[k, :n + m + ncut[k]] = vstat[k]
        status = np.zeros(B, np.int32); obj = np.zeros(B); x = np.zeros((B, n)); y = np.zeros((B, M))
        vout = np.zeros((B, n + M), np.int8); iters = np.zeros(B, np.int32); npiv = np.zeros(B, np.int32)
        L = lib()
        L.mipx_lp_solve_batch_cuts.argtypes = [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp,
                                               C.c_int] + [_vp] * 7
        rc = L.mipx_lp_solve_batch_cuts(self._h, B, _ptr(l), _ptr(u), _ptr(vin), len(cut_pi0), _ptr(cut_pi),
                                        _ptr(cut_pi0), kc, _ptr(ncut), _ptr(ids), int(max_iter), _ptr(status),
                                        _ptr(obj), _ptr(x), _ptr(y), _ptr(vout), _ptr(iters), _ptr(npiv))
        self.ctx.check(rc, 'mipx_lp_solve_batch_cuts')
        return dict(status=status, obj=obj, x=x, iters=iters, npivots=npiv,
                    y=[y[k, :m + ncut[k]].copy() for k in range(B)],
                    vstat=[vout[k, :n + m + ncut[k]].copy() for k in range(B)])

    def dive_batch(self, l, u, vstat, rule, integer_indices, cost_l=None, cost_r=None, has_entry=None,
                   cutoff=float('inf'), max_iter=0, depth=1):
        """Node LPs with the in-place dive (mipx_lp_dive_batch; depth > 1: mipx_lp_plunge_batch): arrays
        of (depth + 1) * batch rows (nodes, then dive children level by level, status -1 where none)
        plus dive_var / dive_dir / dive_val with depth * batch entries (the decision after level p at
        p * batch + node)."""
        n, m = self.n, self.m
        l = np.ascontiguousarray(l, dtype=np.float64).reshape(-1, n)
        B = l.shape[0]
        D = int(depth)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(B, n)
        if vstat is not None:
            vstat = np.ascontiguousarray(vstat, dtype=np.int8).reshape(B, n + m)
        ii = np.ascontiguousarray(integer_indices, dtype=np.int32)
        cl = None if cost_l is None else np.ascontiguousarray(cost_l, np.float64)
        cr = None if cost_r is None else np.ascontiguousarray(cost_r, np.float64)
        he = None if has_entry is None else np.ascontiguousarray(has_entry, np.uint8)
        R = (D + 1) * B
        status = np.zeros(R, np.int32); obj = np.zeros(R); x = np.zeros((R, n))
        vout = np.zeros((R, n + m), np.int8); iters = np.zeros(R, np.int32)
        npiv = np.zeros(R, np.int32)
        dvar = np.zeros(D * B, np.int32); ddir = np.zeros(D * B, np.int32); dval = np.zeros(D * B)
        L = lib()
        L.mipx_lp_plunge_batch.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int,
                                           _vp, _vp, _vp, C.c_double] + [_vp] * 9
        rc = L.mipx_lp_plunge_batch(self._h, B, D, _ptr(l), _ptr(u), _ptr(vstat), int(max_iter), int(rule),
                                    _ptr(ii), len(ii), _ptr(cl), _ptr(cr), _ptr(he), float(cutoff),
                                    _ptr(status), _ptr(obj), _ptr(x), _ptr(vout), _ptr(iters), _ptr(npiv),
                                    _ptr(dvar), _ptr(ddir), _ptr(dval))
        self.ctx.check(rc, 'mipx_lp_plunge_batch')
        return dict(status=status, obj=obj, x=x, vstat=vout, iters=iters, npivots=npiv, dive_var=dvar,
                    dive_dir=ddir, dive_val=dval)

    def gomory_batch(self, l, u, vstat, x, integer_indices, max_term=1e3):
        """GMI cuts + safe rounding for solved nodes; list (one per node) of dicts with
        row_idx, pi, pi0, safe_pi, safe_pi0 (one row per cut)."""
        n, m = self.n, self.m
        l = np.ascontiguousarray(l, dtype=np.float64).reshape(-1, n)
        B = l.shape[0]
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(B, n)
        vstat = np.ascontiguousarray(vstat, dtype=np.int8).reshape(B, n + m)
        x = np.ascontiguousarray(np.maximum(np.asarray(x, dtype=np.float64), 0)).reshape(B, n)
        is_int = np.zeros(n, np.uint8)
        is_int[np.asarray(integer_indices, dtype=int)] = 1
        ncuts = np.zeros(B, np.int32); row_idx = np.zeros((B, max(m, 1)), np.int32)
        pi = np.zeros((B, max(m, 1), n)); pi0 = np.zeros((B, max(m, 1)))
        spi = np.zeros((B, max(m, 1), n)); spi0 = np.zeros((B, max(m, 1)))
        rc = lib().mipx_gomory_batch(self._h, B, _ptr(l), _ptr(u), _ptr(vstat), _ptr(x),
                                     _ptr(is_int), float(max_term), _ptr(ncuts), _ptr(row_idx),
                                     _ptr(pi), _ptr(pi0), _ptr(spi), _ptr(spi0))
        self.ctx.check(rc, 'mipx_gomory_batch')
        return [dict(row_idx=row_idx[k, :ncuts[k]], pi=pi[k, :ncuts[k]], pi0=pi0[k, :ncuts[k]],
                     safe_pi=spi[k, :ncuts[k]], safe_pi0=spi0[k, :ncuts[k]]) for k in range(B)]

    def solve_batch_dev(self, B, d_l, d_u, d_vstat, max_iter, d_status, d_obj, d_x, d_y, d_vout,
                        d_iters, d_npiv):
        rc = lib().mipx_lp_solve_batch_dev(self._h, int(B), d_l, d_u, d_vstat, int(max_iter),
                                           d_status, d_obj, d_x, d_y, d_vout, d_iters, d_npiv)
        self.ctx.check(rc, 'mipx_lp_solve_batch_dev')

    def close(self):
        if getattr(self, '_h', None) and getattr(self.ctx, '_h', None):
            lib().mipx_problem_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tree:
    """Native frontier engine on one problem (mipx_tree)."""

    def __init__(self, problem, integer_indices, l, u, branch_rule='most fractional',
                 search_rule='best first', strong_branch_iters=5, max_batch=1,
                 pool_capacity=1 << 16, cut_params=None):
        """cut_params: None (no cut rounds) or a dict of mipx_cut_params fields -- Gomory cut rounds
        run inside the engine (mipx_tree_create_ex)."""
        self.problem = problem
        ints = np.ascontiguousarray(integer_indices, dtype=np.int32)
        l = np.ascontiguousarray(l, dtype=np.float64).reshape(problem.n)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(problem.n)
        rule = {'most fractional': 0, 'pseudo cost': 1}[branch_rule]
        search = {'best first': 0, 'depth first': 1}[search_rule]
        h = _vp()
        L = lib()
        L.mipx_tree_create_ex.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int64, C.POINTER(CutParams), C.POINTER(_vp)]
        cp = None
        if cut_params is not None:
            cp = CutParams(max_cut_generation_iterations=10, max_nonzero_coefs=1000000, max_cuts_per_node=0,
                           exact_tableau=1, cutting_plane_progress_tolerance=1e-4, min_cut_depth=1e-8,
                           cos_parallel=0.984807753012208, max_abs_coef=1e6, max_term=1e3,
                           max_dual_bound=float('inf'), store_capacity=0)
            for key, value in cut_params.items():
                assert hasattr(cp, key), f'unknown cut parameter {key}'
                setattr(cp, key, value)
        rc = L.mipx_tree_create_ex(problem._h, _ptr(ints), len(ints), _ptr(l), _ptr(u), rule,
                                   search, int(strong_branch_iters), int(max_batch),
                                   int(pool_capacity), None if cp is None else C.byref(cp), C.byref(h))
        problem.ctx.check(rc, 'mipx_tree_create_ex')
        self.cuts = cp is not None
        self._h = h
        self.max_batch = int(max_batch)

    def solve(self, node_limit=0, mip_gap=1e-4, max_seconds=0.0, frontier_batch=None, max_steps=0):
        st = TreeStats()
        rc = lib().mipx_tree_solve(self._h, int(node_limit), float(mip_gap), float(max_seconds),
                                   int(frontier_batch or self.max_batch), int(max_steps),
                                   C.byref(st))
        err, self._hook_error = getattr(self, '_hook_error', None), None
        if err is not None:  # raised inside the step hook: the engine stopped, re-raise it here
            raise err
        comm = getattr(self, '_comm', None)
        if comm is not None and getattr(comm, '_err', None) is not None:   # raised inside a transport callback
            cerr, comm._err = comm._err, None
            raise cerr
        self.problem.ctx.check(rc, 'mipx_tree_solve')
        return st.as_dict()

    def reanchor(self, max_nodes):
        """Give the first max_nodes open nodes an anchor of their own (mipx_tree_reanchor)."""
        L = lib()
        L.mipx_tree_reanchor.argtypes = [_vp, C.c_int64]
        self.problem.ctx.check(L.mipx_tree_reanchor(self._h, int(max_nodes)), 'mipx_tree_reanchor')

    def peek_anchors(self, max_nodes):
        """Anchor-table entry of each open node, in the order of peek_open (-1: the root's anchor)."""
        a = np.full(int(max_nodes), -1, np.int32)
        L = lib()
        L.mipx_tree_peek_anchors.argtypes = [_vp, C.c_int64, _vp]
        L.mipx_tree_peek_anchors.restype = C.c_int64
        k = L.mipx_tree_peek_anchors(self._h, int(max_nodes), _ptr(a))
        if k < 0:
            self.problem.ctx.check(int(k), 'mipx_tree_peek_anchors')
        return a[:k]

    def anchor_table(self):
        """(T, vec, idx) of the re-anchoring table as host arrays, or None."""
        L = lib()
        L.mipx_tree_anchor_table.argtypes = [_vp, _vp, _vp, _vp]
        L.mipx_tree_anchor_table.restype = C.c_int64
        K = L.mipx_tree_anchor_table(self._h, None, None, None)
        if K <= 0:
            return None
        n, m = self.problem.n, self.problem.m
        T = np.zeros((K, m, n)); vec = np.zeros((K, n + 3 * m)); idx = np.zeros((K, 2 * n + m), np.int32)
        k = L.mipx_tree_anchor_table(self._h, _ptr(T), _ptr(vec), _ptr(idx))
        if k < 0:
            self.problem.ctx.check(int(k), 'mipx_tree_anchor_table')
        return T, vec, idx

    def set_dive(self, on=True):
        """In-place plunge on the tableau a node's workgroup holds (mipx_tree_set_dive): True / 1 one
        dive child per node, an int up to 8 that many in a row, False / 0 off."""
        self.problem.ctx.check(lib().mipx_tree_set_dive(self._h, int(on)), 'mipx_tree_set_dive')

    def set_step_hook(self, fn, every_steps=1):
        """Call fn() every `every_steps` frontier steps inside solve(), while the GPU works on the
        steps already queued (mipx_tree_set_step_hook): the place for a rank's all-reduce.  fn may
        use stats(), pseudo_cost_arrays(), set_primal_bound(), set_pseudo_cost_arrays(); a truthy
        return value or an exception stops the solve.  fn=None removes the hook."""
        L = lib()
        proto = C.CFUNCTYPE(C.c_int, _vp)
        L.mipx_tree_set_step_hook.argtypes = [_vp, proto, _vp, C.c_int]
        if fn is None:
            self._hook = None
            rc = L.mipx_tree_set_step_hook(self._h, proto(), None, 0)
        else:
            def trampoline(_user):
                try:
                    return 1 if fn() else 0
                except BaseException as e:  # never unwind through the C frames
                    self._hook_error = e
                    return 1
            self._hook = proto(trampoline)  # keep the thunk alive as long as it is installed
            rc = L.mipx_tree_set_step_hook(self._h, self._hook, None, int(every_steps))
        self.problem.ctx.check(rc, 'mipx_tree_set_step_hook')

    def set_comm(self, comm, every_steps=5):
        """Attach the communicator: solve() becomes a collective call (mipx_tree_set_comm)."""
        L = lib()
        L.mipx_tree_set_comm.argtypes = [_vp, _vp, C.c_int]
        self._comm = comm
        self.problem.ctx.check(L.mipx_tree_set_comm(self._h, None if comm is None else comm._h, int(every_steps)),
                               'mipx_tree_set_comm')

    def migrate_self(self, amount):
        """Test hook (mipx_tree_migrate_self): up to `amount` open nodes leave and re-enter this rank through
        the communicator's point-to-point path; returns how many moved."""
        L = lib()
        L.mipx_tree_migrate_self.argtypes = [_vp, C.c_int64]
        L.mipx_tree_migrate_self.restype = C.c_int64
        k = L.mipx_tree_migrate_self(self._h, int(amount))
        if k < 0:
            self.problem.ctx.check(int(k), 'mipx_tree_migrate_self')
        return int(k)

    def exchange_record(self):
        """The record this rank would post right now (mipx_tree_exchange_record; needs set_comm)."""
        r = np.zeros(exchange_record_len(self.problem.n))
        L = lib()
        L.mipx_tree_exchange_record.argtypes = [_vp, _vp]
        self.problem.ctx.check(L.mipx_tree_exchange_record(self._h, _ptr(r)), 'mipx_tree_exchange_record')
        return r

    def global_stats(self):
        st = GlobalStats()
        L = lib()
        L.mipx_tree_global_stats.argtypes = [_vp, C.POINTER(GlobalStats)]
        self.problem.ctx.check(L.mipx_tree_global_stats(self._h, C.byref(st)), 'mipx_tree_global_stats')
        return {k: getattr(st, k) for k, _ in st._fields_}

    def kernel_ms(self):
        """Device time by kernel (ms): dict(node_lp, gomory, select) (mipx_tree_kernel_ms)."""
        out = (C.c_double * 4)()
        L = lib()
        L.mipx_tree_kernel_ms.argtypes = [_vp, _vp]
        self.problem.ctx.check(L.mipx_tree_kernel_ms(self._h, out), 'mipx_tree_kernel_ms')
        return dict(node_lp=out[0], gomory=out[1], select=out[2])

    def cut_stats(self):
        """The running GMIC totals of BaseNode._base_bound over every evaluated node (+ 'dropped')."""
        out = (C.c_int64 * 8)()
        L = lib()
        L.mipx_tree_cut_stats.argtypes = [_vp, _vp]
        self.problem.ctx.check(L.mipx_tree_cut_stats(self._h, out), 'mipx_tree_cut_stats')
        d = {k: int(out[i]) for i, k in enumerate(CUT_TOTAL_KEYS)}
        d['dropped'] = int(out[7])
        return d

    def stats(self):
        st = TreeStats()
        self.problem.ctx.check(lib().mipx_tree_get_stats(self._h, C.byref(st)), 'mipx_tree_get_stats')
        return st.as_dict()

    def solution(self):
        x = np.zeros(self.problem.n)
        self.problem.ctx.check(lib().mipx_tree_solution(self._h, _ptr(x)), 'mipx_tree_solution')
        return x

    def set_primal_bound(self, bound):
        self.problem.ctx.check(lib().mipx_tree_set_primal_bound(self._h, float(bound)),
                               'mipx_tree_set_primal_bound')

    def pseudo_costs(self):
        """The table in the reference's layout {idx: {'left'|'right': {'cost', 'times'}}}."""
        n = self.problem.n
        cl, cr = np.zeros(n), np.zeros(n)
        tl, tr = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.problem.ctx.check(lib().mipx_tree_pseudo_costs(self._h, _ptr(cl), _ptr(cr), _ptr(tl),
                                                            _ptr(tr)), 'mipx_tree_pseudo_costs')
        out = {}
        for i in range(n):
            if tl[i] or tr[i]:
                out[i] = {'left': {'cost': float(cl[i]), 'times': int(tl[i])},
                          'right': {'cost': float(cr[i]), 'times': int(tr[i])}}
        return out

    def pseudo_cost_arrays(self):
        """(cost_l, cost_r, times_l, times_r) as arrays over all n variables."""
        n = self.problem.n
        cl, cr = np.zeros(n), np.zeros(n)
        tl, tr = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self.problem.ctx.check(lib().mipx_tree_pseudo_costs(self._h, _ptr(cl), _ptr(cr), _ptr(tl),
                                                            _ptr(tr)), 'mipx_tree_pseudo_costs')
        return cl, cr, tl, tr

    def set_pseudo_cost_arrays(self, cl, cr, tl, tr):
        cl = np.ascontiguousarray(cl, np.float64); cr = np.ascontiguousarray(cr, np.float64)
        tl = np.ascontiguousarray(tl, np.int32); tr = np.ascontiguousarray(tr, np.int32)
        L = lib()
        L.mipx_tree_set_pseudo_costs.argtypes = [_vp, _vp, _vp, _vp, _vp]
        self.problem.ctx.check(L.mipx_tree_set_pseudo_costs(self._h, _ptr(cl), _ptr(cr), _ptr(tl), _ptr(tr)),
                               'mipx_tree_set_pseudo_costs')

    def peek_open(self, max_nodes):
        """(l, u, vstat, dual_bound) of up to max_nodes open nodes, without removing them."""
        n, nv = self.problem.n, self.problem.n + self.problem.m
        l = np.zeros((max_nodes, n)); u = np.zeros((max_nodes, n))
        v = np.zeros((max_nodes, nv), np.int8); db = np.zeros(max_nodes)
        k = lib().mipx_tree_peek_open(self._h, int(max_nodes), _ptr(l), _ptr(u), _ptr(v), _ptr(db))
        if k < 0:
            self.problem.ctx.check(int(k), 'mipx_tree_peek_open')
        return l[:k], u[:k], v[:k], db[:k]

    def keep_shard(self, rank, world):
        self.problem.ctx.check(lib().mipx_tree_keep_shard(self._h, int(rank), int(world)),
                               'mipx_tree_keep_shard')

    def set_anchor_mode(self, on=True):
        self.problem.ctx.check(lib().mipx_tree_set_anchor_mode(self._h, int(on)),
                               'mipx_tree_set_anchor_mode')

    def set_trace(self, on=True):
        self.problem.ctx.check(lib().mipx_tree_set_trace(self._h, int(on)), 'mipx_tree_set_trace')

    def trace(self):
        k = lib().mipx_tree_trace(self._h, 0, None, None, None, None)
        ids = np.zeros(k, np.int64); st = np.zeros(k, np.int32)
        bv = np.zeros(k, np.int32); obj = np.zeros(k)
        lib().mipx_tree_trace(self._h, k, _ptr(ids), _ptr(st), _ptr(bv), _ptr(obj))
        return dict(node_id=ids, status=st, branch_var=bv, objective=obj)

    def trace_cuts(self):
        """(nodes, 8) per evaluated node in trace order: cut rounds, iterations / number of GMICs created,
        added, removed, cut rows at the end (mipx_tree_trace_cuts)."""
        L = lib()
        L.mipx_tree_trace_cuts.argtypes = [_vp, C.c_int64, _vp]
        L.mipx_tree_trace_cuts.restype = C.c_int64
        k = L.mipx_tree_trace_cuts(self._h, 0, None)
        out = np.zeros((max(k, 0), 8), np.int32)
        L.mipx_tree_trace_cuts(self._h, k, _ptr(out))
        return out

    def peek_cuts(self, max_nodes):
        """(node ids, cut counts, cut lists, basis codes of the cut rows) of the open nodes, in the order
        of peek_open (mipx_tree_peek_cuts)."""
        L = lib()
        L.mipx_tree_peek_cuts.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp]
        L.mipx_tree_peek_cuts.restype = C.c_int64
        L.mipx_tree_cut_rows_per_node.argtypes = [_vp]
        kc = max(0, L.mipx_tree_cut_rows_per_node(self._h))
        ids = np.zeros(max_nodes, np.int64); ncut = np.zeros(max_nodes, np.int32)
        lists = np.zeros((max_nodes, max(kc, 1)), np.int32); codes = np.zeros((max_nodes, max(kc, 1)), np.int8)
        k = L.mipx_tree_peek_cuts(self._h, int(max_nodes), _ptr(ids), _ptr(ncut), _ptr(lists), _ptr(codes))
        if k < 0:
            self.problem.ctx.check(int(k), 'mipx_tree_peek_cuts')
        return ids[:k], ncut[:k], lists[:k], codes[:k]

    def cut_store(self):
        """(pi, pi0) of every cut added so far (mipx_tree_cut_store)."""
        L = lib()
        L.mipx_tree_cut_store.argtypes = [_vp, C.c_int64, _vp, _vp]
        L.mipx_tree_cut_store.restype = C.c_int64
        k = L.mipx_tree_cut_store(self._h, 0, None, None)
        if k < 0:
            self.problem.ctx.check(int(k), 'mipx_tree_cut_store')
        pi = np.zeros((k, self.problem.n)); pi0 = np.zeros(k)
        if k:
            L.mipx_tree_cut_store(self._h, k, _ptr(pi), _ptr(pi0))
        return pi, pi0

    def close(self):
        if getattr(self, '_h', None) and getattr(self.problem, '_h', None):
            lib().mipx_tree_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kernel_name(m, n):
    buf = C.create_string_buffer(128)
    rc = lib().mipx_kernel_name(int(m), int(n), buf, 128)
    if rc != MIPX_OK:
        raise MipxError(f'no on-chip kernel for m={m}, n={n}: {ERRORS.get(rc, rc)}')
    return buf.value.decode()


_default_ctx = None


def default_context():
    """Process-wide context on the GPU selected by LOCAL_RANK (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get('LOCAL_RANK', '0')))
    return _default_ctx


def debug_dump(problem, l, u, vstat=None, max_iter=0):
    """Test hook: solve one LP and return the kernel's final tableau state (see mipx.h)."""
    L = lib()
    L.mipx_debug_enable.argtypes = [_vp]
    L.mipx_debug_read.argtypes = [_vp, _vp, _vp, _vp]
    problem.ctx.check(L.mipx_debug_enable(problem._h), 'mipx_debug_enable')
    res = problem.solve_batch(np.asarray(l, float)[None], np.asarray(u, float)[None],
                              None if vstat is None else np.asarray(vstat, np.int8)[None], max_iter)
    m, n = problem.m, problem.n
    T = np.zeros((m, n)); vec = np.zeros(n + 3 * m); idx = np.zeros(2 * n + m, np.int32)
    problem.ctx.check(L.mipx_debug_read(problem._h, _ptr(T), _ptr(vec), _ptr(idx)), 'mipx_debug_read')
    return res, dict(T=T, d=vec[:n], beta0=vec[n:n + m], ba=vec[n + m:n + 2 * m],
                     bb=vec[n + 2 * m:], nvar=idx[:n], bvar=idx[n:n + m], side=idx[n + m:])
