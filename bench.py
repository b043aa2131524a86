#!/usr/bin/env python3
"""bench.py -- node LP-relaxations/s (+ time-to-optimal leg) on the C3 workload of BASELINE.md.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ONE 256 vars x
128 rows random dense MILP (seed 0), PseudoCostBranchNode, best-first, pseudo_costs={},
strong_branch_iters=5, gomory_cuts=False, solved by the native frontier engine (mipx_tree_*): node
records resident in HBM, every step = one frontier batch through the whole node hot path (LP
relaxation to termination, strong-branching probes where the pseudo-cost table has no entry yet,
pseudo-cost update, branching index, child materialisation).  Untimed: the ramp-up until every rank
owns >= B open nodes, then W warm-up steps.  Timed: exactly K steps, barrier + device sync on both
sides, MAX over ranks.  value = node LP relaxations (solved to termination; probes are reported
separately) per second, summed over ranks.

N > 1: `python bench.py --gpus N` spawns N processes (one per GPU; the driver's launcher form with
RANK / WORLD_SIZE in the environment works too).  All ranks run the same deterministic ramp-up,
each keeps its share of the open nodes and searches it with its own best-first queue; incumbent
(value + solution), bounds, counters, pseudo-cost samples and -- when a shard runs dry -- node
records are exchanged over RCCL from inside libmipx.so (no PyTorch anywhere in the product).
Per-GPU frontier batch is fixed -> "weak" scaling.

Also measured inside the run, after the timed region (rank 0): the other BASELINE configs (C2, C4,
C5 single-GPU) under config.others, the time-to-optimal leg on the metric's own instance
(depth-first: first incumbent, gap at a time limit) and the CPU baselines.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
# f64 vector FMA: half the guide's 157.3 TFLOP/s f32 vector rate; scripts/microbench/prim.hip measures
# 64 dependent-free v_fma_f64 in 278 cycles per wave = 72 TFLOP/s over 1024 SIMDs at 2.4 GHz
F64_VECTOR_PEAK_TFLOPS = 78.6


def algorithmic_bytes(m, n, lps, pivots, dives=0):
    """SURVEY.md section 8(d), dense-tableau model, f64: per LP load A,b,c,l,u + store x, obj,
    status, basis; per pivot one read + one write of the bordered tableau.  A dive child continues
    on its parent's tableau: it is charged its outputs and pivots, not a load."""
    per_pivot = 2 * 8 * (m + 1) * (n + m + 1)
    load = 8 * (m * n + m + 3 * n)
    store = 8 * (n + 2) + (n + m)
    return (lps - dives) * load + lps * store + pivots * per_pivot


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8192, help='frontier nodes per step per GPU')
    ap.add_argument('--vars', type=int, default=256)
    ap.add_argument('--cons', type=int, default=128)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cpu-seconds', type=float, default=6.0, help='wall seconds of the CPU port baseline (0: skip)')
    ap.add_argument('--cpu-threads', type=int, default=0, help='threads of the CPU baseline; 0 = every CPU this process may use')
    ap.add_argument('--highs-seconds', type=float, default=5.0, help='wall seconds of the HiGHS baseline B2 (0: skip)')
    ap.add_argument('--tto-seconds', type=float, default=8.0, help='time limit of the time-to-gap leg on the metric\'s instance (0: skip both time legs)')
    ap.add_argument('--others', type=int, default=1, choices=[0, 1], help='1: also measure C2, C4, C5 (config.others)')
    ap.add_argument('--no-dive-leg', type=int, default=1, choices=[0, 1],
                    help='1: also time the same region with --dive 0 (value_no_dive; one GPU only)')
    ap.add_argument('--exchange-every', type=int, default=5, help='steps between exchanges (N > 1)')
    ap.add_argument('--dive', type=int, default=8, choices=range(0, 9),
                    help='dive children solved in a row on the tableau a node\'s workgroup holds (mipx_tree_set_dive; 0: off)')
    ap.add_argument('--reanchor', type=int, default=1, choices=[0, 1],
                    help='1: after sharding every open node gets an anchor of its own (mipx_tree_reanchor)')
    ap.add_argument('--no-anchor', action='store_true',
                    help='refactor every node from the slack basis instead of the root tableau')
    return ap.parse_args()


def spawn_ranks(args):
    """`bench.py --gpus N` from a plain shell: N fresh child processes, one per GPU, before this
    process has touched the GPU (it never does).  Rank 0's JSON line is relayed; any failing child
    fails the run."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread; the ranks are polled: the first one that exits non-zero takes
    # the others down with it (a rank that died would otherwise leave its peers in the next all-gather)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            time.sleep(2.0)   # (let the ranks that were told -- MIPX_EPEER -- print their own errors)
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b''.join(chunks).decode())
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write(f'bench.py: rank exit codes {codes}\n')
        return 1
    return 0


def host_cpus():
    """(logical CPUs of the host, physical cores, CPUs this process may run on, model name)."""
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = logical
    try:   # a cgroup quota narrows it further
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except Exception:
        pass
    cores, model, phys, core = set(), '', None, None
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name') and not model:
                model = ln.split(':', 1)[1].strip()
            elif ln.startswith('physical id'):
                phys = ln.split(':', 1)[1].strip()
            elif ln.startswith('core id'):
                core = ln.split(':', 1)[1].strip()
                cores.add((phys, core))
    except Exception:
        pass
    return logical, len(cores) or logical, usable, model


def cpu_port_baseline(args, tree, A, b, c, l, u, ints, B):
    """B1: the oracle (the build's own C restatement of the path) on the very node LPs the GPU is about
    to solve, OpenMP-style over nodes on every CPU this process may use."""
    from oracle import oracle as O
    from concurrent.futures import ThreadPoolExecutor
    import contextlib
    logical, physical, usable, model = host_cpus()
    threads = args.cpu_threads if args.cpu_threads > 0 else usable
    peek = 4 * B * (3 if threads > 4 else 1)
    L, U, V, _ = tree.peek_open(peek)
    atab = tree.anchor_table() if args.reanchor and not args.no_anchor else None
    asel = tree.peek_anchors(peek) if atab is not None else None
    chunk = max(8, min(128, len(L) // (4 * threads) or 8))
    nchunks = (len(L) + chunk - 1) // chunk
    pc_tab = tree.pseudo_cost_arrays()
    pc_has = ((pc_tab[2] > 0) | (pc_tab[3] > 0)).astype(np.uint8)
    cutoff = tree.stats()['primal_bound']
    deadline = [0.0]

    def work(ci):  # ctypes releases the GIL inside the C oracle: the threads run in parallel
        if time.perf_counter() > deadline[0]:
            return 0
        e0 = min((ci + 1) * chunk, len(L))
        sl = slice(ci * chunk, e0)
        if args.dive:  # like the GPU path: node + one child continued on the node's tableau
            r = O.lp_solve_dive_batch(A, b, c, L[sl], U[sl], V[sl], 1, ints, pc_tab[0], pc_tab[1], pc_has, cutoff,
                                      anchor_table=atab, anchor_sel=None if asel is None else asel[sl], depth=args.dive)
            return e0 - ci * chunk + int((r['dive_var'] >= 0).sum())
        if atab is not None:  # rule -1: no dive, anchors from the table
            O.lp_solve_dive_batch(A, b, c, L[sl], U[sl], V[sl], -1, ints, pc_tab[0], pc_tab[1], pc_has, cutoff,
                                  anchor_table=atab, anchor_sel=asel[sl])
        else:
            O.lp_solve_batch(A, b, c, L[sl], U[sl], V[sl])
        return e0 - ci * chunk

    anchor_cm = contextlib.nullcontext()
    if not args.no_anchor:  # like the GPU path, warm starts refactor from the root's optimal tableau
        root = O.lp_solve(A, b, c, l, u)
        anchor_cm = O.anchored(O.make_anchor(A, b, c, root['vstat']))
    tc = time.perf_counter()
    deadline[0] = tc + args.cpu_seconds
    done, passes = 0, 0
    with anchor_cm, ThreadPoolExecutor(threads) as ex:
        while passes == 0 or time.perf_counter() < deadline[0]:   # whole passes until the time is up
            done += int(sum(ex.map(work, range(nchunks))))
            passes += 1
    t_cpu = time.perf_counter() - tc
    how = 'slack-basis refactorisation' if args.no_anchor else \
        ('anchored like the GPU path: the tableau of the re-anchored ancestor, read back from the device'
         if atab is not None else 'anchored at the root tableau like the GPU path')
    return {'value': done / t_cpu, 'unit': 'node LP-relaxations/s', 'cores': threads, 'kind': 'port',
            'physical_cores': physical, 'logical_cpus': logical, 'cpus_usable': usable, 'cpu_model': model,
            'seconds': t_cpu,
            'sample': f'{done} node LPs in {t_cpu:.1f} s = {passes} pass(es) (the last cut off at the time limit) over '
                      f'{len(L)} open nodes of the same tree{" and their dive children" if args.dive else ""} (the LPs '
                      f'the GPU solves next: bounds + warm-start bases read back from the device pool), '
                      f'oracle/libmipx_oracle.so ({how}), {threads} threads over nodes = every CPU this process may '
                      f'run on (host: {logical} logical / {physical} physical cores, {model})'}, (L, U, V)


def highs_baseline(args, prob, A, b, c, sample):
    """B2: an independent solver -- scipy's HiGHS dual simplex, one LP per call, one core, cold start
    (linprog takes no basis), Python call overhead included -- on a slice of the same node LPs, with
    the objectives cross-checked against the GPU's."""
    from scipy.optimize import linprog
    L, U, V = sample
    k_max = min(len(L), 4096)
    gpu = prob.solve_batch(L[:k_max], U[:k_max], V[:k_max])
    t0 = time.perf_counter()
    k, worst, compared = 0, 0.0, 0
    while k < k_max and time.perf_counter() - t0 < args.highs_seconds:
        res = linprog(c, A_ub=-A, b_ub=-b, bounds=np.stack([L[k], U[k]], axis=1), method='highs-ds')
        if res.status == 0 and gpu['status'][k] == 0:
            worst = max(worst, abs(res.fun - gpu['obj'][k]) / max(1.0, abs(res.fun)))
            compared += 1
        elif res.status == 2:
            assert gpu['status'][k] == 1, 'HiGHS says infeasible, the engine does not'
        k += 1
    el = time.perf_counter() - t0
    assert worst <= 1e-6, f'HiGHS / engine objective mismatch {worst}'
    return {'value': k / el, 'unit': 'node LP-relaxations/s', 'cores': 1, 'kind': 'independent solver',
            'solver': "scipy.optimize.linprog(method='highs-ds')", 'lps': k, 'seconds': el,
            'max_rel_objective_diff_vs_gpu': worst, 'objectives_compared': compared,
            'note': 'cold start per LP (no warm-start interface), one core, Python call overhead included; the '
                    'same node LPs the GPU solved (first %d of the sample)' % k}


GAP_MARKS = (0.01, 0.005, 0.0035, 0.003, 0.0025)


def two_phase(ctx, A, b, c, l, u, ints, dive, dfs_seconds, limit, pool_log2, mip_gap=1e-4, marks=()):
    """The reference's loop ends on current_gap <= mip_gap (branch_and_bound.py:199-229).  Two phases
    on the engine: depth first (DepthFirstSearchNode semantics, nodes/search/depth_first.py:16-28)
    until `dfs_seconds` have passed, for an incumbent; then best first on a fresh tree with that
    incumbent installed as initial_primal_bound (branch_and_bound.py:121, :157), which lifts the dual
    bound.  The gap of the run at any time is |best incumbent - proven dual bound| / |incumbent|
    with the bounds of the phase in progress (both are bounds of the same problem).  Returns the
    times at which the gap first fell below each of `marks`."""
    from simple_mip_solver_amd import _ffi
    p = _ffi.Problem(ctx, A, b, c)
    inf = float('inf')
    t0 = time.perf_counter()
    reached = {}
    nodes = 0
    alloc = [0.0]   # seconds spent creating and freeing the trees' node pools (up to 75 GB: 0.3 s on one box, 3 s on another)

    def clock():
        """Search time: wall time since the start minus the allocations -- a phase's budget is for the search, and
        a time-to-gap mark must not depend on how long hipMalloc takes on the box."""
        return time.perf_counter() - t0 - alloc[0]

    def timed(fn):
        a0 = time.perf_counter()
        out = fn()
        alloc[0] += time.perf_counter() - a0
        return out

    def note(s, extra_nodes=0):
        g = s['gap']
        for mk in marks:
            if mk not in reached and 0 <= g <= mk:
                reached[mk] = {'seconds': clock(), 'nodes': extra_nodes + s['evaluated_nodes'],
                               'primal_bound': s['primal_bound'], 'dual_bound': s['dual_bound']}
    t = timed(lambda: _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', search_rule='depth first', max_batch=1024,
                                pool_capacity=1 << 21))
    t.set_anchor_mode(True)
    t.set_dive(max(1, dive))
    first, s = None, None
    while s is None or clock() < dfs_seconds:
        s = t.solve(mip_gap=mip_gap, frontier_batch=1024, max_steps=2 if first is None else 20)
        if first is None and s['primal_bound'] < inf:
            first = {'seconds': clock(), 'nodes': s['evaluated_nodes'], 'objective': s['primal_bound']}
        note(s)
        if s['status'] != 4:
            break
    phase1 = {'seconds': clock(), 'nodes': s['evaluated_nodes'],
              'primal_bound': None if s['primal_bound'] == inf else s['primal_bound'], 'dual_bound': s['dual_bound']}
    nodes = s['evaluated_nodes']
    pb = s['primal_bound']
    timed(t.close)
    if s['status'] == 4:
        t = timed(lambda: _ffi.Tree(p, ints, l, u, branch_rule='pseudo cost', max_batch=8192, pool_capacity=1 << pool_log2))
        t.set_anchor_mode(True)
        t.set_dive(max(1, dive))
        if pb < inf:
            t.set_primal_bound(pb)
        s = None
        while s is None or clock() < limit:
            s = t.solve(mip_gap=mip_gap, frontier_batch=8192, max_steps=10)
            note(s, nodes)
            if s['status'] != 4 or s['pool_exhausted']:
                break
        nodes += s['evaluated_nodes']
        el = clock()
        timed(t.close)
    else:
        el = clock()
    p.close()
    return {'time_to_first_incumbent': first, 'phase_1_depth_first': phase1, 'status': _ffi.TREE_STATUS[s['status']],
            'seconds': el, 'allocation_seconds': alloc[0], 'time_to_optimal': el if s['status'] == 1 else None,
            'primal_bound': None if s['primal_bound'] == inf else s['primal_bound'], 'dual_bound': s['dual_bound'],
            'gap': None if s['gap'] < 0 else s['gap'], 'nodes': nodes, 'pool_exhausted': bool(s['pool_exhausted']),
            'clock': 'search seconds: wall time minus allocation_seconds (creating and freeing the node pools of the two trees)',
            'time_to_gap': {f'{100 * mk:g}%': reached.get(mk) for mk in marks}}


def time_to_optimal_leg(args, ctx, A, b, c, l, u, ints):
    """The metric's second leg on its own 256 x 128 instance: the times at which the gap falls below
    1 %, 0.5 %, 0.35 % ... inside the time limit (the tree of this instance does not close at 1e-4 at
    any node rate reached so far: DESIGN.md section 5)."""
    out = two_phase(ctx, A, b, c, l, u, ints, args.dive, dfs_seconds=0.25 * args.tto_seconds, limit=args.tto_seconds,
                    pool_log2=24, marks=GAP_MARKS)
    out.update({'instance': f'{len(c)} vars x {len(b)} rows, seed {args.seed} (the metric\'s own instance)',
                'search': f'two phases in the native engine: depth first (1024 nodes per step) for {0.25 * args.tto_seconds:g} s, then '
                          f'best first (8192 nodes per step) with the incumbent installed; in-place dive of depth {max(1, args.dive)}',
                'time_limit_seconds': args.tto_seconds,
                'note': 'time_to_optimal is null unless the gap closed to 1e-4 inside the time limit; time_to_gap: first time '
                        'the gap (incumbent vs proven dual bound) was at or below the mark, null if never'})
    return out


def largest_closing_time_to_optimal(ctx, dive, n=144, m=72, limit=60.0):
    """time_to_optimal on the LARGEST instance of the same generator that closes to mip_gap = 1e-4 inside
    the stated limit (a smaller config than the metric's, labelled as such)."""
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    A2, b2, c2, l2, u2, ints2 = random_dense_milp_arrays(n, m, seed=0)
    out = two_phase(ctx, A2, b2, c2, l2, u2, ints2, dive, dfs_seconds=TTO_DFS_SECONDS, limit=limit, pool_log2=25)
    out.update({'instance': f'{n} vars x {m} rows, seed 0, same generator (a SMALLER config than the metric\'s: the largest '
                            f'that closes inside the limit)', 'time_limit_seconds': limit,
                'search': f'two phases: depth first for {TTO_DFS_SECONDS:g} s, then best first with the incumbent installed'})
    out.pop('time_to_gap', None)
    return out


TTO_DFS_SECONDS = 1.0


def no_dive_leg(args, ctx, prob, l, u, ints, B):
    """The same timed region WITHOUT the plunge (--dive 0): pure best first as north_star states it --
    every LP of a step is one of the B best open nodes.  Same instance, ramp-up, re-anchoring and timing
    brackets and number of steps."""
    from simple_mip_solver_amd import _ffi
    steps = max(10, args.steps)
    n, m = prob.n, prob.m
    pool = min(2 * B * (steps + max(3, args.warmup) + 8) + 4 * B, int(40e9 // (2 * 8 * n + n + m)))
    t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', search_rule='best first', strong_branch_iters=5,
                  max_batch=B, pool_capacity=pool)
    if not args.no_anchor:
        t.set_anchor_mode(True)
    st = t.stats()
    while st['open_nodes'] < B or st['evaluated_nodes'] == 0:
        st = t.solve(mip_gap=0.0, frontier_batch=min(B, 1024), max_steps=1)
    if args.reanchor and not args.no_anchor:
        t.reanchor(t.stats()['open_nodes'])
    t.solve(mip_gap=0.0, frontier_batch=B, max_steps=max(3, args.warmup))
    b0 = t.stats()
    ctx.sync()
    t0 = time.perf_counter()
    st = t.solve(mip_gap=0.0, frontier_batch=B, max_steps=steps)
    ctx.sync()
    el = time.perf_counter() - t0
    d = {k: st[k] - b0[k] for k in ('lp_solved', 'pivots', 'kernel_ms', 'steps', 'dives')}
    t.close()
    assert d['dives'] == 0 and d['steps'] == steps
    return {'value': d['lp_solved'] / el, 'unit': 'node LP-relaxations/s', 'steps': steps, 'ms_per_step': el / steps * 1e3,
            'launch_ms': d['kernel_ms'] / steps, 'lps_per_step': d['lp_solved'] / steps,
            'mean_pivots_per_lp': d['pivots'] / max(1, d['lp_solved']),
            'note': 'dive = 0: every LP of a step is one of the B best open nodes of the queue (pure best first)'}


def other_configs(args, ctx):
    """BASELINE configs C2, C4 and C5 (single GPU), measured in this run after the headline region."""
    from simple_mip_solver_amd import _ffi
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    out = {}
    # ---- C2: 1024 independent 64 x 32 roots, cold start, one launch (host buffers: PCIe inclusive)
    Bn, n2, m2 = 1024, 64, 32
    P = [random_dense_milp_arrays(n2, m2, seed=k) for k in range(Bn)]
    A2 = np.stack([p[0] for p in P]); b2 = np.stack([p[1] for p in P]); c2 = np.stack([p[2] for p in P])
    l2 = np.stack([p[3] for p in P]); u2 = np.stack([p[4] for p in P])
    best = None
    for _ in range(3):
        ctx.sync(); ctx.timer_start()
        t0 = time.perf_counter()
        g = _ffi.solve_multi(ctx, A2, b2, c2, l2, u2)
        wall = time.perf_counter() - t0
        dev_ms = ctx.timer_stop()
        if best is None or wall < best[0]:
            best = (wall, dev_ms, g, ctx.last_kernel_ms())
    wall, dev_ms, g, k_ms = best
    piv = int(g['npivots'].sum())
    byt = algorithmic_bytes(m2, n2, Bn, piv)
    out['C2'] = {'workload': '1024 independent random dense MILPs, 64 vars x 32 rows, seeds 0..1023, root relaxation, cold '
                             'start, one launch (mipx_lp_solve_multi; HOST buffers: the PCIe copies are inside)',
                 'kernel': _ffi.kernel_name(m2, n2), 'lps_per_s': Bn / wall, 'wall_ms': wall * 1e3,
                 'stream_ms_incl_copies': dev_ms, 'kernel_ms': k_ms,
                 'kernel_lps_per_s': None if k_ms <= 0 else Bn / (k_ms * 1e-3),   # (inputs resident in HBM: the launch alone)
                 'mean_pivots_per_lp': piv / Bn,
                 'optimal': int((g['status'] == 0).sum()),
                 'roofline': {'bound': 'hbm', 'model_hbm_GBps': byt / (dev_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBPS,
                              'note': 'algorithmic bytes (SURVEY 8d) over the stream time incl. the PCIe copies; the '
                                      'tableau is register-resident, the launch is bound by the 0.6 MB/LP of copies and '
                                      'the dependent pivots of one cold LP'}}
    # ---- C4: the metric's instance with Gomory cut rounds inside the engine; C4b: the same shape on a family
    # where the reference's selection rules DO add cuts below the root (density 0.1), so that cut rows are
    # carried, re-solved over (K1's cut-row tile) and removed again inside the timed steps
    n, m = 256, 128

    def cut_config(seed, density, B4, steps4, label, n=256, m=128):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
        prob = _ffi.Problem(ctx, A, b, c)
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=B4, pool_capacity=2 * B4 * (steps4 + 14),
                      cut_params=dict(max_abs_coef=1000.0 * float(np.max(np.abs(A))), exact_tableau=0))
        t.set_anchor_mode(True)
        st = t.stats()
        while st['open_nodes'] < B4 or st['evaluated_nodes'] == 0:
            st = t.solve(mip_gap=0.0, frontier_batch=min(B4, 1024), max_steps=1)
        t.reanchor(st['open_nodes'])   # as C3: the open nodes that carry no cut row get anchors of their own
        b0, c0, k0 = t.stats(), t.cut_stats(), t.kernel_ms()
        ctx.sync()
        t0 = time.perf_counter()
        st = t.solve(mip_gap=0.0, frontier_batch=B4, max_steps=steps4)
        ctx.sync()
        el = time.perf_counter() - t0
        c1, k1 = t.cut_stats(), t.kernel_ms()
        d4 = {k: st[k] - b0[k] for k in ('evaluated_nodes', 'lp_solved', 'probes_solved', 'pivots', 'steps')}
        created = c1['total_number_gmic_created'] - c0['total_number_gmic_created']
        k2_s = (k1['gomory'] - k0['gomory']) * 1e-3
        # K2's slack substitution pi + A' pi_s: one multiply-add per (cut, row, column) -- its GEMM-shaped part
        k2_flops = 2.0 * m * n * created
        out = {'workload': label, 'kernel': _ffi.kernel_name(m, n) + ' (cut-row variant)',
               'nodes_per_s': d4['evaluated_nodes'] / el, 'lps_per_s': d4['lp_solved'] / el,
               'ms_per_step': el / max(1, d4['steps']) * 1e3, 'steps': d4['steps'],
               'mean_pivots_per_lp': d4['pivots'] / max(1, d4['lp_solved']),
               'cut_rounds': c1['total_cut_generation_iterations'] - c0['total_cut_generation_iterations'],
               'gmic_created': created,
               'gmic_added': c1['total_number_gmic_added'] - c0['total_number_gmic_added'],
               'gmic_removed': c1['total_number_gmic_removed'] - c0['total_number_gmic_removed'],
               'gmic_dropped_for_capacity': c1['dropped'] - c0['dropped'],
               'kernel_ms_per_step': {'K1_first_solves': (k1['node_lp'] - k0['node_lp']) / max(1, d4['steps']),
                                      'K2_gomory': (k1['gomory'] - k0['gomory']) / max(1, d4['steps']),
                                      'pool_append_K3_select': (k1['select'] - k0['select']) / max(1, d4['steps'])},
               'roofline': {'kernel': 'gomory_cuts<256> (K2)', 'bound': 'mfma',
                            'achieved': None if k2_s <= 0 else k2_flops / k2_s / 1e12, 'peak': F64_VECTOR_PEAK_TFLOPS,
                            'unit': 'TFLOP/s', 'frac': None if k2_s <= 0 else k2_flops / k2_s / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                            'note': 'the multiply-adds of the slack substitution (2 m n per cut created) over K2\'s own time '
                                    '(HIP events); they run as f64 VALU mul + add in the reference\'s row order -- an MFMA '
                                    'formulation would change the summation order (DESIGN.md 4b) -- and the kernel is bound '
                                    'by its instruction stream (profiles/: 5 vector + 3 scalar instructions per multiply-add), '
                                    'the continued-fraction rounding and three barriers per group of 8 cuts'}}
        t.close()
        prob.close()
        return out
    out['C4'] = cut_config(args.seed, 1.0, 4096, 10,
                           'C3\'s instance with gomory_cuts=True (the reference default): every node runs BaseNode._base_bound\'s '
                           'cut loop inside the engine, 4096 nodes per step, best-first, anchored, no dive')
    out['C4']['note'] = ('on this dense family the reference\'s selection rules reject every rounded GMIC (depth >= 0 after the '
                         'outer rounding; tests/golden/base_node_large.npz holds the reference\'s own answer for this root): '
                         'rounds create cuts, add none, and stall after one round')
    out['C4b'] = cut_config(1, 0.1, 2048, 6,
                            'the C4 shape (256 x 128, seed 1) at density 0.1 -- the family of this generator on which the reference\'s '
                            'rules do add cuts at this size (a sweep over density 1 / 0.5 / 0.25 / 0.1, with and without upper '
                            'bounds, scripts/c4_tree.py: everywhere else the rounded GMICs are rejected below the root) -- '
                            'gomory_cuts=True, 2048 nodes per step: nodes carry cut rows (K1\'s <7,7,16,192> tile), gain and lose '
                            'them in the timed steps')
    out['C5_cuts_single_gpu'] = cut_config(0, 1.0, 256, 3,
                                           'C5\'s instance (1024 x 512, seed 0) with the reference\'s default gomory_cuts=True: cut rounds '
                                           'inside the engine on the HBM-streaming kernel (K1b with per-node cut rows), 256 nodes per step, '
                                           'no dive (one GPU of the 8 the config names)', n=1024, m=512)
    # ---- C5 on one GPU: 1024 x 512, the HBM-streaming kernel
    n5, m5, B5, steps5 = 1024, 512, 1024, 10   # (10 steps of ~68 ms: the pipeline's fill and drain are ~3 % of the region)
    A5, b5, c5, l5, u5, ints5 = random_dense_milp_arrays(n5, m5, seed=0)
    p5 = _ffi.Problem(ctx, A5, b5, c5)
    depth5 = max(1, args.dive)
    t5 = _ffi.Tree(p5, ints5, l5, u5, branch_rule='pseudo cost', max_batch=B5,
                   pool_capacity=(2 + 2 * depth5) * B5 * (steps5 + 3) + 16 * B5)
    t5.set_anchor_mode(True)
    t5.set_dive(depth5)
    st = t5.stats()
    while st['open_nodes'] < B5 or st['evaluated_nodes'] == 0:
        st = t5.solve(mip_gap=0.0, frontier_batch=min(B5, 256), max_steps=1)
    t5.reanchor(st['open_nodes'])   # as C3: every open node gets an anchor of its own (4 MB each here)
    b0 = t5.stats()
    ctx.sync()
    t0 = time.perf_counter()
    st = t5.solve(mip_gap=0.0, frontier_batch=B5, max_steps=steps5)
    ctx.sync()
    el = time.perf_counter() - t0
    d5 = {k: st[k] - b0[k] for k in ('lp_solved', 'pivots', 'kernel_ms', 'steps', 'dives', 'probes_solved')}
    ks = d5['kernel_ms'] * 1e-3
    model = algorithmic_bytes(m5, n5, d5['lp_solved'], d5['pivots'], d5['dives'])
    real = d5['pivots'] * 2 * 8 * m5 * n5   # what K1b streams: the condensed m x n tableau, read + written per pivot
    out['C5_single_gpu'] = {
        'workload': f'1024 vars x 512 rows, seed 0, as C3, {B5} nodes per step + in-place dive of depth {depth5}, re-anchored '
                    f'(one GPU of the 8 the config names)', 'kernel': _ffi.kernel_name(m5, n5),
        'lps_per_s': d5['lp_solved'] / el, 'kernel_lps_per_s': d5['lp_solved'] / ks, 'ms_per_step': el / max(1, d5['steps']) * 1e3,
        'mean_pivots_per_lp': d5['pivots'] / max(1, d5['lp_solved']),
        'roofline': {'bound': 'hbm', 'achieved': real / ks / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                     'frac': real / ks / 1e9 / HBM_PEAK_GBPS, 'model_hbm_GBps': model / ks / 1e9,
                     'note': 'achieved = bytes the kernel streams per pivot (condensed m x n tableau, read + write) over '
                             'the K1b launch time (HIP events); model_hbm = SURVEY 8d\'s bordered-tableau figure'}}
    t5.close()
    p5.close()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    # stdout carries the one JSON line and nothing else: libraries that chat on fd 1 (RCCL prints a
    # version banner there) are sent to stderr for the rest of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from simple_mip_solver_amd import _ffi
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    from simple_mip_solver_amd.parallel import env_ranks, global_gap, init_comm

    rank, local_rank, world = env_ranks()
    gpu_index = local_rank % max(1, _ffi.lib().mipx_device_count())  # (a launcher may narrow the visible devices)
    n, m, B = args.vars, args.cons, args.batch
    if _ffi.kernel_name(m, n) == 'lp_dual_simplex_big' and '--reanchor' not in sys.argv:
        args.reanchor = 0  # (HBM-streaming kernel: a per-node 4 MB anchor costs more than the pivots it saves at this depth)
    ctx = _ffi.Context(gpu_index)
    comm, rehearsal = None, False
    if world > 1 and os.environ.get('MIPX_BENCH_TRANSPORT') == 'gloo':
        # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (RCCL refuses two ranks
        # on one device): the same exchange protocol over the test transport.  Never a measurement.
        import torch.distributed as dist
        dist.init_process_group('gloo')
        from tests.support.gloo_comm import make_comm
        comm, rehearsal = make_comm(ctx), True
    elif world > 1:
        comm = init_comm(ctx, rank, world)
    elif os.environ.get('MIPX_BENCH_FORCE_COMM'):   # the RCCL path with one rank (rehearsal on a one-GPU box)
        comm = _ffi.Comm(ctx, 0, 1, unique_id=_ffi.comm_unique_id())
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=args.seed)
    prob = _ffi.Problem(ctx, A, b, c)
    # room for every child the timed steps can queue (a step's claim: two children per output level of each
    # node), but never more than 160 GB of HBM for the node pool (a record: l, u, the basis)
    pool_nodes = (2 + 2 * args.dive) * B * (args.steps + args.warmup + 8) + 4 * B * world
    pool_nodes = min(pool_nodes, int(160e9 // (2 * 8 * n + n + m)))
    tree = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', search_rule='best first',
                     strong_branch_iters=5, max_batch=B, pool_capacity=pool_nodes)
    if not args.no_anchor:
        tree.set_anchor_mode(True)  # warm starts refactor from the root's optimal tableau
    if args.dive:
        tree.set_dive(args.dive)

    # ---- untimed: replicated ramp-up, then sharding ------------------------------------------
    st = tree.stats()
    while st['open_nodes'] < B * world or st['evaluated_nodes'] == 0:
        st = tree.solve(mip_gap=0.0, frontier_batch=min(B, 1024), max_steps=1)
        assert st['status'] == 4, f'tree finished during ramp-up: {st}'
    ramp = dict(st)
    tree.keep_shard(rank, world)
    if args.reanchor and not args.no_anchor:
        tree.reanchor(tree.stats()['open_nodes'])
    if comm is not None:
        tree.set_comm(comm, args.exchange_every)

    def barrier():
        ctx.sync()
        if comm is not None:
            comm.barrier()

    if args.warmup > 0:
        tree.solve(mip_gap=0.0, frontier_batch=B, max_steps=args.warmup)

    # CPU baseline sample: the very node LPs the GPU is about to solve (rank 0 only; the others wait
    # at the barrier below)
    cpu, sample = None, None
    if rank == 0 and args.cpu_seconds > 0:
        try:
            cpu, sample = cpu_port_baseline(args, tree, A, b, c, l, u, ints, B)
        except Exception as e:   # noqa: BLE001 -- a reported baseline, never worth the measurement (or a hang of the other ranks)
            import traceback
            traceback.print_exc(file=sys.stderr)
            cpu, sample = {'error': f'cpu_port_baseline failed: {type(e).__name__}: {e}'}, None

    before = tree.stats()
    barrier()
    t0 = time.perf_counter()
    st = tree.solve(mip_gap=0.0, frontier_batch=B, max_steps=args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    after = tree.stats()

    d = {k: after[k] - before[k] for k in ('lp_solved', 'probes_solved', 'pivots', 'evaluated_nodes',
                                           'kernel_ms', 'steps', 'dives')}
    mine = np.array([elapsed, d['lp_solved'], d['probes_solved'], d['pivots'], after['open_nodes'],
                     after['evaluated_nodes'] - ramp['evaluated_nodes'], after['primal_bound'], after['dual_bound'],
                     d['steps']], dtype=np.float64)
    allr = comm.allgather(mine) if comm is not None else mine[None]
    gstats = tree.global_stats()
    if comm is not None:
        tree.set_comm(None)

    if rank == 0:
        # exactly K steps on every rank, or the line says it is not a measurement (it still goes out: a missing line
        # tells the reader less than a flagged one)
        steps_done = int(allr[:, 8].min())
        invalid = None
        if steps_done < args.steps:
            invalid = (f'a rank ran {steps_done} of the {args.steps} steps asked for (steps per rank: {allr[:, 8].astype(int).tolist()}): '
                       'the node pool is capped at 160 GB -- fewer --steps, or a smaller --dive / --batch')
            sys.stderr.write('bench.py: ' + invalid + '\n')
        elapsed_max = float(allr[:, 0].max())
        lps_total, probes_total = float(allr[:, 1].sum()), float(allr[:, 2].sum())
        gp = float(allr[:, 6].min()); gd = float(allr[:, 7].min())
        launch_s = d['kernel_ms'] * 1e-3 / max(1, d['steps'])
        model_gbps = algorithmic_bytes(m, n, d['lp_solved'], d['pivots'], d['dives']) / max(1, d['steps']) / launch_s / 1e9
        flops = 2.0 * m * n * d['pivots'] / max(1, d['steps']) / launch_s / 1e12
        traffic, pmc_src, issue, stale = None, None, None, None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic, pmc_src, issue = pj.get('hbm_bytes_per_launch'), pj.get('command'), pj.get('issue')
                # the counters were collected on the kernel sources with this hash: another hash now means the
                # figure describes an older kernel -- it is reported as stale and no fraction is derived from it
                stale = pj.get('csrc_sha256') != _ffi.source_hash()
            except Exception:
                traffic = None
        hbm_gbps = None if (traffic is None or stale) else traffic / launch_s / 1e9
        tree.close()   # (its node pool -- up to 160 GB -- makes room for the legs below)
        def leg(name, fn, *a):
            """A side leg of the line (never the timed region): if it fails, the line still goes out and says so."""
            try:
                return fn(*a)
            except Exception as e:   # noqa: BLE001 -- reported, not hidden
                import traceback
                traceback.print_exc(file=sys.stderr)
                return {'error': f'{name} failed: {type(e).__name__}: {e}'}
        nodive = leg('no_dive_leg', no_dive_leg, args, ctx, prob, l, u, ints, B) if (world == 1 and args.dive and args.no_dive_leg) else None
        tto = leg('time_to_optimal_leg', time_to_optimal_leg, args, ctx, A, b, c, l, u, ints) if args.tto_seconds > 0 and (n, m) == (256, 128) else None
        tto_small = leg('largest_closing_time_to_optimal', largest_closing_time_to_optimal, ctx, args.dive) if args.tto_seconds > 0 else None
        highs = leg('highs_baseline', highs_baseline, args, prob, A, b, c, sample) if (sample is not None and args.highs_seconds > 0) else None
        others = leg('other_configs', other_configs, args, ctx) if args.others else None
        out = {
            'metric': 'node LP-relaxations/s', 'value': lps_total / elapsed_max,
            'unit': 'node LP-relaxations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed_max / max(1, min(args.steps, steps_done)) * 1e3,
            **({'invalid': invalid} if invalid else {}),
            'value_no_dive': None if nodive is None else nodive.get('value'), 'no_dive': nodive,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic' + (' (REHEARSAL: ranks share GPUs, gloo transport -- not a measurement)' if rehearsal else ''),
            'config': {
                'workload': f'{ {(256, 128): "C3", (1024, 512): "C5"}.get((n, m), "custom") }: {n} vars x {m} rows random dense MILP (BASELINE.md sec. 4, seed '
                            f'{args.seed}), PseudoCostBranchNode, best-first, strong_branch_iters=5, '
                            f'gomory_cuts=False, native frontier engine, {B} open nodes per step per GPU'
                            + (', every open node re-anchored after sharding' if args.reanchor and not args.no_anchor else '')
                            + (f' + in-place dive (each node and, where the rule needs no probes, up to {args.dive} child(ren) in a row on the same register tableau): '
                               f'of the {d["lp_solved"] / max(1, d["steps"]):.0f} LPs of a step {d["dives"] / max(1, d["steps"]):.0f} are plunge children, '
                               f'{(d["lp_solved"] - d["dives"]) / max(1, d["steps"]):.0f} are nodes popped from the best-first queue (value_no_dive: the same region without the plunge)' if args.dive else ''),
                'frontier_batch_per_gpu': B, 'kernel': _ffi.kernel_name(m, n),
                'anchored_refactorisation': not args.no_anchor, 'reanchored_after_sharding': bool(args.reanchor and not args.no_anchor), 'dive': args.dive,
                'dive_children_per_step': d['dives'] / max(1, d['steps']),
                'mean_pivots_per_lp': d['pivots'] / max(1, d['lp_solved']),
                'sb_probes_per_s': probes_total / elapsed_max,
                'nodes_evaluated_total': ramp['evaluated_nodes'] + int(allr[:, 5].sum()),
                'nodes_evaluated_note': 'the replicated ramp-up counted once + every rank\'s nodes since sharding',
                'open_nodes_total': int(allr[:, 4].sum()),
                'ramp_up_nodes': ramp['evaluated_nodes'],
                'primal_bound': None if gp == float('inf') else gp, 'dual_bound': gd,
                'gap': global_gap(gp, gd),
                'time_to_optimal': tto, 'time_to_optimal_largest_closing_config': tto_small,
                'exchanges': gstats['exchanges'], 'nodes_sent_rank0': gstats['nodes_sent'],
                'nodes_received_rank0': gstats['nodes_received'],
                'parallelism': f'open nodes sharded x{world}, per-GPU best-first queue; one all-gather per rank every '
                               f'{args.exchange_every} steps inside libmipx.so (RCCL: ncclAllGather on its own stream, posted '
                               f'in the step loop and applied one interval later): incumbent value + solution, shard dual '
                               f'bounds, open counts, stop flags, counters, pseudo-cost samples; node records to a dry rank by '
                               f'ncclSend/ncclRecv; no scaling curve is claimed until the driver\'s SCALE record exists',
                'others': others},
            'roofline': {
                'bound': 'hbm',
                # what the kernel really moves: PMC bytes (profiles/, same command) over the live launch time
                'achieved': hbm_gbps, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                'frac': None if hbm_gbps is None else hbm_gbps / HBM_PEAK_GBPS,
                'traffic': traffic, 'traffic_stale': stale, 'traffic_source': pmc_src, 'launch_ms': launch_s * 1e3,
                'model_hbm': {'achieved': model_gbps, 'frac': model_gbps / HBM_PEAK_GBPS, 'unit': 'GB/s',
                              'note': 'SURVEY 8d\'s dense-tableau HBM model (algorithmic bytes / launch time): the '
                                      'tableau is register-resident, this traffic does not happen -- not a roofline '
                                      'fraction, kept for comparison with round 1'},
                'vector_f64': {'achieved': flops, 'peak': F64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': flops / F64_VECTOR_PEAK_TFLOPS,
                               'note': 'rank-1 tableau updates (2mn flop per pivot) against the v_fma_f64 rate'},
                'issue': issue,
                'note': 'the node-LP kernel (HIP events on its stream, rank 0).  It is bound by neither roof: one '
                        'workgroup per CU (the 256 KiB tableau fills half its register file), the control wave\'s '
                        'dependent selections between the rank-1 sweeps'},
            'cpu_baseline': cpu,
            'cpu_baseline_highs': highs,
        }
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    tree.close()
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == '__main__':
    main()
