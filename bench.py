#!/usr/bin/env python3
"""bench.py -- node LP-relaxations/s on the C3 workload of BASELINE.md.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ONE 256 vars x
128 rows random dense MILP (seed 0), PseudoCostBranchNode, best-first, pseudo_costs={},
strong_branch_iters=5, gomory_cuts=False, solved by the native frontier engine
(BranchAndBound(..., frontier_batch=B) -> mipx_tree_*): node records resident in HBM, every step
= one frontier batch through the whole node hot path (LP relaxation to termination, strong-
branching probes where the pseudo-cost table has no entry yet, pseudo-cost update, branching index,
child materialisation).  Untimed: the ramp-up until every rank owns >= B open nodes, then W warm-up
steps.  Timed: exactly K steps, barrier + device sync on both sides, MAX over ranks.
value = node LP relaxations (solved to termination; probes are reported separately) per second,
summed over ranks.

N > 1: all ranks run the same deterministic ramp-up, each keeps its share of the open nodes
(Tree.keep_shard) and searches it with its own best-first queue; incumbent / global dual bound /
counters are exchanged by all-reduce over RCCL every few steps (simple_mip_solver_amd/parallel.py).
Per-GPU frontier batch is fixed -> "weak" scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from simple_mip_solver_amd import _ffi  # noqa: E402
from simple_mip_solver_amd.generators import random_dense_milp_arrays  # noqa: E402
from simple_mip_solver_amd.parallel import PipelinedExchange, exchange, global_gap  # noqa: E402

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8192, help='frontier nodes per step per GPU')
    ap.add_argument('--vars', type=int, default=256)
    ap.add_argument('--cons', type=int, default=128)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--cpu-threads', type=int, default=16, help='threads of the CPU baseline (over nodes)')
    ap.add_argument('--exchange-every', type=int, default=5, help='steps between all-reduces (N > 1)')
    ap.add_argument('--dive', type=int, default=1, choices=[0, 1],
                    help='1: one-level plunge on the register tableau (mipx_tree_set_dive)')
    ap.add_argument('--reanchor', type=int, default=1, choices=[0, 1],
                    help='1: after sharding every open node gets an anchor of its own (mipx_tree_reanchor)')
    ap.add_argument('--no-anchor', action='store_true',
                    help='refactor every node from the slack basis instead of the root tableau')
    args = ap.parse_args()
    # stdout carries the one JSON line and nothing else: libraries that chat on fd 1 (RCCL prints a
    # version banner there) are sent to stderr for the rest of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist, device = None, 'cpu'
    gpu_index = local_rank
    if world > 1 or os.environ.get('MIPX_BENCH_FORCE_DIST'):  # the env: RCCL path with one rank
        import torch
        import torch.distributed as dist_
        dist = dist_
        if os.environ.get('MIPX_BENCH_BACKEND') == 'gloo':
            # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks: the ranks
            # share the GPUs that exist and exchange over gloo (never a measurement)
            gpu_index = local_rank % max(1, _ffi.lib().mipx_device_count())
            dist.init_process_group('gloo')
        else:
            # a launcher may narrow the visible devices per rank: index among those this rank sees
            gpu_index = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(gpu_index)
            device = torch.device('cuda', gpu_index)
            dist.init_process_group('nccl', device_id=device)
            # first collective now: communicator set-up stays out of the timed region
            warm = torch.zeros(1, dtype=torch.float64, device=device)
            dist.all_reduce(warm)
            torch.cuda.synchronize()

    n, m, B = args.vars, args.cons, args.batch
    tto_dive = args.dive
    if _ffi.kernel_name(m, n) == 'lp_dual_simplex_big' and '--reanchor' not in sys.argv:
        args.reanchor = 0  # (HBM-streaming kernel: a per-node 4 MB anchor costs more than the pivots it saves at this depth)
    ctx = _ffi.Context(gpu_index)
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=args.seed)
    prob = _ffi.Problem(ctx, A, b, c)
    tree = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', search_rule='best first',
                     strong_branch_iters=5, max_batch=B,
                     pool_capacity=(2 + 2 * args.dive) * B * (args.steps + args.warmup + 4) + 4 * B * world)

    if not args.no_anchor:
        tree.set_anchor_mode(True)  # warm starts refactor from the root's optimal tableau
    if args.dive:
        tree.set_dive(True)

    # ---- untimed: replicated ramp-up, then sharding ------------------------------------------
    st = tree.stats()
    while st['open_nodes'] < B * world or st['evaluated_nodes'] == 0:
        st = tree.solve(mip_gap=0.0, frontier_batch=min(B, 1024), max_steps=1)
        assert st['status'] == 4, f'tree finished during ramp-up: {st}'
    ramp = dict(st)
    tree.keep_shard(rank, world)
    if args.reanchor and not args.no_anchor:
        tree.reanchor(tree.stats()['open_nodes'])
    pex = PipelinedExchange(dist, device, n, n_counters=1)
    pex.start(*tree.pseudo_cost_arrays())  # identical on every rank after the replicated ramp-up

    def run_steps(k):
        # inside one call the engine overlaps the host half of step i with the GPU half of i+1
        return tree.solve(mip_gap=0.0, frontier_batch=B, max_steps=k)

    def barrier():
        ctx.sync()
        if dist is not None:
            if device != 'cpu':
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    if args.warmup > 0:
        st = run_steps(args.warmup)

    # CPU baseline sample: the very node LPs the GPU is about to solve (rank 0 only)
    cpu = None
    if rank == 0 and args.cpu_seconds > 0:
        from oracle import oracle as O
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        peek = 4 * B * (3 if threads > 4 else 1)
        L, U, V, _ = tree.peek_open(peek)
        # the GPU path's anchors (one per re-anchored ancestor), read back for the CPU port
        atab = tree.anchor_table() if args.reanchor and not args.no_anchor else None
        asel = tree.peek_anchors(peek) if atab is not None else None
        chunk = 128
        nchunks = (len(L) + chunk - 1) // chunk
        deadline = time.perf_counter() + args.cpu_seconds
        done_chunks = []

        pc_tab = tree.pseudo_cost_arrays()
        pc_has = ((pc_tab[2] > 0) | (pc_tab[3] > 0)).astype(np.uint8)
        cutoff = tree.stats()['primal_bound']

        def work(ci):  # ctypes releases the GIL inside the C oracle: the threads run in parallel
            if time.perf_counter() > deadline:
                return 0
            e0 = min((ci + 1) * chunk, len(L))
            if args.dive:  # like the GPU path: node + one child continued on the node's tableau
                r = O.lp_solve_dive_batch(A, b, c, L[ci * chunk:e0], U[ci * chunk:e0], V[ci * chunk:e0], 1,
                                          ints, pc_tab[0], pc_tab[1], pc_has, cutoff, anchor_table=atab,
                                          anchor_sel=None if asel is None else asel[ci * chunk:e0])
                return e0 - ci * chunk + int((r['dive_var'] >= 0).sum())
            if atab is not None:  # rule -1: no dive, anchors from the table
                O.lp_solve_dive_batch(A, b, c, L[ci * chunk:e0], U[ci * chunk:e0], V[ci * chunk:e0], -1,
                                      ints, pc_tab[0], pc_tab[1], pc_has, cutoff, anchor_table=atab,
                                      anchor_sel=asel[ci * chunk:e0])
            else:
                O.lp_solve_batch(A, b, c, L[ci * chunk:e0], U[ci * chunk:e0], V[ci * chunk:e0])
            return e0 - ci * chunk

        # like the GPU path, warm starts refactor from the root's optimal tableau when anchoring is on
        import contextlib
        anchor_cm = contextlib.nullcontext()
        if not args.no_anchor:
            root = O.lp_solve(A, b, c, l, u)
            anchor_cm = O.anchored(O.make_anchor(A, b, c, root['vstat']))
        deadline = time.perf_counter() + args.cpu_seconds
        tc = time.perf_counter()
        done, passes = 0, 0
        with anchor_cm, ThreadPoolExecutor(threads) as ex:
            # whole passes over the peeked nodes until about cpu_seconds x 2 of CPU work is done
            while passes == 0 or ((time.perf_counter() - tc) * threads < 2 * args.cpu_seconds
                                  and time.perf_counter() < deadline):
                done += int(sum(ex.map(work, range(nchunks))))
                passes += 1
        t_cpu = time.perf_counter() - tc
        model = ''
        try:
            model = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
        except Exception:
            pass
        cpu = {'value': done / t_cpu, 'unit': 'node LP-relaxations/s', 'cores': threads, 'kind': 'port',
               'sample': f'{done} node LPs = {passes} pass(es) over {len(L)} open nodes of the same tree{" and their dive children" if args.dive else ""} (the LPs the GPU solves next: bounds + '
                         f'warm-start bases read back from the device pool), oracle/libmipx_oracle.so '
                         f'({("anchored like the GPU path: the tableau of the re-anchored ancestor, read back from the device" if atab is not None else "anchored at the root tableau like the GPU path") if not args.no_anchor else "slack-basis refactorisation"}), '
                         f'{threads} threads over nodes, {t_cpu:.1f} s wall; host: {os.cpu_count()} logical '
                         f'CPUs, {model}'}

    before = tree.stats()
    trace = [] if os.environ.get('MIPX_BENCH_TRACE') else None
    barrier()
    t0 = time.perf_counter()
    def do_exchange():
        # Runs inside the engine's step loop while the GPU works on the steps already queued.
        # Pipelined: applies the all-reduce posted at the previous call (incumbent / bound MIN,
        # pseudo-cost updates SUM) and posts the next one without waiting for it.
        te = time.perf_counter()
        s_ = tree.stats()
        got = pex.step(s_['primal_bound'], s_['dual_bound'], [s_['evaluated_nodes']],
                       *tree.pseudo_cost_arrays())
        if got is not None:
            if got[0] < s_['primal_bound']:
                tree.set_primal_bound(got[0])
            tree.set_pseudo_cost_arrays(*got[3])
        if trace is not None:
            trace.append((time.perf_counter() - te) * 1e3)

    if dist is not None:
        tree.set_step_hook(do_exchange, args.exchange_every)
    st = run_steps(args.steps)
    tree.set_step_hook(None)
    if dist is not None:  # apply the exchange still in flight and share what came after it
        s_ = tree.stats()
        got = pex.drain(s_['primal_bound'], s_['dual_bound'], [s_['evaluated_nodes']],
                        *tree.pseudo_cost_arrays())
        if got[0] < s_['primal_bound']:
            tree.set_primal_bound(got[0])
        tree.set_pseudo_cost_arrays(*got[3])
    tc_ = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    if trace is not None:
        sys.stderr.write('[bench rank %d] exchanges (ms): %s; closing barrier %.2f ms\n' % (
            rank, ', '.join('%.2f' % x for x in trace), (time.perf_counter() - tc_) * 1e3))
    after = tree.stats()

    # time-to-optimal leg of the metric: the 256x128 tree cannot be closed in a bench run, so the
    # same engine solves a small instance of the same family to proven optimality (rank 0, untimed
    # with respect to `value`)
    tto = None
    if rank == 0:
        A2, b2, c2, l2, u2, ints2 = random_dense_milp_arrays(80, 40, seed=0)
        p2 = _ffi.Problem(ctx, A2, b2, c2)
        best = None
        for _ in range(3):  # a 25 ms solve: the fastest of three (allocation and first-touch effects)
            t2 = _ffi.Tree(p2, ints2, l2, u2, branch_rule='pseudo cost', max_batch=4096, pool_capacity=1 << 21)
            if tto_dive:
                t2.set_dive(True)
            tt = time.perf_counter()
            s2 = t2.solve(mip_gap=1e-4, frontier_batch=4096, max_seconds=30.0)
            el2 = time.perf_counter() - tt
            t2.close()
            if best is None or el2 < best[0]:
                best = (el2, s2)
        el2, s2 = best
        tto = {'instance': '80 vars x 40 rows, seed 0, same generator, PseudoCostBranchNode best-first'
                           + (' + one-level dive' if tto_dive else ''),
               'seconds': el2,
               'status': _ffi.TREE_STATUS[s2['status']], 'objective': s2['primal_bound'],
               'nodes': s2['evaluated_nodes']}
        p2.close()

    d = {k: after[k] - before[k] for k in ('lp_solved', 'probes_solved', 'pivots', 'evaluated_nodes',
                                           'kernel_ms', 'steps', 'dives')}
    gp, gd, sums, _ = exchange(dist, device, after['primal_bound'], after['dual_bound'],
                               [d['lp_solved'], d['probes_solved'], d['pivots'], after['open_nodes'],
                                after['evaluated_nodes']])
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        assert d['steps'] == args.steps, 'frontier ran dry inside the timed region'
        launch_s = d['kernel_ms'] * 1e-3 / args.steps
        achieved = algorithmic_bytes(m, n, d['lp_solved'], d['pivots'], d['dives']) / args.steps / launch_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        gap = global_gap(gp, gd)
        out = {
            'metric': 'node LP-relaxations/s', 'value': sums[0] / elapsed,
            'unit': 'node LP-relaxations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': f'{ {(256, 128): "C3", (1024, 512): "C5"}.get((n, m), "custom") }: {n} vars x {m} rows random dense MILP (BASELINE.md sec. 4, seed '
                            f'{args.seed}), PseudoCostBranchNode, best-first, strong_branch_iters=5, '
                            f'gomory_cuts=False, native frontier engine, {B} open nodes per step per GPU'
                            + (', every open node re-anchored after sharding' if args.reanchor and not args.no_anchor else '')
                            + (' + one-level dive (each node and, where the rule needs no probes, one child on the same register tableau)' if args.dive else ''),
                'frontier_batch_per_gpu': B, 'kernel': _ffi.kernel_name(m, n),
                'anchored_refactorisation': not args.no_anchor, 'reanchored_after_sharding': bool(args.reanchor and not args.no_anchor), 'dive': bool(args.dive),
                'dive_children_per_step': d['dives'] / args.steps,
                'mean_pivots_per_lp': d['pivots'] / max(1, d['lp_solved']),
                'sb_probes_per_s': sums[1] / elapsed,
                'nodes_evaluated_total': sums[4], 'open_nodes_total': sums[3],
                'ramp_up_nodes': ramp['evaluated_nodes'],
                'primal_bound': None if gp == float('inf') else gp, 'dual_bound': gd,
                'gap': gap, 'time_to_optimal': tto,
                'parallelism': f'open nodes sharded x{world}, per-GPU best-first queue, '
                               f'allreduce(MIN) incumbent/bound + allreduce(SUM) pseudo-cost updates every '
                               f'{args.exchange_every} steps, posted inside the step loop and applied one interval later'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'launch_ms': launch_s * 1e3,
                         'vector_f64': {'achieved': 2.0 * m * n * d['pivots'] / args.steps / launch_s / 1e12,
                                        'peak': F64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                        'frac': 2.0 * m * n * d['pivots'] / args.steps / launch_s / 1e12
                                                / F64_VECTOR_PEAK_TFLOPS,
                                        'note': 'rank-1 tableau updates (2mn flop per pivot) against the '
                                                'v_fma_f64 rate: the kernel is bound by the dependent '
                                                'selection chains between the updates, not by either roof'},
                         'note': 'algorithmic bytes of the dense-tableau HBM model (SURVEY 8d) over '
                                 'the node-LP kernel time (HIP events on its stream), rank 0; the '
                                 'tableau is register-resident, so real HBM traffic is far below'},
            'cpu_baseline': cpu,
        }
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    tree.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
