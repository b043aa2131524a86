#!/usr/bin/env python3
"""bench.py -- node LP-relaxations/s on the C3 workload of BASELINE.md (256 vars x 128 rows).

A "step" is one pass of the node hot path over one frontier batch: B open nodes of a real
best-first branch-and-bound tree on the synthetic 256x128 random dense MILP (their bounds and
warm-start bases already resident in HBM), each solved to termination by the batched dual simplex
kernel through the C ABI (mipx_lp_solve_batch_dev).  value = node LP relaxations per second over
all ranks.  One process per GPU; for N > 1 every rank owns its own shard (its own tree) and the
only collectives are the timing barrier/MAX (weak scaling).

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit/... plus
  roofline     -- algorithmic bytes (SURVEY.md section 8d dense-tableau model) / kernel time
                  measured with HIP events on the library's stream, against the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (kind "port") timed on a bounded sample of the same node LPs
"""
import argparse
import heapq
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from simple_mip_solver_amd import _ffi  # noqa: E402
from simple_mip_solver_amd.generators import random_dense_milp_arrays  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def bytes_per_lp(m, n, pivots):
    """dense-tableau model, f64."""
    per_pivot = 2 * 8 * (m + 1) * (n + m + 1)
    return 8 * (m * n + m + 3 * n) + pivots * per_pivot + 8 * (n + 2) + (n + m)


def ramp_up_frontier(p, l, u, ints, want, chunk=256):
    """Best-first B&B (most-fractional branching, base_node.py:544-562) on the GPU until at least
    `want` open nodes exist; returns their (l, u, warm-start basis) arrays, best bound first."""
    n = len(l)
    ints = np.asarray(ints)
    root = p.solve_batch(l[None], u[None])
    assert root['status'][0] == 0, 'root LP must be feasible'
    heap = []
    cnt = 0

    def push_children(lk, uk, res_x, res_obj, res_v):
        nonlocal cnt
        x = res_x[ints]
        frac = np.minimum(x - np.floor(x), np.ceil(x) - x)
        k = int(np.argmax(frac))  # first max == lowest index, as the reference's strict '>'
        if frac[k] <= 1e-4:
            return
        j = int(ints[k])
        l2, u2 = lk.copy(), uk.copy()
        u2[j] = np.floor(res_x[j])
        heapq.heappush(heap, (res_obj, cnt, lk, u2, res_v)); cnt += 1
        l2[j] = np.ceil(res_x[j])
        heapq.heappush(heap, (res_obj, cnt, l2, uk.copy(), res_v)); cnt += 1

    push_children(l, u, root['x'][0], float(root['obj'][0]), root['vstat'][0])
    solved = 1
    while len(heap) < want:
        take = [heapq.heappop(heap) for _ in range(min(chunk, len(heap)))]
        L = np.stack([t[2] for t in take]); U = np.stack([t[3] for t in take])
        V = np.stack([t[4] for t in take])
        g = p.solve_batch(L, U, V)
        solved += len(take)
        for k in range(len(take)):
            if g['status'][k] == 0:
                push_children(L[k], U[k], g['x'][k], float(g['obj'][k]), g['vstat'][k])
        assert heap, 'tree exhausted during ramp-up'
    nodes = heapq.nsmallest(want, heap)
    L = np.stack([t[2] for t in nodes]); U = np.stack([t[3] for t in nodes])
    V = np.stack([t[4] for t in nodes])
    return L, U, V, solved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=8192, help='frontier nodes per step per GPU')
    ap.add_argument('--vars', type=int, default=256)
    ap.add_argument('--cons', type=int, default=128)
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_
        dist = dist_
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    n, m, B = args.vars, args.cons, args.batch
    ctx = _ffi.Context(local_rank)
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=rank)  # one tree per rank
    p = _ffi.Problem(ctx, A, b, c)
    L, U, V, ramp_solved = ramp_up_frontier(p, l, u, ints, B)

    d_l, d_u, d_v = ctx.to_device(L), ctx.to_device(U), ctx.to_device(V)
    d_st, d_obj = ctx.alloc(B * 4), ctx.alloc(B * 8)
    d_x, d_y = ctx.alloc(B * n * 8), ctx.alloc(B * m * 8)
    d_vo, d_it, d_np = ctx.alloc(B * (n + m)), ctx.alloc(B * 4), ctx.alloc(B * 4)

    def step():
        p.solve_batch_dev(B, d_l, d_u, d_v, 0, d_st, d_obj, d_x, d_y, d_vo, d_it, d_np)

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    kernel_ms = ctx.timer_stop()  # HIP events on the stream the kernel is launched on
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f'cuda:{local_rank}')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); npv = np.zeros(B, np.int32)
    ctx.d2h(st, d_st); ctx.d2h(it, d_it); ctx.d2h(npv, d_np)

    out = None
    if rank == 0:
        lps = world * B * args.steps
        launch_s = kernel_ms * 1e-3 / args.steps
        algo_bytes = float(sum(bytes_per_lp(m, n, int(k)) for k in npv))
        achieved = algo_bytes / launch_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        # CPU baseline: the oracle on a bounded sample of the same node LPs, one thread
        cpu = None
        if world == 1 or rank == 0:
            from oracle import oracle as O
            done, t_cpu, chunk = 0, 0.0, 256
            while t_cpu < args.cpu_seconds and done < 4 * B:
                s0 = done % B
                e0 = min(s0 + chunk, B)
                tc = time.perf_counter()
                o = O.lp_solve_batch(A, b, c, L[s0:e0], U[s0:e0], V[s0:e0])
                t_cpu += time.perf_counter() - tc
                assert np.array_equal(o['status'], st[s0:e0]), 'GPU/oracle status mismatch'
                assert np.array_equal(o['npivots'], npv[s0:e0]), 'GPU/oracle pivot-count mismatch'
                done += e0 - s0
            cpu = {'value': done / t_cpu, 'unit': 'node LP-relaxations/s', 'cores': 1,
                   'kind': 'port',
                   'sample': f'{done} of the same warm-started {n}x{m} frontier node LPs, '
                             f'oracle/libmipx_oracle.so single thread, {t_cpu:.1f} s'}
        out = {
            'metric': 'node LP-relaxations/s', 'value': lps / elapsed,
            'unit': 'node LP-relaxations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': f'C3: {n} vars x {m} rows random dense MILP (BASELINE.md sec. 4, seed=rank), '
                            f'frontier batch of {B} warm-started open nodes per GPU from a best-first '
                            f'most-fractional ramp-up ({ramp_solved} nodes), each LP solved to termination',
                'frontier_batch_per_gpu': B, 'kernel': _ffi.kernel_name(m, n),
                'mean_dual_iters_per_lp': float(it.mean()), 'mean_pivots_per_lp': float(npv.mean()),
                'status_counts': np.bincount(st, minlength=4).tolist(),
                'parallelism': f'node-sharded x{world} (one tree per GPU)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'launch_ms': launch_s * 1e3,
                         'note': 'algorithmic bytes of the dense-tableau model (SURVEY 8d); the '
                                 'tableau is register-resident so HBM traffic is far below them'},
            'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
