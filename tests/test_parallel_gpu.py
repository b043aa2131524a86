"""The sharded search on a real GPU.

(1) RCCL inside libmipx.so with one rank (the pool's boxes have one GPU): the communicator's
    stream / pinned buffers / ncclAllGather, and mipx_tree_solve as a collective call -- same search
    as without a communicator.
(2) Two PROCESSES sharing the one GPU -- a rehearsal, never a measurement: RCCL refuses two ranks on
    one device, so the very same protocol runs over the custom transport (gloo underneath,
    tests/support/gloo_comm.py).  Two sharded engines reach the proven optimum AND the solution of
    the single-rank run on a 40 x 16 instance; both ranks end with the same incumbent; a rank that
    starts without open nodes is fed by the other (node records move); every rank's stop is the
    joint decision.  The same through BranchAndBound(comm=...).
No scaling curve exists until the driver's SCALE record does."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from tests.test_parallel_cpu import ROOT, run_two_ranks

pytestmark = pytest.mark.gpu


def test_rccl_one_rank_collective_solve():
    from simple_mip_solver_amd import _ffi
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    ctx = _ffi.default_context()
    comm = _ffi.Comm(ctx, 0, 1, unique_id=_ffi.comm_unique_id())
    assert comm.transport == 'rccl'
    got = comm.allgather(np.arange(7, dtype=np.float64))
    assert got.shape == (1, 7) and np.array_equal(got[0], np.arange(7))
    comm.barrier()
    n, m, B = 64, 32, 256
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
    prob = _ffi.Problem(ctx, A, b, c)

    def run(with_comm):
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=1 << 16)
        t.set_anchor_mode(True); t.set_dive(True)
        st = t.stats()
        while st['open_nodes'] < B:
            st = t.solve(mip_gap=0.0, frontier_batch=64, max_steps=1)
        t.keep_shard(0, 1)
        if with_comm:
            t.set_comm(comm, 2)
        st = t.solve(mip_gap=0.0, frontier_batch=B, max_steps=8)
        return t, st
    t0, a = run(False)
    t1, b_ = run(True)
    assert a['evaluated_nodes'] == b_['evaluated_nodes'] and a['steps'] == b_['steps']
    assert abs(a['dual_bound'] - b_['dual_bound']) < 1e-6 and b_['status'] == 4
    g = t1.global_stats()
    # 8 steps, an exchange every 2: four in the loop (the first has nothing to collect), the one the
    # rank waits in when its step limit is reached, the closing one
    assert g['world'] == 1 and g['exchanges'] >= 4 and g['evaluated_nodes'] == b_['evaluated_nodes']
    assert g['nodes_sent'] == g['nodes_received'] == 0
    # one rank: the merged pseudo-cost table is its own (up to the mean <-> sum round trip)
    own0, own1 = t0.pseudo_cost_arrays(), t1.pseudo_cost_arrays()
    assert np.allclose(own0[0], own1[0], rtol=1e-12) and np.array_equal(own0[2], own1[2])
    t1.set_comm(None)
    # a tree run to the end: the decision to stop is the exchange's
    A, b, c, l, u, ints = random_dense_milp_arrays(40, 16, seed=3)
    prob = _ffi.Problem(ctx, A, b, c)
    out = []
    for with_comm in (False, True):
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=16, pool_capacity=1 << 15)
        t.set_anchor_mode(True); t.set_dive(True)
        if with_comm:
            t.solve(mip_gap=0.0, frontier_batch=16, max_steps=3)
            t.set_comm(comm, 3)
        st = t.solve(mip_gap=1e-4, frontier_batch=16)
        out.append((st, t.solution()))
        if with_comm:
            g = t.global_stats()
            assert g['incumbent_rank'] == 0 and g['open_nodes'] >= 0 and g['evaluated_nodes'] == st['evaluated_nodes']
            t.set_comm(None)
    (ref, xr), (st, x) = out
    assert st['status'] == ref['status'] == 1 and abs(st['primal_bound'] - ref['primal_bound']) < 1e-9
    assert abs(float(c @ x) - st['primal_bound']) < 1e-6 and abs(float(c @ xr) - ref['primal_bound']) < 1e-6


@pytest.mark.parametrize('transport', ['rccl', 'custom'])
def test_point_to_point_path_to_self(transport):
    """The wrappers a migration goes through -- pack_nodes, ncclSend + ncclRecv (one group), unpack_nodes --
    on RCCL with the one rank a one-GPU box allows: open nodes leave the queue, travel to the own rank and
    re-enter under new ids in fresh pool rows.  The records arrive intact (bounds, bases, inherited bounds)
    and the search ends exactly like the one that moved nothing."""
    from simple_mip_solver_amd import _ffi
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    ctx = _ffi.default_context()
    if transport == 'rccl':
        comm = _ffi.Comm(ctx, 0, 1, unique_id=_ffi.comm_unique_id())
    else:
        comm = _ffi.Comm(ctx, 0, 1, allgather=lambda b: [b], send=lambda p, d: None, recv=lambda p, k: b'')
    assert comm.transport == transport
    n, m, B = 40, 16, 16
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=3)
    prob = _ffi.Problem(ctx, A, b, c)

    def tree():
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=1 << 15)
        t.set_anchor_mode(True); t.set_dive(True)
        st = t.stats()
        while st['open_nodes'] < 6 * B:
            st = t.solve(mip_gap=0.0, frontier_batch=B, max_steps=1)
        return t
    ref = tree()
    end_ref = ref.solve(mip_gap=1e-4, frontier_batch=B)
    t = tree()
    with pytest.raises(_ffi.MipxError, match='no communicator'):
        t.migrate_self(4)
    t.keep_shard(0, 1)
    t.set_comm(comm, 3)
    N = t.stats()['open_nodes']
    L0, U0, V0, D0 = t.peek_open(N)
    key = lambda L, U, V, D: sorted((D[k], L[k].tobytes(), U[k].tobytes(), V[k].tobytes()) for k in range(len(D)))
    moved = t.migrate_self(20)
    assert moved == 20
    L1, U1, V1, D1 = t.peek_open(N)
    assert t.stats()['open_nodes'] == N and key(L0, U0, V0, D0) == key(L1, U1, V1, D1)   # the same records, bit for bit
    assert t.migrate_self(10 ** 6) == min(4096, N // 2)   # every second node of the whole queue
    L2, U2, V2, D2 = t.peek_open(N)
    assert key(L0, U0, V0, D0) == key(L2, U2, V2, D2)
    g = t.global_stats()
    assert g['nodes_sent'] == g['nodes_received'] == 20 + min(4096, N // 2)
    end = t.solve(mip_gap=1e-4, frontier_batch=B)
    assert end['status'] == end_ref['status'] == 1 and abs(end['primal_bound'] - end_ref['primal_bound']) < 1e-9
    assert abs(float(c @ t.solution()) - end['primal_bound']) < 1e-6
    t.set_comm(None)
    comm.close()


def test_record_dual_bound_covers_the_steps_in_flight():
    """The exchange record's dual bound ([1]) includes the nodes popped into the steps in flight ([12]):
    read from a step hook (steps ARE in flight there) it never decreases and never exceeds the optimum,
    also in the tail of the search where the queue is empty and the last nodes are being solved; and a
    one-rank collective solve with an exchange at every step ends like the plain solve (the decision
    'gap closed' is taken from those records)."""
    from simple_mip_solver_amd import _ffi
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    ctx = _ffi.default_context()
    comm = _ffi.Comm(ctx, 0, 1, allgather=lambda b: [b], send=lambda p, d: None, recv=lambda p, k: b'')
    covered = 0
    for seed in range(6):
        n, m, B = 40, 16, 8
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=seed)
        prob = _ffi.Problem(ctx, A, b, c)
        ref_t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=1 << 15)
        ref_t.set_anchor_mode(True); ref_t.set_dive(True)
        ref = ref_t.solve(mip_gap=1e-4, frontier_batch=B)
        assert ref['status'] == 1
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=1 << 15)
        t.set_anchor_mode(True); t.set_dive(True)
        t.solve(mip_gap=0.0, frontier_batch=B, max_steps=2)
        t.keep_shard(0, 1)
        t.set_comm(comm, 1)
        seen = []
        t.set_step_hook(lambda: seen.append(t.exchange_record()[[1, 2, 12]]) and False, 1)
        st = t.solve(mip_gap=1e-4, frontier_batch=B)
        t.set_step_hook(None)
        assert st['status'] == 1 and abs(st['primal_bound'] - ref['primal_bound']) < 1e-9, (seed, st, ref)
        seen = np.array(seen)
        assert len(seen) > 3
        # a proven bound never decreases (1e-9: a child's LP value may undercut its parent's by rounding noise)
        steps_down = np.diff(seen[:, 0])
        assert np.all(steps_down >= -1e-9), (seed, steps_down[steps_down < 0], np.where(steps_down < 0)[0], len(seen))
        assert np.all(seen[:, 0] <= ref['primal_bound'] + 1e-9), (seed, seen[:, 0])
        assert np.all(seen[:, 0] <= seen[:, 2])                               # [1] includes [12]
        covered += int(np.sum(np.isfinite(seen[:, 2]) & (seen[:, 0] == seen[:, 2])))
        # the merged pseudo-cost table of one rank is the table of the plain recurrence (a sample that counts
        # without a cost -- an infeasible child -- weighs in with the mean)
        own = t.pseudo_cost_arrays()
        assert np.all(np.isfinite(own[0])) and np.all(own[2] >= 0)
        t.set_comm(None)
    assert covered > 0   # the in-flight nodes did hold the shard's bound at some exchange


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group('gloo')
    os.environ['LOCAL_RANK'] = '0'                 # (one GPU on the box: both ranks use device 0)
    from simple_mip_solver_amd import _ffi, BranchAndBound, PseudoCostBranchNode, MILPInstance
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    from simple_mip_solver_amd.parallel import shard_and_attach
    from tests.support.gloo_comm import make_comm
    rank = dist.get_rank()
    ctx = _ffi.Context(0)                          # both ranks on the one GPU: a rehearsal
    comm = make_comm(ctx)
    n, m, B = 40, 16, 16
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=3)
    prob = _ffi.Problem(ctx, A, b, c)

    def tree():
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=1 << 15)
        t.set_anchor_mode(True); t.set_dive(True)
        return t
    one = tree()
    ref = one.solve(mip_gap=1e-4, frontier_batch=B)
    assert ref['status'] == 1
    xref = one.solution()

    # --- two sharded engines, the ordinary way ----------------------------------------------------
    t = tree()
    ramp = shard_and_attach(t, comm, B, exchange_every=3)
    assert ramp['status'] == 4 and t.stats()['open_nodes'] >= B // 2
    st = t.solve(mip_gap=1e-4, frontier_batch=B)
    g = t.global_stats()
    assert st['status'] == 1, st
    assert abs(st['primal_bound'] - ref['primal_bound']) < 1e-9
    x = t.solution()                                # every rank holds the incumbent SOLUTION
    assert np.max(np.abs(x[ints] - np.round(x[ints]))) <= 1e-4 and np.all(A @ x >= b - 1e-6)
    assert abs(float(c @ x) - st['primal_bound']) < 1e-6
    both = comm.allgather(np.concatenate([[st['primal_bound'], st['dual_bound'], g['evaluated_nodes'], g['exchanges'],
                                           g['incumbent_rank'], st['status']], x]))
    assert np.array_equal(both[0], both[1]), both   # the same answer, bit for bit, on both ranks
    assert g['world'] == 2 and g['evaluated_nodes'] >= ramp['evaluated_nodes']
    moved = comm.allgather(np.array([g['nodes_sent'], g['nodes_received']], float))
    assert moved[:, 0].sum() == moved[:, 1].sum()

    # --- a rank that starts dry is fed by the other -------------------------------------------------
    t = tree()
    stx = t.stats()
    while stx['open_nodes'] < 4 * B:
        stx = t.solve(mip_gap=0.0, frontier_batch=B, max_steps=1)
    if rank == 0:
        t.keep_shard(0, 1)                          # everything
    else:
        t.keep_shard(999983, 1000003)               # nothing
        assert t.stats()['open_nodes'] == 0
    t.set_comm(comm, 3)
    st = t.solve(mip_gap=1e-4, frontier_batch=B)
    g = t.global_stats()
    assert st['status'] == 1 and abs(st['primal_bound'] - ref['primal_bound']) < 1e-9
    moved = comm.allgather(np.array([g['nodes_sent'], g['nodes_received'], st['evaluated_nodes']], float))
    assert moved[0, 0] > 0 and moved[1, 1] == moved[0, 0] and moved[1, 0] == moved[0, 1] == 0
    assert moved[1, 2] > stx['evaluated_nodes']     # the fed rank did evaluate nodes of its own
    assert np.array_equal(t.solution(), comm.allgather(t.solution())[0])

    # --- limits: max_steps is a per-rank quota (each rank does its own), a node limit ends it for both
    t = tree()
    r2 = shard_and_attach(t, comm, B, exchange_every=2)
    st = t.solve(mip_gap=0.0, frontier_batch=B, max_steps=[4, 9][rank])
    assert st['status'] == 4 and st['steps'] - r2['steps'] == [4, 9][rank], st
    st = t.solve(mip_gap=0.0, frontier_batch=B, node_limit=[st['evaluated_nodes'] + 2 * B, 10 ** 9][rank])
    steps_after = comm.allgather(np.array([st['steps'] - r2['steps'], st['status']], float))
    assert np.all(steps_after[:, 1] == 4) and steps_after[1, 0] <= 9 + 12, steps_after
    t.set_comm(None)

    # --- a rank that fails tells the other: nobody is left waiting in an all-gather ---------------------
    t = tree()
    r3 = shard_and_attach(t, comm, B, exchange_every=2)
    if rank == 1:
        os.environ['MIPX_FAULT_STEP'] = str(t.stats()['steps'] + 3)
    try:
        t.solve(mip_gap=0.0, frontier_batch=B, max_steps=40)
        raise SystemExit('rank %d: expected an error' % rank)
    except _ffi.MipxError as e:
        assert ('injected fault' in str(e)) if rank == 1 else ('MIPX_EPEER' in str(e) or 'another rank failed' in str(e)), str(e)
    os.environ.pop('MIPX_FAULT_STEP', None)
    t.set_comm(None)
    comm.barrier()                                  # the all-gather sequence is still aligned

    # --- through the driver -----------------------------------------------------------------------
    make = lambda: MILPInstance(A=A, b=b, c=c, l=l, u=u, sense=['Min', '>='], integerIndices=ints, numVars=n)
    single = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={{}}, gomory_cuts=False, frontier_batch=B)
    single.solve()
    bb = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={{}}, gomory_cuts=False, frontier_batch=B,
                        comm=make_comm(_ffi.default_context()), exchange_every=3)
    bb.solve()
    assert bb.status == single.status == 'optimal' and abs(bb.objective_value - single.objective_value) < 1e-9
    assert bb.solution is not None and abs(float(c @ bb.solution) - bb.objective_value) < 1e-6
    ev = comm.allgather(np.array([bb.evaluated_nodes, bb.objective_value], float))
    assert np.array_equal(ev[0], ev[1]) and bb._native_global['world'] == 2
    # with cut rounds the ranks still agree (no migration in that mode)
    bc = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={{}}, frontier_batch=B,
                        comm=make_comm(_ffi.default_context()), exchange_every=3)
    bc.solve()
    assert bc.status == 'optimal' and bc.solution is not None
    agree = comm.allgather(np.concatenate([[bc.objective_value], bc.solution]))
    assert np.array_equal(agree[0], agree[1])
    comm.barrier()
    dist.destroy_process_group()
    sys.stdout.write('rank%dok\\n' % rank)
    sys.stdout.flush()
''')


def test_two_processes_share_the_gpu_rehearsal(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    res = run_two_ranks(script, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-6000:]
    assert 'rank0ok' in res.stdout and 'rank1ok' in res.stdout


def test_bench_with_four_ranks_rehearsal():
    """`bench.py --gpus 4` as the driver launches it, rehearsed on the one GPU of the box: four processes share the
    card, the exchange protocol runs over the test transport (RCCL refuses several ranks on one device).  Never a
    measurement -- the line says so -- but every rank goes through the replicated ramp-up, the sharding, the
    pipelined all-gathers of four records and the joint bookkeeping of the timed region, and rank 0's line comes
    out."""
    import json
    env = dict(os.environ, MIPX_BENCH_TRANSPORT='gloo')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '12', '--warmup', '2',
                          '--batch', '512', '--exchange-every', '3', '--cpu-seconds', '0', '--highs-seconds', '0',
                          '--tto-seconds', '0', '--others', '0', '--no-dive-leg', '0'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 4 and line['steps'] == 12 and 'REHEARSAL' in line['data']
    assert line['config']['exchanges'] >= 3 and line['value'] > 0
    assert line['config']['open_nodes_total'] > 4 * 512
