"""The N > 1 path on CPU: two real processes over gloo (no GPU needed).

What runs here is the product's own code in libmipx.so -- the communicator's all-gather / barrier
(through the custom transport: tests/support/gloo_comm.py) and `mipx_exchange_decide`, the pure
function by which every rank turns one gathered set of records into the same incumbent, bounds,
termination decision and migration plan.  The parts that need a device (node LPs, moving pool rows)
are covered by tests/test_parallel_gpu.py."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from simple_mip_solver_amd import _ffi
from simple_mip_solver_amd.parallel import env_ranks, global_gap, share_unique_id

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INF = float('inf')


def record(n, primal=INF, dual=-INF, open_nodes=0, stop=0, counts=(0, 0, 0, 0), x=None, batch=8, samples=None,
           room=1 << 30):
    r = np.zeros(_ffi.exchange_record_len(n))
    r[0], r[1], r[2], r[3] = primal, dual, open_nodes, stop
    r[4:8] = counts
    r[8] = 0.0 if x is None else 1.0
    r[10] = batch
    r[11] = room
    r[12] = INF
    if x is not None:
        r[16:16 + n] = x
    if samples is not None:
        r[16 + n:] = samples
    return r


def test_single_process_helpers():
    assert global_gap(-2, -2.25) == .125 and global_gap(INF, -3) is None and global_gap(0, 0) == 0
    assert env_ranks() == (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
                           int(os.environ.get('WORLD_SIZE', 1)))
    assert share_unique_id(0, 1, lambda: b'x' * 128) == b'x' * 128
    assert _ffi.exchange_record_len(256) == 16 + 5 * 256


def test_decision_rules():
    n = 3
    # nobody has an incumbent; rank 1's shard has the weaker bound; both have work
    d = _ffi.exchange_decide(np.stack([record(n, dual=-10.0, open_nodes=50, counts=(3, 3, 0, 30)),
                                       record(n, dual=-12.5, open_nodes=40, counts=(4, 4, 2, 40))]), n)
    assert (d['primal'], d['dual'], d['gap'], d['incumbent_rank'], d['done']) == (INF, -12.5, None, -1, False)
    assert d['sums'] == [7, 7, 2, 70] and d['open_nodes'] == 90 and d['moves'] == []
    # rank 1 finds an incumbent; the lowest rank HOLDING A SOLUTION for the best value is the source
    d = _ffi.exchange_decide(np.stack([record(n, primal=-9.0, dual=-9.5, open_nodes=9),          # value only
                                       record(n, primal=-9.0, dual=-9.5, open_nodes=9, x=[1, 2, 3]),
                                       record(n, primal=-8.0, dual=-9.7, open_nodes=9, x=[0, 0, 1])]), n, mip_gap=1e-4)
    assert d['primal'] == -9.0 and d['incumbent_rank'] == 1 and d['dual'] == -9.7 and not d['done']
    assert abs(d['gap'] - 0.7 / 9.0) < 1e-15
    # termination: the gap of the GLOBAL bounds, a rank's stop flag, nobody has an open node
    recs = np.stack([record(n, primal=-9.0, dual=-9.0005, open_nodes=5, x=[1, 2, 3]), record(n, dual=-9.0008, open_nodes=5)])
    assert _ffi.exchange_decide(recs, n, mip_gap=1e-4)['reason'] == 3
    assert _ffi.exchange_decide(recs, n, mip_gap=1e-5)['reason'] == 0
    recs[1, 3] = 2.0                                                       # rank 1 has done its steps: rank 0 goes on
    assert _ffi.exchange_decide(recs, n, mip_gap=1e-5)['reason'] == 0
    recs[0, 3] = 2.0                                                       # both have
    assert _ffi.exchange_decide(recs, n, mip_gap=1e-5)['reason'] == 4
    recs[0, 3] = 0.0
    recs[0, 2] = 0                                                         # rank 0 ran dry instead: nobody can feed it
    assert _ffi.exchange_decide(recs, n, mip_gap=1e-5)['reason'] == 4
    recs[0, 2] = 5
    recs[1, 3] = 1.0                                                       # a limit that ends the search
    assert _ffi.exchange_decide(recs, n, mip_gap=1e-5)['reason'] == 2
    recs[:, 2] = 0
    assert _ffi.exchange_decide(recs, n, mip_gap=1e-5)['reason'] == 1     # idle outranks the flag
    # migration: a rank that cannot fill its batch gets half the surplus of the fullest rank
    recs = np.stack([record(n, open_nodes=1000, batch=64), record(n, open_nodes=3, batch=64),
                     record(n, open_nodes=0, batch=64), record(n, open_nodes=200, batch=64)])
    d = _ffi.exchange_decide(recs, n)
    assert d['moves'] == [(0, 1, 498), (0, 2, 251)] and not d['done']
    assert _ffi.exchange_decide(recs, n, allow_migration=False)['moves'] == []
    recs[0, 2] = 100                                                       # nobody has two batches to spare ...
    recs[3, 2] = 100
    assert _ffi.exchange_decide(recs, n)['moves'] == []
    recs[0, 2] = 40000                                                     # a donation is capped
    assert _ffi.exchange_decide(recs, n)['moves'][0] == (0, 1, 4096)
    # ... and never exceeds the room its receiver reported (the receiver cannot refuse after the send);
    # a rank without room receives nothing
    recs[1, 11], recs[2, 11] = 100, 0
    assert _ffi.exchange_decide(recs, n)['moves'] == [(0, 1, 100)]
    # a rank that is returning an error stops everybody (reason 5 outranks everything, no moves)
    recs[3, 3] = 3.0
    d = _ffi.exchange_decide(recs, n)
    assert d['done'] and d['reason'] == 5 and d['moves'] == []
    recs[:, 2] = 0
    assert _ffi.exchange_decide(recs, n)['reason'] == 5


def test_tail_of_the_search_is_not_a_closed_gap():
    """A rank whose queue is empty while its last nodes are in flight reports THEIR bound ([12], folded
    into [1] by the engine), not the value of its closed leaves: the gap of such records stays open."""
    n = 3
    # rank 0 holds the incumbent -9, queue empty, 64 nodes in flight with bounds down to -9.4
    r0 = record(n, primal=-9.0, dual=-9.4, open_nodes=64, x=[1, 2, 3])
    r0[12] = -9.4
    r1 = record(n, primal=-9.0, dual=-9.0, open_nodes=0)
    d = _ffi.exchange_decide(np.stack([r0, r1]), n, mip_gap=1e-4)
    assert not d['done'] and d['dual'] == -9.4


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch.distributed as dist
    dist.init_process_group('gloo')
    from simple_mip_solver_amd import _ffi
    from tests.support.gloo_comm import make_comm
    from tests.test_parallel_cpu import record
    INF = float('inf')
    rank = dist.get_rank()
    comm = make_comm(None)                       # host-only communicator inside libmipx.so, gloo underneath
    assert (comm.rank, comm.world, comm.transport) == (rank, 2, 'custom')
    got = comm.allgather(np.arange(5, dtype=np.float64) + 10 * rank)
    assert got.shape == (2, 5) and np.array_equal(got[1], np.arange(5) + 10.0)
    comm.barrier()
    n = 4
    # round 1: rank 1 holds an incumbent with its solution, rank 0 is almost out of nodes
    mine = [record(n, dual=-10.0, open_nodes=2, counts=(5, 5, 0, 40), batch=16),
            record(n, primal=-7.0, dual=-12.0, open_nodes=400, counts=(9, 9, 4, 80), x=[1, 0, 2, 0], batch=16)][rank]
    recs = comm.allgather(mine)
    d = _ffi.exchange_decide(recs, n)
    assert d['primal'] == -7.0 and d['incumbent_rank'] == 1 and d['dual'] == -12.0 and not d['done']
    assert np.array_equal(recs[d['incumbent_rank'], 16:16 + n], [1, 0, 2, 0])      # the solution travels with the value
    assert d['moves'] == [(1, 0, 199)] and d['sums'] == [14, 14, 4, 120]
    # both ranks reached the same conclusion from the same bytes
    both = comm.allgather(np.array([d['primal'], d['dual'], d['incumbent_rank'], len(d['moves']), d['moves'][0][2]], float))
    assert np.array_equal(both[0], both[1])
    # round 2: everything closed up
    mine = record(n, primal=-7.0, dual=[-7.0, -7.0005][rank], open_nodes=[0, 3][rank], x=[1, 0, 2, 0])
    d = _ffi.exchange_decide(comm.allgather(mine), n, mip_gap=1e-4)
    assert d['done'] and d['reason'] == 3
    # a failing transport surfaces as the Python exception, on the rank it happened on
    def boom(data):
        raise RuntimeError('link down')
    bad = _ffi.Comm(None, rank, 2, allgather=boom, send=lambda p, d: None, recv=lambda p, k: b'')
    try:
        bad.allgather(np.zeros(2))
        raise SystemExit('expected a failure')
    except RuntimeError as e:
        assert 'link down' in str(e)
    comm.barrier()
    dist.destroy_process_group()
    sys.stdout.write('rank%dok\\n' % rank)
    sys.stdout.flush()
''')


RENDEZVOUS_CLASH = ('address already in use', 'eaddrinuse', 'errno: 98')


def run_two_ranks(script, timeout=300, extra_env=None):
    """Two ranks under torch.distributed.run.  The free port is probed, not reserved, so a run is
    repeated ONLY when its stderr shows a rendezvous clash on the port; any other failure -- a rank that
    crashed, an assertion, a GPU fault -- is returned at once (a retry must not be able to hide it)."""
    res = None
    for attempt in range(3):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), **(extra_env or {}))
        res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                              '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port',
                              str(port), str(script)], env=env, capture_output=True, text=True,
                             timeout=timeout)
        if res.returncode == 0 or not any(k in res.stderr.lower() for k in RENDEZVOUS_CLASH):
            break
        sys.stderr.write(f'run_two_ranks: rendezvous clash on port {port} (attempt {attempt + 1}), stderr was:\n{res.stderr}\n')
    return res


def test_two_ranks_over_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    res = run_two_ranks(script)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'rank0ok' in res.stdout and 'rank1ok' in res.stdout


def test_unique_id_travels_over_tcp(tmp_path):
    """share_unique_id: rank 0's 128 bytes reach the other rank (what init_comm does with RCCL's id)."""
    script = tmp_path / 'idw.py'
    script.write_text(textwrap.dedent('''
        import os, sys
        sys.path.insert(0, ROOT_DIR)
        from simple_mip_solver_amd.parallel import share_unique_id, env_ranks
        rank, _, world = env_ranks()
        uid = share_unique_id(rank, world, lambda: bytes(range(128)))
        assert uid == bytes(range(128)), uid
        sys.stdout.write('id%dok\\n' % rank); sys.stdout.flush()
    ''').replace('ROOT_DIR', repr(ROOT)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2',
                                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)))
             for r in (1, 0)]       # (the receiver first: it has to retry until rank 0 listens)
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'id1ok' in outs[0][0] and 'id0ok' in outs[1][0]


def test_package_is_torch_free():
    """north_star: 'no PyTorch' -- the product never imports it (tests and bench.py's launcher may)."""
    pkg = os.path.join(ROOT, 'simple_mip_solver_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                assert 'torch' not in open(os.path.join(d, f)).read(), os.path.join(d, f)
