"""The N > 1 exchange path on CPU: two processes, gloo backend (no GPU needed)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from simple_mip_solver_amd.parallel import exchange, global_gap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INF = float('inf')


def test_single_process_passthrough():
    assert exchange(None, 'cpu', -3.0, -5.0, [1, 2]) == (-3.0, -5.0, [1, 2], 0)
    assert global_gap(-2, -2.25) == .125 and global_gap(INF, -3) is None and global_gap(0, 0) == 0


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from simple_mip_solver_amd.parallel import exchange
    dist.init_process_group('gloo')
    rank = dist.get_rank()
    INF = float('inf')
    # step 1: nobody has an incumbent; rank 1's shard has the weaker dual bound
    out = exchange(dist, 'cpu', INF, [-10.0, -12.5][rank], [3 + rank, 100 * (rank + 1)])
    assert out == (INF, -12.5, [7, 300], 2), out
    # step 2: rank 1 finds an incumbent, rank 0 has run out of open nodes
    out = exchange(dist, 'cpu', [INF, -9.0][rank], [INF, -9.5][rank], [1, 1])
    assert out == (-9.0, -9.5, [2, 2], 1), out
    # step 3: both hold the same incumbent value: the lowest rank is reported
    out = exchange(dist, 'cpu', -9.0, -9.0, [0, 0])
    assert out == (-9.0, -9.0, [0, 0], 0), out
    # pseudo-cost tables: a shared start (replicated ramp-up), then each rank adds its own samples
    import numpy as np
    from simple_mip_solver_amd.parallel import PseudoCostExchange
    n = 4
    px = PseudoCostExchange(n)
    start = (np.array([2.0, 0, 0, 1.0]), np.array([4.0, 0, 0, 0]), np.array([1, 0, 0, 2], np.int32),
             np.array([1, 0, 0, 0], np.int32))
    px.start(*start)
    cl, cr, tl, tr = [a.copy() for a in start]
    if rank == 0:   # variable 0 left: one more sample of 4 -> mean 3 over 2; variable 1 left: new, 5
        cl[0], tl[0] = 3.0, 2
        cl[1], tl[1] = 5.0, 1
    else:           # variable 0 left: one more sample of 6 -> mean 4 over 2; variable 3 right: new, 7
        cl[0], tl[0] = 4.0, 2
        cr[3], tr[3] = 7.0, 1
    ml, mr, ntl, ntr = px.merge(dist, 'cpu', cl, cr, tl, tr)
    assert np.allclose(ml, [4.0, 5.0, 0, 1.0]) and list(ntl) == [3, 1, 0, 2], (ml, ntl)   # (2+4+6)/3
    assert np.allclose(mr, [4.0, 0, 0, 7.0]) and list(ntr) == [1, 0, 0, 1], (mr, ntr)
    # a second exchange with no new samples changes nothing
    ml2, mr2, ntl2, ntr2 = px.merge(dist, 'cpu', ml, mr, ntl, ntr)
    assert np.array_equal(ml2, ml) and np.array_equal(ntl2, ntl) and np.array_equal(mr2, mr)
    # the pipelined exchange: what is posted at one call is applied at the next
    from simple_mip_solver_amd.parallel import PipelinedExchange
    pe = PipelinedExchange(dist, 'cpu', n, n_counters=2)
    pe.start(*start)
    tab = [a.copy() for a in start]
    def add(tab, side, var, sample):   # running mean, like pseudo_cost.py:97-98
        c, t = tab[side], tab[2 + side]
        c[var] = (c[var] * t[var] + sample) / (t[var] + 1); t[var] += 1
    add(tab, 0, 0, [4.0, 6.0][rank])
    assert pe.step(INF, [-10.0, -12.5][rank], [1, 10], *tab) is None          # nothing to apply yet
    add(tab, 0, 1 if rank == 0 else 2, 5.0)                                      # not yet shared
    gp, gd, cnt, merged = pe.step([INF, -9.0][rank], -9.5, [2, 20], *tab)
    assert (gp, gd, cnt) == (INF, -12.5, [2, 20]), (gp, gd, cnt)                 # as of the first call
    # agreed: variable 0 left (2 + 4 + 6) / 3; own unshared sample kept, the other rank's not yet seen
    assert np.allclose(merged[0], [[4.0, 5.0, 0, 1.0], [4.0, 0, 5.0, 1.0]][rank]), merged[0]
    assert list(merged[2]) == [[3, 1, 0, 2], [3, 0, 1, 2]][rank]
    tab = [np.array(a, dtype=b.dtype) for a, b in zip(merged, start)]
    gp, gd, cnt, merged = pe.drain(-9.0, -9.0, [0, 0], *tab)
    assert (gp, gd, cnt) == (-9.0, -9.0, [0, 0])
    assert np.allclose(merged[0], [4.0, 5.0, 5.0, 1.0]) and list(merged[2]) == [3, 1, 1, 2], merged
    assert np.allclose(merged[1], [4.0, 0, 0, 0]) and list(merged[3]) == [1, 0, 0, 0]
    assert PipelinedExchange(None, 'cpu', n).step(0, 0, [0], *tab) is None        # single process
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write('rank%dok\\n' % rank)  # one write: the two ranks share the pipe
    sys.stdout.flush()
''')


def test_two_rank_exchange_over_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    res = None
    for attempt in range(3):  # the free port is probed, not reserved: retry on a rendezvous clash
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                              '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port',
                              str(port), str(script)], env=env, capture_output=True, text=True,
                             timeout=300)
        if res.returncode == 0:
            break
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'rank0ok' in res.stdout and 'rank1ok' in res.stdout
