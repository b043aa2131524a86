"""The RCCL path of the exchange on a real GPU: one rank (the pool's boxes have one GPU), the
pinned-buffer / side-stream code of PipelinedExchange and the step hook with collectives inside."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=dev)
    from simple_mip_solver_amd import _ffi
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    from simple_mip_solver_amd.parallel import PipelinedExchange, exchange
    INF = float('inf')
    assert exchange(dist, dev, INF, -12.5, [3, 100]) == (INF, -12.5, [3, 100], 1)   # nobody holds an incumbent
    assert exchange(dist, dev, -9.0, -9.5, [1, 1]) == (-9.0, -9.5, [1, 1], 0)
    # the engine with the pipelined exchange inside its step loop (what bench.py does for N > 1)
    n, m, B = 64, 32, 256
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
    ctx = _ffi.Context(0)
    prob = _ffi.Problem(ctx, A, b, c)
    def run(with_hook):
        t = _ffi.Tree(prob, ints, l, u, branch_rule='pseudo cost', max_batch=B, pool_capacity=1 << 16)
        t.set_anchor_mode(True); t.set_dive(True)
        st = t.stats()
        while st['open_nodes'] < B:
            st = t.solve(mip_gap=0.0, frontier_batch=64, max_steps=1)
        t.keep_shard(0, 1)
        pe = PipelinedExchange(dist, dev, n, n_counters=1)
        pe.start(*t.pseudo_cost_arrays())
        calls = []
        def hook():
            s_ = t.stats()
            got = pe.step(s_['primal_bound'], s_['dual_bound'], [s_['evaluated_nodes']], *t.pseudo_cost_arrays())
            calls.append(got)
            if got is not None:
                assert got[0] >= s_['primal_bound'] and got[2][0] <= s_['evaluated_nodes']   # one rank: its own, one interval old
                t.set_pseudo_cost_arrays(*got[3])
        if with_hook:
            t.set_step_hook(hook, 2)
        st = t.solve(mip_gap=0.0, frontier_batch=B, max_steps=8)
        t.set_step_hook(None)
        if with_hook:
            assert len(calls) == 4 and calls[0] is None and all(c is not None for c in calls[1:])
            s_ = t.stats()
            got = pe.drain(s_['primal_bound'], s_['dual_bound'], [s_['evaluated_nodes']], *t.pseudo_cost_arrays())
            assert got[2] == [s_['evaluated_nodes']] and got[1] == s_['dual_bound']
            # with one rank the merged table is the rank's own table (up to the mean <-> sum round trip)
            own = t.pseudo_cost_arrays()
            assert np.allclose(got[3][0], own[0], rtol=1e-12) and np.array_equal(got[3][2], own[2])
        return st
    a, b_ = run(False), run(True)
    assert a['evaluated_nodes'] == b_['evaluated_nodes'] and abs(a['dual_bound'] - b_['dual_bound']) < 1e-6
    dist.barrier()
    dist.destroy_process_group()
    print('rccl_ok')
''')


def test_pipelined_exchange_over_rccl_one_rank(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0')
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'rccl_ok' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
