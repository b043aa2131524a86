"""Multi-GPU launch helpers for one sharded branch-and-bound tree.

One process per GPU.  Every rank runs the same deterministic ramp-up on its own GPU (replicated,
so no data has to move), keeps its share of the open nodes (`Tree.keep_shard`), attaches the
communicator (`Tree.set_comm`) and from then on `Tree.solve` is a collective call: the ranks
exchange incumbent value + solution, shard dual bounds, open-node counts, stop flags, counters and
pseudo-cost samples every few steps, decide termination together and move node records to a rank
that runs dry -- all inside libmipx.so over RCCL / xGMI (csrc/comm.hip.h, csrc/tree_engine.hip.h).
This module only finds out who the ranks are and hands rank 0's RCCL id to the others over a TCP
socket.  The reference has no counterpart (single process).
"""
import os
import socket
import time

INF = float('inf')
_ID_PORT_OFFSET = 23     # the id travels on MASTER_PORT + 23 (a launcher's own store sits on MASTER_PORT)


def env_ranks():
    """(rank, local_rank, world) as the launcher exports them (RANK, LOCAL_RANK, WORLD_SIZE:
    `bench.py --gpus N` or any elastic launcher); (0, 0, 1) without."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def share_unique_id(rank, world, make_id, addr=None, port=None, timeout=120.0):
    """Rank 0 calls make_id() and sends the bytes to every other rank; all return them."""
    addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(port if port is not None else int(os.environ.get('MASTER_PORT', '29500')) + _ID_PORT_OFFSET)
    if world == 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(uid)
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                chunks, got = [], 0
                while got < 128:
                    part = conn.recv(128 - got)
                    if not part:
                        break
                    chunks.append(part)
                    got += len(part)
                if got == 128:
                    return b''.join(chunks)
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError(f'rank {rank}: no RCCL id from rank 0 at {addr}:{port}')
        time.sleep(0.05)


def init_comm(ctx, rank=None, world=None):
    """The RCCL communicator of this process's rank (None for a single rank)."""
    from simple_mip_solver_amd import _ffi
    r, _, w = env_ranks()
    rank = r if rank is None else rank
    world = w if world is None else world
    if world <= 1:
        return None
    uid = share_unique_id(rank, world, _ffi.comm_unique_id)
    return _ffi.Comm(ctx, rank, world, unique_id=uid)


def shard_and_attach(tree, comm, frontier_batch, exchange_every=5, ramp_batch=None):
    """The replicated ramp-up, then this rank's shard and the communicator.  Returns the stats of the
    last ramp-up step (status != 4: the tree was finished before it could be sharded -- every rank
    holds the same result and nothing is attached)."""
    world = comm.world if comm is not None else 1
    st = tree.stats()
    while st['open_nodes'] < frontier_batch * world or st['evaluated_nodes'] == 0:
        st = tree.solve(mip_gap=0.0, frontier_batch=min(frontier_batch, ramp_batch or 1024), max_steps=1)
        if st['status'] != 4 or st['open_nodes'] == 0:
            return st
    if comm is not None:
        tree.keep_shard(comm.rank, world)
        tree.set_comm(comm, exchange_every)
    return st


def global_gap(primal_bound, dual_bound):
    """current_gap of the reference (branch_and_bound.py:203-213) on the exchanged bounds."""
    if primal_bound == dual_bound == 0:
        return 0
    if primal_bound == 0:
        return INF
    if primal_bound == INF:
        return None
    return abs(primal_bound - dual_bound) / abs(primal_bound)
