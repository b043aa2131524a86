"""torch.distributed/gloo as the transport of a custom mipx communicator -- FOR TESTS ONLY.

The product binds RCCL inside libmipx.so and never imports torch; RCCL refuses two ranks on one
device and needs a GPU, so the CPU tests and the two-processes-on-one-GPU rehearsal run the very
same exchange protocol over these three host-buffer primitives instead."""
import numpy as np
import torch
import torch.distributed as dist

from simple_mip_solver_amd import _ffi


def make_comm(ctx):
    """An _ffi.Comm over the initialised gloo process group (ctx None: host-only, no GPU)."""
    rank, world = dist.get_rank(), dist.get_world_size()

    def allgather(data):
        mine = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        return [p.numpy().tobytes() for p in parts]

    def send(peer, data):
        dist.send(torch.frombuffer(bytearray(data), dtype=torch.uint8), dst=peer)

    def recv(peer, nbytes):
        buf = torch.empty(nbytes, dtype=torch.uint8)
        dist.recv(buf, src=peer)
        return buf.numpy().tobytes()

    return _ffi.Comm(ctx, rank, world, allgather=allgather, send=send, recv=recv)
