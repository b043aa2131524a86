"""Multi-GPU sharding of one branch-and-bound tree.

One process per GPU.  Every rank runs the same deterministic ramp-up on its own GPU (replicated,
so no data has to move), keeps its share of the open nodes (`Tree.keep_shard`), then searches its
shard with its own best-first queue.  The only cross-rank traffic is a small fused exchange every
few steps: incumbent value (MIN), global dual bound (MIN), node/LP counters (SUM) -- latency-bound
messages of a few dozen bytes over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm; "gloo"
in the CPU tests).  The reference has no counterpart: it is single-process.
"""
import numpy as np

INF = float('inf')


def exchange(dist, device, primal_bound, dual_bound, counters):
    """All-reduce one step's worth of search state.

    primal_bound, dual_bound: this rank's values (dual bound of its shard; +inf if it has no open
    node and closed nothing).  counters: sequence of ints summed over ranks.
    Returns (global primal bound, global dual bound, summed counters, rank holding the incumbent).
    With dist None (single process) the inputs are returned unchanged.
    """
    counters = [int(c) for c in counters]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return primal_bound, dual_bound, counters, 0
    import torch
    # one MIN all-reduce carries both bounds (finite encoding: +inf -> 1.7e308)
    big = np.finfo(np.float64).max
    mins = torch.tensor([min(primal_bound, big), min(dual_bound, big)], dtype=torch.float64,
                        device=device)
    dist.all_reduce(mins, op=dist.ReduceOp.MIN)
    sums = torch.tensor(counters, dtype=torch.int64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    gp, gd = float(mins[0].item()), float(mins[1].item())
    gp = INF if gp >= big else gp
    gd = INF if gd >= big else gd
    # who holds the incumbent: lowest rank whose bound equals the global one
    rank = dist.get_rank()
    cand = torch.tensor([rank if (primal_bound == gp and gp < INF) else dist.get_world_size()],
                        dtype=torch.int64, device=device)
    dist.all_reduce(cand, op=dist.ReduceOp.MIN)
    return gp, gd, [int(v) for v in sums.tolist()], int(cand.item())


def global_gap(primal_bound, dual_bound):
    """current_gap of the reference (branch_and_bound.py:203-213) on the exchanged bounds."""
    if primal_bound == dual_bound == 0:
        return 0
    if primal_bound == 0:
        return INF
    if primal_bound == INF:
        return None
    return abs(primal_bound - dual_bound) / abs(primal_bound)


class PseudoCostExchange:
    """Merges the ranks' pseudo-cost tables (collective C3).

    An entry is a running mean of branch-cost samples (branch/pseudo_cost.py:97-98), i.e.
    sum / count: each rank all-reduces (SUM) what it added since the last exchange --
    delta(cost * times) as f64 and delta(times) as i64, 32 bytes per variable -- and rebuilds the
    means from the global sums.  `base` is the table every rank agreed on last time.
    """

    def __init__(self, n):
        self.base_sum = np.zeros((2, n))
        self.base_times = np.zeros((2, n), np.int64)

    def start(self, cost_l, cost_r, times_l, times_r):
        """The table all ranks share at sharding time (replicated ramp-up): counted once."""
        times = np.stack([times_l, times_r]).astype(np.int64)
        self.base_sum = np.stack([cost_l, cost_r]).astype(np.float64) * times
        self.base_times = times

    def merge(self, dist, device, cost_l, cost_r, times_l, times_r):
        """Local table in, merged table out (same four arrays); single process: unchanged."""
        if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
            return cost_l, cost_r, times_l, times_r
        import torch
        cost = np.stack([cost_l, cost_r]).astype(np.float64)
        times = np.stack([times_l, times_r]).astype(np.int64)
        d_sum = torch.tensor(cost * times - self.base_sum, dtype=torch.float64, device=device)
        d_times = torch.tensor(times - self.base_times, dtype=torch.int64, device=device)
        dist.all_reduce(d_sum, op=dist.ReduceOp.SUM)
        dist.all_reduce(d_times, op=dist.ReduceOp.SUM)
        self.base_sum = self.base_sum + d_sum.cpu().numpy()
        self.base_times = self.base_times + d_times.cpu().numpy()
        mean = np.divide(self.base_sum, self.base_times, out=np.zeros_like(self.base_sum),
                         where=self.base_times > 0)
        t32 = self.base_times.astype(np.int32)
        return mean[0], mean[1], t32[0], t32[1]
