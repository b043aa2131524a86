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
    if dist is None or not dist.is_initialized():
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
        if dist is None or not dist.is_initialized():
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


class PipelinedExchange:
    """The same exchange, software-pipelined: posted at one hook call, applied at the next.

    A rank's GPU is saturated by node-LP launches, so a collective whose result the host waits
    for sits behind 2 ms kernels (measured: 2-5 ms per synchronous RCCL all-reduce from inside the
    step loop, against 20-50 us for the collective itself).  `step()` therefore (1) applies the
    all-reduce posted at the previous call -- long finished -- and (2) posts the next one without
    waiting: host -> device copy, all-reduce(MIN) of the bounds, all-reduce(SUM) of the pseudo-cost
    deltas and counters, device -> pinned host copy, all on one side stream.  Bounds and pseudo
    costs from other ranks arrive one exchange interval late, which branch and bound tolerates (an
    incumbent prunes from the moment it is known; a pseudo cost is a heuristic).

    Pseudo-cost bookkeeping in sum form (cost * times, times): `base` is the table all ranks agree
    on, `sent` what this rank's table was when it last posted.  On completion of a posted reduction
    R (the sum over ranks of each rank's table - base): base += R, and this rank's table becomes
    base + (table now - sent), i.e. everything agreed plus its own samples not yet shared.
    """

    def __init__(self, dist, device, n, n_counters=1):
        import torch
        self.dist, self.device, self.n, self.nc = dist, device, int(n), int(n_counters)
        self.active = dist is not None and dist.is_initialized()
        self.base = np.zeros(4 * n)       # [sum_l | sum_r | times_l | times_r]
        self.sent = np.zeros(4 * n)
        self.pending = None
        self.on_gpu = self.active and str(device) != 'cpu'
        if self.active:
            k = 4 * n + self.nc
            if self.on_gpu:
                self.stream = torch.cuda.Stream(device=device)
                self.event = torch.cuda.Event()
                pin = lambda size: torch.empty(size, dtype=torch.float64).pin_memory()
                self.h_min_in, self.h_min_out, self.h_sum_in, self.h_sum_out = pin(2), pin(2), pin(k), pin(k)
                self.d_min = torch.empty(2, dtype=torch.float64, device=device)
                self.d_sum = torch.empty(k, dtype=torch.float64, device=device)
            else:
                self.d_min = torch.empty(2, dtype=torch.float64)
                self.d_sum = torch.empty(k, dtype=torch.float64)

    @staticmethod
    def _sums(cost_l, cost_r, times_l, times_r):
        tl, tr = np.asarray(times_l, np.float64), np.asarray(times_r, np.float64)
        return np.concatenate([np.asarray(cost_l, np.float64) * tl, np.asarray(cost_r, np.float64) * tr, tl, tr])

    def _table(self, s):
        n = self.n
        mean = np.divide(s[:2 * n], s[2 * n:], out=np.zeros(2 * n), where=s[2 * n:] > 0)
        t32 = np.rint(s[2 * n:]).astype(np.int32)
        return mean[:n], mean[n:], t32[:n], t32[n:]

    def start(self, cost_l, cost_r, times_l, times_r):
        """The table every rank holds at sharding time (replicated ramp-up): counted once."""
        self.base = self._sums(cost_l, cost_r, times_l, times_r)
        self.sent = self.base.copy()

    def _post(self, primal_bound, dual_bound, counters, sums):
        import torch
        big = np.finfo(np.float64).max
        mins = np.array([min(primal_bound, big), min(dual_bound, big)])
        vec = np.concatenate([sums - self.base, np.asarray(counters, np.float64)])
        dist = self.dist
        if self.on_gpu:
            self.h_min_in.copy_(torch.from_numpy(mins))
            self.h_sum_in.copy_(torch.from_numpy(vec))
            with torch.cuda.stream(self.stream):
                self.d_min.copy_(self.h_min_in, non_blocking=True)
                self.d_sum.copy_(self.h_sum_in, non_blocking=True)
                w1 = dist.all_reduce(self.d_min, op=dist.ReduceOp.MIN, async_op=True)
                w2 = dist.all_reduce(self.d_sum, op=dist.ReduceOp.SUM, async_op=True)
                w1.wait(); w2.wait()      # RCCL: orders the side stream after the collectives, no host wait
                self.h_min_out.copy_(self.d_min, non_blocking=True)
                self.h_sum_out.copy_(self.d_sum, non_blocking=True)
                self.event.record(self.stream)
            self.pending = ()
        else:
            self.d_min.copy_(torch.from_numpy(mins))
            self.d_sum.copy_(torch.from_numpy(vec))
            self.pending = (dist.all_reduce(self.d_min, op=dist.ReduceOp.MIN, async_op=True),
                            dist.all_reduce(self.d_sum, op=dist.ReduceOp.SUM, async_op=True))
        self.sent = sums

    def _collect(self):
        if self.on_gpu:
            self.event.synchronize()
            mins, vec = self.h_min_out.numpy().copy(), self.h_sum_out.numpy().copy()
        else:
            for w in self.pending:
                w.wait()
            mins, vec = self.d_min.numpy().copy(), self.d_sum.numpy().copy()
        self.pending = None
        big = np.finfo(np.float64).max
        gp = INF if mins[0] >= big else float(mins[0])
        gd = INF if mins[1] >= big else float(mins[1])
        return gp, gd, vec[:4 * self.n], [int(round(v)) for v in vec[4 * self.n:]]

    def step(self, primal_bound, dual_bound, counters, cost_l, cost_r, times_l, times_r, post=True):
        """Apply what was posted last time, post this rank's state.  Returns None on the first call
        (nothing to apply yet), else (global primal bound, global dual bound, summed counters,
        merged table as (cost_l, cost_r, times_l, times_r)) -- all as of the previous call."""
        if not self.active:
            return None
        assert len(counters) == self.nc
        sums = self._sums(cost_l, cost_r, times_l, times_r)
        out = None
        if self.pending is not None:
            gp, gd, reduced, cnt = self._collect()
            self.base = self.base + reduced
            sums = self.base + (sums - self.sent)
            out = (gp, gd, cnt, self._table(sums))
        if post:
            self._post(primal_bound, dual_bound, counters, sums)
        return out

    def drain(self, primal_bound, dual_bound, counters, cost_l, cost_r, times_l, times_r):
        """End of the search: apply the posted exchange, then one more, waited for."""
        if not self.active:
            return None
        self.step(primal_bound, dual_bound, counters, cost_l, cost_r, times_l, times_r)
        t = self._table(self.sent)
        return self.step(primal_bound, dual_bound, counters, *t, post=False)
