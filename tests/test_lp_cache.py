"""HipBackend's cache of resident row sets: eviction must never free a Problem somebody still uses.

Every cut round of the per-node path makes a new row set; a live native tree
(BranchAndBound._native) keeps the raw mipx_problem* of its own.  Eviction therefore only drops the
cache's reference (the device buffers go when the last Python reference does) and the least
recently USED row set leaves first."""
import gc

import numpy as np
import pytest

from simple_mip_solver_amd import _ffi, lp as lpmod


class FakeProblem:
    alive = 0
    closed = []

    def __init__(self, ctx, A, b, c):
        self.key = float(A[0, 0])
        FakeProblem.alive += 1

    def close(self):
        FakeProblem.closed.append(self.key)

    def __del__(self):
        FakeProblem.alive -= 1


@pytest.fixture
def backend(monkeypatch):
    FakeProblem.alive, FakeProblem.closed = 0, []
    monkeypatch.setattr(_ffi, 'Problem', FakeProblem)
    be = lpmod.HipBackend()
    be._ctx = object()   # never touched by the fake
    return be


def rowset(k):
    return np.full((1, 2), float(k)), np.zeros(1), np.zeros(2), k


def test_eviction_drops_the_reference_only(backend):
    root = backend._problem(*rowset(0))          # what a native Tree would pin
    for k in range(1, 70):
        backend._problem(*rowset(k))
    gc.collect()
    assert FakeProblem.closed == []              # nothing was destroyed explicitly
    assert len(backend._problems) == lpmod.HipBackend.MAX_RESIDENT_ROWSETS
    assert 0 not in backend._problems            # the root row set left the cache ...
    assert root.key == 0.0 and FakeProblem.alive == lpmod.HipBackend.MAX_RESIDENT_ROWSETS + 1  # ... and lives on
    del root
    gc.collect()
    assert FakeProblem.alive == lpmod.HipBackend.MAX_RESIDENT_ROWSETS


def test_least_recently_used_goes_first(backend):
    first = backend._problem(*rowset(0))
    for k in range(1, 70):
        backend._problem(*rowset(k))
        assert backend._problem(*rowset(0)) is first   # the root row set is hit by every cut-free node
    assert 0 in backend._problems and 1 not in backend._problems and 69 in backend._problems


@pytest.mark.gpu
def test_native_tree_survives_70_row_sets():
    """The advisor's scenario on the device: a native BranchAndBound, then 70 other row sets through
    the same backend (enough to evict its problem from the cache), then its solve continues."""
    from simple_mip_solver_amd import BranchAndBound, PseudoCostBranchNode, MILPInstance
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    lpmod.set_backend(None)
    A, b, c, l, u, ints = random_dense_milp_arrays(24, 10, seed=10)
    make = lambda: MILPInstance(A=A, b=b, c=c, l=l, u=u, sense=['Min', '>='], integerIndices=ints, numVars=24)
    ref = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=1)
    ref.solve()
    bb = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=1,
                        node_limit=5)
    bb.solve()
    assert bb.status == 'stopped on iterations or time'
    be = lpmod.get_backend()
    for k in range(70):
        A2, b2, c2, l2, u2, _ = random_dense_milp_arrays(6, 3, seed=100 + k)
        be.solve(A2, b2, c2, l2[None], u2[None], None, 0, ('evict', k))
    gc.collect()
    bb.node_limit = float('inf')
    bb.solve()
    assert bb.status == 'optimal' and bb.objective_value == ref.objective_value
    assert bb.evaluated_nodes == ref.evaluated_nodes
