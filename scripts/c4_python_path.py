"""Config C4 (Gomory cut rounds on) on the per-node Python path: every bound() is a few GPU calls
(K1 solve, K2 cut rows, K3 selection, K1 re-solve).  Reports node LP relaxations per second."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import BranchAndBound, PseudoCostBranchNode, MILPInstance
from simple_mip_solver_amd.generators import random_dense_milp_arrays
from simple_mip_solver_amd import lp as lpmod


class Counting(lpmod.HipBackend):
    lps = 0

    def solve(self, A, b, c, l, u, vstat, max_iter, cache_key):
        Counting.lps += len(l)
        return super().solve(A, b, c, l, u, vstat, max_iter, cache_key)


lpmod.set_backend(Counting())
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 128
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=0)
model = MILPInstance(A=A, b=b, c=c, l=l, u=u, sense=['Min', '>='], integerIndices=list(ints), numVars=n)
# (context creation, library load and first launches stay out of the timing)
BranchAndBound(model, PseudoCostBranchNode, pseudo_costs={}, strong_branch_iters=5, gomory_cuts=False, node_limit=5).solve()
for cuts in (False, True):
    bb = BranchAndBound(model, PseudoCostBranchNode, pseudo_costs={}, strong_branch_iters=5,
                        gomory_cuts=cuts, node_limit=limit)
    be = lpmod.get_backend()
    n0 = getattr(be, 'lps', 0)
    t0 = time.perf_counter()
    bb.solve()
    dt = time.perf_counter() - t0
    solved = getattr(be, 'lps', 0) - n0
    print('gomory_cuts=%s: %d nodes evaluated, %s LP solves in %.2f s -> %.0f nodes/s; status %s, dual bound %.4f' % (
        cuts, bb.evaluated_nodes, solved if solved else 'n/a', dt, bb.evaluated_nodes / dt, bb.status, bb.dual_bound))
