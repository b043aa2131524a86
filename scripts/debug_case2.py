import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from simple_mip_solver_amd import lp as lpmod, BaseNode
from tests.support.example_models import std_model

class Spy(lpmod.HipBackend):
    def solve(self, A, b, c, l, u, vstat, max_iter, cache_key):
        g = super().solve(A, b, c, l, u, vstat, max_iter, cache_key)
        o = O.lp_solve_batch(A, b, c, l, u, vstat, max_iter)
        print('key', cache_key, 'm', A.shape[0], 'vstat', None if vstat is None else vstat.tolist(), 'max_iter', max_iter)
        print('   gpu', g['status'], g['iters'], g['npivots'], g['x'], g['vstat'])
        print('   ora', o['status'], o['iters'], o['npivots'], o['x'], o['vstat'])
        return g
lpmod.set_backend(Spy())
m = std_model('cut2')
node = BaseNode(m.lp, m.integerIndices, idx=0)
node._bound_lp(track_dual_bound=True)
node._base_bound(gomory_cuts=True, track_dual_bound=True)
