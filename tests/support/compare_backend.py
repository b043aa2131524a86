"""HIP backend that cross-checks EVERY solve against the CPU oracle -- FOR GPU TESTS ONLY.

Returns the engine's result (the thing under test); the oracle is only the checker.  Bar: status,
iteration counts, basis, x, objective and duals bit-identical (== so that -0.0 == 0.0)."""
import numpy as np

from simple_mip_solver_amd.lp import HipBackend
from oracle import oracle as O


class CompareBackend(HipBackend):

    def __init__(self):
        super().__init__()
        self.lps = 0

    def solve(self, A, b, c, l, u, vstat, max_iter, cache_key):
        g = super().solve(A, b, c, l, u, vstat, max_iter, cache_key)
        o = O.lp_solve_batch(A, b, c, l, u, vstat, max_iter)
        self.lps += len(l)
        for key in ('status', 'iters', 'npivots', 'vstat'):
            assert np.array_equal(g[key], o[key]), \
                f'engine/oracle mismatch in {key} (m={A.shape[0]}, n={A.shape[1]}, key={cache_key}):\n' \
                f'engine {dict((k, v.tolist()) for k, v in g.items())}\n' \
                f'oracle {dict((k, v.tolist()) for k, v in o.items())}'
        fin = g['status'] != 1
        for key in ('x', 'obj', 'y'):
            assert np.array_equal(g[key][fin], o[key][fin]), f'engine/oracle mismatch in {key}'
        return g

    def gomory(self, A, b, c, l, u, vstat, x, integer_indices, max_term, cache_key):
        g = super().gomory(A, b, c, l, u, vstat, x, integer_indices, max_term, cache_key)
        o = O.gomory(A, b, c, l, u, vstat, x, integer_indices, max_term)
        assert np.array_equal(g['row_idx'], o['row_idx']), 'engine/oracle Gomory rows differ'
        for key in ('pi', 'pi0', 'safe_pi', 'safe_pi0'):
            assert np.array_equal(g[key], o[key]), f'engine/oracle mismatch in Gomory {key}'
        return g

    def select_cuts(self, pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef):
        g = super().select_cuts(pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef)
        o = O.select_cuts(pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef)
        assert np.array_equal(g[0], o[0]) and g[1] == o[1] and np.array_equal(g[2], o[2]), \
            'engine/oracle cut selection differs'
        return g
