"""LP backend that routes DenseLP solves to the CPU oracle -- FOR TESTS ONLY.

Lets the `-m "not gpu"` suite exercise the host logic (node classes, driver, kwargs protocol)
in a container without a GPU.  The product never installs this backend: the default is HipBackend
(simple_mip_solver_amd/lp.py), which fails loudly when libmipx.so or the device is missing.
"""
from simple_mip_solver_amd.lp import LPBackend
from oracle import oracle as O


class OracleBackend(LPBackend):

    def __init__(self):
        self.calls = 0
        self.lps = 0

    def solve(self, A, b, c, l, u, vstat, max_iter, cache_key):
        self.calls += 1
        self.lps += len(l)
        return O.lp_solve_batch(A, b, c, l, u, vstat, max_iter)

    def gomory(self, A, b, c, l, u, vstat, x, integer_indices, max_term, cache_key):
        return O.gomory(A, b, c, l, u, vstat, x, integer_indices, max_term)

    def select_cuts(self, pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef):
        return O.select_cuts(pi, pi0, x, max_nonzero_coefs, min_cut_depth, cos_parallel, max_abs_coef)
