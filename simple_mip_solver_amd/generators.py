"""Synthetic random dense MILPs.

Restates the distribution of the reference's wrapper around grumpy's GenerateRandomMIP
(test_simple_mip_solver/example_models.py:12-25): integer c_j ~ U{1..maxObjCoeff},
A_ij ~ U{1..maxConsCoeff} with probability `density` (else 0),
b_i ~ U{floor(n*density*maxConsCoeff/tightness) .. floor(n*density*maxConsCoeff/1.5)};
problem  max c'x, Ax <= b, 0 <= x <= maxObjCoeff, x integer, handed to the solver as
min -c'x, -Ax >= -b (example_models.py:24-25).  RNG: numpy Generator(PCG64(seed)).
"""
import numpy as np


def random_dense_milp_arrays(num_vars, num_cons, density=1.0, max_obj_coeff=10,
                             max_cons_coeff=10, tightness=2, seed=0):
    """Return (A, b, c, l, u, integer_indices) already in `min c'x, Ax >= b` form."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n, m = int(num_vars), int(num_cons)
    c = rng.integers(1, max_obj_coeff + 1, n).astype(np.float64)
    A = rng.integers(1, max_cons_coeff + 1, (m, n)).astype(np.float64)
    if density < 1.0:
        A *= rng.random((m, n)) < density
    lo = int(n * density * max_cons_coeff / tightness)
    hi = int(n * density * max_cons_coeff / 1.5)
    b = rng.integers(lo, hi + 1, m).astype(np.float64)
    l = np.zeros(n)
    u = np.full(n, float(max_obj_coeff))
    return -A, -b, -c, l, u, list(range(n))
