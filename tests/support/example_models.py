"""The reference's inline test instances restated as data (SURVEY.md Appendix C;
test_simple_mip_solver/example_models.py:89-311).  Each call builds a fresh instance."""
import numpy as np

from simple_mip_solver_amd import MILPInstance

_MODELS = {
    # name: (A, b, c, l, u, sense, integer indices) exactly as passed in the reference
    'no_branch': (-np.eye(3), [-1, -1, -1], [-1, -1, 0], [0, 0, 0], None, ['Min', '>='], [0, 1]),
    'small_branch': ([[-1, 0, -1], [0, -1, 0]], [-1.5, -1.25], [-1, -1, -1], [0, 0, 0],
                     [10, 10, 10], ['Min', '>='], [0, 1, 2]),
    'small_branch_max': ([[1, 0, 1], [0, 1, 0]], [1.5, 1.25], [1, 1, 1], [0, 0, 0], [10, 10, 10],
                         ['Max', '<='], [0, 1, 2]),
    'infeasible': ([[-1, -1, 0]], [1], [-1, -1, 0], [0, 0, 0], None, ['Min', '>='], [0, 1]),
    'infeasible2': ([[-1, -1, 0], [0, 0, -1]], [1, -1], [-1, -1, 0], [0, 0, 0], None,
                    ['Min', '>='], [0, 1]),
    'unbounded': ([[-1, 1], [1, -1]], [-.5, -.5], [-1, -1], [0, 0], None, ['Min', '>='], [0, 1]),
    'cut1': ([[8, -30], [14, -8], [-10, -10]], [-115, -1, -127], [0, -1], [0, 0], None,
             ['Min', '>='], [0, 1]),
    'cut2': ([[-4, -1], [-1, -4], [-1, 1]], [-28, -27, -1], [-2, -5], [0, 0], None,
             ['Min', '>='], [0, 1]),
    'cut3': ([[-3, -4], [-5, -10], [-1, -2]], [-10, -8, -1.2], [-8, -12], [0, 0], None,
             ['Min', '>='], [0, 1]),
    'square': (-np.eye(2), [-1.5, -1.5], [-1, -1], [0, 0], None, ['Min', '>='], [0, 1]),
    'negative': (-np.eye(2), [.5, -.5], [-1, -1], [-1, -1], None, ['Min', '>='], [0, 1]),
    'lift_project': ([[-1, 1], [1, 1]], [-1, 2], [1, 2], [0, 0], None, ['Min', '>='], [0, 1]),
    # ISE 418 HW 3 problem 1 and its right-hand-side family (example_models.py:195-278)
    'h3p1': ([[2, 5, -2, -2, 5, 5], [-2, -5, 2, 2, -5, -5]], [3.5, -3.5], [1, 4, 6, 4, 5, 7], [0] * 6, None,
             ['Min', '>='], [0, 1, 3]),
    **{f'h3p1_{beta}': ([[2, 5, -2, -2, 5, 5], [-2, -5, 2, 2, -5, -5]], [beta, -beta], [1, 4, 6, 4, 5, 7], [0] * 6,
                        None, ['Min', '>='], [0, 1, 3]) for beta in range(6)},
}


def model(name):
    A, b, c, l, u, sense, ints = _MODELS[name]
    return MILPInstance(A=np.array(A, float), b=b, c=c, l=l, u=u, sense=sense,
                        integerIndices=ints, numVars=len(c))


def std_model(name):
    """The per-test copies the reference's setUp builds (test_base_node.py:26-35,
    test_branch_and_bound.py:28-35): same rows and objective but NO upper bounds on x."""
    A, b, c, l, u, sense, ints = _MODELS[name]
    m = MILPInstance(A=np.array(A, float), b=b, c=c, l=l, sense=sense, integerIndices=ints,
                     numVars=len(c))
    from simple_mip_solver_amd.algorithms.base_algorithm import BaseAlgorithm
    return BaseAlgorithm._convert_constraints_to_greq(m)
