"""MILPInstance stand-in, the MPS reader and the cylp-style modelling sugar of DenseLP."""
import os

import numpy as np
import pytest

from simple_mip_solver_amd import CyLPArray, DenseLP, MILPInstance
from simple_mip_solver_amd.lp import COIN_INFINITY
from simple_mip_solver_amd.milp_instance import read_mps
from simple_mip_solver_amd.algorithms.base_algorithm import BaseAlgorithm
from tests.support.example_models import model

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'example_models')


def test_from_arrays():
    m = model('small_branch')
    assert m.sense == '>=' and m.integerIndices == [0, 1, 2] and m.numVars == 3 and m.numCons == 2
    assert m.lp.nVariables == 3 and m.lp.nConstraints == 2 and m.lp.nCols == 3
    assert all(m.lp.variablesUpper == 10) and all(m.lp.variablesLower == 0)
    assert all(m.lp.constraintsLower == [-1.5, -1.25]) and all(m.lp.constraintsUpper == COIN_INFINITY)
    assert all(m.lp.objective == [-1, -1, -1])
    assert len(m.lp.variables) == 1 and m.lp.variables[0].name == 'x'
    assert np.array_equal(m.lp.coefMatrix.toarray(), [[-1, 0, -1], [0, -1, 0]])
    mx = model('small_branch_max')
    assert mx.sense == '<=' and all(mx.lp.objective == [-1, -1, -1]) and all(mx.c == [1, 1, 1])
    assert all(mx.lp.constraintsUpper == [1.5, 1.25])


def test_convert_constraints_to_greq():  # test_base_algorithm.py:19-35
    mx = model('small_branch_max')
    ge = BaseAlgorithm._convert_constraints_to_greq(mx)
    assert ge is not mx and ge.sense == '>='
    assert np.array_equal(ge.A, -mx.A) and all(ge.b == -mx.b) and all(ge.lp.objective == [-1, -1, -1])
    assert all(ge.l == mx.l) and all(ge.u == mx.u) and ge.integerIndices == mx.integerIndices
    same = model('small_branch')
    assert BaseAlgorithm._convert_constraints_to_greq(same) is same


def test_mps_reader_on_reference_fixture():
    f = 'constraints_low_variables_low_density_low_max_obj_coeff_low_max_cons_coeff_low_tightness_low.mps'
    A, b, c, l, u, sense, ints = read_mps(os.path.join(GOLD, f))
    assert np.array_equal(A, [[3, 0], [0, 0]]) and all(b == [2, 2]) and all(c == [-1, -2])
    assert all(l == 0) and all(u == 10) and sense == ['Min', '<='] and ints == [0, 1]
    m = MILPInstance(file_name=os.path.join(GOLD, f))
    assert m.sense == '<=' and m.integerIndices == [0, 1] and m.lp.nConstraints == 2
    assert len([f for f in os.listdir(GOLD) if f.endswith('.mps')]) == 64


def test_mps_markers_and_bound_kinds(tmp_path):
    text = """NAME T
ROWS
 N obj
 G r1
 G r2
COLUMNS
    MARKER                 'MARKER'                 'INTORG'
    a  obj 1.0  r1 1.0
    b  obj 2.0  r1 1.0
    MARKER                 'MARKER'                 'INTEND'
    c  obj 3.0  r2 1.0
    c  r1 -1.0
RHS
    rhs r1 1.5 r2 0.25
BOUNDS
 UP bnd a 4.0
 LO bnd b 1.0
 FX bnd c 0.5
ENDATA
"""
    p = tmp_path / 't.mps'
    p.write_text(text)
    A, b, c, l, u, sense, ints = read_mps(str(p))
    assert np.array_equal(A, [[1, 1, -1], [0, 0, 1]]) and all(b == [1.5, .25]) and all(c == [1, 2, 3])
    assert all(l == [0, 1, .5]) and u[0] == 4 and u[1] == COIN_INFINITY and u[2] == .5
    assert sense == ['Min', '>='] and ints == [0, 1]


def test_modelling_sugar():
    lp = DenseLP()
    x = lp.addVariable('x', 3)
    lp += CyLPArray([0, 0, 0]) <= x <= CyLPArray([10, 10, 1])
    lp.addConstraint(np.array([[-1, 0, -1], [0, -1, 0.]]) * x >= CyLPArray([-1.5, -1.25]), 'R_1')
    lp.addConstraint(CyLPArray([0, -1, 0]) * x >= -2, 'cut_gomory_0_1_0')
    lp.objective = CyLPArray([-1, -1, -1])
    assert lp.nConstraints == 3 and [c.name for c in lp.constraints] == ['R_1', 'cut_gomory_0_1_0']
    assert all(lp.variablesUpper == [10, 10, 1])
    c0 = lp.constraints[0]
    assert np.array_equal(c0.varCoefs[c0.variables[0]], [[-1, 0, -1], [0, -1, 0]])
    assert lp.getVarByName('x') is x
    # the rebuild idiom of the reference's _base_branch (base_node.py:602-606)
    lp2 = DenseLP()
    y = lp2.addVariable('x', 3)
    for con in lp.constraints:
        lp2.addConstraint(CyLPArray(con.lower.copy()) <= con.varCoefs[con.variables[0]] * y
                          <= CyLPArray(con.upper.copy()), name=con.name)
    assert np.array_equal(lp2.dense_rows(), lp.dense_rows())
    assert all(lp2.constraintsLower == lp.constraintsLower)
    lp.removeConstraint('cut_gomory_0_1_0')
    assert lp.nConstraints == 2
    with pytest.raises(Exception, match='Constraint "cut_gomory_0_1_0" does not exist'):
        lp.removeConstraint('cut_gomory_0_1_0')


def test_engine_form_handles_le_and_ranged_rows():
    lp = DenseLP()
    x = lp.addVariable('x', 2)
    lp.addConstraint(CyLPArray([1, 1]) * x <= 4, 'le')
    lp.addConstraint(CyLPArray([1.0]) <= CyLPArray([1, -1]) * x <= CyLPArray([2.0]), 'rng')
    rs = lp._engine_form()
    assert np.array_equal(rs.A, [[-1, -1], [1, -1], [-1, 1]]) and all(rs.b == [-4, 1, -2])


def test_mps_writer_round_trips_every_fixture(tmp_path):
    """write_mps -> read_mps gives back the arrays of all 64 example models (the reference's
    on-disk format, test_simple_mip_solver/example_models/*.mps), plus the awkward bound kinds."""
    import glob
    from simple_mip_solver_amd.milp_instance import read_mps, write_mps
    files = sorted(glob.glob(os.path.join(GOLD, '*.mps')))
    assert len(files) == 64
    for k, f in enumerate(files):
        A, b, c, l, u, sense, ints = read_mps(f)
        out = tmp_path / f'rt_{k}.mps'
        write_mps(out, A, b, c, l, u, sense, ints)
        A2, b2, c2, l2, u2, sense2, ints2 = read_mps(out)
        assert np.array_equal(A, A2) and np.array_equal(b, b2) and np.array_equal(c, c2), f
        assert np.array_equal(l, l2) and np.array_equal(u, u2) and sense == sense2 and ints == ints2, f
    A = np.array([[1.5, 0, -2], [0, 0, 1e-3]]); b = [0, -7.25]; c = [0, 3, -1e9]
    l = [-np.inf, 2, 1]; u = [4, 2, np.inf]
    out = tmp_path / 'odd.mps'
    write_mps(out, A, b, c, l, u, ['Min', '>='], [2])
    A2, b2, c2, l2, u2, sense2, ints2 = read_mps(out)
    assert np.array_equal(A, A2) and np.array_equal(b2, b) and np.array_equal(c2, c)
    assert l2[0] < -1e300 and l2[1] == u2[1] == 2 and l2[2] == 1 and u2[2] > 1e300 and u2[0] == 4
    assert sense2 == ['Min', '>='] and ints2 == [2]
