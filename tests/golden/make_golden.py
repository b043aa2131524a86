"""Generate golden vectors from the REFERENCE's own pure-Python arithmetic.

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py
It imports simple_mip_solver/utils/floating_point.py, nodes/base_node.py and
nodes/branch/pseudo_cost.py from /root/reference with empty stand-ins for the third-party
packages that are absent (cylp: only the CyClpSimplex class name and the CyLPArray ndarray
subclass are needed by the code paths exercised), following SURVEY.md Appendix B, and records
(inputs, outputs) as JSON next to this script.  Nothing from the reference's source is written.
LP solutions / bases fed to the reference's tableau / Gomory code come from this repo's CPU
oracle (the reference delegates that part to Clp, which is not available).

Also writes example_models_optima.json: optimal objectives of the 64 example .mps instances
computed with HiGHS via scipy (NOT reference output; the reference compares against Gurobi at test
time, test_simple_mip_solver/helpers.py:39-49, and stores no expected values).
"""
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)


def load_reference():
    class CyLPArray(np.ndarray):
        def __new__(cls, a):
            return np.asarray(a, dtype=float).view(cls)

    class CyClpSimplex:
        pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('cylp'); mod('cylp.cy'); mod('cylp.py'); mod('cylp.py.modeling')
    mod('cylp.cy.CyClpSimplex', CyClpSimplex=CyClpSimplex, CyLPArray=CyLPArray)
    mod('cylp.py.modeling.CyLPModel', CyLPArray=CyLPArray)
    for name, sub in [('simple_mip_solver', 'simple_mip_solver'),
                      ('simple_mip_solver.utils', 'simple_mip_solver/utils'),
                      ('simple_mip_solver.nodes', 'simple_mip_solver/nodes'),
                      ('simple_mip_solver.nodes.branch', 'simple_mip_solver/nodes/branch'),
                      ('test_simple_mip_solver', 'test_simple_mip_solver'),
                      ('test_simple_mip_solver.test_utils', 'test_simple_mip_solver/test_utils')]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    sys.path.insert(0, REF)
    fp = importlib.import_module('simple_mip_solver.utils.floating_point')
    bn = importlib.import_module('simple_mip_solver.nodes.base_node')
    pc = importlib.import_module('simple_mip_solver.nodes.branch.pseudo_cost')
    return fp, bn, pc, CyLPArray


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        return [jsonable(v) for v in o.tolist()]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    return o


def golden_floating_point(fp, CyLPArray):
    rng = np.random.default_rng(20261004)
    out = {'get_fraction': [], 'numerically_safe_cut': [], 'scale_cut': []}
    xs = [0.0, 1.0, -1.0, 0.5, -0.5, 1 / 3, -1 / 3, 0.9991, 0.00000001, 1e-15, 3.14159265358979,
          -2.718281828, 1000.0, 1000.5, -1000.5, 999.999, 12345.678, -12345.678, 0.1, 0.7, 1.25 + 1e-15,
          2.375 + 1e-15, 0.999999999, 1.000000001, 5, -7, 0.333, 0.667, 100.01, 0.0099]
    xs += list(rng.uniform(-5, 5, 60)) + list(rng.uniform(0, 1, 60)) + list(10 ** rng.uniform(-6, 4, 40))
    for x in xs:
        x = float(x) if not isinstance(x, int) else x
        for est in (None, 'over', 'under'):
            for mt in (1000.0, 10, 1e6):
                n, d = fp.get_fraction(x, max_term=mt, estimate=est)
                out['get_fraction'].append({'x': x, 'max_term': mt, 'estimate': est,
                                            'n': int(n), 'd': int(d)})
    cases = [([1.25 + 1e-15, 2.375 + 1e-15, 4 + 1e-15], 4.0, 'over', True, {}),
             ([1, 100, 10000], 100, 'over', False, {}),
             ([100, 9999, 10000], 100, 'under', False, {}),
             ([0, 0, 0], 3.0, 'over', False, {}),
             ([0, -1, 0], -2, 'over', False, {})]
    for _ in range(120):
        k = int(rng.integers(2, 9))
        pi = rng.uniform(-10, 10, k) * (rng.random(k) < 0.8)
        if rng.random() < 0.3:
            pi = np.round(pi * 4) / 4
        cases.append((pi.tolist(), float(rng.uniform(-20, 20)), str(rng.choice(['over', 'under'])),
                      bool(rng.random() < 0.3), {}))
    for pi, pi0, est, mk, kw in cases:
        spi, spi0 = fp.numerically_safe_cut(CyLPArray(pi), pi0, estimate=est, make_integer=mk, **kw)
        out['numerically_safe_cut'].append({'pi': pi, 'pi0': pi0, 'estimate': est, 'make_integer': mk,
                                            'safe_pi': jsonable(np.asarray(spi)), 'safe_pi0': float(spi0)})
        a, a0 = fp.scale_cut(np.asarray(pi, float), pi0)
        out['scale_cut'].append({'pi': pi, 'pi0': pi0,
                                 'out_pi': None if a is None else jsonable(a),
                                 'out_pi0': None if a0 is None else float(a0)})
    return out


class DuckLP:
    """The members the reference's tableau / Gomory / selection / pseudo-cost code reads."""

    def __init__(self, A, b, l, u, vstat, CyLPArray):
        from scipy.sparse import csc_matrix
        self._A = np.asarray(A, float)
        self.coefMatrix = csc_matrix(self._A)
        self.constraintsLower = np.asarray(b, float)
        self.variablesLower = np.asarray(l, float)
        self.variablesUpper = np.asarray(u, float)
        self.nVariables = self._A.shape[1]
        self.nConstraints = self._A.shape[0]
        self._vstat = np.asarray(vstat)
        self.added = []
        self._status = 0
        self.objectiveValue = 0.0

    def getBasisStatus(self):
        n = self.nVariables
        return self._vstat[:n], self._vstat[n:]

    def getVarByName(self, name):
        return 1.0  # pi * 1.0 >= pi0 evaluates to a boolean array; only the call is recorded

    def addConstraint(self, cut, name):
        self.added.append(name)

    def getStatusCode(self):
        return self._status


def golden_base_node(bn, pc, CyLPArray):
    import re
    from oracle import oracle as O
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    INF = np.inf
    insts = {
        'cut2': (np.array([[-4, -1], [-1, -4], [-1, 1.]]), [-28, -27, -1.], [-2, -5.], [0, 0.], [INF] * 2, [0, 1]),
        'cut3': (np.array([[-3, -4], [-5, -10], [-1, -2.]]), [-10, -8, -1.2], [-8, -12.], [0, 0.], [INF] * 2, [0, 1]),
        'small_branch': (np.array([[-1, 0, -1], [0, -1, 0.]]), [-1.5, -1.25], [-1, -1, -1.], [0, 0, 0.], [10.] * 3, [0, 1, 2]),
    }
    for seed, (n, m) in enumerate([(6, 4), (8, 5), (10, 6), (12, 8), (16, 8), (20, 10)]):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, seed=100 + seed)
        u = np.full(n, INF) if seed % 2 == 0 else u   # half without upper bounds (nonbasic at 0)
        insts[f'rand{n}x{m}'] = (A, b, c, l, u, ints[: max(2, n - 2)])
    out = []
    for name, (A, b, c, l, u, ints) in insts.items():
        A = np.asarray(A, float)
        r = O.lp_solve(A, b, c, l, u)
        if r['status'] != 0:
            continue
        node = bn.BaseNode.__new__(bn.BaseNode)
        node.lp = DuckLP(A, b, l, u, r['vstat'], CyLPArray)
        node._integer_indices = list(ints)
        node.solution = np.maximum(r['x'].copy(), 0)
        node.lp_feasible = True
        node.idx = 0
        node.cut_generation_iterations = 1
        node.cut_name_pattern = re.compile('^cut_')
        node.gmic_name_pattern = re.compile('^cut_gomory_')
        for op in ('created', 'added', 'removed'):
            setattr(node, f'iterations_gmic_{op}', 0)
            setattr(node, f'number_gmic_{op}', 0)
        node._cut_pool = {}
        node.cut_generation_terminator = None
        node.max_term = float(np.max(np.abs(A)))
        tab = node.tableau
        cuts = node._find_gomory_cuts()
        pool = node._generate_cuts(gomory_cuts=True)
        node.cut_pool = dict(pool)
        added = node._select_cuts()
        rec = {
            'name': name, 'A': jsonable(A), 'b': jsonable(np.asarray(b, float)),
            'c': jsonable(np.asarray(c, float)), 'l': jsonable(np.asarray(l, float)),
            'u': [None if np.isinf(v) else float(v) for v in np.asarray(u, float)],
            'integer_indices': list(map(int, ints)), 'vstat': jsonable(r['vstat']),
            'x': jsonable(r['x']), 'obj': float(r['obj']),
            'basic_variable_indices': jsonable(node.basic_variable_indices),
            'tableau': None if tab is None else jsonable(tab),
            'gomory': {str(k): {'pi': jsonable(np.asarray(v[0])), 'pi0': float(v[1])} for k, v in cuts.items()},
            'generated': {k: {'pi': jsonable(np.asarray(v[0])), 'pi0': float(v[1])} for k, v in pool.items()},
            'selected': list(added.keys()), 'left_in_pool': list(node.cut_pool.keys()),
            'terminator': node.cut_generation_terminator,
            'counters': {f'{a}_{op}': getattr(node, f'{a}_gmic_{op}') for a in ('iterations', 'number')
                         for op in ('created', 'added', 'removed')},
            'most_fractional_index': node._most_fractional_index,
        }
        out.append(rec)
    # the reference's own selection fixtures (test_base_node.py:568-652) run through its code
    sel = []
    A, b, c, l, u, ints = insts['small_branch']
    r = O.lp_solve(A, b, c, l, u)
    pools = {
        'default': {'cut_1': ([-1, -1, -1], -2), 'cut_2': ([-1, 0, -1], -1), 'cut_3': ([0, -1, 0], -1),
                    'cut_4': ([0, 0, 0], 0), 'cut_5': ([-99, 0, -101], -110), 'cut_6': ([-1, 0, 0], -2),
                    'cut_7': ([-10000, -10000, -10000], -10000)},
    }
    for kw in ({}, {'max_nonzero_coefs': 2, 'parallel_cut_tolerance': .0001}, {'min_cut_depth': .5},
               {'max_relative_cut_term_ratio': 50}):
        node = bn.BaseNode.__new__(bn.BaseNode)
        node.lp = DuckLP(A, b, l, u, r['vstat'], CyLPArray)
        node._integer_indices = list(ints)
        node.solution = r['x'].copy()
        node.lp_feasible = True
        node.cut_name_pattern = re.compile('^cut_')
        node.gmic_name_pattern = re.compile('^cut_gomory_')
        for op in ('created', 'added', 'removed'):
            setattr(node, f'iterations_gmic_{op}', 0)
            setattr(node, f'number_gmic_{op}', 0)
        node.cut_generation_terminator = None
        node.max_term = float(np.max(np.abs(A)))
        node._cut_pool = {k: (CyLPArray(p), p0) for k, (p, p0) in pools['default'].items()}
        added = node._select_cuts(**kw)
        sel.append({'kwargs': kw, 'pool': {k: {'pi': p, 'pi0': p0} for k, (p, p0) in pools['default'].items()},
                    'x': jsonable(r['x']), 'selected': list(added.keys()),
                    'left_in_pool': list(node.cut_pool.keys()), 'terminator': node.cut_generation_terminator})
    # pseudo-cost arithmetic (pseudo_cost.py:68-133)
    pcs = []
    rng = np.random.default_rng(7)
    for _ in range(40):
        n = 6
        node = pc.PseudoCostBranchNode.__new__(pc.PseudoCostBranchNode)
        node._integer_indices = [0, 1, 2, 4]
        node.solution = np.round(rng.uniform(0, 5, n), int(rng.integers(0, 3)))
        table = {}
        for i in node._integer_indices:
            if rng.random() < 0.85:
                table[i] = {d: {'cost': float(np.round(rng.uniform(0, 3), 2)), 'times': int(rng.integers(0, 5))}
                            for d in ('right', 'left')}
        frac = [i for i in node._integer_indices if node._is_fractional(node.solution[i])]
        rec = {'integer_indices': node._integer_indices, 'x': jsonable(node.solution), 'table': jsonable(table)}
        if frac and all(i in table for i in frac):
            rec['best_index'] = int(node._best_pseudo_costs_index(table))
        # one running-mean update
        child = pc.PseudoCostBranchNode.__new__(pc.PseudoCostBranchNode)
        child._b_idx = int(rng.choice(node._integer_indices))
        child._b_dir = str(rng.choice(['left', 'right']))
        child._b_val = float(rng.uniform(0.1, 4.9))
        l = np.zeros(n); u = np.full(n, 10.)
        if child._b_dir == 'left':
            u[child._b_idx] = np.floor(child._b_val)
        else:
            l[child._b_idx] = np.ceil(child._b_val)
        child.lp = DuckLP(np.zeros((1, n)), [0.], l, u, np.zeros(n + 1), CyLPArray)
        child.lp._status = int(rng.choice([0, 1, 3]))
        child.dual_bound = float(rng.uniform(-10, 0))
        child.lp.objectiveValue = child.dual_bound + float(rng.uniform(-0.5, 3))
        node.pseudo_costs = json.loads(json.dumps(table), object_hook=lambda d: {
            (int(k) if k.lstrip('-').isdigit() else k): v for k, v in d.items()})
        node._calculate_costs(child)
        rec['update'] = {'b_idx': child._b_idx, 'b_dir': child._b_dir, 'b_val': child._b_val,
                         'l': jsonable(l), 'u': jsonable(u), 'status': child.lp._status,
                         'dual_bound': child.dual_bound, 'objective': child.lp.objectiveValue,
                         'table_after': jsonable(node.pseudo_costs)}
        pcs.append(rec)
    return {'nodes': out, 'select_cuts': sel, 'pseudo_costs': pcs}


LARGE_CASES = [
    # (n, m, seed, density, boxed): the sizes the numbers are quoted on (BASELINE configs C2 / C4) and the
    # shapes on which the reference's own rules DO select cuts (sparser rows, or no upper bounds)
    (64, 32, 0, 1.0, True), (64, 32, 5, 1.0, False), (64, 32, 2, 0.25, True),
    (256, 128, 0, 1.0, True), (256, 128, 1, 1.0, False), (256, 128, 3, 0.25, True),
]


def golden_base_node_large(bn, CyLPArray):
    """Roots and children (left branches: a nonbasic-at-upper column each, base_node.py:496-503; one right
    branch) of BASELINE-sized instances through the reference's tableau / _find_gomory_cuts /
    _generate_cuts / _select_cuts (base_node.py:365-530).  The instance is not stored: the generator
    (simple_mip_solver_amd/generators.py) rebuilds it from (n, m, seed, density).  Node LPs are solved
    by this repo's CPU oracle (the reference delegates that to Clp)."""
    import re
    from oracle import oracle as O
    from simple_mip_solver_amd.generators import random_dense_milp_arrays
    INF = np.inf
    out = {'cases': np.array(LARGE_CASES, float)}
    for ci, (n, m, seed, density, boxed) in enumerate(LARGE_CASES):
        A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
        if not boxed:
            u = np.full(n, INF)
        root = O.lp_solve(A, b, c, l, u)
        assert root['status'] == 0
        x = root['x']
        frac = np.minimum(x - np.floor(x), np.ceil(x) - x)
        L, U = [l], [u]
        order = [j for j in np.argsort(-frac, kind='stable') if frac[j] > 1e-4]
        for j in order[:3]:
            u2 = u.copy(); u2[j] = np.floor(x[j]); L.append(l); U.append(u2)
        if n >= 256:   # (the big shapes: root, one left child, one right child -- the fixture stays small)
            L, U = L[:2], U[:2]
        if order:
            l2 = l.copy(); l2[order[0]] = np.ceil(x[order[0]]); L.append(l2); U.append(u)
        k = 0
        for lk, uk in zip(L, U):
            r = O.lp_solve(A, b, c, lk, uk, None if k == 0 else root['vstat'])
            if r['status'] != 0:
                continue
            node = bn.BaseNode.__new__(bn.BaseNode)
            node.lp = DuckLP(A, b, lk, uk, r['vstat'], CyLPArray)
            node._integer_indices = list(ints)
            node.solution = np.maximum(r['x'].copy(), 0)
            node.lp_feasible = True
            node.idx = 0
            node.cut_generation_iterations = 1
            node.cut_name_pattern = re.compile('^cut_')
            node.gmic_name_pattern = re.compile('^cut_gomory_')
            for op in ('created', 'added', 'removed'):
                setattr(node, f'iterations_gmic_{op}', 0)
                setattr(node, f'number_gmic_{op}', 0)
            node._cut_pool = {}
            node.cut_generation_terminator = None
            node.max_term = float(np.max(np.abs(A)))
            cuts = node._find_gomory_cuts()
            pool = node._generate_cuts(gomory_cuts=True)
            node.cut_pool = dict(pool)
            added = node._select_cuts()
            rows = np.array(sorted(cuts), np.int32)
            key = f'c{ci}_k{k}_'
            out[key + 'l'] = np.asarray(lk, float); out[key + 'u'] = np.asarray(uk, float)
            out[key + 'vstat'] = np.asarray(r['vstat'], np.int8); out[key + 'x'] = np.asarray(r['x'], float)
            out[key + 'rows'] = rows
            out[key + 'pi'] = np.array([np.asarray(cuts[q][0], float) for q in rows]).reshape(len(rows), n)
            out[key + 'pi0'] = np.array([float(cuts[q][1]) for q in rows])
            out[key + 'safe_pi'] = np.array([np.asarray(pool[f'cut_gomory_0_1_{q}'][0], float) for q in rows]).reshape(len(rows), n)
            out[key + 'safe_pi0'] = np.array([float(pool[f'cut_gomory_0_1_{q}'][1]) for q in rows])
            # selected cuts in the order they were added (their tableau rows), terminator 0 None / 1 'no cuts' /
            # 2 'no improving cuts' / 3 'no sufficient cuts'
            out[key + 'selected'] = np.array([int(name.rsplit('_', 1)[1]) for name in added], np.int32)
            out[key + 'terminator'] = np.array([{None: 0, 'no cuts': 1, 'no improving cuts': 2,
                                                 'no sufficient cuts': 3}[node.cut_generation_terminator]], np.int32)
            out[key + 'counters'] = np.array([getattr(node, f'{a}_gmic_{op}') for a in ('iterations', 'number')
                                              for op in ('created', 'added', 'removed')], np.int32)
            print(f'  large case {ci} ({n}x{m} seed {seed} density {density} boxed {boxed}) node {k}: '
                  f'{len(rows)} cuts, {len(added)} selected, terminator {node.cut_generation_terminator!r}')
            k += 1
        out[f'c{ci}_count'] = np.array([k], np.int32)
    return out


def golden_example_models():
    from scipy.optimize import milp, LinearConstraint, Bounds, linprog
    from simple_mip_solver_amd.milp_instance import read_mps
    src = os.path.join(REF, 'test_simple_mip_solver', 'example_models')
    dst = os.path.join(HERE, 'example_models')
    os.makedirs(dst, exist_ok=True)
    table = {}
    for f in sorted(os.listdir(src)):
        if not f.endswith('.mps'):
            continue
        data = open(os.path.join(src, f)).read()
        open(os.path.join(dst, f), 'w').write(data)  # data fixture (an instance file), verbatim
        A, b, c, l, u, sense, ints = read_mps(os.path.join(dst, f))
        u = np.where(u > 1e300, np.inf, u)
        cons = LinearConstraint(A, -np.inf, b) if sense[1] == '<=' else LinearConstraint(A, b, np.inf)
        integrality = np.zeros(len(c)); integrality[ints] = 1
        res = milp(c, constraints=cons, bounds=Bounds(l, u), integrality=integrality)
        rel = linprog(c, A_ub=A if sense[1] == '<=' else -A, b_ub=b if sense[1] == '<=' else -b,
                      bounds=list(zip(l, u)), method='highs')
        assert res.status == 0 and rel.status == 0, f
        table[f] = {'rows': int(A.shape[0]), 'cols': int(A.shape[1]), 'milp_opt': float(res.fun),
                    'lp_opt': float(rel.fun)}
    return {'provenance': 'HiGHS via scipy %s (NOT reference output)' % __import__('scipy').__version__,
            'models': table}


if __name__ == '__main__':
    fp, bn, pc, CyLPArray = load_reference()
    json.dump(golden_floating_point(fp, CyLPArray), open(os.path.join(HERE, 'floating_point.json'), 'w'))
    json.dump(golden_base_node(bn, pc, CyLPArray), open(os.path.join(HERE, 'base_node.json'), 'w'))
    json.dump(golden_example_models(), open(os.path.join(HERE, 'example_models_optima.json'), 'w'), indent=1)
    np.savez_compressed(os.path.join(HERE, 'base_node_large.npz'), **golden_base_node_large(bn, CyLPArray))
    print('golden vectors written to', HERE)
