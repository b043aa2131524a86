import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from simple_mip_solver_amd import BaseNode, BranchAndBound, MILPInstance, _ffi
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests')
TABLE = json.load(open(os.path.join(HERE, 'golden', 'example_models_optima.json')))['models']
for f in sorted(TABLE):
    path = os.path.join(HERE, 'golden', 'example_models', f)
    py = BranchAndBound(MILPInstance(file_name=path), BaseNode, gomory_cuts=False)
    tr = []
    inner = py._evaluate_node
    def spy(node):
        b = py.evaluated_nodes; inner(node)
        if py.evaluated_nodes > b: tr.append((node.idx, node.lp.getStatusCode(), node.lp.objectiveValue, node.lp.iteration))
    py._evaluate_node = spy
    py.solve()
    real = _ffi.Tree
    class T(real):
        def __init__(self, *a, **k):
            super().__init__(*a, **k); self.set_trace(True)
    _ffi.Tree = T
    nb = BranchAndBound(MILPInstance(file_name=path), BaseNode, gomory_cuts=False, frontier_batch=1)
    nb.solve()
    _ffi.Tree = real
    t = nb._native.trace()
    bad = [k for k in range(min(len(tr), len(t['node_id']))) if tr[k][2] != t['objective'][k] and tr[k][1] in (0, 3)]
    if bad or len(tr) != len(t['node_id']):
        print(f[12:70], 'len', len(tr), len(t['node_id']), 'first bad', bad[:3])
        for k in bad[:3]:
            print('   py', tr[k], 'native', t['node_id'][k], t['status'][k], repr(t['objective'][k]))
        break
print('done')
