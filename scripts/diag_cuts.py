import numpy as np, sys
sys.path.insert(0, '.')
from simple_mip_solver_amd import BranchAndBound, PseudoCostBranchNode, BaseNode, MILPInstance
from simple_mip_solver_amd.generators import random_dense_milp_arrays
def random_model(n, m, seed, density=1.0, inf_u=False):
    A, b, c, l, u, ints = random_dense_milp_arrays(n, m, density=density, seed=seed)
    if inf_u: u = np.full(n, np.inf)
    return MILPInstance(A=A, b=b, c=c, l=l, u=u, sense=['Min', '>='], integerIndices=ints, numVars=n)
for inf_u in (False, True):
  for n, m, seed, density in ((20, 10, 1, 1.0), (24, 10, 12, 1.0), (30, 15, 2, 0.3 if not inf_u else 1.0)):
    make = lambda: random_model(n, m, seed, density, inf_u)
    ref = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=False, frontier_batch=1); ref.solve()
    py = BranchAndBound(make(), PseudoCostBranchNode, pseudo_costs={}, gomory_cuts=True); py.solve()
    out = [ref.objective_value, py.objective_value]
    for Node in (BaseNode, PseudoCostBranchNode):
        for batch in (1, 4, 64):
            for anchor in (True, False):
                if batch == 1 and anchor: continue
                bb = BranchAndBound(make(), Node, pseudo_costs={}, gomory_cuts=True, frontier_batch=batch, pool_capacity=1 << 15, anchor=anchor)
                bb.solve()
                out.append((Node.__name__[:4], batch, anchor, bb.objective_value, bb.evaluated_nodes, bb._kwargs['total_number_gmic_added']))
    print(inf_u, n, m, seed, out, flush=True)
