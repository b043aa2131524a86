"""BaseAlgorithm: shared start-up of the tree-search drivers.

Mirror of simple_mip_solver/algorithms/base_algorithm.py:12-72: converts the model to
`min c'x, Ax >= b`, instantiates and validates the root Node, and owns the `_kwargs` dictionary
that is splatted into every node call and updated from every returned dict (the plugin ABI).
"""
import inspect

import numpy as np

from simple_mip_solver_amd.milp_instance import MILPInstance


class BaseAlgorithm:

    def __init__(self, model, Node, node_attributes, node_funcs, **kwargs):
        assert isinstance(model, MILPInstance), 'model must be cuppy MILPInstance'
        # the stored model may differ from the one passed in (constraints flipped to >=)
        self.model = self._convert_constraints_to_greq(model)
        self._swapped_constraint_direction = model.sense != self.model.sense

        assert inspect.isclass(Node), 'Node must be a class'
        root_node = Node(lp=self.model.lp, integer_indices=self.model.integerIndices, idx=0,
                         **kwargs)
        for attribute in node_attributes:
            assert hasattr(root_node, attribute), f'Node needs a {attribute} attribute'
        for func in node_funcs:
            assert callable(getattr(root_node, func, None)), f'Node needs a {func} function'

        assert 'next_node_idx' not in kwargs, 'key next_node_idx is reserved for use by solver'

        self._Node = Node
        self.root_node = root_node
        self.evaluated_nodes = 0
        kwargs['next_node_idx'] = 1  # node methods advance this through their returned dicts
        self._kwargs = kwargs
        self._M = 999999999

    @staticmethod
    def _convert_constraints_to_greq(model):
        """A x <= b becomes -A x >= -b in a fresh instance; >= models pass through
        (reference :47-61).  The objective is taken from model.lp, i.e. already a minimisation."""
        if model.sense != '<=':
            return model
        return MILPInstance(A=-np.asarray(model.A), b=-np.asarray(model.b), c=model.lp.objective,
                            l=model.l, u=model.u, integerIndices=model.integerIndices,
                            sense=['Min', '>='], numVars=len(model.c))

    def _process_rtn(self, rtn):
        """Merge a node method's returned dict into the kwargs of all later calls."""
        assert isinstance(rtn, dict), 'rtn must be a dictionary'
        assert all(isinstance(k, str) for k in rtn), 'rtn keys must be strings'
        self._kwargs.update(rtn)
