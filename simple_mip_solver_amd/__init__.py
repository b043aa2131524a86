"""simple_mip_solver_amd -- MI355X-native branch-and-bound node engine behind the
Node / BranchAndBound plugin surface of spkelle2/simple_mip_solver.

Exports mirror simple_mip_solver/__init__.py:1-9 plus the problem container the reference takes
from coinor.cuppy.
"""
from simple_mip_solver_amd.nodes.base_node import BaseNode
from simple_mip_solver_amd.algorithms.branch_and_bound import BranchAndBound
from simple_mip_solver_amd.nodes.search.depth_first import DepthFirstSearchNode
from simple_mip_solver_amd.nodes.branch.pseudo_cost import PseudoCostBranchNode
from simple_mip_solver_amd.nodes.bound.disjunctive_cut import DisjunctiveCutBoundNode
from simple_mip_solver_amd.nodes.nodes import PseudoCostBranchDepthFirstSearchNode, \
    DisjunctiveCutBoundPseudoCostBranchNode
from simple_mip_solver_amd.milp_instance import MILPInstance
from simple_mip_solver_amd.lp import CyLPArray, DenseLP

__version__ = '0.1.0'
__all__ = ['BaseNode', 'BranchAndBound', 'DepthFirstSearchNode', 'PseudoCostBranchNode',
           'PseudoCostBranchDepthFirstSearchNode', 'DisjunctiveCutBoundNode',
           'DisjunctiveCutBoundPseudoCostBranchNode', 'MILPInstance', 'CyLPArray', 'DenseLP']
