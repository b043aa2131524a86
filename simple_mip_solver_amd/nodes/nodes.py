"""Mix-in combinations (simple_mip_solver/nodes/nodes.py:9-14)."""
from simple_mip_solver_amd.nodes.bound.disjunctive_cut import DisjunctiveCutBoundNode
from simple_mip_solver_amd.nodes.branch.pseudo_cost import PseudoCostBranchNode
from simple_mip_solver_amd.nodes.search.depth_first import DepthFirstSearchNode


class PseudoCostBranchDepthFirstSearchNode(PseudoCostBranchNode, DepthFirstSearchNode):
    pass


class DisjunctiveCutBoundPseudoCostBranchNode(DisjunctiveCutBoundNode, PseudoCostBranchNode):
    pass
