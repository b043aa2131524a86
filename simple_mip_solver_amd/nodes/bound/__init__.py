from simple_mip_solver_amd.nodes.bound.disjunctive_cut import DisjunctiveCutBoundNode

__all__ = ['DisjunctiveCutBoundNode']
