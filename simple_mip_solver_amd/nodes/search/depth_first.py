"""DepthFirstSearchNode: deeper nodes first (simple_mip_solver/nodes/search/depth_first.py:8-28)."""
from simple_mip_solver_amd.nodes.base_node import BaseNode


class DepthFirstSearchNode(BaseNode):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.search_method = 'depth first'

    def __eq__(self, other):
        if not isinstance(other, DepthFirstSearchNode):
            raise TypeError('A DFS Node can only be compared with another DFS Node')
        return self.depth == other.depth

    def __lt__(self, other):
        # "less than" == popped earlier == deeper in the tree
        if not isinstance(other, DepthFirstSearchNode):
            raise TypeError('A DFS Node can only be compared with another DFS Node')
        return self.depth > other.depth

    __hash__ = object.__hash__
