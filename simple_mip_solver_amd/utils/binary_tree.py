"""Minimal binary tree with attribute dictionaries on its vertices.

Stands in for the part of `coinor.gimpy.tree.BinaryTree` that the reference's
BranchAndBoundTree builds on (simple_mip_solver/algorithms/branch_and_bound.py:19, :193,
:286-287): add_root / add_left_child / add_right_child with keyword attributes, `nodes`
(id -> vertex with `.attr`), membership test, children / parent queries.
"""


class TreeVertex:
    __slots__ = ('name', 'attr')

    def __init__(self, name, attr):
        self.name = name
        self.attr = attr

    def get_attr(self, key):
        return self.attr.get(key)

    def set_attr(self, key, value):
        self.attr[key] = value


class BinaryTree:

    def __init__(self):
        self.nodes = {}
        self.root = None

    def __contains__(self, name):
        return name in self.nodes

    def _add(self, name, parent, direction, attr):
        assert name not in self.nodes, f'vertex {name} already exists'
        attr = dict(attr)
        attr.update(parent=parent, direction=direction, Lchild=None, Rchild=None)
        self.nodes[name] = TreeVertex(name, attr)
        return self.nodes[name]

    def add_root(self, root, **attr):
        assert self.root is None, 'tree already has a root'
        self.root = root
        return self._add(root, None, None, attr)

    def _add_child(self, name, parent, side, attr):
        assert parent in self.nodes, f'parent {parent} is not in the tree'
        slot = 'Lchild' if side == 'L' else 'Rchild'
        assert self.nodes[parent].attr[slot] is None, f'{parent} already has that child'
        vertex = self._add(name, parent, side, attr)
        self.nodes[parent].attr[slot] = name
        return vertex

    def add_left_child(self, n, parent, **attr):
        return self._add_child(n, parent, 'L', attr)

    def add_right_child(self, n, parent, **attr):
        return self._add_child(n, parent, 'R', attr)

    def get_node(self, name):
        return self.nodes.get(name)

    def get_node_attr(self, name, key):
        return self.nodes[name].attr.get(key)

    def set_node_attr(self, name, key, value):
        self.nodes[name].attr[key] = value

    def get_parent(self, n):
        return self.nodes[n].attr['parent']

    def get_left_child(self, n):
        return self.nodes[n].attr['Lchild']

    def get_right_child(self, n):
        return self.nodes[n].attr['Rchild']

    def get_children(self, n):
        a = self.nodes[n].attr
        return [c for c in (a['Lchild'], a['Rchild']) if c is not None]
