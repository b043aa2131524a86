"""simple_mip_solver_amd -- MI355X-native branch-and-bound node engine behind the
Node / BranchAndBound plugin surface of spkelle2/simple_mip_solver
(simple_mip_solver/__init__.py:1-9 lists the names mirrored here)."""
__version__ = '0.1.0'
