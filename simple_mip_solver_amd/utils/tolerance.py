"""Numeric tolerances and limits of the node hot path.

Same names and values as simple_mip_solver/utils/tolerance.py:2-43 (they are part of the
plugin surface: node methods take them as keyword defaults and users override them through
BranchAndBound(**kwargs)).
"""

# integrality: |x - round(x)| <= variable_epsilon counts as integer (base_node.py:283, :656)
variable_epsilon = 1e-4

# cut rounding (utils/floating_point.py): a rational estimate within 1 % is "good";
# an exact continued-fraction match within 1e-14 replaces a poor directed estimate
good_coefficient_approximation_epsilon = 1e-2
exact_coefficient_approximation_epsilon = 1e-14
cut_tolerance = 1e-14
max_term = 1e3                      # largest numerator / denominator in a rational estimate

# cut selection (base_node.py:387-466)
max_nonzero_coefs = 1000000         # cuts with more "nonzero" (> 1e-2) coefficients are skipped
parallel_cut_tolerance = 10         # degrees; closer cuts to an already added one are skipped
max_relative_cut_term_ratio = 1000  # max |pi| allowed relative to the root LP's max |A|
min_cut_depth = 1e-8                # euclidean violation a cut needs to be added

# cut loop control (base_node.py:137-230, :292-324)
cutting_plane_progress_tolerance = 1e-4
max_cut_generation_iterations = 10

# disjunctive cuts (out of scope here, kept so user kwargs resolve)
min_cglp_norm = 1e-4
