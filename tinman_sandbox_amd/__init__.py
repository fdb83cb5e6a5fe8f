"""MI355X-native compute_and_apply_rhs (HOMME CAAR) — host-side Python surface.

The product is the HIP library (csrc/, C ABI in include/caar.h); this package is
the thin host layer the tests and bench.py drive it through.
"""
from .caar import (ARRAY_NAMES, CaarLibrary, Constants, Control, Derivative, ElementArrays,  # noqa: F401
                   HVCoord, TestData, algorithmic_bytes, array_shapes, compute_and_apply_rhs, compute_and_apply_rhs_steps,
                   gll_derivative_matrix, library, placement, print_results_2norm, euler_step, shard_range, sphere_operator, sphere_operator_all, sphere_operator_ex, SPHERE_OPERATORS,
                   state_norms)
