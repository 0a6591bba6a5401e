"""CPU oracle for the SOR hot path -- TEST INFRASTRUCTURE, never the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  See oracle/xinv_oracle.c for what is restated and how it is pinned.
"""
from .oracle import (  # noqa: F401
    LEX, COLOUR_AUTO, COLOUR_2, COLOUR_4, FMA, BC_CODES,
    build, lib, use_native, use_portable, standard_2d, general_2d, standard_3d, general_3d, general_bih_2d,
    standard_2d_test, abs_norm,
)
