"""Leg 1: the product's DEFAULT configuration -- the one-launch solve (csrc/small_solve.hpp) on, as
`bench.py`, `smoke()` and every user get it.  tests/conftest.py switches it off for the main suite so
that the streamed sweep kernels keep their solve-level coverage; here every eligible solve of every
parity case goes through `k_solve_small` instead (coordinate_descent.jl:65-92 in one launch)."""
from _legs import adopt

SKIP = {}
adopt(globals(), {"CDH_SMALL_PATH": "1"}, SKIP)
