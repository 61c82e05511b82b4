"""Leg 2: the gradient cache and its covariance-form visits forced on for every handle
(`CDH_GRADIENT_CACHE=3`: engaged from the first full pass, no size guard; csrc/grad_cache.hpp) -- every
parity case's full passes are served from (g, Gram columns) instead of reading X
(coordinate_descent.jl:94-110 over 1..p)."""
from _legs import adopt

SKIP = {}
adopt(globals(), {"CDH_GRADIENT_CACHE": "3"}, SKIP)
