#!/usr/bin/env python3
"""Streaming rate of k_cross (the gradient cache's Gram-column kernel) at a given shape: a short warm-started
path with the cache forced on, so that a few batches of columns are fetched.  Run under
`rocprofv3 --kernel-trace --stats` to read the kernel's own time (tools/kstats-style)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402

n, p = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("COLS", 1000))
f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=40, noise=6.0)
f.set_gradient_cache(3)
x = cd.SparseIterate(p)
cd.initialize_(f, x)
lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
t0 = time.perf_counter()
for frac in (0.5, 0.3, 0.2, 0.1):
    cd.coordinateDescent_(x, f, cd.ProxL1(frac * lmax), cd.CDOptions(optTol=1e-7, randomize=False))
f._L.cdh_synchronize(f._h)
print({"n": n, "p": p, "seconds": time.perf_counter() - t0, "nnz": x.nnz, "cache": f.cache_stats()})
