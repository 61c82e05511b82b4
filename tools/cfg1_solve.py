#!/usr/bin/env python3
"""BASELINE.json configs[0] (n = 1000, p = 200, lambda = 0.1) solved REPS times on a resident X: wall time per solve as
the host sees it.  Run under `rocprofv3 --kernel-trace --stats` for the kernels' own time (tools/README.md)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402

reps = int(os.environ.get("REPS", "200"))
rng = np.random.default_rng(123)
n, p, s, lam = 1000, 200, 10, 0.1
X = np.asfortranarray(rng.standard_normal((n, p)))
y = X[:, :s] @ (rng.standard_normal(s) * (1.0 + rng.random(s))) + rng.standard_normal(n)
opt = cd.CDOptions(maxIter=2000, optTol=1e-7, randomize=bool(int(os.environ.get("RAND", "0"))))
f = cd.CDLeastSquaresLoss(y, X)
g = cd.ProxL1(lam)
for _ in range(3):
    cd.coordinateDescent_(cd.SparseIterate(p), f, g, opt)
t0 = time.perf_counter()
for _ in range(reps):
    x = cd.SparseIterate(p)
    cd.coordinateDescent_(x, f, g, opt)
dt = (time.perf_counter() - t0) / reps
# the library call alone (penalty and iterate already on the device side of the ABI)
import ctypes as C  # noqa: E402
o, st = opt._c(), cd._lib.cdh_stats()
t0 = time.perf_counter()
for _ in range(reps):
    f._L.cdh_set_iterate(f._h, p, 0, None, None)
    f._L.cdh_coordinate_descent(f._h, C.byref(o), C.byref(st))
dt_c = (time.perf_counter() - t0) / reps
print({"ms_per_solve_python_api": dt * 1e3, "ms_per_solve_c_abi_calls_only": dt_c * 1e3, "passes": f.last_stats["passes"],
       "visits": f.last_stats["visits"], "nnz": x.nnz, "onchip": f.onchip_stats(), "last": f.onchip_last()})
