#!/usr/bin/env python3
"""LassoPath (src/lasso.jl:229-260; omega = _stdX!, optTol 1e-7, ordered) at the reference's own benchmark shape
(benchmark/cd_bench.jl: n = 3000, p = 5000, s = 100, noise 6): 60 lambdas from 0.95 to 0.03 lambda_max (774 non-zeros at the
end), the GPU path with the gradient cache in its default mode / forced / off, and with the device-resident pass loop off
(CDH_COV_SOLVE=0 in the environment).  Prints one line per mode; the CPU port's time with CPU=1; shuffled sweeps with RANDOMIZE=1."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402

rng = np.random.default_rng(123)
n, p, s = 3000, 5000, 100
X = np.asfortranarray(rng.standard_normal((n, p)))
Y = X[:, :s] @ (rng.standard_normal(s) * (1.0 + rng.random(s))) + 6.0 * rng.standard_normal(n)
lmax = float(np.max(np.abs(X.T @ Y) / np.sqrt((X * X).mean(axis=0)))) / n
lams = lmax * np.exp(np.linspace(np.log(0.95), np.log(0.03), 60))
# RANDOMIZE=1: shuffled sweeps, the reference's default order (CDOptions.randomize = true, src/utils.jl:18), seeded on both sides
o = dict(maxIter=2000, optTol=1e-7, randomize=bool(int(os.environ.get("RANDOMIZE", "0"))), seed=11)
ref = None
for mode in (1, 3, 0):
    f = cd.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(mode)
    times = []
    for rep in range(3):                      # the first run on a handle fetches the Gram columns (and, on the first handle, loads the kernels)
        t0 = time.perf_counter()
        path = cd.LassoPath(f, None, lams, cd.CDOptions(**o))
        f._L.cdh_synchronize(f._h)
        times.append(time.perf_counter() - t0)
    tg = min(times[1:])
    b = path.betapath[-1].dense()
    if ref is None:
        ref = b
    print("cache mode", mode, "path %.4f s" % tg, "(first run on the handle %.4f s)" % times[0], "nnz", path.betapath[-1].nnz, "max |dbeta| vs mode 1 %.2e" % float(np.max(np.abs(b - ref))),
          f.cache_stats(), f.device_loop_stats(), flush=True)
    f.close()
if os.environ.get("CPU"):
    import oracle as O  # noqa: E402
    t0 = time.perf_counter()
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(**o))
    print("CPU port, one core: %.3f s" % (time.perf_counter() - t0), "max |dbeta| %.2e" % float(np.max(np.abs(bo[-1] - ref))))
