#!/usr/bin/env python3
"""The reference's own benchmark workload (benchmark/cd_bench.jl: n = 3000, p = 5000, s = 100, noise 6) on the GPU path and
on the CPU port of the oracle (one core), piece by piece: scaledLasso! (sigma loop, optTol 1e-3), coordinateDescent! at
lambda = 0.001 warm from zero (a DENSE solution: thousands of non-zeros), and the same cold (51 continuation solves).
Environment: CFG_CACHE (gradient-cache mode of the GPU handle, default 1), MAXITER (default 200)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402
import oracle as O  # noqa: E402

rng = np.random.default_rng(123)
n, p, s = 3000, 5000, 100
X = np.asfortranarray(rng.standard_normal((n, p)))
Y = X[:, :s] @ (rng.standard_normal(s) * (1.0 + rng.random(s))) + 6.0 * rng.standard_normal(n)
lam = float(np.sqrt(2.0 * np.log(p) / n))
maxit = int(os.environ.get("MAXITER", "200"))
out = {"cache_mode": int(os.environ.get("CFG_CACHE", "1"))}

f = cd.CDLeastSquaresLoss(Y, X)
f.set_gradient_cache(out["cache_mode"])
if os.environ.get("CFG_GRAPH"):
    f.set_use_graph(True)
    out["graph"] = True
if os.environ.get("CFG_BLOCK"):
    f.set_sweep_mode("block", int(os.environ["CFG_BLOCK"]))
    out["block"] = int(os.environ["CFG_BLOCK"])
if os.environ.get("ONLY"):
    out["only"] = os.environ["ONLY"]
sx = cd.stdX(f)
fo = O.CDLeastSquaresLoss(Y, X)


def timed(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        f._L.cdh_synchronize(f._h)
        best = min(best, time.perf_counter() - t0)
    return best, r


cdo = dict(randomize=False, optTol=1e-7)
t, sol = timed(lambda: cd.scaledLasso_(cd.SparseIterate(p), f, None, lam, sx, cd.IterLassoOptions(optTol=1e-3, maxIter=50, optionsCD=cd.CDOptions(**cdo))))
t0 = time.perf_counter()
so = O.scaledLasso_(O.SparseIterate(p), X, Y, lam, sx, O.IterLassoOptions(optTol=1e-3, maxIter=50, optionsCD=O.CDOptions(**cdo)))
tc = time.perf_counter() - t0
out["scaledLasso"] = {"gpu_s": t, "cpu_port_s": tc, "sigma_gpu": sol.sigma, "sigma_cpu": so.sigma, "nnz": int(sol.x.nnz),
                      "max_abs_dbeta": float(np.max(np.abs(sol.x.dense() - so.x.dense()))), "cache": f.cache_stats()}
for name, warm in (("cd_lambda0.001_warm", True), ("cd_lambda0.001_cold", False)):
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    o = dict(randomize=False, optTol=1e-7, maxIter=maxit, warmStart=warm)
    def run():
        x = cd.SparseIterate(p)
        cd.coordinateDescent_(x, f, cd.ProxL1(0.001), cd.CDOptions(**o))
        return x
    t, x = timed(run, reps=2)
    st = dict(f.last_stats)
    xo = O.SparseIterate(p)
    t0 = time.perf_counter()
    sto = O.coordinateDescent_(xo, fo, O.ProxL1(0.001), O.CDOptions(**o))
    tc = time.perf_counter() - t0
    out[name] = {"gpu_s": t, "cpu_port_s": tc, "passes_gpu": st["passes"], "passes_cpu": sto["passes"], "visits": st["visits"],
                 "nnz": int(x.nnz), "max_abs_dbeta": float(np.max(np.abs(x.dense() - xo.dense())))}
print(json.dumps(out))
