#!/usr/bin/env python3
"""Timings of BASELINE.json's other configurations on ONE GPU (parity-test cases, not bench
lines): cfg3 warm-started lambda path, a 1/8 row shard of cfg4 (sqrt-lasso) and of cfg5
(fp32 weighted-l1 scaled lasso).  Prints one JSON object per configuration."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402


def sync(f):
    f._L.cdh_synchronize(f._h)


def cfg3(n=2_000_000, p=5000, nlam=100):
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0)
    f.set_sweep_mode(os.environ.get("CFG_MODE", "block"), int(os.environ.get("CFG_BLOCK", "16")))
    f.set_use_graph(bool(int(os.environ.get("CFG_GRAPH", "0"))))
    f.set_gradient_cache(int(os.environ.get("CFG_CACHE", "2")))     # a path: what cd.LassoPath asks for
    f.set_screening(int(os.environ.get("CFG_SCREEN", "1")))
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    om = cd.stdX(f)
    lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0, om))
    lams = np.exp(np.linspace(np.log(lmax), np.log(1e-2 * lmax), nlam))
    opt = cd.CDOptions(optTol=1e-7, randomize=False)
    # what cd.LassoPath does around its loop: the carried residual is reused by the warm starts
    cd._lib.check(f._L.cdh_set_reuse_residual(f._h, int(os.environ.get("CFG_REUSE", "1"))), f._h)
    t0 = time.perf_counter()
    passes = visits = 0
    nnz = []
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), opt)
        passes += f.last_stats["passes"]
        visits += f.last_stats["visits"]
        nnz.append(x.nnz)
    sync(f)
    dt = time.perf_counter() - t0
    return dict(config="cfg3 lasso path", n=n, p=p, lambdas=nlam, seconds=dt, passes=passes, visits=visits,
                visits_per_s=visits / dt, nnz_last=nnz[-1], nnz_max=max(nnz), cache=f.cache_stats())


def cfg4_shard(n=5_000_000, p=1000):
    f, _ = cd.CDSqrtLassoLoss.generate(n, p, seed=123, s=100, noise=1.0)
    f.set_sweep_mode("block", int(os.environ.get("CFG_BLOCK", "16")))
    x = cd.SparseIterate(p)
    t0 = time.perf_counter()
    cd.coordinateDescent_(x, f, cd.ProxL1(4.4612 * np.sqrt(1.0)), cd.CDOptions(optTol=1e-8, randomize=False))
    sync(f)
    dt = time.perf_counter() - t0
    st = f.last_stats
    return dict(config="cfg4 sqrt-lasso, 1/8 row shard on one GPU", n=n, p=p, seconds=dt, passes=st["passes"],
                visits=st["visits"], visits_per_s=st["visits"] / dt, nnz=x.nnz, converged=st["converged"])


def cfg5_shard(n=10_000_000, p=2000):
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0, dtype=np.float32)
    f.set_sweep_mode("block", int(os.environ.get("CFG_BLOCK", "16")))
    om = cd.stdX(f)
    x = cd.SparseIterate(p)
    lam = float(np.sqrt(2 * np.log(p) / n))
    t0 = time.perf_counter()
    sol = cd.scaledLasso_(x, f, None, lam, om, cd.IterLassoOptions(optTol=1e-2, optionsCD=cd.CDOptions(randomize=False)))
    sync(f)
    dt = time.perf_counter() - t0
    return dict(config="cfg5 fp32 scaled lasso, 1/8 row shard on one GPU", n=n, p=p, seconds=dt, sigma=sol.sigma,
                nnz=x.nnz, last_solve=f.last_stats)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]
    for w in which:
        fn = {"cfg3": cfg3, "cfg4": cfg4_shard, "cfg5": cfg5_shard}[w]
        print(json.dumps(fn()), flush=True)
