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
                visits_per_s=visits / dt, nnz_last=nnz[-1], nnz_max=max(nnz), cache=f.cache_stats(), device_loop=f.device_loop_stats(),
                drift=f.cache_drift(rereference_now=True), beta_checksum=float(np.sum(np.abs(x.dense()))))


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
    f.set_gradient_cache(int(os.environ.get("CFG_CACHE", "1")))
    om = cd.stdX(f)
    x = cd.SparseIterate(p)
    lam = float(np.sqrt(2 * np.log(p) / n))
    tol = float(os.environ.get("CFG_SIGMA_TOL", "1e-2"))              # IterLassoOptions' default; 1e-6 runs the loop to its fixed point
    t0 = time.perf_counter()
    sol = cd.scaledLasso_(x, f, None, lam, om, cd.IterLassoOptions(maxIter=50, optTol=tol, optionsCD=cd.CDOptions(randomize=False)))
    sync(f)
    dt = time.perf_counter() - t0
    return dict(config="cfg5 fp32 scaled lasso, 1/8 row shard on one GPU", n=n, p=p, seconds=dt, sigma=sol.sigma,
                sigma_tol=tol, nnz=x.nnz, last_solve=f.last_stats, cache_mode=int(os.environ.get("CFG_CACHE", "1")),
                cache=f.cache_stats(), beta_checksum=float(np.sum(np.abs(x.dense()))))


def wls_path(n=4_000_000, p=2000, nlam=30):
    """CDWeightedLSLoss (cd_differentiable_function.jl:118-194) on a warm-started path: observation weights in [0.5, 1.5)."""
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=50, noise=4.0)
    cd._lib.check(f._L.cdh_set_loss(f._h, cd.CDH_WLS), f._h)
    w = 0.5 + np.random.default_rng(3).random(n)
    cd._lib.check(f._L.cdh_set_obs_weights(f._h, w.ctypes.data), f._h)
    f.set_sweep_mode("block", int(os.environ.get("CFG_BLOCK", "32")))
    f.set_gradient_cache(int(os.environ.get("CFG_CACHE", "2")))
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
    lams = np.exp(np.linspace(np.log(lmax), np.log(3e-2 * lmax), nlam))
    opt = cd.CDOptions(optTol=1e-7, randomize=False)
    cd._lib.check(f._L.cdh_set_reuse_residual(f._h, 1), f._h)
    t0 = time.perf_counter()
    passes = visits = 0
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), opt)
        passes += f.last_stats["passes"]
        visits += f.last_stats["visits"]
    sync(f)
    dt = time.perf_counter() - t0
    return dict(config="weighted-LS loss, warm-started path", n=n, p=p, lambdas=nlam, seconds=dt, passes=passes, visits=visits,
                nnz_last=x.nnz, cache_mode=int(os.environ.get("CFG_CACHE", "2")), cache=f.cache_stats(), device_loop=f.device_loop_stats(),
                beta_checksum=float(np.sum(np.abs(x.dense()))))


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg4", "cfg5"]
    for w in which:
        fn = {"cfg3": cfg3, "cfg4": cfg4_shard, "cfg5": cfg5_shard, "wls": wls_path}[w]
        print(json.dumps(fn()), flush=True)
