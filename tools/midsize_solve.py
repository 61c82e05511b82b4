#!/usr/bin/env python3
"""Solves on problems between the reference's own shapes and BASELINE.json's: p <= 1024 columns, X from tens of MB to GB.
For each shape: a warm solve from zero, a cold start (51 continuation solves) and a 20-lambda warm-started path, on the
streamed kernels (CDH_SMALL_MAX_BYTES=0: the Gram form never built) and with the Gram form under the cost comparison of
small_worth_building (csrc/small_solve.hpp).  Same iterates either way (max |d beta| printed).  One JSON line per shape."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402

SHAPES = [(50_000, 200), (100_000, 500), (50_000, 1000), (400_000, 1000), (1_000_000, 100), (2_000_000, 200), (4_000_000, 500)]


def run(n, p, gram):
    if gram:
        os.environ.pop("CDH_SMALL_MAX_BYTES", None)
    else:
        os.environ["CDH_SMALL_MAX_BYTES"] = "0"
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=min(20, p // 4), noise=3.0)
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
    sync = lambda: f._L.cdh_synchronize(f._h)  # noqa: E731
    out = {}
    o = dict(maxIter=2000, optTol=1e-8, randomize=False)
    for name, fn in (
        ("first_warm_solve", lambda: cd.coordinateDescent_(cd.SparseIterate(p), f, cd.ProxL1(0.1 * lmax), cd.CDOptions(**o))),
        ("second_warm_solve", lambda: cd.coordinateDescent_(cd.SparseIterate(p), f, cd.ProxL1(0.05 * lmax), cd.CDOptions(**o))),
        ("cold_start_51", lambda: cd.coordinateDescent_(cd.SparseIterate(p), f, cd.ProxL1(0.02 * lmax), cd.CDOptions(warmStart=False, **o))),
    ):
        sync()
        t0 = time.perf_counter()
        xs = fn()
        sync()
        out[name] = {"s": time.perf_counter() - t0, "passes": f.last_stats["passes"], "nnz": int(xs.nnz)}
        out[name + "_beta"] = xs.dense()
    xp = cd.SparseIterate(p)
    lams = np.exp(np.linspace(np.log(0.9 * lmax), np.log(0.02 * lmax), 20))
    sync()
    t0 = time.perf_counter()
    for lam in lams:
        cd.coordinateDescent_(xp, f, cd.ProxL1(lam), cd.CDOptions(**o))
    sync()
    out["path_20"] = {"s": time.perf_counter() - t0, "nnz": int(xp.nnz)}
    out["path_20_beta"] = xp.dense()
    out["onchip"] = f.onchip_stats()
    f.close()
    return out


if __name__ == "__main__":
    shapes = SHAPES if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for n, p in shapes:
        a, b = run(n, p, False), run(n, p, True)
        line = {"n": n, "p": p, "X_MB": n * p * 8 / 1e6, "gram_built": b["onchip"]["gram_matrices"]}
        for k in ("first_warm_solve", "second_warm_solve", "cold_start_51", "path_20"):
            line[k] = {"streamed_s": round(a[k]["s"], 6), "gram_s": round(b[k]["s"], 6), "passes": a[k].get("passes"),
                       "nnz": a[k]["nnz"], "max_abs_dbeta": float(np.max(np.abs(a[k + "_beta"] - b[k + "_beta"])))}
        print(json.dumps(line), flush=True)
