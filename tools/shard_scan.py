#!/usr/bin/env python3
"""Sweep time of the blocked kernels against shard length (strong-scaling regime): for each rows-per-GPU
and block size, ms per all-move sweep, us per launch of the column-streaming kernel (HIP events) and
the fixed + per-row decomposition of that launch time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402

p = 1000
rows = [int(v) for v in os.environ.get("ROWS", "156250,312500,625000,1250000,2500000,5000000").split(",")]
blocks = [int(v) for v in os.environ.get("BLOCKS", "16,32,64").split(",")]
res = {}
for n in rows:
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0, dtype=np.float64)
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    g = cd.ProxL1(1e-6 * cd.findLambdaMax(x, f, cd.ProxL1(1.0)))
    visit = list(range(1, p + 1))
    for B in blocks:
        f.set_sweep_mode("block", B)
        ts, ev = [], []
        for i in range(7):
            x.fill_(0.0)
            cd.initialize_(f, x)
            f._L.cdh_synchronize(f._h)
            f.profile_begin()
            t0 = time.perf_counter()
            cd.cdPass_(x, f, g, visit)
            f._L.cdh_synchronize(f._h)
            ts.append((time.perf_counter() - t0) * 1e3)
            ms, nl, by = f.profile_end()
            ev.append(ms * 1e3 / max(nl, 1))
        res[(n, B)] = (float(np.median(ts[2:])), float(np.median(ev[2:])))
        print(f"rows {n:8d} B {B:2d}: sweep {res[(n, B)][0]:7.3f} ms  stream-kernel {res[(n, B)][1]:8.1f} us/launch "
              f"({res[(n, B)][1] / B:6.2f} us/visit)  alg {2 * n * 8 * B * (1 + 1 / B) / res[(n, B)][1] / 1e6:5.2f} TB/s",
              flush=True)
    f.close()
for B in blocks:
    xs = np.array([n for n in rows], dtype=float)
    ys = np.array([res[(n, B)][1] for n in rows])
    A = np.vstack([np.ones_like(xs), xs]).T
    (c0, c1), *_ = np.linalg.lstsq(A, ys, rcond=None)
    print(f"B {B}: launch us ~= {c0:.1f} + {c1 * 1e6:.1f} per 1e6 rows  (asymptotic {2 * 8 * B * (1 + 1 / B) / c1 / 1e6:.2f} TB/s)")
