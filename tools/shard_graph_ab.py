#!/usr/bin/env python3
"""A/B at row-shard lengths: the all-move sweep (p = 1000) with plain launches and replayed from a captured hipGraph,
interleaved ABBA, B = 32 and 64.  One GPU standing in for one rank (no exchange).  ROWS / BLOCKS as tools/shard_scan.py."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402

p = 1000
rows = [int(v) for v in os.environ.get("ROWS", "1250000,2500000").split(",")]
blocks = [int(v) for v in os.environ.get("BLOCKS", "32,64").split(",")]
for n in rows:
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0, dtype=np.float64)
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    g = cd.ProxL1(1e-6 * cd.findLambdaMax(x, f, cd.ProxL1(1.0)))
    visit = np.arange(1, p + 1, dtype=np.int64)
    for B in blocks:
        f.set_sweep_mode("block", B)
        t = {False: [], True: []}
        for graph in (False, True, True, False, False, True, True, False):
            f.set_use_graph(graph)
            for i in range(6):
                x.fill_(0.0)
                cd.initialize_(f, x)
                f._L.cdh_synchronize(f._h)
                t0 = time.perf_counter()
                cd.cdPass_(x, f, g, visit)
                f._L.cdh_synchronize(f._h)
                if i >= 2:
                    t[graph].append((time.perf_counter() - t0) * 1e3)
        print(f"rows {n:8d} B {B:2d}: plain {np.median(t[False]):7.3f} ms   graph {np.median(t[True]):7.3f} ms per sweep", flush=True)
    f.close()
