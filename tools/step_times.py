#!/usr/bin/env python3
"""Per-sweep wall times inside one process (is run-to-run spread a per-process or a per-sweep effect?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd
n, p, B = 10_000_000, 1000, int(os.environ.get("B", "16"))
f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0)
f.set_sweep_mode("block", B)
x = cd.SparseIterate(p); cd.initialize_(f, x)
g = cd.ProxL1(1e-6 * cd.findLambdaMax(x, f, cd.ProxL1(1.0)))
ts = []
for i in range(int(os.environ.get("STEPS", "12"))):
    x.fill_(0.0); cd.initialize_(f, x)
    t0 = time.perf_counter(); cd.cdPass_(x, f, g, range(1, p + 1)); ts.append((time.perf_counter() - t0) * 1e3)
print("B=%d per-sweep ms:" % B, " ".join("%.2f" % t for t in ts))
