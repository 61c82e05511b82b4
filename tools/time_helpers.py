#!/usr/bin/env python3
"""Times the one-off helper passes over X at cfg2 size: lambda_max / stdX (k_col_dots),
initialize! with a 100-column support (k_init_resid), the generator."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd
n, p = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("COLS", 1000))
t0 = time.perf_counter(); f, b = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0); t_gen = time.perf_counter() - t0
x = cd.SparseIterate(p); cd.initialize_(f, x)
for name, fn in [("lambda_max", lambda: cd.findLambdaMax(x, f, cd.ProxL1(1.0))), ("stdX", lambda: cd.stdX(f))]:
    fn(); t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name}: {dt*1e3:.2f} ms -> {n*p*8/dt/1e12:.2f} TB/s")
xs = cd.SparseIterate(p, np.concatenate([b, np.zeros(p - len(b))]))
cd.initialize_(f, xs); t0 = time.perf_counter(); cd.initialize_(f, xs); dt = time.perf_counter() - t0
print(f"initialize!(nnz=100): {dt*1e3:.2f} ms -> {n*102*8/dt/1e12:.2f} TB/s"); print(f"generate: {t_gen*1e3:.1f} ms")
