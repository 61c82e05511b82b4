#!/usr/bin/env python3
"""Host-side cost of one bench step at a shard size: x.fill_, initialize_, cdPass_ (wall) against the HIP-event
span of the pass's kernels."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402

n, p, B = int(os.environ.get("ROWS", 1_250_000)), 1000, int(os.environ.get("BLOCK", 64))
f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0)
f.set_sweep_mode("block", B)
x = cd.SparseIterate(p)
cd.initialize_(f, x)
g = cd.ProxL1(1e-6 * cd.findLambdaMax(x, f, cd.ProxL1(1.0)))
visit = np.arange(1, p + 1, dtype=np.int64)
sync = lambda: f._L.cdh_synchronize(f._h)  # noqa: E731
rows = []
for i in range(12):
    sync()
    t0 = time.perf_counter()
    x.fill_(0.0)
    t1 = time.perf_counter()
    cd.initialize_(f, x)          # not waited for: the pass is queued behind it on the same stream
    t2 = time.perf_counter()
    f.profile_begin()
    cd.cdPass_(x, f, g, visit)
    sync()
    t3 = time.perf_counter()
    ev_ms, nl, _ = f.profile_end()
    rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ev_ms))
a = np.median(np.array(rows[2:]), axis=0)
print(f"rows {n} B {B}: fill {a[0]*1e3:.0f} us  initialize {a[1]*1e3:.0f} us  cdPass wall {a[2]*1e3:.0f} us  "
      f"of which kernels (event span) {a[3]*1e3:.0f} us  -> host-side {((a[0]+a[1]+a[2])-a[3])*1e3:.0f} us per step")
