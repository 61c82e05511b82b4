#!/usr/bin/env python3
"""Does a re-read of a just-streamed column block come from the 256 MB Infinity Cache?  Times cdh_gram
(one k_gramstep launch, r only read) over the SAME m columns repeatedly against a rotation over
disjoint column sets, for several block footprints."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402
from coordinatedescent_jl_amd import _lib  # noqa: E402

C = _lib.C
n, p = int(os.environ.get("ROWS", 1_250_000)), 512
f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=10, noise=1.0, dtype=np.float64)
x = cd.SparseIterate(p)
cd.initialize_(f, x)


def gram_ms(m, sets, reps=40):
    G, c, q = np.zeros((m, m)), np.zeros(m), C.c_double()
    idxs = [np.ascontiguousarray(np.arange(s * m, s * m + m) + 1, dtype=np.int64) for s in range(sets)]
    ts = []
    for r in range(reps + 5):
        idx = idxs[r % sets]
        t0 = time.perf_counter()
        _lib.check(f._L.cdh_gram(f._h, m, idx.ctypes.data, G.ctypes.data, c.ctypes.data, C.byref(q)), f._h)
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[5:]))


for m in (4, 8, 16, 32, 64):
    mb = m * n * 8 / 1e6
    same, rot = gram_ms(m, 1), gram_ms(m, p // m)
    print(f"m {m:2d} ({mb:7.1f} MB): same columns {same * 1e3:8.1f} us   rotating {rot * 1e3:8.1f} us   "
          f"-> {mb / same / 1e3:5.2f} vs {mb / rot / 1e3:5.2f} TB/s", flush=True)
