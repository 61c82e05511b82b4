"""Per-exchange overhead on ONE GPU: the B=64 blocked sweep of a 1.25M-row shard with (a) no exchange,
(b) a forced 1-rank RCCL communicator, (c) the opt-in direct exchange connected to itself.  What this
measures is the launch-level cost each exchange adds to the dependent kernel chain (no wire time)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import coordinatedescent_jl_amd as cd  # noqa: E402
from coordinatedescent_jl_amd import _lib  # noqa: E402

n, p, B = int(os.environ.get("ROWS", 1_250_000)), 1000, int(os.environ.get("BLOCK", 64))
C = _lib.C


def sweep_ms(kind, reps=8):
    os.environ.pop("CDH_FORCE_RCCL", None)
    f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=100, noise=6.0, dtype=np.float64, device=0,
                                          n_total=n, row_offset=0)
    if kind == "rccl":
        os.environ["CDH_FORCE_RCCL"] = "1"
        buf = C.create_string_buffer(128)
        _lib.check(f._L.cdh_comm_unique_id(buf), None)
        f.comm_init(buf.raw, 0, 1)
    elif kind == "p2p":
        f.p2p_connect(f.p2p_local_handle(), 0, 1)
        f.p2p_enable(True)
    f.set_sweep_mode("block", B)
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    g = cd.ProxL1(1e-6 * cd.findLambdaMax(x, f, cd.ProxL1(1.0)))
    visit = list(range(1, p + 1))
    out = []
    for r in range(reps + 2):
        x.fill_(0.0)
        cd.initialize_(f, x)
        f._L.cdh_synchronize(f._h)
        t = time.perf_counter()
        cd.cdPass_(x, f, g, visit)
        f._L.cdh_synchronize(f._h)
        out.append((time.perf_counter() - t) * 1e3)
    f.close()
    return float(np.median(out[2:]))


if __name__ == "__main__":
    res = {}
    for rnd in range(2):                       # ABBA-style: two rounds, order reversed
        order = ["none", "rccl", "p2p"] if rnd == 0 else ["p2p", "rccl", "none"]
        for k in order:
            res.setdefault(k, []).append(sweep_ms(k))
    nex = (p + B - 1) // B
    for k, v in res.items():
        print(f"{k:5s} sweep ms {v}  mean {np.mean(v):.3f}")
    base = np.mean(res["none"])
    for k in ("rccl", "p2p"):
        print(f"{k}: +{(np.mean(res[k]) - base) / nex * 1e3:.1f} us per exchange ({nex} exchanges/sweep)")
