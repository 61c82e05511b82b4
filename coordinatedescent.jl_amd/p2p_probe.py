"""First contact of an exchange with a set of GPUs, in processes of its own.  Two modes.

`rccl`: the communicator of the row-sharded sweep -- unique id over a private control plane, ncclCommInitRank, three probe
records whose exact sums every rank checks (sharded.connect_checked), then the sweeps and the latency probe below.  bench.py
runs it before building its own communicator: RCCL has never run across GPUs in this pipeline, and a bring-up that never
returns would otherwise take the whole run with it -- here it is waited for with a timeout.  Prints `RCCL_PROBE_OK <us>` or
`RCCL_PROBE_FAILED <why>`.

`p2p` (default): the direct exchange (csrc/p2p_exchange.hpp) -- IPC handles over a private control plane, the mapped-inbox self-test, a few sharded sweeps whose
beta must be bit-identical on every rank, a latency probe.  bench.py runs one of these per rank (same
GPUs, fresh rendezvous port) BEFORE its own process touches the direct exchange, so that whatever a
never-validated transport can do on new hardware -- an IPC mapping error, a peer store that faults, a
hang -- happens here and costs nothing but this probe.  Prints `P2P_PROBE_OK <microseconds>` and exits 0
only if every rank validated every stage.

usage (under the usual RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT environment):
    python p2p_probe.py <device index> [p2p|rccl]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    if os.environ.get("CDH_P2P_PROBE_ABORT"):      # TEST ONLY: stands in for a probe that dies (a GPU fault aborts the process)
        os._exit(3)
    import numpy as np
    import coordinatedescent_jl_amd as cd
    from importlib import import_module
    sharded = import_module("coordinatedescent_jl_amd.sharded")
    cp = sharded.ControlPlane(backend="gloo")
    device = int(sys.argv[1]) if len(sys.argv) > 1 else cp.local_rank
    what = sys.argv[2] if len(sys.argv) > 2 else "p2p"
    tag = "RCCL_PROBE" if what == "rccl" else "P2P_PROBE"
    n_local, p = 65536, 128
    f, _ = cd.CDLeastSquaresLoss.generate(n_local, p, seed=7, s=8, noise=1.0, device=device,
                                          n_total=n_local * cp.world, row_offset=cp.rank * n_local)
    ok, lat, why = False, float("nan"), ""
    try:
        if what == "rccl":
            if os.environ.get("CDH_RCCL_PROBE_HANG"):  # TEST ONLY: stands in for a communicator whose construction never returns
                import time
                time.sleep(3600)
            ok, why = sharded.connect_checked(f, cp)
        else:
            ok = sharded.connect_p2p(f, cp, selftest=True)
    except Exception as e:
        ok, why = False, str(e)[:200]
    if cp.sum_over_ranks(1.0 if ok else 0.0) == cp.world:
        try:
            f.set_sweep_mode("block", 64)
            x = cd.SparseIterate(p)
            cd.initialize_(f, x)
            lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))         # a 2p-long record through the inboxes
            visit = np.arange(1, p + 1, dtype=np.int64)
            for _ in range(3):
                x.fill_(0.0)
                cd.initialize_(f, x)
                cd.cdPass_(x, f, cd.ProxL1(1e-3 * lmax), visit)
            beta = x.dense()
            lat = f.exchange_latency(2625, 100)
        except Exception:
            ok, beta = False, np.zeros(p)
        every = np.frombuffer(cp.all_gather_bytes(beta.tobytes()), dtype=np.float64).reshape(cp.world, p)
        ok = ok and bool(np.any(beta != 0.0)) and all(np.array_equal(every[0], every[q]) for q in range(1, cp.world))
    else:
        ok = False
    ok_all = cp.sum_over_ranks(1.0 if ok else 0.0) == cp.world
    try:
        f.close()
        cp.shutdown()
    except Exception:
        pass
    print((tag + "_OK %.2f" % lat) if ok_all else (tag + "_FAILED " + (why or "a later stage")).rstrip(), flush=True)
    sys.exit(0 if ok_all else 4)


if __name__ == "__main__":
    main()
