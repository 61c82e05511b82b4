"""First contact of the direct exchange (csrc/p2p_exchange.hpp) with a set of GPUs, in processes of its
own: IPC handles over a private control plane, the mapped-inbox self-test, a few sharded sweeps whose
beta must be bit-identical on every rank, a latency probe.  bench.py runs one of these per rank (same
GPUs, fresh rendezvous port) BEFORE its own process touches the direct exchange, so that whatever a
never-validated transport can do on new hardware -- an IPC mapping error, a peer store that faults, a
hang -- happens here and costs nothing but this probe.  Prints `P2P_PROBE_OK <microseconds>` and exits 0
only if every rank validated every stage.

usage (under the usual RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT environment):
    python p2p_probe.py <device index>
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
    n_local, p = 65536, 128
    f, _ = cd.CDLeastSquaresLoss.generate(n_local, p, seed=7, s=8, noise=1.0, device=device,
                                          n_total=n_local * cp.world, row_offset=cp.rank * n_local)
    ok, lat = False, float("nan")
    try:
        ok = sharded.connect_p2p(f, cp, selftest=True)
    except Exception:
        ok = False
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
    print(("P2P_PROBE_OK %.2f" % lat) if ok_all else "P2P_PROBE_FAILED", flush=True)
    sys.exit(0 if ok_all else 4)


if __name__ == "__main__":
    main()
