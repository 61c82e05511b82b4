"""Worker of test_gpu_parity.py::test_p2p_exchange_two_ranks_one_gpu (run under
torch.distributed.run, 2 ranks, BOTH on cuda:0).  Each rank holds a row shard; the only exchange
is the opt-in direct exchange of csrc/p2p_exchange.hpp (RCCL refuses two ranks on one device, so
no communicator exists and longer vectors are chunked through the same inboxes).  Checks: the
transport self-test, beta against the oracle's unsharded solve (1e-10), and bit-identical beta on
the two ranks.  Prints P2P_OK from rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import coordinatedescent_jl_amd as cd  # noqa: E402
from coordinatedescent_jl_amd import sharded  # noqa: E402
import oracle  # noqa: E402  (the checker)


def main():
    cp = sharded.ControlPlane(backend="gloo")
    assert cp.world in (2, 3, 4)
    n, p = 60000, 150
    rng = np.random.default_rng(17)
    X = rng.standard_normal((n, p))
    beta0 = np.zeros(p)
    beta0[:12] = rng.standard_normal(12) * 2
    y = X @ beta0 + rng.standard_normal(n)
    row0, nl = sharded.shard_rows(n, cp.rank, cp.world)
    lam = 0.05
    opt = cd.CDOptions(randomize=False, optTol=1e-11, maxIter=500)
    xo = oracle.SparseIterate(p)
    fo = oracle.CDLeastSquaresLoss(y, X)
    oracle.coordinateDescent_(xo, fo, oracle.ProxL1(lam), oracle.CDOptions(randomize=False, optTol=1e-11, maxIter=500))
    want = xo.dense()
    for mode, block in (("coord", 0), ("block", 16), ("block", 64)):
        f = cd.CDLeastSquaresLoss(y[row0:row0 + nl], X[row0:row0 + nl], device=0, n_total=n, row_offset=row0)
        assert sharded.connect_p2p(f, cp, selftest=True), "p2p self-test failed"
        f.set_gradient_cache(0)     # the graph checks below compare bit for bit: both runs must take the same path
        if mode == "block":
            f.set_sweep_mode("block", block)
        else:
            f.set_sweep_mode("coord")
        x = cd.SparseIterate(p)
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), opt)
        got = x.dense()
        err = float(np.max(np.abs(got - want)))
        assert err < 1e-10, (mode, block, err)
        every = np.frombuffer(cp.all_gather_bytes(got.tobytes()), dtype=np.float64).reshape(cp.world, p)
        assert all(np.array_equal(every[0], every[q]) for q in range(1, cp.world)), "ranks disagree"
        lm = cd.findLambdaMax(cd.SparseIterate(p), f, cd.ProxL1(1.0))
        lo = oracle.findLambdaMax(oracle.SparseIterate(p), fo, oracle.ProxL1(1.0))
        assert abs(lm - lo) < 1e-12 * max(1.0, lo), (lm, lo)
        # the same solve replayed from captured hipGraphs: the exchange kernels are graph nodes that take
        # their epoch from device memory (set before each replay) -- bit-identical to node-by-node launches
        f.set_use_graph(True)
        xg = cd.SparseIterate(p)
        calls0 = f.exchange_stats()["p2p_calls"]
        cd.coordinateDescent_(xg, f, cd.ProxL1(lam), opt)
        assert np.array_equal(xg.dense(), got), ("graph replay differs", mode, block)
        assert f.exchange_stats()["p2p_calls"] > calls0
        xg = cd.SparseIterate(p)
        cd.coordinateDescent_(xg, f, cd.ProxL1(0.5 * lam), opt)      # warm graphs of the same lengths, new epochs
        f.set_use_graph(False)
        xn = cd.SparseIterate(p)
        cd.coordinateDescent_(xn, f, cd.ProxL1(0.5 * lam), opt)
        assert np.array_equal(xg.dense(), xn.dense()), ("graph replay differs at the second lambda", mode, block)
        # longer than one inbox slot and no communicator: chunked through the same inboxes
        v = np.arange(4096, dtype=np.float64) * (1 + cp.rank)
        assert np.array_equal(f.exchange_probe(v), (cp.world * (cp.world + 1) / 2) * np.arange(4096, dtype=np.float64))
        lat = f.exchange_latency(2625, 100)          # collective; informational
        if cp.rank == 0:
            print(f"P2P_LATENCY world={cp.world} mode={mode}{block} us={lat:.1f}")
        cp.barrier()
        del f
    # sqrt-lasso with penalty weights on shards (the exchange then carries q = r'r as well)
    om = 0.5 + rng.random(p)
    xo = oracle.SparseIterate(p)
    fo2 = oracle.CDSqrtLassoLoss(y, X)
    oracle.coordinateDescent_(xo, fo2, oracle.ProxL1(3.8, om), oracle.CDOptions(randomize=False, optTol=1e-11, maxIter=500))
    for block in (0, 32):
        f = cd.CDSqrtLassoLoss(y[row0:row0 + nl], X[row0:row0 + nl], device=0, n_total=n, row_offset=row0)
        assert sharded.connect_p2p(f, cp, selftest=False)
        f.set_sweep_mode("block" if block else "coord", block or 8)
        x = cd.SparseIterate(p)
        cd.coordinateDescent_(x, f, cd.ProxL1(3.8, om), opt)
        err = float(np.max(np.abs(x.dense() - xo.dense())))
        assert err < 1e-10, ("sqrt", block, err)
        assert abs(cd.objective(f) - oracle.objective(fo2, oracle.ProxL1(3.8, om), xo)) <= 1e-12 * abs(cd.objective(f))
        cp.barrier()
        del f
    # cfg5-shaped on shards: fp32 storage, omega = _stdX!, scaledLasso! with the :Screening init, B = 32, every
    # pass replayed from a hipGraph whose exchange nodes read their epoch from device memory -- bit-identical to
    # node-by-node launches, and within fp32 storage tolerance of the fp64 oracle
    lam5 = float(np.sqrt(2 * np.log(p) / n)) * 2
    cdo = dict(randomize=False, optTol=1e-7, maxIter=2000)
    outs = []
    for graph in (False, True):
        f = cd.CDLeastSquaresLoss(y[row0:row0 + nl].astype(np.float32), X[row0:row0 + nl].astype(np.float32),
                                  device=0, n_total=n, row_offset=row0)
        assert sharded.connect_p2p(f, cp, selftest=False)
        f.set_sweep_mode("block", 32)
        f.set_use_graph(graph)
        om5 = cd.stdX(f)
        x = cd.SparseIterate(p)
        sol = cd.scaledLasso_(x, f, None, lam5, om5, cd.IterLassoOptions(maxIter=50, optTol=1e-6, optionsCD=cd.CDOptions(**cdo)))
        outs.append((x.dense(), sol.sigma, f.r))
        cp.barrier()
        del f
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][2], outs[1][2]) and outs[0][1] == outs[1][1]
    xo = oracle.SparseIterate(p)
    so = oracle.scaledLasso_(xo, X, y, lam5, oracle.stdX(X),
                             oracle.IterLassoOptions(maxIter=50, optTol=1e-6, optionsCD=oracle.CDOptions(**cdo)))
    assert abs(outs[1][1] - so.sigma) / so.sigma < 1e-4, (outs[1][1], so.sigma)
    assert float(np.max(np.abs(outs[1][0] - xo.dense()))) < 3e-4
    # a peer that never arrives: the wait is bounded, the call fails, and a shard whose only exchange
    # is gone refuses to continue (no silent unsharded arithmetic)
    os.environ["CDH_P2P_SPIN_LIMIT"] = "200000"
    f = cd.CDLeastSquaresLoss(y[row0:row0 + nl], X[row0:row0 + nl], device=0, n_total=n, row_offset=row0)
    assert sharded.connect_p2p(f, cp, selftest=True)
    if cp.rank == 0:
        for attempt in range(2):
            try:
                f.exchange_probe(np.ones(8))
                raise AssertionError("a lone rank's exchange must time out")
            except cd._lib.HipError as e:
                assert ("timed out" if attempt == 0 else "lost its exchange") in str(e), str(e)
        # ... in EVERY sweep mode: the per-coordinate and narrow-block sweeps must not fall into the
        # single-process kernels on their local rows and return numbers
        for mode, block in (("coord", 8), ("block", 8), ("block", 32)):
            f.set_sweep_mode(mode, block)
            try:
                cd.cdPass_(cd.SparseIterate(p), f, cd.ProxL1(0.1), [1, 2, 3])
                raise AssertionError(f"a shard without its exchange swept anyway ({mode}{block})")
            except cd._lib.HipError as e:
                assert "lost its exchange" in str(e), str(e)
    cp.barrier()
    del f
    if cp.rank == 0:
        print("P2P_OK")
    cp.shutdown()


if __name__ == "__main__":
    main()
