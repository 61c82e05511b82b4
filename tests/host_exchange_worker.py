"""Worker of test_gpu_parity.py::test_sharded_launch_sequence_over_host_exchange (run under
torch.distributed.run, every rank on cuda:0).  Each rank holds a row shard of the same problem and
the ONLY exchange is the host-staged one (cdh_set_host_exchange -> a gloo all-reduce): no RCCL (it
refuses two ranks on one device), no IPC inboxes.  So what runs here is the library's own sharded
launch sequence -- k_finalize<false> -> allreduce() -> k_scalar_update, k_block_finalize<B,false> ->
allreduce() -> k_block_scalar, k_gram_reduce -> allreduce() -> k_gram_scalar, the 2p-long column-dot
records of _findLambdaMax / _stdX!, the moments of sigma -- with a real multi-rank sum behind the
seam, checked against the oracle's unsharded solve.  Prints HOSTX_OK from rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import coordinatedescent_jl_amd as cd  # noqa: E402
from coordinatedescent_jl_amd import sharded  # noqa: E402
import oracle as O  # noqa: E402  (the checker)


def main():
    cp = sharded.ControlPlane(backend="gloo")
    assert cp.world >= 2
    n, p = 20011, 70
    rng = np.random.default_rng(23)
    X = np.asfortranarray(rng.standard_normal((n, p)))
    y = X[:, :9] @ (2 * rng.standard_normal(9)) + rng.standard_normal(n)
    w = rng.uniform(0.5, 2.0, size=n)
    om = 0.5 + rng.random(p)
    row0, nl = sharded.shard_rows(n, cp.rank, cp.world)
    rows = slice(row0, row0 + nl)
    o = dict(randomize=False, optTol=1e-11, maxIter=800)

    def shard(cls, *extra):
        f = cls(y[rows], X[rows], *[e[rows] for e in extra], device=0, n_total=n, row_offset=row0)
        sharded.connect_host(f, cp)
        return f

    def same_on_all_ranks(v):
        every = np.frombuffer(cp.all_gather_bytes(np.ascontiguousarray(v).tobytes()), dtype=np.float64).reshape(cp.world, -1)
        return all(np.array_equal(every[0], every[q]) for q in range(1, cp.world))

    cases = [("ls", "coord", 0, 0.05, None), ("ls", "block", 8, 0.05, om), ("ls", "block", 16, 0.05, None),
             ("ls", "block", 32, 0.03, om), ("ls", "block", 64, 0.05, None),
             ("sqrt", "coord", 0, 3.5, om), ("sqrt", "block", 8, 3.5, None), ("sqrt", "block", 32, 3.5, om),
             ("wls", "coord", 0, 0.05, None), ("wls", "block", 16, 0.05, om)]
    for loss, mode, block, lam, omega in cases:
        if loss == "ls":
            f, fo = shard(cd.CDLeastSquaresLoss), O.CDLeastSquaresLoss(y, X)
        elif loss == "sqrt":
            f, fo = shard(cd.CDSqrtLassoLoss), O.CDSqrtLassoLoss(y, X)
        else:
            f, fo = shard(cd.CDWeightedLSLoss, w), O.CDWeightedLSLoss(y, X, w)
        f.set_sweep_mode(mode, block or 8)
        f.set_use_graph(True)       # the host-staged exchange is never recorded: node-by-node launches
        for warm in (True, False):  # the cold start adds _findLambdaMax: a 2p-long record through the seam
            x, xo = cd.SparseIterate(p), O.SparseIterate(p)
            before = f.exchange_stats()["host_calls"]
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, omega), cd.CDOptions(warmStart=warm, **o))
            st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, omega), O.CDOptions(warmStart=warm, **o))
            err = float(np.max(np.abs(x.dense() - xo.dense())))
            assert err < 1e-10, (loss, mode, block, warm, err)
            assert sorted(x.nzval2ind.tolist()) == sorted(xo.nzval2ind.tolist()), (loss, mode, block, warm)
            if mode == "coord":     # same visit order, support order and pass count as the reference state machine
                assert x.nzval2ind.tolist() == xo.nzval2ind.tolist(), (loss, warm)
                assert f.last_stats["passes"] == st["passes"], (loss, warm)
            assert same_on_all_ranks(x.dense()), "ranks disagree"
            es = f.exchange_stats()
            assert es["host_calls"] > before and es["rccl_calls"] == 0 and es["p2p_calls"] == 0 and es["nranks"] == cp.world
        np.testing.assert_allclose(cd.objective(f), O.objective(fo, O.ProxL1(lam, omega), xo), rtol=1e-12)
        np.testing.assert_allclose(f.r, fo.r[rows], rtol=0, atol=1e-9)       # each rank holds its own residual rows
        np.testing.assert_allclose(cd.stdX(f), O.stdX(X), rtol=1e-13)        # _stdX! over all shards
        cp.barrier()
        f.close()

    # a warm-started lambda path on shards with the gradient cache from the first full pass: the dots-only
    # reference pass and the Gram-column batches (80 K cross products per all-reduce) go through the seam too
    n2, p2 = 6000, 520
    X2 = np.asfortranarray(rng.standard_normal((n2, p2)))
    y2 = X2[:, :9] @ (2 * rng.standard_normal(9)) + rng.standard_normal(n2)
    r0, nl2 = sharded.shard_rows(n2, cp.rank, cp.world)
    f = cd.CDLeastSquaresLoss(y2[r0:r0 + nl2], X2[r0:r0 + nl2], device=0, n_total=n2, row_offset=r0)
    sharded.connect_host(f, cp)
    f.set_gradient_cache(3)
    fo = O.CDLeastSquaresLoss(y2, X2)
    x, xo = cd.SparseIterate(p2), O.SparseIterate(p2)
    for lam in (0.4, 0.2, 0.1, 0.05):
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        assert float(np.max(np.abs(x.dense() - xo.dense()))) < 1e-10, lam
        assert f.last_stats["passes"] == st["passes"] and sorted(x.nzval2ind.tolist()) == sorted(xo.nzval2ind.tolist())
        assert same_on_all_ranks(x.dense())
    cs = f.cache_stats()
    assert cs["passes"] >= 4 and cs["settled_visits"] > cs["exact_visits"] > 0 and cs["gram_columns"] >= xo.nnz, cs
    cp.barrier()
    f.close()

    # scaledLasso! on shards: the :Screening init (X'y scores, the s x s normal equations through cdh_gram,
    # std of the OLS residuals) and the sigma loop all go through the seam
    f = shard(cd.CDLeastSquaresLoss)
    f.set_sweep_mode("block", 32)
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    io = dict(maxIter=50, optTol=1e-6)
    sol = cd.scaledLasso_(x, f, None, 0.08, om, cd.IterLassoOptions(optionsCD=cd.CDOptions(**o), **io))
    so = O.scaledLasso_(xo, X, y, 0.08, om, O.IterLassoOptions(optionsCD=O.CDOptions(**o), **io))
    np.testing.assert_allclose(sol.sigma, so.sigma, rtol=1e-9)
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-9)
    assert same_on_all_ranks(x.dense())
    cp.barrier()
    # taking the transport away from a shard of a multi-rank problem must not leave it sweeping its local rows
    # as if they were the whole problem: every later exchange refuses
    f.set_host_exchange(None, 0, 1)
    try:
        cd.coordinateDescent_(x, f, cd.ProxL1(0.08, om), cd.CDOptions(**o))
        raise AssertionError("a shard without its exchange kept sweeping")
    except cd.HipError as e:
        assert "exchange" in str(e)
    cp.barrier()
    # ... and an exchange installed AFTER the loss puts the shard back to work (only the host-exchange install
    # used to clear the refusal; here the direct exchange replaces the transport that was taken away)
    assert sharded.connect_p2p(f, cp, selftest=True), "p2p self-test failed"
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.08, om), cd.CDOptions(**o))
    O.coordinateDescent_(xo, O.CDLeastSquaresLoss(y, X), O.ProxL1(0.08, om), O.CDOptions(**o))
    assert float(np.max(np.abs(x.dense() - xo.dense()))) < 1e-10 and same_on_all_ranks(x.dense())
    assert f.exchange_stats()["p2p_calls"] > 0
    cp.barrier()
    f.close()
    # Near-equal shards on opposite sides of the default-width cut (cdh_create: B = 64 below 262 144 local rows, else 32):
    # 524 287 rows over two ranks are 262 144 + 262 143.  Nobody sets a width here -- the handles agree on one through the
    # exchange before their first streamed chunk; ranks with different B would issue different numbers and sizes of
    # all-reduces per pass (a hang, or wrong sums).
    if cp.world == 2:
        n3, p3 = 524287, 80
        rng3 = np.random.default_rng(29)
        X3 = np.asfortranarray(rng3.standard_normal((n3, p3)))
        y3 = X3[:, :6] @ (2 * rng3.standard_normal(6)) + rng3.standard_normal(n3)
        r3, nl3 = sharded.shard_rows(n3, cp.rank, cp.world)
        assert nl3 == (262144 if cp.rank == 0 else 262143)
        f = cd.CDLeastSquaresLoss(y3[r3:r3 + nl3], X3[r3:r3 + nl3], device=0, n_total=n3, row_offset=r3)
        sharded.connect_host(f, cp)
        f.set_gradient_cache(0)
        fo = O.CDLeastSquaresLoss(y3, X3)
        x, xo = cd.SparseIterate(p3), O.SparseIterate(p3)
        before = f.exchange_stats()["host_calls"]
        cd.coordinateDescent_(x, f, cd.ProxL1(0.05), cd.CDOptions(**o))
        O.coordinateDescent_(xo, fo, O.ProxL1(0.05), O.CDOptions(**o))
        assert float(np.max(np.abs(x.dense() - xo.dense()))) < 1e-10
        assert same_on_all_ranks(x.dense())
        calls = float(f.exchange_stats()["host_calls"] - before)
        assert same_on_all_ranks(np.array([calls])), "ranks issued different numbers of all-reduces"
        cp.barrier()
        f.close()
    if cp.rank == 0:
        print("HOSTX_OK")
    cp.shutdown()


if __name__ == "__main__":
    main()
