"""N > 1 path on CPU: two processes over gloo.

What can run without a GPU is the control plane (coordinatedescent.jl_amd/sharded.py:
row partition, rendezvous, byte broadcast of the RCCL unique id, max/sum over ranks) and
the ARITHMETIC of row sharding: every rank reduces its own rows to the per-coordinate sums
(a, b, q), the sums are all-reduced, every rank applies the same scalar update and updates
its own residual rows -- exactly what csrc/cdhip.hip does with RCCL between k_step /
k_finalize and k_scalar_update.  Here the per-rank sums come from numpy (standing in for
the HIP kernels, test infrastructure only) and the exchange is a real gloo all-reduce; the
result must equal the unsharded oracle solve, and all ranks must hold bit-identical beta.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, p, lam, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from importlib import import_module
    import coordinatedescent_jl_amd  # noqa: F401
    sharded = import_module("coordinatedescent_jl_amd.sharded")
    cp = sharded.ControlPlane(backend="gloo")
    assert (cp.rank, cp.world) == (rank, world)
    # byte broadcast (the RCCL unique id travels this way)
    blob = bytes(range(128)) if rank == 0 else None
    assert cp.broadcast_bytes(blob, 128) == bytes(range(128))
    assert cp.max_over_ranks(float(rank)) == world - 1
    assert cp.sum_over_ranks(1.0) == world
    # rank-ordered gather (the IPC handles of the opt-in direct exchange travel this way)
    assert cp.all_gather_bytes(bytes([rank]) * 64) == b"".join(bytes([q]) * 64 for q in range(world))
    # row-sharded coordinate descent with a real all-reduce per coordinate
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, p))
    y = X[:, :4] @ rng.standard_normal(4) + rng.standard_normal(n)
    row0, nl = sharded.shard_rows(n, rank, world)
    Xl, rl = X[row0:row0 + nl], y[row0:row0 + nl].copy()
    beta = np.zeros(p)
    for _ in range(60):
        for k in range(p):
            sums = torch.tensor([Xl[:, k] @ Xl[:, k], Xl[:, k] @ rl])
            dist.all_reduce(sums)                      # the path's one exchange step
            a, b = sums.tolist()
            v = beta[k] + b / a
            t = lam * n / a
            new = v - t if v > t else (v + t if v < -t else 0.0)
            rl -= Xl[:, k] * (new - beta[k])
            beta[k] = new
    gathered = [torch.zeros(p, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(beta))
    if rank == 0:
        np.save(out, np.stack([g.numpy() for g in gathered]))
    cp.barrier()
    cp.shutdown()


def test_shard_rows_partition():
    sys.path.insert(0, ROOT)
    from importlib import import_module
    import coordinatedescent_jl_amd  # noqa: F401
    sharded = import_module("coordinatedescent_jl_amd.sharded")
    for n, w in [(10, 1), (10, 3), (10_000_000, 8), (17, 8), (8, 8)]:
        spans = [sharded.shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(s[1] for s in spans) == n
        for (o1, l1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + l1 == o2
        assert max(s[1] for s in spans) - min(s[1] for s in spans) <= 1
    with pytest.raises(ValueError):
        sharded.shard_rows(3, 0, 8)


def test_two_rank_gloo_sharded_solve_matches_oracle(tmp_path):
    import oracle as O
    n, p, lam, world = 301, 12, 0.05, 2
    out = str(tmp_path / "beta.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, p, lam, out), nprocs=world, join=True)
    betas = np.load(out)
    np.testing.assert_array_equal(betas[0], betas[1])   # every rank derives the identical iterate
    rng = np.random.default_rng(5)
    X = np.asfortranarray(rng.standard_normal((n, p)))
    y = X[:, :4] @ rng.standard_normal(4) + rng.standard_normal(n)
    xo = O.SparseIterate(p)
    O.coordinateDescent_(xo, O.CDLeastSquaresLoss(y, X), O.ProxL1(lam),
                         O.CDOptions(maxIter=500, optTol=1e-13, randomize=False))
    np.testing.assert_allclose(betas[0], xo.dense(), rtol=0, atol=1e-10)
