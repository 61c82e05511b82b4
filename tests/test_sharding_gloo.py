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


class _FakeShard:
    """Stands in for a loss handle in connect_checked: what matters here is the sequence of control-plane
    collectives when a stage fails on ONE rank only (no rank may be left waiting in a collective)."""

    def __init__(self, rank, world, fail_init_on=None, wrong_sums_on=None, dist=None):
        self.rank, self.world, self.fail_init_on, self.wrong_sums_on, self.dist = rank, world, fail_init_on, wrong_sums_on, dist
        self.uid = None

    def comm_init(self, uid, rank, world):
        self.uid = uid
        if self.fail_init_on == rank:
            raise RuntimeError("ncclCommInitRank failed: unhandled system error")

    def exchange_probe(self, values):
        t = torch.from_numpy(np.array(values, dtype=np.float64))
        self.dist.all_reduce(t)
        out = t.numpy().copy()
        if self.wrong_sums_on == self.rank:
            out[7] += 1.0
        return out


def _connect_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from importlib import import_module
    import coordinatedescent_jl_amd  # noqa: F401
    sharded = import_module("coordinatedescent_jl_amd.sharded")
    cp = sharded.ControlPlane(backend="gloo")
    uid = lambda: bytes(range(128))      # noqa: E731

    def boom():
        raise OSError("dlopen librccl failed")

    got = []
    good = _FakeShard(rank, world, dist=dist)
    got.append(sharded.connect_checked(good, cp, unique_id=uid))
    assert good.uid == bytes(range(128))                       # every rank received rank 0's id
    got.append(sharded.connect_checked(_FakeShard(rank, world, dist=dist), cp, unique_id=boom))
    got.append(sharded.connect_checked(_FakeShard(rank, world, fail_init_on=1, dist=dist), cp, unique_id=uid))
    got.append(sharded.connect_checked(_FakeShard(rank, world, wrong_sums_on=0, dist=dist), cp, unique_id=uid))
    gathered = [None] * world
    dist.all_gather_object(gathered, got)
    if rank == 0:
        import json
        with open(out, "w") as fh:
            json.dump(gathered, fh)
    cp.barrier()
    cp.shutdown()


def test_checked_connect_agrees_on_every_rank_whatever_fails_locally(tmp_path):
    """bench.py's RCCL bring-up (sharded.connect_checked): a failure on one rank at any stage -- no unique id, no
    communicator, wrong probe sums -- must come back as the SAME (False, reason) on every rank, with nobody left
    waiting, so that all ranks switch to the host-staged exchange together."""
    import json
    world, out = 2, str(tmp_path / "connect.json")
    mp.spawn(_connect_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = json.load(open(out))
    assert r0 == r1
    assert r0[0] == [True, ""]
    assert r0[1][0] is False and r0[1][1].startswith("rank 0: cdh_comm_unique_id:") and "dlopen" in r0[1][1]
    assert r0[2][0] is False and r0[2][1].startswith("rank 1: cdh_comm_init:")
    assert r0[3][0] is False and r0[3][1].startswith("rank 0: probe record 0")
