"""Row-sharded multi-process driver: one process per GPU, the n (sample) dimension
split into contiguous row blocks, beta / omega / every scalar replicated.

The data path's only exchange step is the sum of the per-coordinate (or per-block)
gradient scalars, done INSIDE the HIP library with an RCCL all-reduce on the sweep
stream (csrc/cdhip.hip `allreduce`).  This module is the control plane around it:
row partition, rendezvous, broadcasting RCCL's unique id, and max-over-ranks timing.
`torch.distributed` is used for that only (gloo works; no tensors of the path go
through it).  The reference has no distributed code; the
sharding follows SURVEY.md section 8(e).
"""
from __future__ import annotations

import ctypes as C
import os


def shard_rows(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, near-equal row blocks: returns (row_offset, n_local).  Every row
    belongs to exactly one rank; the first n_total % world ranks get one extra row."""
    if not (0 <= rank < world) or n_total < world:
        raise ValueError("need 0 <= rank < world <= n_total")
    base, extra = divmod(n_total, world)
    n_local = base + (1 if rank < extra else 0)
    row_offset = rank * base + min(rank, extra)
    return row_offset, n_local


def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


class ControlPlane:
    """Rendezvous + tiny host-side collectives.  world == 1 needs no torch at all."""

    def __init__(self, backend: str = "gloo"):
        self.rank, self.local_rank, self.world = env_rank_world()
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def broadcast_bytes(self, payload: bytes | None, nbytes: int, src: int = 0) -> bytes:
        if self.dist is None:
            return payload
        import torch
        t = torch.zeros(nbytes, dtype=torch.uint8)
        if self.rank == src:
            t[:] = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        self.dist.broadcast(t, src=src)
        return bytes(t.numpy().tobytes())

    def all_gather_bytes(self, payload: bytes) -> bytes:
        """Concatenation of every rank's equal-length payload, in rank order."""
        if self.dist is None:
            return payload
        import torch
        mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        return b"".join(bytes(t.numpy().tobytes()) for t in parts)

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def shutdown(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def connect(loss, cp: ControlPlane):
    """Give `loss` (a row shard) its RCCL communicator: rank 0 draws the unique id, the
    control plane broadcasts the 128 bytes, every rank calls cdh_comm_init."""
    from . import _lib
    L = _lib.lib()
    uid = None
    force = bool(os.environ.get("CDH_FORCE_RCCL"))
    if cp.world == 1 and not force:
        return loss
    if cp.rank == 0:
        buf = C.create_string_buffer(128)
        _lib.check(L.cdh_comm_unique_id(buf), None)
        uid = buf.raw
    uid = cp.broadcast_bytes(uid, 128, src=0)
    loss.comm_init(uid, cp.rank, cp.world)
    return loss


def connect_checked(loss, cp: ControlPlane, unique_id=None) -> tuple[bool, str]:
    """connect() for a caller that can go on without RCCL (bench.py): builds the communicator, sums three probe
    records through it and compares them with the exact expected sums.  Returns (True, "") only if every stage
    succeeded on EVERY rank; otherwise (False, reason) -- the reason of the lowest rank that failed -- and the
    caller picks another exchange (connect_host).  Collective: every rank goes through the same sequence of
    control-plane collectives whatever fails locally, so one rank's failure cannot leave the others waiting.
    What it cannot catch is a communicator whose construction never returns.  `unique_id` (tests): a callable
    standing in for cdh_comm_unique_id on rank 0."""
    import numpy as np
    from . import _lib
    if cp.world == 1 and not os.environ.get("CDH_FORCE_RCCL"):     # (forced: a 1-rank communicator, as connect())
        return True, ""

    def agree(ok: bool, why: str) -> tuple[bool, str]:
        if cp.sum_over_ranks(1.0 if ok else 0.0) == cp.world:
            return True, ""
        blob = cp.all_gather_bytes((b"" if ok else (why or "failed").encode()[:200]).ljust(200, b"\0"))
        for q in range(cp.world):
            msg = blob[200 * q:200 * (q + 1)].rstrip(b"\0").decode(errors="replace")
            if msg:
                return False, f"rank {q}: {msg}"
        return False, "a rank failed without saying why"

    ok, why, uid = True, "", bytes(128)
    if cp.rank == 0:
        try:
            if unique_id is not None:
                uid = bytes(unique_id())
            else:
                buf = C.create_string_buffer(128)
                _lib.check(_lib.lib().cdh_comm_unique_id(buf), None)
                uid = buf.raw
        except Exception as e:
            ok, why = False, f"cdh_comm_unique_id: {e}"
    uid = cp.broadcast_bytes(uid if cp.rank == 0 else None, 128, src=0)
    ok, why = agree(ok, why)
    if not ok:
        return False, why
    try:
        loss.comm_init(uid, cp.rank, cp.world)
    except Exception as e:
        ok, why = False, f"cdh_comm_init: {e}"
    ok, why = agree(ok, why)
    if not ok:
        return False, why
    try:
        w, base = cp.world, np.arange(801, dtype=np.float64)
        for rep in range(3):
            got = loss.exchange_probe((cp.rank + 1) * (base + rep) + 0.25 * cp.rank)
            want = (w * (w + 1) / 2) * (base + rep) + 0.25 * (w * (w - 1) / 2)
            if ok and not np.array_equal(got, want):     # (no early exit: the peers are in the next all-reduce)
                ok, why = False, f"probe record {rep}: sums over the communicator are wrong"
    except Exception as e:
        ok, why = False, f"cdh_exchange_probe: {e}"
    return agree(ok, why)


def connect_host(loss, cp: ControlPlane):
    """Row-shard exchange over the control plane itself (cdh_set_host_exchange): every all-reduce of
    the sweep is staged through pinned host memory and summed by torch.distributed (gloo).  Far slower
    than RCCL or the direct exchange -- a stream round trip per exchange -- but it needs nothing from
    the GPUs' interconnect, so it is the transport that works anywhere (and the one the tests use to
    run the library's sharded launch sequence with several ranks on one GPU)."""
    if cp.world == 1:
        return loss
    import torch

    def allreduce(buf):
        t = torch.from_numpy(buf)          # shares memory with the library's staging buffer
        cp.dist.all_reduce(t, op=cp.dist.ReduceOp.SUM)

    loss.set_host_exchange(allreduce, cp.rank, cp.world)
    return loss


def connect_p2p(loss, cp: ControlPlane, selftest: bool = True) -> bool:
    """OPT-IN direct exchange for the short per-block records (csrc/p2p_exchange.hpp): each
    rank's inbox is IPC-mapped into every peer, handles travel over the control plane.  With
    `selftest`, probe records are exchanged and compared with the exact expected sums on every
    rank; the exchange is enabled only if EVERY stage succeeded on ALL ranks (else False is
    returned and the handle stays on RCCL -- unless the self-test ended in a timeout, after which
    the library refuses further exchanges on this handle: the ranks may no longer be in step).  Collective: every rank must call it, and every rank
    goes through the same sequence of control-plane collectives whatever fails locally."""
    import numpy as np
    if cp.world == 1:
        return False

    def everyone(ok: bool) -> bool:
        return cp.sum_over_ranks(1.0 if ok else 0.0) == cp.world

    ok, mine = True, bytes(64)
    try:
        mine = loss.p2p_local_handle()
    except Exception:
        ok = False
    handles = cp.all_gather_bytes(mine)
    if not everyone(ok):
        return False
    try:
        loss.p2p_connect(handles, cp.rank, cp.world)
    except Exception:
        ok = False
    if not everyone(ok):           # also the barrier: every inbox is mapped before anyone stores
        return False
    try:
        loss.p2p_enable(True)
        if selftest:
            count, w = 2625, cp.world
            base = np.arange(count, dtype=np.float64)
            for rep in range(3):   # three epochs: both slots and the reuse of the first
                got = loss.exchange_probe((cp.rank + 1) * (base + rep) + 0.25 * cp.rank)
                want = (w * (w + 1) / 2) * (base + rep) + 0.25 * (w * (w - 1) / 2)
                ok = ok and bool(np.array_equal(got, want))
    except Exception:
        ok = False
    if not everyone(ok):
        try:
            loss.p2p_enable(False)
        except Exception:
            pass
        return False
    return True
