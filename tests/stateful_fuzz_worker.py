"""Worker of test_gpu_stateful_fuzz.py::test_random_call_sequences_on_row_shards (run under torch.distributed.run, every
rank on cuda:0): the module's call sequences with the loss object replaced by a ROW SHARD per rank, the host-staged
exchange (cdh_set_host_exchange -> a gloo all-reduce) behind the library's all-reduce seam.  Every rank draws the same
sequence, checks the same replicated quantities against the oracle's unsharded run, and its own rows of the residual.
Prints FUZZ_SHARDS_OK <sequences> from rank 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("CDH_SMALL_PATH", "0")
import coordinatedescent_jl_amd as cd  # noqa: E402,F401
from coordinatedescent_jl_amd import sharded  # noqa: E402
import test_gpu_stateful_fuzz as T  # noqa: E402


class Shard:
    sharded = True

    def __init__(self, cp):
        self.cp = cp

    def rows(self, n):
        row0, nl = sharded.shard_rows(n, self.cp.rank, self.cp.world)
        self.row0 = row0
        return slice(row0, row0 + nl)

    def make(self, cls, n, y, X, *extra):
        f = cls(np.ascontiguousarray(y), np.asfortranarray(X), *[np.ascontiguousarray(e) for e in extra],
                device=0, n_total=n, row_offset=self.row0)
        sharded.connect_host(f, self.cp)
        return f


    def check_replicas(self, x, log):
        mine = np.ascontiguousarray(x.dense())
        every = np.frombuffer(self.cp.all_gather_bytes(mine.tobytes()), dtype=np.float64).reshape(self.cp.world, -1)
        assert all(np.array_equal(every[0], every[q]) for q in range(1, self.cp.world)), "ranks disagree\n" + "\n".join(log)


def main():
    cp = sharded.ControlPlane(backend="gloo")
    assert cp.world >= 2
    first, count = int(sys.argv[1]), int(sys.argv[2])
    for seed in range(first, first + count):
        for k in ("CDH_GC_REFRESH", "CDH_GC_INJECT_ROLLBACK"):
            os.environ.pop(k, None)
        T.run_sequence(seed, os.environ.__setitem__, Shard(cp))
        cp.barrier()
    got = T.REACHED
    assert got["sequences"] == count
    if cp.rank == 0:
        print("FUZZ_SHARDS_OK", count, {k: got.get(k, 0) for k in ("solves", "passes", "device_passes", "covariance_visits",
                                                                      "gram_batches", "rollbacks", "cold_start_ties")}, flush=True)
    cp.shutdown()


if __name__ == "__main__":
    main()
