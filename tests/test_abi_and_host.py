"""CPU-side checks: the C-ABI library builds/loads and exports every symbol that
include/cdhip.h declares; the host-side mirror (SparseIterate, iterators, options,
error mapping) behaves as the reference's tests require; without a GPU the product
fails loudly instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import coordinatedescent_jl_amd as cd


def test_library_exports_every_declared_symbol():
    assert not cd.needs_build(), "libcdhip.so is stale: run __graft_entry__.build()"
    names = cd.declared_symbols()
    assert len(names) >= 30 and "cdh_coordinate_descent" in names
    L = C.CDLL(cd.SO_PATH)
    for n in names:
        assert hasattr(L, n), n


def test_header_cites_reference_for_every_entry_point():
    txt = open(os.path.join(os.path.dirname(cd.SO_PATH), "..", "..", "include", "cdhip.h")).read()
    assert len(re.findall(r"\.jl:\d+", txt)) >= 20


def test_product_never_imports_oracle():
    pkg = os.path.dirname(cd.__file__)
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(root, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "cd_oracle" not in src, fn


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(cd.HipError):
        cd.CDLeastSquaresLoss(np.zeros(8), np.zeros((8, 2)))


def test_loss_constructor_checks_before_touching_the_device():
    with pytest.raises(cd.DimensionMismatch):  # cd_differentiable_function.jl:53
        cd.CDLeastSquaresLoss(np.zeros(7), np.zeros((8, 2)))
    with pytest.raises(TypeError):  # y::AbstractVector{T}, X::AbstractMatrix{T}
        cd.CDLeastSquaresLoss(np.zeros(8, dtype=np.float32), np.zeros((8, 2)))


def test_cdoptions_defaults():  # src/utils.jl:14-20
    o = cd.CDOptions()
    assert (o.maxIter, o.optTol, o.randomize, o.warmStart, o.numSteps) == (2000, 1e-7, True, True, 50)
    io = cd.IterLassoOptions()  # src/utils.jl:33-39
    assert (io.maxIter, io.optTol, io.initProcedure, io.sinit, io.sigmainit) == (20, 1e-2, "Screening", 5, 1.0)


def test_ordered_iterator():  # reference test/atom_iterator.jl:11-48
    x = cd.SparseIterate(5)
    x[2] = 1.0
    x[1] = 2.0
    it = cd.OrderedIterator(x)
    assert list(it) == [1, 2, 3, 4, 5]
    cd.reset_(it, True)
    assert list(it) == [1, 2, 3, 4, 5]
    cd.reset_(it, False)
    assert list(it) == [2, 1]


def test_random_iterator_matches_oracle_stream():  # reference test/atom_iterator.jl:50-85
    import oracle as O
    rng = np.random.default_rng(0)
    p = 50
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    for _ in range(10):
        k, v = int(rng.integers(1, p + 1)), float(rng.standard_normal())
        x[k] = v
        xo[k] = v
    it, ito = cd.RandomIterator(x, seed=77), O.Iterator(xo, randomize=True, seed=77)
    assert list(it) == list(range(1, p + 1))
    for full in (True, False, True, False):
        cd.reset_(it, full)
        ito.reset(full)
        got = list(it)
        assert got == ito.collect().tolist()
        if full:
            assert got == it.order
        else:
            assert got == [int(x.nzval2ind[o - 1]) for o in it.order[: x.nnz]]


def test_sparse_iterate_matches_oracle_bookkeeping():
    import oracle as O
    rng = np.random.default_rng(1)
    p = 30
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    for step in range(400):
        k = int(rng.integers(1, p + 1))
        v = 0.0 if rng.random() < 0.4 else float(rng.standard_normal())
        x[k] = v
        xo[k] = v
        if step % 37 == 36:
            x.dropzeros_()
            xo.dropzeros()
        assert x.nnz == xo.nnz
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()
    np.testing.assert_array_equal(x.dense(), xo.dense())
    y = x.copy()
    x.fill_(0.0)
    assert x.nnz == 0 and y.nnz == xo.nnz and y == y.copy() and not (y == x)


def test_every_export_rejects_a_null_handle():
    """status codes, never an abort.  Every entry point that takes a handle returns
    CDH_BAD_ARG for NULL (cdh_destroy(NULL) is a no-op, cdh_last_error(NULL) is the create error);
    nothing here touches a device, so it runs without a GPU."""
    L = cd._lib.lib()
    BAD = cd._lib.CDH_BAD_ARG
    no_handle = {"cdh_create", "cdh_destroy", "cdh_last_error", "cdh_device_count", "cdh_comm_unique_id"}
    checked = 0
    for name in cd.declared_symbols():
        if name in no_handle:
            continue
        fn = getattr(L, name)
        args = [None]
        for t in fn.argtypes[1:]:
            if t in (C.c_int32, C.c_int64, C.c_uint64):
                args.append(0)
            elif t is C.c_double:
                args.append(0.0)
            elif t is cd._lib.HOST_ALLREDUCE_FN:
                args.append(cd._lib.HOST_ALLREDUCE_FN(0))
            else:
                args.append(None)
        assert fn(*args) == BAD, name
        assert b"NULL" in L.cdh_last_error(None), name
        checked += 1
    assert checked >= 38
    assert L.cdh_destroy(None) == cd._lib.CDH_OK
    assert L.cdh_device_count(None) == BAD and L.cdh_comm_unique_id(None) == BAD
    assert L.cdh_create(None, 0, 0, 8, 8, 0, 2, 0) == BAD


def test_k_cross_lds_image_is_conflict_free_and_consistent():
    """csrc/gram_kernels.hpp, k_cross: an LDS-DMA piece writes lane l at (wave-uniform base) + 16 l, so the image cannot
    be padded; bank conflicts are removed on the SOURCE side -- lane (column cl = l // 8, slot j = l % 8) fetches vector
    j ^ (cl & 6) of its 128-byte line, i.e. vector v of column cc sits in 16-byte slot 8 cc + (v ^ (cc & 6)).  Checked
    here exhaustively: (a) the write side and the read side agree on where (column, vector) lives; (b) every
    ds_read_b128 of the MFMA fragment pattern -- lane (c, g) reads column 16 t + c, vector 4 u + g -- hits 16 distinct
    bank quads in each of the instruction's four lane groups (MI355X_MICROARCH.md, LDS table)."""
    groups = [[*range(0, 4), *range(12, 16), *range(20, 28)], [*range(4, 12), *range(16, 20), *range(28, 32)],
              [*range(32, 36), *range(44, 48), *range(52, 60)], [*range(36, 44), *range(48, 52), *range(60, 64)]]
    slot = lambda cc, v: cc * 8 + (v ^ (cc & 6))        # noqa: E731  (f_slot / a_slot in the kernel)
    for piece in range(8):                              # 8 pieces of 8 columns x 8 vectors: 64 columns
        for lane in range(64):
            cl, j = lane // 8, lane % 8
            cc, v = 8 * piece + cl, j ^ (cl & 6)        # what the lane fetches (sj in the kernel)
            assert slot(cc, v) == 64 * piece + lane     # ... lands where the fragment reads look for it
    for t in range(4):
        for u in range(2):
            for grp in groups:
                quads = {slot(16 * t + (lane & 15), 4 * u + (lane >> 4)) % 16 for lane in grp}
                assert len(quads) == 16, (t, u, sorted(quads))
