"""Host-side mirror of CoordinateDescent.jl's interface for the hot path, over the
C-ABI of the HIP library (include/cdhip.h).

The reference is Julia and no Julia toolchain exists in this image, so this mirror
is Python (julia/CoordinateDescentHIP.jl holds the thin ccall binding a Julia
maintainer would add).  Names, argument meaning and error behaviour follow the
reference: `coordinateDescent!` is `coordinateDescent_`, and so on (a trailing
underscore stands for Julia's `!`).  Coordinates are 1-based, as in the reference.
All arithmetic of the path runs in the HIP library; nothing here computes on the
CPU beyond the O(p) bookkeeping the reference also does on the host.

Reference files mirrored (relative to the reference's src/):
  utils.jl:7-39                      CDOptions, IterLassoOptions
  cd_differentiable_function.jl      the loss operators and the 4-function plugin API
  coordinate_descent.jl              coordinateDescent!, _findLambdaMax
  atom_iterator.jl                   OrderedIterator, RandomIterator
  lasso.jl                           lasso, sqrtLasso, scaledLasso!, LassoPath
  ProximalBase 0.3.0 (not vendored)  ProxL1, SparseIterate (contract: SURVEY.md App. B)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import (ArgumentError, DimensionMismatch, DomainError, HipError,  # noqa: F401
                   CDH_F32, CDH_F64, CDH_LS, CDH_SQRT, CDH_WLS, CDH_SWEEP_BLOCK, CDH_SWEEP_COORD,
                   cdh_options, cdh_stats, check)


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------
# Options (src/utils.jl:7-39)
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class CDOptions:
    """CDOptions(;maxIter=2000, optTol=1e-7, randomize=true, warmStart=true, numSteps=50)
    (src/utils.jl:7-20).  `seed` seeds the documented substitute for Julia's global RNG
    used by RandomIterator (src/atom_iterator.jl:60)."""
    maxIter: int = 2000
    optTol: float = 1e-7
    randomize: bool = True
    warmStart: bool = True
    numSteps: int = 50
    seed: int = 0

    def _c(self):
        return cdh_options(int(self.maxIter), float(self.optTol), int(bool(self.randomize)),
                           int(bool(self.warmStart)), int(self.numSteps), int(self.seed))


@dataclass(frozen=True)
class IterLassoOptions:
    """IterLassoOptions (src/utils.jl:24-39); initProcedure in {"Screening","InitStd","WarmStart"}."""
    maxIter: int = 20
    optTol: float = 1e-2
    initProcedure: str = "Screening"
    sinit: int = 5
    sigmainit: float = 1.0
    optionsCD: CDOptions = field(default_factory=CDOptions)


# --------------------------------------------------------------------------------------
# ProximalBase: ProxL1, SparseIterate
# --------------------------------------------------------------------------------------
class ProxL1:
    """ProxL1(λ0) / ProxL1(λ0, λ::AbstractArray): fields lambda0 and lam (None = unweighted)."""

    def __init__(self, lambda0, lam=None):
        self.lambda0 = float(lambda0)
        self.lam = None if lam is None else np.ascontiguousarray(lam, dtype=np.float64).copy()


class SparseIterate:
    """SparseIterate(p): dense-indexable iterate with an insertion-ordered support
    (nzval2ind[1:nnz]); zeros written to stored coordinates keep their slot until
    dropzeros_ (test/atom_iterator.jl:13-28)."""

    def __init__(self, p, values=None):
        if isinstance(p, np.ndarray):
            values, p = p, p.shape[0]
        self.p = int(p)
        self._val = np.zeros(self.p)
        self._slot2ind = np.zeros(self.p, dtype=np.int64)
        self._ind2slot = np.zeros(self.p, dtype=np.int64)
        self._nnz = 0
        self._version = 0
        if values is not None:
            for k, v in enumerate(np.asarray(values, dtype=np.float64)):
                if v != 0.0:
                    self[k + 1] = v

    # -- AbstractArray surface ---------------------------------------------------------
    def __len__(self):
        return self.p

    def __getitem__(self, k1):
        s = self._ind2slot[k1 - 1]
        return float(self._val[s - 1]) if s else 0.0

    def __setitem__(self, k1, v):
        k = int(k1) - 1
        if not 0 <= k < self.p:
            raise IndexError(k1)
        v = float(v)
        s = self._ind2slot[k]
        self._version += 1
        if s:
            self._val[s - 1] = v
        elif v != 0.0:
            self._val[self._nnz] = v
            self._slot2ind[self._nnz] = k
            self._nnz += 1
            self._ind2slot[k] = self._nnz

    @property
    def nnz(self):
        return self._nnz

    @property
    def nzval2ind(self):
        """1-based support in insertion order."""
        return self._slot2ind[: self._nnz] + 1

    @property
    def nzval(self):
        return self._val[: self._nnz].copy()

    def dense(self):
        out = np.zeros(self.p)
        out[self._slot2ind[: self._nnz]] = self._val[: self._nnz]
        return out

    __array__ = lambda self, dtype=None, copy=None: self.dense()  # noqa: E731  Vector(x)

    def __eq__(self, other):
        return isinstance(other, SparseIterate) and self.p == other.p and \
            np.array_equal(self.dense(), other.dense())

    def fill_(self, v=0.0):
        """fill!(x, 0) empties the iterate."""
        if v != 0.0:
            raise ArgumentError("only fill!(x, 0) is used on this path")
        self._ind2slot[self._slot2ind[: self._nnz]] = 0
        self._nnz = 0
        self._version += 1

    def dropzeros_(self):
        """dropzeros!(x): swap-with-last compaction (order unpinned by the reference's tests)."""
        i = 0
        while i < self._nnz:
            if self._val[i] == 0.0:
                self._ind2slot[self._slot2ind[i]] = 0
                last = self._nnz - 1
                if i != last:
                    self._val[i] = self._val[last]
                    self._slot2ind[i] = self._slot2ind[last]
                    self._ind2slot[self._slot2ind[i]] = i + 1
                self._nnz -= 1
            else:
                i += 1
        self._version += 1

    def copy(self):
        y = SparseIterate(self.p)
        y._val[:] = self._val
        y._slot2ind[:] = self._slot2ind
        y._ind2slot[:] = self._ind2slot
        y._nnz = self._nnz
        return y

    # -- sync with a device handle -------------------------------------------------------
    def _load(self, idx1, val):
        self._ind2slot[:] = 0
        n = len(idx1)
        self._slot2ind[:n] = np.asarray(idx1, dtype=np.int64) - 1
        self._val[:n] = val
        self._ind2slot[self._slot2ind[:n]] = np.arange(1, n + 1)
        self._nnz = n
        self._version += 1


def numCoordinates(obj):
    """numCoordinates(f) / ProximalBase.numCoordinates(x)."""
    return obj.p


# --------------------------------------------------------------------------------------
# Loss operators = the plugin API (src/cd_differentiable_function.jl)
# --------------------------------------------------------------------------------------
class CoordinateDifferentiableFunction:
    """abstract type CoordinateDifferentiableFunction (src/cd_differentiable_function.jl:1)."""


class _HipLoss(CoordinateDifferentiableFunction):
    """A loss whose X, y, r live in HBM behind one cdh_handle."""
    _kind = CDH_LS

    def __init__(self, y, X, w=None, *, device=0, n_total=None, row_offset=0):
        X = np.asarray(X)
        y = np.asarray(y)
        if X.dtype not in (np.float64, np.float32) or y.dtype != X.dtype or X.ndim != 2:
            raise TypeError("MethodError: y::AbstractVector{T}, X::AbstractMatrix{T}, T<:AbstractFloat")
        if y.shape[0] != X.shape[0]:  # cd_differentiable_function.jl:53,212
            raise DimensionMismatch("length(y) != size(X, 1)")
        if w is not None and np.asarray(w).shape[0] != X.shape[0]:  # :129
            raise DimensionMismatch("length(w) != size(X, 1)")
        n, p = X.shape
        self._create(X.dtype, n, p, device, n_total, row_offset)
        step = max(1, (64 << 20) // max(1, n * X.itemsize))
        for j0 in range(0, p, step):
            blk = np.asfortranarray(X[:, j0:j0 + step])
            check(self._L.cdh_set_X_cols(self._h, j0, blk.shape[1], _vp(blk), n), self._h)
        yy = np.ascontiguousarray(y)
        check(self._L.cdh_set_y(self._h, _vp(yy)), self._h)
        if w is not None:
            ww = np.ascontiguousarray(w, dtype=X.dtype)
            check(self._L.cdh_set_obs_weights(self._h, _vp(ww)), self._h)

    def _create(self, dtype, n, p, device, n_total, row_offset):
        self._L = _lib.lib()
        self.dtype = np.dtype(dtype)
        self.n, self.p = int(n), int(p)
        self.n_total = int(n if n_total is None else n_total)
        self.row_offset = int(row_offset)
        h = C.c_void_p()
        st = self._L.cdh_create(C.byref(h), CDH_F64 if self.dtype == np.float64 else CDH_F32,
                                self._kind, self.n, self.n_total, self.row_offset, self.p, int(device))
        check(st, None)
        self._h = h
        self._synced = None  # (id(x), version) of the iterate the handle currently mirrors
        self._penalty = None
        self.last_stats = None

    @classmethod
    def generate(cls, n, p, *, seed=123, s=0, noise=1.0, dtype=np.float64, device=0, n_total=None,
                 row_offset=0):
        """Synthetic Gaussian problem generated on the device (benchmark/cd_bench.jl:8-14
        shapes); returns (loss, planted beta*)."""
        self = cls.__new__(cls)
        self._create(dtype, n, p, device, n_total, row_offset)
        bstar = np.zeros(max(int(s), 1))
        check(self._L.cdh_generate(self._h, int(seed), int(s), float(noise), _vp(bstar)), self._h)
        return self, bstar[: int(s)]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.cdh_destroy(self._h)
                self._h = None
        except Exception:
            pass

    close = __del__

    # -- data views ------------------------------------------------------------------------
    @property
    def r(self):
        """f.r: the residual vector (copied from HBM)."""
        out = np.zeros(self.n, dtype=self.dtype)
        check(self._L.cdh_get_residual(self._h, _vp(out)), self._h)
        return out

    @property
    def y(self):
        out = np.zeros(self.n, dtype=self.dtype)
        check(self._L.cdh_get_y(self._h, _vp(out)), self._h)
        return out

    def X_cols(self, j0, ncols):
        """Columns [j0, j0+ncols) (0-based bulk helper) copied back from HBM."""
        out = np.zeros((self.n, ncols), dtype=self.dtype, order="F")
        check(self._L.cdh_get_X_cols(self._h, int(j0), int(ncols), _vp(out), self.n), self._h)
        return out

    # -- execution control -----------------------------------------------------------------
    def set_sweep_mode(self, mode, block=32):
        mode = {"coord": CDH_SWEEP_COORD, "block": CDH_SWEEP_BLOCK}.get(mode, mode)
        check(self._L.cdh_set_sweep_mode(self._h, int(mode), int(block)), self._h)

    def set_screening(self, on=True):
        """0 / False: never; 1 / True: full passes of the solves (default); 2: cdPass_ as well."""
        check(self._L.cdh_set_screening(self._h, int(on)), self._h)

    def set_gradient_cache(self, mode=1):
        """0 off, 1 rent-or-buy (default), 2 from the first full pass, 3 = 2 without the tall-problem guard."""
        check(self._L.cdh_set_gradient_cache(self._h, int(mode)), self._h)

    def set_onchip_solve(self, on=True):
        """One-launch solves of problems that fit on chip (cdh_set_onchip_solve; on by default)."""
        check(self._L.cdh_set_onchip_solve(self._h, int(bool(on))), self._h)

    def onchip_stats(self):
        out = (C.c_int64 * 2)()
        check(self._L.cdh_onchip_stats(self._h, out), self._h)
        return {"solves": int(out[0]), "gram_matrices": int(out[1])}

    def onchip_last(self):
        """Of the last one-launch solve: visit steps, kernel time in microseconds, the clock (GHz) the chip held."""
        out = (C.c_int64 * 3)()
        check(self._L.cdh_onchip_last(self._h, out), self._h)
        return {"steps": int(out[0]), "kernel_us": out[2] / 100.0, "clock_GHz": (out[1] / out[2] * 0.1) if out[2] else 0.0}

    def gradient_cache_mode(self):
        out = C.c_int32()
        check(self._L.cdh_get_gradient_cache(self._h, C.byref(out)), self._h)
        return out.value

    def cache_drift(self, rereference_now=False):
        """max_k |g_carried - X_k'r| / thr_k at the gradient cache's re-references (cdh_cache_drift); with
        rereference_now the carried gradient is taken afresh from X first (one dots-only pass) and measured."""
        out = (C.c_double * 3)()
        check(self._L.cdh_cache_drift(self._h, int(bool(rereference_now)), out), self._h)
        return {"last": out[0], "max": out[1], "measured": int(out[2])}

    def cache_stats(self):
        out = (C.c_int64 * 10)()
        check(self._L.cdh_cache_stats(self._h, out), self._h)
        return dict(zip(("passes", "settled_visits", "exact_visits", "reference_passes", "gram_batches", "gram_columns",
                         "covariance_visits", "residual_catchups", "rollbacks", "device_passes"), [int(v) for v in out]))

    def cache_gram_column(self, k):
        """(X'X_k as the gradient cache holds it, the relative error its entries are declared to carry); k 1-based."""
        out, eps = np.zeros(self.p), C.c_double()
        check(self._L.cdh_cache_gram_column(self._h, int(k), _vp(out), C.byref(eps)), self._h)
        return out, eps.value

    def set_device_loop(self, on=True, helpers=None):
        """The pass loop of a cache-served solve on the device (cdh_set_device_loop; on by default).  helpers: 0 keeps the
        loop to one workgroup (large visit lists then run from its Gram table only), n > 2 sets the number of helper
        workgroups a launch that expects large visit lists brings (default 31)."""
        v = int(bool(on))
        if on and helpers is not None:
            v = 2 if int(helpers) == 0 else max(3, int(helpers))
        check(self._L.cdh_set_device_loop(self._h, v), self._h)

    def device_loop_stats(self):
        out = (C.c_int64 * 12)()
        check(self._L.cdh_device_loop_stats(self._h, out), self._h)
        d = dict(zip(("launches", "passes", "folds", "exact_rechecks"), [int(x) for x in out[:4]]))
        d["phase_us"] = dict(zip(("list", "scan", "exact_g", "visits", "recheck", "accept", "bookkeeping", "dropzeros_rest"),
                                 [int(x) / 100.0 for x in out[4:]]))
        t = (C.c_int64 * 8)()
        check(self._L.cdh_device_loop_table(self._h, t), self._h)
        d["table"] = dict(zip(("passes", "rows_filled", "coordinates", "capacity"), [int(x) for x in t[:4]]))
        d["forced_rounds"] = {"host_pass": int(t[4]), "loop": int(t[5])}
        d["crew"] = {"passes": int(t[6]), "jobs": int(t[7])}
        return d

    def set_use_graph(self, on=True):
        check(self._L.cdh_set_use_graph(self._h, int(bool(on))), self._h)

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        check(self._L.cdh_comm_init(self._h, buf, int(rank), int(nranks)), self._h)

    def comm_drop(self):
        """Give the RCCL communicator up so that another exchange can be installed (cdh_comm_drop)."""
        check(self._L.cdh_comm_drop(self._h), self._h)

    def p2p_local_handle(self) -> bytes:
        buf = C.create_string_buffer(64)
        check(self._L.cdh_p2p_local_handle(self._h, buf), self._h)
        return buf.raw

    def p2p_connect(self, handles: bytes, rank: int, nranks: int):
        if len(handles) != 64 * int(nranks):
            raise ArgumentError("p2p_connect needs 64 bytes per rank")
        buf = C.create_string_buffer(bytes(handles), len(handles))
        check(self._L.cdh_p2p_connect(self._h, buf, int(rank), int(nranks)), self._h)

    def p2p_enable(self, on=True):
        check(self._L.cdh_p2p_enable(self._h, int(bool(on))), self._h)

    def set_host_exchange(self, fn, rank, nranks):
        """Bring-your-own transport: `fn(array_of_doubles)` must sum the array in place over all ranks
        (cdh_set_host_exchange).  The ctypes trampoline is kept alive on the loss."""
        if fn is None:
            check(self._L.cdh_set_host_exchange(self._h, _lib.HOST_ALLREDUCE_FN(0), None, 0, 1), self._h)
            self._host_cb = None
            return

        def tramp(_user, ptr, count):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(count,)))
                return 0
            except Exception:      # nothing may unwind into the library
                return 1
        cb = _lib.HOST_ALLREDUCE_FN(tramp)
        check(self._L.cdh_set_host_exchange(self._h, cb, None, int(rank), int(nranks)), self._h)
        self._host_cb = cb

    def exchange_stats(self):
        """All-reduces issued through each exchange so far, and the rank count the active one reports."""
        a, b, c, n = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        check(self._L.cdh_exchange_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)), self._h)
        return {"rccl_calls": a.value, "p2p_calls": b.value, "host_calls": c.value, "nranks": n.value}

    def exchange_probe(self, values):
        """All-reduce (sum) up to 4096 doubles through the active exchange; returns the sums."""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(self._L.cdh_exchange_probe(self._h, v.ctypes.data_as(C.c_void_p), v.size), self._h)
        return v

    def exchange_latency(self, count, iters=200):
        """Average microseconds per all-reduce of `count` doubles through the active exchange (collective)."""
        out = C.c_double()
        check(self._L.cdh_exchange_latency(self._h, int(count), int(iters), C.byref(out)), self._h)
        return out.value

    def profile_begin(self):
        check(self._L.cdh_profile_begin(self._h), self._h)

    def profile_end(self):
        ms, nl, by = C.c_double(), C.c_int64(), C.c_double()
        check(self._L.cdh_profile_end(self._h, C.byref(ms), C.byref(nl), C.byref(by)), self._h)
        return ms.value, nl.value, by.value

    # -- sync helpers ----------------------------------------------------------------------
    def _set_penalty(self, g):
        if not isinstance(g, ProxL1):
            raise TypeError("MethodError: descendCoordinate! is defined for g::ProxL1 only")
        n_om = 0 if g.lam is None else g.lam.shape[0]
        check(self._L.cdh_set_penalty(self._h, g.lambda0, _vp(g.lam), n_om), self._h)

    def _push(self, x, rebuild):
        idx = np.ascontiguousarray(x.nzval2ind, dtype=np.int64)
        val = np.ascontiguousarray(x._val[: x.nnz], dtype=np.float64)
        fn = self._L.cdh_initialize if rebuild else self._L.cdh_set_iterate
        check(fn(self._h, len(x), x.nnz, _vp(idx), _vp(val)), self._h)
        self._synced = (id(x), x._version)

    def _ensure_synced(self, x):
        if self._synced != (id(x), x._version):
            self._push(x, rebuild=False)

    def _pull(self, x):
        if getattr(self, "_pull_idx", None) is None:      # (two p-sized host buffers, kept: a path pulls once per lambda)
            self._pull_idx, self._pull_beta = np.zeros(max(self.p, 1), dtype=np.int64), np.zeros(self.p)
        idx, beta = self._pull_idx, self._pull_beta
        nnz = C.c_int64()
        check(self._L.cdh_get_support(self._h, _vp(idx), C.byref(nnz)), self._h)
        check(self._L.cdh_get_beta(self._h, _vp(beta)), self._h)
        sup = idx[: nnz.value]
        x._load(sup, beta[sup - 1])
        self._synced = (id(x), x._version)


class CDLeastSquaresLoss(_HipLoss):
    """CDLeastSquaresLoss(y, X): |y - Xβ|²/(2n) (src/cd_differentiable_function.jl:43-111)."""
    _kind = CDH_LS


class CDSqrtLassoLoss(_HipLoss):
    """CDSqrtLassoLoss(y, X): |y - Xβ|₂ (src/cd_differentiable_function.jl:202-291)."""
    _kind = CDH_SQRT


class CDWeightedLSLoss(_HipLoss):
    """CDWeightedLSLoss(y, X, w): Σ w_i (y_i - X_iβ)²/(2n) (src/cd_differentiable_function.jl:118-194)."""
    _kind = CDH_WLS

    def __init__(self, y, X, w, **kw):
        super().__init__(y, X, w, **kw)


def initialize_(f, x):
    """initialize!(f, x): r = y - Xβ (src/cd_differentiable_function.jl:59-72)."""
    f._push(x, rebuild=True)


def gradient(f, x, k):
    """gradient(f, x, k) (src/cd_differentiable_function.jl:75-76, 234-235)."""
    f._ensure_synced(x)
    out = C.c_double()
    check(f._L.cdh_gradient(f._h, int(k), C.byref(out)), f._h)
    return out.value


def descendCoordinate_(f, g, x, k):
    """descendCoordinate!(f, g, x, k) -> h (src/cd_differentiable_function.jl:83-111, 242-291)."""
    f._set_penalty(g)
    f._ensure_synced(x)
    out = C.c_double()
    check(f._L.cdh_descend(f._h, int(k), C.byref(out)), f._h)
    f._pull(x)
    return out.value


# --------------------------------------------------------------------------------------
# Coordinate schedulers (src/atom_iterator.jl) -- host objects; the library has its own
# copy of the same logic for whole solves.
# --------------------------------------------------------------------------------------
class OrderedIterator:
    def __init__(self, iterate):
        self.iterate, self.fullPass = iterate, True

    def __iter__(self):
        if self.fullPass:
            return iter(range(1, numCoordinates(self.iterate) + 1))
        return iter(self.iterate.nzval2ind.tolist())

    def __len__(self):
        return numCoordinates(self.iterate) if self.fullPass else self.iterate.nnz


class RandomIterator:
    """Fisher-Yates with the documented splitmix64 substitute for Julia's global RNG."""

    def __init__(self, iterate, seed=0):
        self.iterate, self.fullPass = iterate, True
        self.order = list(range(1, numCoordinates(iterate) + 1))
        self._state = int(seed) & 0xFFFFFFFFFFFFFFFF

    def _next(self):
        m = 0xFFFFFFFFFFFFFFFF
        self._state = (self._state + 0x9E3779B97F4A7C15) & m
        z = self._state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
        return z ^ (z >> 31)

    def __iter__(self):
        L = len(self)
        if self.fullPass:
            return iter(self.order[:L])
        sup = self.iterate.nzval2ind
        return iter([int(sup[o - 1]) for o in self.order[:L]])

    def __len__(self):
        return numCoordinates(self.iterate) if self.fullPass else self.iterate.nnz


def reset_(it, fullPass):
    """reset!(it, fullPass) (src/atom_iterator.jl:34-37, 53-64)."""
    it.fullPass = bool(fullPass)
    if isinstance(it, RandomIterator):
        L = len(it)
        for i in range(L):
            it.order[i] = i + 1
        for i in range(L - 1):
            j = i + it._next() % (L - i)
            it.order[i], it.order[j] = it.order[j], it.order[i]
    return it


# --------------------------------------------------------------------------------------
# Driver (src/coordinate_descent.jl)
# --------------------------------------------------------------------------------------
def _check_dims(x, f, g):
    if numCoordinates(x) != numCoordinates(f):  # :13
        raise DimensionMismatch("numCoordinates(x) != numCoordinates(f)")
    if g.lam is not None and g.lam.shape[0] != numCoordinates(f):  # :14-16
        raise DimensionMismatch("length(g.λ) != numCoordinates(f)")


def _stats(st):
    return {"passes": st.passes, "full_passes": st.full_passes, "visits": st.visits,
            "converged": bool(st.converged), "domain_error": bool(st.domain_error),
            "maxH": st.maxH, "lambda_max": st.lambda_max}


def coordinateDescent_(x, f, g, options=None):
    """coordinateDescent!(x, f, g::ProxL1, options=CDOptions()) (src/coordinate_descent.jl:7-39).
    Returns x; pass/convergence statistics are left in f.last_stats."""
    options = options or CDOptions()
    _check_dims(x, f, g)
    f._set_penalty(g)
    f._ensure_synced(x)
    o, st = options._c(), cdh_stats()
    status = f._L.cdh_coordinate_descent(f._h, C.byref(o), C.byref(st))
    check(status, f._h)
    f.last_stats = _stats(st)
    f._pull(x)
    if st.domain_error:
        raise DomainError("sqrt-lasso update took sqrt of a negative (λ² ≥ ‖X_k‖² ?)")
    return x


def cdPass_(x, f, g, visit):
    """_cdPass!(x, f, g, it) (src/coordinate_descent.jl:94-110) over an explicit 1-based
    visit list; returns maxH."""
    f._set_penalty(g)
    f._ensure_synced(x)
    if isinstance(visit, np.ndarray) and visit.dtype == np.int64 and visit.flags.c_contiguous:
        idx = visit                      # a prepared list crosses as is (sweep loops reuse one)
    else:
        idx = np.ascontiguousarray(list(visit), dtype=np.int64)
    out = C.c_double()
    check(f._L.cdh_pass(f._h, idx.shape[0], _vp(idx), C.byref(out)), f._h)
    f._pull(x)
    return out.value


def findLambdaMax(x, f, g):
    """_findLambdaMax(x, f, g) (src/coordinate_descent.jl:118-149)."""
    f._set_penalty(g)
    f._ensure_synced(x)
    out = C.c_double()
    check(f._L.cdh_lambda_max(f._h, C.byref(out)), f._h)
    return out.value


def stdX(f):
    """_stdX!(out, X) (src/utils.jl:127-138) for the X resident behind loss f."""
    out = np.zeros(f.p)
    check(f._L.cdh_col_rms(f._h, _vp(out)), f._h)
    return out


def objective(f, g=None):
    """f(β) + λ0 Σ ω|β| at the handle's current state (src/coordinate_descent.jl:1-3)."""
    if g is not None:
        f._set_penalty(g)
    out = C.c_double()
    check(f._L.cdh_objective(f._h, C.byref(out)), f._h)
    return out.value


# --------------------------------------------------------------------------------------
# Front-ends (src/lasso.jl)
# --------------------------------------------------------------------------------------
@dataclass
class LassoSolution:
    """LassoSolution (src/lasso.jl:7-17)."""
    x: SparseIterate
    residuals: np.ndarray
    penalty: ProxL1
    sigma: float


def _std_resid(f):
    """std(f.r) as Statistics.std computes it: the mean, then the centred sum of squares (two passes on the device),
    Bessel-corrected (cdh_resid_std)."""
    out = C.c_double()
    check(f._L.cdh_resid_std(f._h, C.byref(out), None), f._h)
    return float(out.value)


def solve_screening_ols(G, c, xt_r_of, refinements=2):
    """Xs \\ y (src/utils.jl:70; a QR in the reference) from the Gram block G = Xs'Xs and c = Xs'y: the minimum-norm solution
    of the normal equations through the symmetric eigendecomposition of G (directions whose eigenvalue is below 1e-13 of the
    largest are left out, as a rank-revealing QR leaves them out), then refined against the normal equations' own residual
    Xs'(y - Xs b), which `xt_r_of(b)` evaluates on the device -- so that near-collinear screening columns do not cost the
    squared condition number in the fitted values."""
    w, V = np.linalg.eigh(G)
    keep = w > 1e-13 * max(w[-1], 0.0)
    Vk = V[:, keep]
    pinv = (Vk / w[keep]) @ Vk.T
    b = pinv @ c
    for _ in range(refinements):
        b = b + pinv @ xt_r_of(b)
    return b


def _as_loss(cls, X, y):
    return X if isinstance(X, _HipLoss) else cls(y, X)


def lasso(X, y, lam, omega=None, options=None):
    """lasso(X, y, λ[, ω], options) (src/lasso.jl:26-53).  X may also be an existing
    CDLeastSquaresLoss (data already resident in HBM); y is then ignored."""
    f = _as_loss(CDLeastSquaresLoss, X, y)
    x = SparseIterate(f.p)
    g = ProxL1(lam, omega)
    coordinateDescent_(x, f, g, options)
    return LassoSolution(x, f.r, g, _std_resid(f))


def sqrtLasso(X, y, lam, omega=None, options=None, standardizeX=True):
    """sqrtLasso (src/lasso.jl:62-98).  standardizeX=True uses ω = _stdX!(X), the behaviour
    the reference intends (its branch is dead on Julia >= 1.0)."""
    f = _as_loss(CDSqrtLassoLoss, X, y)
    x = SparseIterate(f.p)
    if omega is None and standardizeX:
        omega = stdX(f)
    g = ProxL1(lam, omega)
    coordinateDescent_(x, f, g, options)
    return LassoSolution(x, f.r, g, _std_resid(f))


def _find_init_sigma(f, s):
    """_findInitSigma! (src/utils.jl:60-77, 96-106): std of the OLS residuals on the s
    columns most correlated with y.  Everything n-sized stays in HBM: the screening scores
    X'y and the s x s normal equations (X_S'X_S, X_S'y) come from one pass each over the
    columns involved, the host solves the s x s system, and the residual y - X_S b is formed
    by initialize! on the device (s is 5 by default)."""
    L = f._L
    check(L.cdh_initialize(f._h, f.p, 0, None, None), f._h)  # beta = 0, r = y: X'r == X'y
    f._synced = None
    out = np.zeros(f.p)
    check(L.cdh_xt_r(f._h, _vp(out)), f._h)
    xty = np.abs(out)
    thr = np.sort(xty)[::-1][s - 1]
    S = np.nonzero(xty >= thr)[0]            # `storage .>= nlargest(s, storage)[end]`: ties kept
    if len(S) > 4096:
        raise ArgumentError("screening set larger than 4096 columns")
    idx1 = np.ascontiguousarray(S + 1, dtype=np.int64)
    G, c = np.zeros((len(S), len(S))), np.zeros(len(S))
    check(L.cdh_gram(f._h, len(S), _vp(idx1), _vp(G), _vp(c), None), f._h)

    def normal_residual(b):                  # Xs'(y - Xs b): initialize! forms the residual, one pass over the s columns dots it
        check(L.cdh_initialize(f._h, f.p, len(S), _vp(idx1), _vp(np.ascontiguousarray(b))), f._h)
        out = np.zeros(len(S))
        check(L.cdh_xt_r_cols(f._h, len(S), _vp(idx1), _vp(out)), f._h)
        return out

    coef = solve_screening_ols(G, c, normal_residual)      # Xs \ y
    check(L.cdh_initialize(f._h, f.p, len(S), _vp(idx1), _vp(np.ascontiguousarray(coef))), f._h)
    sigma = _std_resid(f)                    # std(y - Xs * (Xs \ y))
    f._synced = None
    return sigma


def scaledLasso_(x, X, y, lam, omega, options=None):
    """scaledLasso!(x, X, y, λ, ω, options) (src/lasso.jl:107-144)."""
    o = options or IterLassoOptions()
    f = _as_loss(CDLeastSquaresLoss, X, y)
    n = f.n_total
    if o.initProcedure == "Screening":
        sigma = _find_init_sigma(f, o.sinit)
    elif o.initProcedure == "InitStd":
        sigma = o.sigmainit
    elif o.initProcedure == "WarmStart":
        initialize_(f, x)
        sigma = _std_resid(f)
    else:
        raise ArgumentError("Incorrect initialization Symbol")  # :128
    g = ProxL1(lam * sigma, omega)
    for _ in range(o.maxIter):
        coordinateDescent_(x, f, g, o.optionsCD)
        s, ss = C.c_double(), C.c_double()
        check(f._L.cdh_resid_moments(f._h, C.byref(s), C.byref(ss)), f._h)
        sigmanew = float(np.sqrt(ss.value / n))  # :134
        if abs(sigmanew - sigma) / sigma < o.optTol:
            break
        sigma = sigmanew
        g = ProxL1(lam * sigma, omega)
    return LassoSolution(x, f.r, g, _std_resid(f))


@dataclass
class LassoPathResult:
    """LassoPath{T} (src/lasso.jl:201-204)."""
    lambdapath: list
    betapath: list


def LassoPath(X, Y, lambdapath, options=None, max_hat_s=np.inf, standardizeX=True, reuse_residual=True):
    """LassoPath(X, Y, λpath, options; max_hat_s, standardizeX) (src/lasso.jl:229-260):
    one x and one f shared by all λ (warm starts), βpath[i] = copy(x)."""
    f = _as_loss(CDLeastSquaresLoss, X, Y)
    sx = stdX(f) if standardizeX else np.ones(f.p)
    x = SparseIterate(f.p)
    lambdapath = list(lambdapath)
    betapath = []
    # x and f are shared by all lambdas; after the first solve the carried residual already
    # equals y - X x, so later warm starts skip the redundant initialize! (rounding-level effect)
    check(f._L.cdh_set_reuse_residual(f._h, 1 if reuse_residual else 0), f._h)
    # a path is many solves on one X: the gradient cache need not wait for evidence of that (1 -> 2 for the
    # duration of the path).  Whatever else the caller -- or CDH_GRADIENT_CACHE -- set on this loss stays: 0 is off.
    mode_before = f.gradient_cache_mode()
    if mode_before == 1:
        check(f._L.cdh_set_gradient_cache(f._h, 2), f._h)
    try:
        for i, lam in enumerate(lambdapath):
            coordinateDescent_(x, f, ProxL1(lam, sx), options)
            betapath.append(x.copy())
            if x.nnz > max_hat_s:
                lambdapath = lambdapath[: i + 1]
                break
    finally:
        check(f._L.cdh_set_reuse_residual(f._h, 0), f._h)
        if mode_before == 1:
            check(f._L.cdh_set_gradient_cache(f._h, 1), f._h)
    return LassoPathResult(lambdapath, betapath)
