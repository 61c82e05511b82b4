"""CPU oracle for the coordinate-descent hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``coordinatedescent.jl_amd``) never
does; it fails loudly when its HIP library is missing instead of falling back.

``cd_oracle.c`` is a plain-C restatement of the reference
(mlakolar/CoordinateDescent.jl v0.3.0: src/coordinate_descent.jl,
src/cd_differentiable_function.jl, src/atom_iterator.jl, src/utils.jl:127-138,
plus the ProximalBase 0.3.0 contract of SURVEY.md Appendix B).  The reference is
Julia and cannot run in this image; the oracle is pinned by the reference's own
known-answer test and test properties and by scikit-learn (see cd_oracle.c's
header and tests/test_oracle_*.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "cd_oracle.c")
_OUT_DIR = os.path.join(_HERE, "_build")
_SO = os.path.join(_OUT_DIR, "libcd_oracle.so")

LS, SQRT, WLS, QUAD = 0, 1, 2, 3


def build(force: bool = False) -> str:
    """Compile cd_oracle.c -> oracle/_build/libcd_oracle.so (gcc, x86-64-v3)."""
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(_SRC):
        return _SO
    os.makedirs(_OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-std=c11",
           "-Wall", "-Wextra", "-o", _SO, _SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return _SO


_lib = None


class _Options(C.Structure):
    _fields_ = [("maxIter", C.c_int64), ("optTol", C.c_double), ("randomize", C.c_int32),
                ("warmStart", C.c_int32), ("numSteps", C.c_int64), ("seed", C.c_uint64)]


class _Stats(C.Structure):
    _fields_ = [("passes", C.c_int64), ("full_passes", C.c_int64), ("visits", C.c_int64),
                ("converged", C.c_int32), ("last_maxH", C.c_double)]


class _Prox(C.Structure):
    _fields_ = [("lambda0", C.c_double), ("omega", C.c_void_p)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    try:
        L = C.CDLL(build())
    except OSError:
        L = C.CDLL(build(force=True))
    vp, i64, f64, i32 = C.c_void_p, C.c_int64, C.c_double, C.c_int32
    sig = {
        "cdo_iterate_new": (vp, [i64]), "cdo_iterate_free": (None, [vp]),
        "cdo_iterate_length": (i64, [vp]), "cdo_iterate_nnz": (i64, [vp]),
        "cdo_iterate_get": (f64, [vp, i64]), "cdo_iterate_set": (None, [vp, i64, f64]),
        "cdo_iterate_support": (None, [vp, vp]), "cdo_iterate_dense": (None, [vp, vp]),
        "cdo_iterate_fill_zero": (None, [vp]), "cdo_iterate_dropzeros": (None, [vp]),
        "cdo_iterate_copy": (None, [vp, vp]),
        "cdo_loss_new": (vp, [C.c_int, i64, i64, vp, i64, vp, vp]),
        "cdo_loss_free": (None, [vp]), "cdo_loss_residual": (vp, [vp]),
        "cdo_num_coordinates": (i64, [vp]), "cdo_loss_domain_error": (i32, [vp]),
        "cdo_initialize": (None, [vp, vp]), "cdo_gradient": (f64, [vp, vp, i64]),
        "cdo_descend": (f64, [vp, vp, vp, i64]),
        "cdo_iter_new": (vp, [i64, C.c_int, C.c_uint64]), "cdo_iter_free": (None, [vp]),
        "cdo_iter_reset": (None, [vp, vp, C.c_int]), "cdo_iter_collect": (i64, [vp, vp, vp]),
        "cdo_pass": (f64, [vp, vp, vp, i64, vp]),
        "cdo_lambda_max": (f64, [vp, vp, vp]),
        "cdo_coordinate_descent": (i32, [vp, vp, i64, vp, vp, vp]),
        "cdo_std_x": (None, [i64, i64, vp, i64, vp]),
        "cdo_objective": (f64, [vp, vp, vp]),
        "cdo_bench_ls_visits": (f64, [i64, i64, vp, i64, vp, vp, f64, i64, i32]),
        "cdo_max_threads": (i32, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class CDOptions:
    """CDOptions (src/utils.jl:7-20) + seed for the substitute RNG."""
    maxIter: int = 2000
    optTol: float = 1e-7
    randomize: bool = True
    warmStart: bool = True
    numSteps: int = 50
    seed: int = 0

    def _c(self):
        return _Options(self.maxIter, self.optTol, int(self.randomize), int(self.warmStart),
                        self.numSteps, self.seed)


class ProxL1:
    """ProxL1(lambda0[, omega]) (ProximalBase 0.3.0)."""

    def __init__(self, lambda0, omega=None):
        self.lambda0 = float(lambda0)
        self.omega = None if omega is None else _f64(omega).copy()
        self._c = _Prox(self.lambda0, _ptr(self.omega))

    def with_lambda0(self, lambda0):
        return ProxL1(lambda0, self.omega)


class SparseIterate:
    """SparseIterate(p) (ProximalBase 0.3.0 contract)."""

    def __init__(self, p, values=None):
        self._h = lib().cdo_iterate_new(int(p))
        self.p = int(p)
        if values is not None:
            for k, v in enumerate(np.asarray(values, dtype=np.float64)):
                if v != 0.0:
                    self[k + 1] = v

    def __del__(self):
        try:
            lib().cdo_iterate_free(self._h)
        except Exception:
            pass

    def __len__(self):
        return self.p

    def __getitem__(self, k1):  # 1-based, as in the reference
        return lib().cdo_iterate_get(self._h, int(k1))

    def __setitem__(self, k1, v):
        lib().cdo_iterate_set(self._h, int(k1), float(v))

    @property
    def nnz(self):
        return lib().cdo_iterate_nnz(self._h)

    @property
    def nzval2ind(self):
        out = np.zeros(max(self.nnz, 1), dtype=np.int64)
        lib().cdo_iterate_support(self._h, _ptr(out))
        return out[: self.nnz]

    def dense(self):
        out = np.zeros(self.p)
        lib().cdo_iterate_dense(self._h, _ptr(out))
        return out

    def fill_zero(self):
        lib().cdo_iterate_fill_zero(self._h)

    def dropzeros(self):
        lib().cdo_iterate_dropzeros(self._h)

    def copy(self):
        y = SparseIterate(self.p)
        lib().cdo_iterate_copy(y._h, self._h)
        return y


class Loss:
    """One of the reference's CoordinateDifferentiableFunction subtypes."""

    def __init__(self, kind, y, X, w=None):
        self.kind = kind
        self.X = np.asfortranarray(X, dtype=np.float64)
        self.y = _f64(y)
        self.w = None if w is None else _f64(w)
        if kind == QUAD:
            p = self.y.shape[0]
            # issymmetric(A) && length(b)==size(A,2) (cd_differentiable_function.jl:306)
            if self.X.shape != (p, p) or not np.array_equal(self.X, self.X.T):
                raise ValueError("ArgumentError")
            n = p
        else:
            n, p = self.X.shape
            if self.y.shape[0] != n or (self.w is not None and self.w.shape[0] != n):
                raise ValueError("DimensionMismatch")  # :53,129,212
        self.n, self.p = n, p
        self._h = lib().cdo_loss_new(kind, n, p, _ptr(self.X), self.X.shape[0], _ptr(self.y),
                                     _ptr(self.w))

    def __del__(self):
        try:
            lib().cdo_loss_free(self._h)
        except Exception:
            pass

    @property
    def r(self):
        m = self.p if self.kind == QUAD else self.n
        addr = lib().cdo_loss_residual(self._h)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_double)), shape=(m,))

    @property
    def domain_error(self):
        return bool(lib().cdo_loss_domain_error(self._h))


def CDLeastSquaresLoss(y, X):
    return Loss(LS, y, X)


def CDSqrtLassoLoss(y, X):
    return Loss(SQRT, y, X)


def CDWeightedLSLoss(y, X, w):
    return Loss(WLS, y, X, w)


def CDQuadraticLoss(A, b):
    return Loss(QUAD, b, A)


def numCoordinates(f):
    return lib().cdo_num_coordinates(f._h)


def initialize_(f, x):
    lib().cdo_initialize(f._h, x._h)


def gradient(f, x, k1):
    return lib().cdo_gradient(f._h, x._h, int(k1))


def descendCoordinate_(f, g, x, k1):
    return lib().cdo_descend(f._h, C.byref(g._c), x._h, int(k1))


def cdPass_(x, f, g, idx1):
    idx = np.ascontiguousarray(idx1, dtype=np.int64)
    return lib().cdo_pass(f._h, C.byref(g._c), x._h, idx.shape[0], _ptr(idx))


def findLambdaMax(x, f, g):
    return lib().cdo_lambda_max(f._h, x._h, C.byref(g._c))


def coordinateDescent_(x, f, g, options=None):
    """coordinateDescent!(x, f, g::ProxL1, options) (src/coordinate_descent.jl:7-39).
    Returns a stats dict (the reference returns x and reports nothing, Q7)."""
    o = (options or CDOptions())._c()
    st = _Stats()
    n_omega = 0 if g.omega is None else g.omega.shape[0]
    rc = lib().cdo_coordinate_descent(f._h, C.byref(g._c), n_omega, x._h, C.byref(o), C.byref(st))
    if rc == 1:
        raise ValueError("DimensionMismatch")
    if rc == 2:
        raise ValueError("ArgumentError: cold-start lambda range has zero step")
    return {"passes": st.passes, "full_passes": st.full_passes, "visits": st.visits,
            "converged": bool(st.converged), "maxH": st.last_maxH}


def objective(f, g, x):
    return lib().cdo_objective(f._h, C.byref(g._c), x._h)


def stdX(X):
    """_stdX!(out, X) (src/utils.jl:127-138)."""
    X = np.asfortranarray(X, dtype=np.float64)
    out = np.zeros(X.shape[1])
    lib().cdo_std_x(X.shape[0], X.shape[1], _ptr(X), X.shape[0], _ptr(out))
    return out


class Iterator:
    """OrderedIterator / RandomIterator (src/atom_iterator.jl)."""

    def __init__(self, x, randomize=False, seed=0):
        self.x = x
        self._h = lib().cdo_iter_new(x.p, int(randomize), seed)

    def __del__(self):
        try:
            lib().cdo_iter_free(self._h)
        except Exception:
            pass

    def reset(self, fullPass):
        lib().cdo_iter_reset(self._h, self.x._h, int(fullPass))

    def collect(self):
        out = np.zeros(max(self.x.p, 1), dtype=np.int64)
        m = lib().cdo_iter_collect(self._h, self.x._h, _ptr(out))
        return out[:m].copy()


# ---------------------------------------------------------------------------
# Front-ends (src/lasso.jl) restated on top of the C core.
# ---------------------------------------------------------------------------
@dataclass
class LassoSolution:  # src/lasso.jl:7-17
    x: SparseIterate
    residuals: np.ndarray
    penalty: ProxL1
    sigma: float


def _std(r):  # Statistics.std: mean-removed, Bessel-corrected
    return float(np.std(r, ddof=1))


def lasso(X, y, lam, omega=None, options=None):
    """lasso(X, y, λ[, ω], options) (src/lasso.jl:26-53)."""
    x = SparseIterate(np.shape(X)[1])
    f = CDLeastSquaresLoss(y, X)
    g = ProxL1(lam, omega)
    coordinateDescent_(x, f, g, options)
    return LassoSolution(x, f.r.copy(), g, _std(f.r))


def sqrtLasso(X, y, lam, omega=None, options=None, standardizeX=True):
    """sqrtLasso (src/lasso.jl:62-98); standardizeX=True uses the intended
    ω = _stdX!(X) (the reference's branch is dead on Julia >= 1.0, quirk Q1)."""
    x = SparseIterate(np.shape(X)[1])
    f = CDSqrtLassoLoss(y, X)
    if omega is None and standardizeX:
        omega = stdX(X)
    g = ProxL1(lam, omega)
    coordinateDescent_(x, f, g, options)
    return LassoSolution(x, f.r.copy(), g, _std(f.r))


@dataclass
class IterLassoOptions:  # src/utils.jl:24-39
    maxIter: int = 20
    optTol: float = 1e-2
    initProcedure: str = "Screening"
    sinit: int = 5
    sigmainit: float = 1.0
    optionsCD: CDOptions = None


def findInitResiduals(X, y, s):
    """_findInitResiduals! (src/utils.jl:65-77, 96-106): top-s |X'y| columns
    (ties included, `storage .>= nlargest(s)[end]`), OLS on them, residuals."""
    X = np.asarray(X, dtype=np.float64)
    c = np.abs(X.T @ y)
    thr = np.sort(c)[::-1][s - 1]
    S = c >= thr
    Xs = X[:, S]
    coef, *_ = np.linalg.lstsq(Xs, y, rcond=None)
    return y - Xs @ coef


def scaledLasso_(x, X, y, lam, omega, options=None):
    """scaledLasso! (src/lasso.jl:107-144)."""
    o = options or IterLassoOptions()
    ocd = o.optionsCD or CDOptions()
    n = np.shape(X)[0]
    f = CDLeastSquaresLoss(y, X)
    if o.initProcedure == "Screening":
        f.r[:] = findInitResiduals(X, np.asarray(y, dtype=np.float64), o.sinit)
        sigma = _std(f.r)
    elif o.initProcedure == "InitStd":
        sigma = o.sigmainit
    elif o.initProcedure == "WarmStart":
        initialize_(f, x)
        sigma = _std(f.r)
    else:
        raise ValueError("ArgumentError: Incorrect initialization Symbol")
    g = ProxL1(lam * sigma, omega)
    for _ in range(o.maxIter):
        coordinateDescent_(x, f, g, ocd)
        sigmanew = float(np.sqrt(np.sum(f.r ** 2) / n))
        if abs(sigmanew - sigma) / sigma < o.optTol:
            break
        sigma = sigmanew
        g = ProxL1(lam * sigma, omega)
    return LassoSolution(x, f.r.copy(), g, _std(f.r))


def LassoPath(X, Y, lambdas, options=None, max_hat_s=np.inf, standardizeX=True):
    """LassoPath (src/lasso.jl:229-260): returns (λpath, [dense β per λ])."""
    p = np.shape(X)[1]
    sx = stdX(X) if standardizeX else np.ones(p)
    x = SparseIterate(p)
    f = CDLeastSquaresLoss(Y, X)
    lambdas = list(lambdas)
    betas = []
    for i, lam in enumerate(lambdas):
        coordinateDescent_(x, f, ProxL1(lam, sx), options)
        betas.append(x.dense())
        if x.nnz > max_hat_s:
            lambdas = lambdas[: i + 1]
            break
    return lambdas, betas
