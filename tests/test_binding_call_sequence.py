"""The Julia binding (julia/CoordinateDescentHIP.jl) cannot be executed in this pipeline (no `julia` in
the image), so this file REPLAYS it: every method of the binding is restated below as the exact
sequence of C-ABI calls it makes -- plain ctypes on libcdhip.so, none of coordinatedescent.jl_amd/api.py
in between -- and the reference's front-ends (src/lasso.jl) are driven on top of those methods in the
order lasso.jl calls them.  Results are compared with the oracle.  What this proves: the C entry
points, called the way the binding calls them, are sufficient and correct for lasso, sqrtLasso,
scaledLasso! (:Screening and :WarmStart), LassoPath (standardizeX) and CDWeightedLSLoss, including the
cases round 1's binding got wrong (a second y on the same matrix, a sqrt-lasso loss on a handle
created for least squares, two live loss objects on one matrix).

Each replayed method names the binding function it mirrors.  That the .jl's ccall signatures and structs agree
with include/cdhip.h is checked mechanically by tests/test_julia_binding_static.py (CPU).
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("CDHIP_SO") or os.path.join(ROOT, "coordinatedescent.jl_amd", "csrc", "libcdhip.so")
CDH_LS, CDH_SQRT, CDH_WLS = 0, 1, 2


class CdhOptions(C.Structure):       # struct CdhOptions of the binding
    _fields_ = [("maxIter", C.c_int64), ("optTol", C.c_double), ("randomize", C.c_int32),
                ("warmStart", C.c_int32), ("numSteps", C.c_int64), ("seed", C.c_uint64)]


class CdhStats(C.Structure):         # mutable struct CdhStats of the binding
    _fields_ = [("passes", C.c_int64), ("full_passes", C.c_int64), ("visits", C.c_int64),
                ("converged", C.c_int32), ("domain_error", C.c_int32), ("maxH", C.c_double),
                ("lambda_max", C.c_double)]


class DimensionMismatch(Exception):
    pass


class ArgumentError(Exception):
    pass


_L = None


def lib():
    global _L
    if _L is None:
        _L = C.CDLL(SO)
        _L.cdh_last_error.restype = C.c_char_p
        _L.cdh_last_error.argtypes = [C.c_void_p]
    return _L


def ccall(name, h, *args):
    """ccall((name, libcdhip), Int32, ...) followed by the binding's check(h, st)."""
    st = getattr(lib(), name)(h, *args)
    if st == 0:
        return
    msg = (lib().cdh_last_error(h) or b"").decode()
    if st == 1:
        raise DimensionMismatch(msg)
    if st == 2:
        raise ArgumentError(msg)
    raise RuntimeError(f"cdhip status {st}: {msg}")


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


i64, i32, f64 = C.c_int64, C.c_int32, C.c_double


# ---- the binding's types --------------------------------------------------------------------------
class HipMatrix:
    """mutable struct HipMatrix + HipMatrix(X; device): cdh_create(LS) + cdh_set_X_cols, nothing about y."""

    def __init__(self, X):
        X = np.asfortranarray(X)
        self.n, self.p = X.shape
        self.dtype = X.dtype
        h = C.c_void_p()
        st = lib().cdh_create(C.byref(h), i32(0 if X.dtype == np.float64 else 1), i32(CDH_LS), i64(self.n),
                              i64(self.n), i64(0), i64(self.p), i32(0))
        assert st == 0, lib().cdh_last_error(None)
        self.handle = h
        ccall("cdh_set_X_cols", h, i64(0), i64(self.p), ptr(X), i64(X.strides[1] // X.itemsize))
        self.owner = None

    def close(self):
        lib().cdh_destroy(self.handle)


class Loss:
    """What the reference's loss constructors build (cd_differentiable_function.jl:52-55, 128-131, 211-214):
    y, X, (w,) r = copy(y).  kind is the Julia TYPE of the loss, not a flag on the matrix."""

    def __init__(self, kind, y, X, w=None):
        if len(y) != X.n or (w is not None and len(w) != X.n):
            raise DimensionMismatch()
        self.kind, self.y, self.X, self.w, self.r = kind, y, X, w, y.copy()


class SparseIterate:
    """ProximalBase's iterate as far as the binding touches it: nzval2ind / nzval in insertion order."""

    def __init__(self, p):
        self.p, self.idx, self.val = p, [], []

    def dense(self):
        out = np.zeros(self.p)
        out[np.asarray(self.idx, dtype=np.int64) - 1] = self.val
        return out


# ---- the binding's methods, call for call ---------------------------------------------------------
def rebind(f):                       # rebind!(f)
    X = f.X
    ccall("cdh_set_loss", X.handle, i32(f.kind))
    yy = np.ascontiguousarray(f.y, dtype=X.dtype)
    ccall("cdh_set_y", X.handle, ptr(yy))
    if f.kind == CDH_WLS:
        ww = np.ascontiguousarray(f.w, dtype=X.dtype)
        ccall("cdh_set_obs_weights", X.handle, ptr(ww))
    X.owner = f.r                    # a strong reference to the loss's residual vector, compared with ===


def bind(f):                         # bind!(f)
    if f.X.owner is not f.r:
        rebind(f)


def is_synced(X, x):                 # is_synced(X, x): the handle already holds exactly this iterate
    return getattr(X, "synced", False) and list(x.idx) == X.synced_idx and [float(v) for v in x.val] == X.synced_val


def remember_iterate(X, x):          # remember_iterate!(X, x)
    X.synced_idx, X.synced_val, X.synced = list(x.idx), [float(v) for v in x.val], True


N_ITERATE_PUSHES = [0]               # cdh_set_iterate / cdh_initialize calls made by push_iterate!


def push_iterate(f, x, rebuild):     # push_iterate!(f, x, rebuild)
    if not rebuild and is_synced(f.X, x):
        return
    idx = np.ascontiguousarray(x.idx, dtype=np.int64)
    val = np.ascontiguousarray(x.val, dtype=np.float64)
    N_ITERATE_PUSHES[0] += 1
    ccall("cdh_initialize" if rebuild else "cdh_set_iterate", f.X.handle, i64(x.p), i64(len(idx)),
          ptr(idx) if len(idx) else None, ptr(val) if len(idx) else None)
    remember_iterate(f.X, x)


def pull_residual(f):                # pull_residual!(f)
    ccall("cdh_get_residual", f.X.handle, ptr(f.r))


def pull_support(f, x):              # pull_support!(f, x)
    p = f.X.p
    idx, nz, beta = np.zeros(p, dtype=np.int64), i64(0), np.zeros(p)
    ccall("cdh_get_support", f.X.handle, ptr(idx), C.byref(nz))
    ccall("cdh_get_beta", f.X.handle, ptr(beta))
    x.idx = idx[: nz.value].tolist()
    x.val = [float(beta[k - 1]) for k in x.idx]
    remember_iterate(f.X, x)
    return x


def pull_iterate(f, x):              # pull_iterate!(f, x)
    pull_support(f, x)
    pull_residual(f)
    return x


def set_penalty(f, lam0, omega):     # set_penalty!(f, g)
    om = None if omega is None else np.ascontiguousarray(omega, dtype=np.float64)
    ccall("cdh_set_penalty", f.X.handle, f64(lam0), None if om is None else ptr(om), i64(0 if om is None else len(om)))


def initialize(f, x):                # initialize!(f, x)
    bind(f)
    push_iterate(f, x, True)
    pull_residual(f)


def solve_resident(x, f, lam0, omega, opt):        # solve_resident!(x, f, g, options)
    if x.p != f.X.p:
        raise DimensionMismatch()
    bind(f)
    set_penalty(f, lam0, omega)
    push_iterate(f, x, False)
    o = CdhOptions(opt["maxIter"], opt["optTol"], int(opt["randomize"]), int(opt.get("warmStart", True)),
                   opt.get("numSteps", 50), opt.get("seed", 0))
    st = CdhStats()
    ccall("cdh_coordinate_descent", f.X.handle, C.byref(o), C.byref(st))
    assert st.domain_error == 0
    pull_support(f, x)
    return st


def coordinateDescent(x, f, lam0, omega, opt):     # coordinateDescent!(x, f, g, options)
    st = solve_resident(x, f, lam0, omega, opt)
    pull_residual(f)
    return st


N_RESIDUAL_COPIES = [0]               # how many n-sized device -> host copies the replayed methods have made
_pull_residual_plain = pull_residual


def pull_residual(f):                # noqa: F811  (the same call, counted)
    N_RESIDUAL_COPIES[0] += 1
    _pull_residual_plain(f)


def resid_moments(X):                # resid_moments(X)
    s, ss = f64(0), f64(0)
    ccall("cdh_resid_moments", X.handle, C.byref(s), C.byref(ss))
    return s.value, ss.value


def resid_std(X):                    # resid_std(X)
    out = f64(0)
    ccall("cdh_resid_std", X.handle, C.byref(out), None)
    return out.value


def descendCoordinate(f, lam0, omega, x, k):       # descendCoordinate!(f, g, x, k): x refreshed, f.r NOT copied back
    bind(f)
    set_penalty(f, lam0, omega)
    push_iterate(f, x, False)
    out = f64(0)
    ccall("cdh_descend", f.X.handle, i64(k), C.byref(out))
    pull_support(f, x)
    return out.value


def screening_ols(X, idx):           # screening_ols!(X, idx)
    m = len(idx)
    if m > 4096:
        raise ArgumentError("screening set larger than 4096 columns")
    G, c = np.zeros((m, m)), np.zeros(m)
    ccall("cdh_gram", X.handle, i64(m), ptr(idx), ptr(G), ptr(c), None)
    w, V = np.linalg.eigh(G)                      # eigen(Symmetric(G))
    keep = w > 1e-13 * max(w[-1], 0.0)
    Vk = V[:, keep]
    pinvG = (Vk / w[keep]) @ Vk.T
    coef = np.ascontiguousarray(pinvG @ c)
    res = np.zeros(m)
    for _ in range(2):
        ccall("cdh_initialize", X.handle, i64(X.p), i64(m), ptr(idx), ptr(coef))
        ccall("cdh_xt_r_cols", X.handle, i64(m), ptr(idx), ptr(res))
        coef = np.ascontiguousarray(coef + pinvG @ res)
    ccall("cdh_initialize", X.handle, i64(X.p), i64(m), ptr(idx), ptr(coef))
    X.synced = False
    return coef


def gradient_cache_mode(X):          # gradient_cache_mode(X)
    m = i32(0)
    ccall("cdh_get_gradient_cache", X.handle, C.byref(m))
    return m.value


def scaledLasso_hip(x, X, y, lam, omega, init, sinit, sigmainit, maxIter, optTol, optCD):
    """scaledLasso!(x, X::HipMatrix, y, λ, ω, options): the binding's own method (device-resident σ loop)."""
    n = X.n
    f = Loss(CDH_LS, y, X)
    if init == "Screening":
        S = findLargestCorrelations(X, y, sinit)
        screening_ols(X, np.ascontiguousarray(np.nonzero(S)[0] + 1, dtype=np.int64))
        sigma = resid_std(X)
        X.owner = f.r
    elif init == "InitStd":
        sigma = sigmainit
    else:
        bind(f)
        push_iterate(f, x, True)
        sigma = resid_std(X)
    lam0 = lam * sigma
    for _ in range(maxIter):
        solve_resident(x, f, lam0, omega, optCD)
        sigmanew = float(np.sqrt(resid_moments(X)[1] / n))
        if abs(sigmanew - sigma) / sigma < optTol:
            break
        sigma = sigmanew
        lam0 = lam * sigma
    pull_residual(f)
    return x, f.r, resid_std(X)


def LassoPath_hip(X, Y, lams, opt, standardizeX=True, max_hat_s=np.inf):
    """LassoPath(X::HipMatrix, Y, λpath, options; max_hat_s, standardizeX): the binding's own method."""
    sx = stdX(X) if standardizeX else np.ones(X.p, dtype=X.dtype)
    x = SparseIterate(X.p)
    f = Loss(CDH_LS, Y, X)
    lams = list(lams)
    path = []
    mode_before = gradient_cache_mode(X)
    ccall("cdh_set_reuse_residual", X.handle, i32(1))
    if mode_before == 1:
        ccall("cdh_set_gradient_cache", X.handle, i32(2))
    try:
        for i, lam in enumerate(lams):
            solve_resident(x, f, lam, sx, opt)
            path.append(x.dense())
            if len(x.idx) > max_hat_s:
                lams = lams[: i + 1]
                break
    finally:
        ccall("cdh_set_reuse_residual", X.handle, i32(0))
        if mode_before == 1:
            ccall("cdh_set_gradient_cache", X.handle, i32(1))
    return lams, path


def stdX(X):                         # _stdX!(out, X::HipMatrix)
    buf = np.zeros(X.p)
    ccall("cdh_col_rms", X.handle, ptr(buf))
    return buf.astype(X.dtype)


def xt_y(X, y):                      # xt_y(X, y)
    ccall("cdh_set_loss", X.handle, i32(CDH_LS))
    yy = np.ascontiguousarray(y, dtype=X.dtype)
    ccall("cdh_set_y", X.handle, ptr(yy))
    X.owner = None
    ccall("cdh_initialize", X.handle, i64(X.p), i64(0), None, None)
    X.synced = False
    out = np.zeros(X.p)
    ccall("cdh_xt_r", X.handle, ptr(out))
    return out


def findLargestCorrelations(X, y, s):            # _findLargestCorrelations(X::HipMatrix, y, s)
    storage = np.abs(xt_y(X, y))
    return storage >= np.sort(storage)[::-1][s - 1]


def findInitResiduals(X, y, s, storage):         # _findInitResiduals!(X::HipMatrix, y, s, storage)
    S = findLargestCorrelations(X, y, s)
    screening_ols(X, np.ascontiguousarray(np.nonzero(S)[0] + 1, dtype=np.int64))
    ccall("cdh_get_residual", X.handle, ptr(storage))
    return storage


# ---- the REFERENCE's front-ends (src/lasso.jl), untouched logic, on top of the methods above ---------
def lasso(X, y, lam, omega=None, opt=None):                      # lasso.jl:26-53
    x = SparseIterate(X.p)
    f = Loss(CDH_LS, y, X)
    coordinateDescent(x, f, lam, omega, opt)
    return x, f.r, float(np.std(f.r, ddof=1))


def sqrtLasso(X, y, lam, omega, opt):                            # lasso.jl:84-98
    x = SparseIterate(X.p)
    f = Loss(CDH_SQRT, y, X)
    coordinateDescent(x, f, lam, omega, opt)
    return x, f.r, float(np.std(f.r, ddof=1))


def scaledLasso(x, X, y, lam, omega, init, sinit, sigmainit, maxIter, optTol, optCD):    # lasso.jl:107-144
    n = X.n
    f = Loss(CDH_LS, y, X)
    if init == "Screening":
        sigma = float(np.std(findInitResiduals(X, y, sinit, f.r), ddof=1))     # _findInitSigma! (utils.jl:60-64)
    elif init == "InitStd":
        sigma = sigmainit
    else:
        initialize(f, x)
        sigma = float(np.std(f.r, ddof=1))
    lam0 = lam * sigma
    for _ in range(maxIter):
        coordinateDescent(x, f, lam0, omega, optCD)
        sigmanew = float(np.sqrt(np.sum(f.r.astype(np.float64) ** 2) / n))
        if abs(sigmanew - sigma) / sigma < optTol:
            break
        sigma = sigmanew
        lam0 = lam * sigma
    return x, f.r, float(np.std(f.r, ddof=1))


def LassoPath(X, Y, lams, opt, standardizeX=True):               # lasso.jl:229-260
    sx = stdX(X) if standardizeX else np.ones(X.p, dtype=X.dtype)
    x = SparseIterate(X.p)
    f = Loss(CDH_LS, Y, X)
    path = []
    for lam in lams:
        coordinateDescent(x, f, lam, sx, opt)
        path.append(x.dense())
    return path


# ---- tests ----------------------------------------------------------------------------------------
OPT = dict(maxIter=5000, optTol=1e-11, randomize=False)


def _problem(seed, n, p, s, noise=1.0):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ rng.standard_normal(s) + noise * rng.standard_normal(n)
    return rng, X, Y


def test_lasso_twice_on_one_matrix_uses_the_y_it_is_given():
    rng, X, Y1 = _problem(1, 2000, 80, 8)
    Y2 = X[:, 40:46] @ rng.standard_normal(6) + rng.standard_normal(2000)
    Xh = HipMatrix(X)
    for Y, om in ((Y1, None), (Y2, None), (Y1, rng.random(80) + 0.5)):
        x, r, sig = lasso(Xh, Y, 0.07, om, OPT)
        so = O.lasso(X, Y, 0.07, om, O.CDOptions(**OPT))
        np.testing.assert_allclose(x.dense(), so.x.dense(), rtol=0, atol=1e-10)
        np.testing.assert_allclose(r, Y - X @ x.dense(), rtol=0, atol=1e-9)      # f.r came back to the host
        np.testing.assert_allclose(sig, so.sigma, rtol=1e-9)
    Xh.close()


def test_loss_kind_follows_the_julia_type_not_the_upload():
    """A CDSqrtLassoLoss on a matrix uploaded by HipMatrix(X) (created as least squares) runs the
    sqrt-lasso update; a least-squares loss right after it runs least squares again."""
    rng, X, Y = _problem(2, 1500, 60, 6)
    Xh = HipMatrix(X)
    om = np.ones(60)
    x, r, _ = sqrtLasso(Xh, Y, 2.5, om, OPT)
    so = O.sqrtLasso(X, Y, 2.5, om, O.CDOptions(**OPT))
    np.testing.assert_allclose(x.dense(), so.x.dense(), rtol=0, atol=1e-10)
    rr = Y - X @ x.dense()
    assert max(0.0, np.max(np.abs(X.T @ rr / np.linalg.norm(rr))) - 2.5) / 2.5 < 1e-6     # test/lasso.jl:123
    x2, _, _ = lasso(Xh, Y, 0.05, None, OPT)
    np.testing.assert_allclose(x2.dense(), O.lasso(X, Y, 0.05, None, O.CDOptions(**OPT)).x.dense(), rtol=0, atol=1e-10)
    Xh.close()


def test_two_live_losses_on_one_matrix_rebind_each_other():
    rng, X, Y1 = _problem(3, 1200, 40, 5)
    Y2 = X[:, 20:25] @ rng.standard_normal(5) + rng.standard_normal(1200)
    Xh = HipMatrix(X)
    f1, f2 = Loss(CDH_LS, Y1, Xh), Loss(CDH_SQRT, Y2, Xh)
    x1, x2 = SparseIterate(40), SparseIterate(40)
    for lam1, lam2 in ((0.2, 3.0), (0.05, 2.0)):         # alternate: every call finds the other loss bound
        coordinateDescent(x1, f1, lam1, None, OPT)
        coordinateDescent(x2, f2, lam2, None, OPT)
    xo1, xo2 = O.SparseIterate(40), O.SparseIterate(40)
    fo1, fo2 = O.CDLeastSquaresLoss(Y1, X), O.CDSqrtLassoLoss(Y2, X)
    for lam1, lam2 in ((0.2, 3.0), (0.05, 2.0)):
        O.coordinateDescent_(xo1, fo1, O.ProxL1(lam1), O.CDOptions(**OPT))
        O.coordinateDescent_(xo2, fo2, O.ProxL1(lam2), O.CDOptions(**OPT))
    np.testing.assert_allclose(x1.dense(), xo1.dense(), rtol=0, atol=1e-10)
    np.testing.assert_allclose(x2.dense(), xo2.dense(), rtol=0, atol=1e-10)
    np.testing.assert_allclose(f1.r, fo1.r, rtol=0, atol=1e-9)
    Xh.close()


@pytest.mark.parametrize("standardize", [True, False])
def test_lasso_path_sequence(standardize):
    rng, X, Y = _problem(4, 1000, 300, 30)
    X *= rng.uniform(0.3, 3.0, size=300)                 # uneven column scales: stdX matters
    Xh = HipMatrix(X)
    lams = [0.3, 0.1, 0.03]
    path = LassoPath(Xh, Y, lams, OPT, standardizeX=standardize)
    np.testing.assert_allclose(stdX(Xh), np.sqrt((X * X).sum(0) / 1000), rtol=1e-13)
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(**OPT), standardizeX=standardize)
    for got, want in zip(path, bo):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    Xh.close()


@pytest.mark.parametrize("init", ["Screening", "InitStd", "WarmStart"])
def test_scaled_lasso_sequence(init):
    rng, X, Y = _problem(5, 1000, 200, 20)
    Xh = HipMatrix(X)
    om = rng.random(200) + 0.5
    x = SparseIterate(200)
    if init == "WarmStart":
        x.idx, x.val = [3, 1], [0.4, -0.2]
    optCD = dict(maxIter=5000, optTol=1e-10, randomize=False)
    xg, r, sig = scaledLasso(x, Xh, Y, 0.1, om, init, 5, 2.0, 50, 1e-6, optCD)
    xo = O.SparseIterate(200)
    if init == "WarmStart":
        xo[3] = 0.4
        xo[1] = -0.2
    so = O.scaledLasso_(xo, X, Y, 0.1, om, O.IterLassoOptions(maxIter=50, optTol=1e-6, initProcedure=init, sinit=5,
                                                             sigmainit=2.0, optionsCD=O.CDOptions(**optCD)))
    np.testing.assert_allclose(sig, so.sigma, rtol=1e-8)
    np.testing.assert_allclose(xg.dense(), xo.dense(), rtol=0, atol=1e-8)
    if init == "Screening":      # the screening init itself against the reference's formula (utils.jl:60-77, Xs \ y)
        c = np.abs(X.T @ Y)
        S = c >= np.sort(c)[::-1][4]
        coef, *_ = np.linalg.lstsq(X[:, S], Y, rcond=None)
        storage = np.zeros(1000)
        np.testing.assert_allclose(findInitResiduals(Xh, Y, 5, storage), Y - X[:, S] @ coef, rtol=0, atol=1e-9)
        assert findLargestCorrelations(Xh, Y, 5).tolist() == S.tolist()
    Xh.close()


def test_weighted_ls_loss_sequence_and_errors():
    rng, X, Y = _problem(6, 3000, 45, 5)
    w = rng.random(3000) + 0.5
    Xh = HipMatrix(X)
    f = Loss(CDH_WLS, Y, Xh, w)
    x = SparseIterate(45)
    coordinateDescent(x, f, 0.05, None, OPT)
    xo = O.SparseIterate(45)
    fo = O.CDWeightedLSLoss(Y, X, w)
    O.coordinateDescent_(xo, fo, O.ProxL1(0.05), O.CDOptions(**OPT))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-10)
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
    # a plain LS loss afterwards must not see the weights
    x2, _, _ = lasso(Xh, Y, 0.05, None, OPT)
    np.testing.assert_allclose(x2.dense(), O.lasso(X, Y, 0.05, None, O.CDOptions(**OPT)).x.dense(), rtol=0, atol=1e-10)
    with pytest.raises(DimensionMismatch):      # length(g.λ) != p (coordinate_descent.jl:14-16), from cdh_set_penalty
        coordinateDescent(SparseIterate(45), Loss(CDH_LS, Y, Xh), 0.1, np.ones(44), OPT)
    with pytest.raises(DimensionMismatch):      # numCoordinates(x) != numCoordinates(f) (:13)
        coordinateDescent(SparseIterate(44), Loss(CDH_LS, Y, Xh), 0.1, None, OPT)
    with pytest.raises(DimensionMismatch):      # the loss constructor's own check (:53)
        Loss(CDH_LS, Y[:-1], Xh)
    Xh.close()


def test_fp32_path_sequence():
    """LassoPath is the front-end that works in Float32 in the reference (SparseIterate(T, p), quirk Q3)."""
    rng, X, Y = _problem(7, 4000, 64, 8)
    Xh = HipMatrix(X.astype(np.float32))
    path = LassoPath(Xh, Y.astype(np.float32), [0.2, 0.05], dict(maxIter=500, optTol=1e-6, randomize=False))
    lo, bo = O.LassoPath(X, Y, [0.2, 0.05], O.CDOptions(maxIter=500, optTol=1e-10, randomize=False))
    np.testing.assert_allclose(path[1], bo[1], rtol=0, atol=2e-4)     # fp32 storage against the fp64 oracle
    Xh.close()


@pytest.mark.parametrize("init", ["Screening", "InitStd", "WarmStart"])
def test_scaled_lasso_hipmatrix_method_keeps_the_sigma_loop_on_the_device(init):
    """scaledLasso!(x, X::HipMatrix, ...) of the binding: same sigma, beta and residual as the reference's loop,
    with ONE n-sized copy back (the LassoSolution's residuals) however many sigma iterations run."""
    rng, X, Y = _problem(15, 1000, 200, 20)
    Xh = HipMatrix(X)
    om = rng.random(200) + 0.5
    x = SparseIterate(200)
    if init == "WarmStart":
        x.idx, x.val = [3, 1], [0.4, -0.2]
    optCD = dict(maxIter=5000, optTol=1e-10, randomize=False)
    N_RESIDUAL_COPIES[0] = 0
    xg, r, sig = scaledLasso_hip(x, Xh, Y, 0.1, om, init, 5, 2.0, 50, 1e-6, optCD)
    assert N_RESIDUAL_COPIES[0] == 1
    xo = O.SparseIterate(200)
    if init == "WarmStart":
        xo[3] = 0.4
        xo[1] = -0.2
    so = O.scaledLasso_(xo, X, Y, 0.1, om, O.IterLassoOptions(maxIter=50, optTol=1e-6, initProcedure=init, sinit=5,
                                                             sigmainit=2.0, optionsCD=O.CDOptions(**optCD)))
    np.testing.assert_allclose(sig, so.sigma, rtol=1e-8)
    np.testing.assert_allclose(xg.dense(), xo.dense(), rtol=0, atol=1e-8)
    np.testing.assert_allclose(r, Y - X @ xg.dense(), rtol=0, atol=1e-8)
    # a plain lasso on the same matrix afterwards binds its own y (the owner reference is this call's f.r)
    Y2 = X[:, 50:55] @ rng.standard_normal(5) + rng.standard_normal(1000)
    x2, _, _ = lasso(Xh, Y2, 0.05, None, OPT)
    np.testing.assert_allclose(x2.dense(), O.lasso(X, Y2, 0.05, None, O.CDOptions(**OPT)).x.dense(), rtol=0, atol=1e-10)
    Xh.close()


def test_lasso_path_hipmatrix_method_moves_nothing_n_sized_and_restores_the_cache_mode():
    rng, X, Y = _problem(16, 4000, 300, 30)
    X *= rng.uniform(0.3, 3.0, size=300)
    Xh = HipMatrix(X)
    lams = [0.3, 0.2, 0.1, 0.05, 0.03]
    N_RESIDUAL_COPIES[0] = 0
    got_lams, path = LassoPath_hip(Xh, Y, lams, OPT)
    assert N_RESIDUAL_COPIES[0] == 0 and gradient_cache_mode(Xh) == 1
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(**OPT))
    for got, want in zip(path, bo):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    # max_hat_s stops the path early and truncates lambdapath (lasso.jl:253-256)
    got_lams, path = LassoPath_hip(Xh, Y, lams, OPT, max_hat_s=int(np.count_nonzero(bo[1])))
    assert len(got_lams) == len(path) < len(lams)
    # a caller who switched the cache off keeps it off through the path
    ccall("cdh_set_gradient_cache", Xh.handle, i32(0))
    LassoPath_hip(Xh, Y, lams[:2], OPT)
    assert gradient_cache_mode(Xh) == 0
    Xh.close()


def test_user_written_pass_over_descend_coordinate_copies_no_residual():
    """A caller driving the operator interface directly -- the reference's own `_cdPass!` (coordinate_descent.jl:94-110:
    `for k in it: h = descendCoordinate!(f, g, x, k)`, then dropzeros!) written against the binding -- must not pay an
    n-sized copy per visit (80 MB at cfg2) nor an upload of the iterate it was just handed: zero residual copies and
    zero pushes inside the loop; f.r is current again after one explicit pull_residual!."""
    rng, X, Y = _problem(17, 3000, 40, 6)
    Xh = HipMatrix(X)
    f, x = Loss(CDH_LS, Y, Xh), SparseIterate(40)
    fo, xo, go = O.CDLeastSquaresLoss(Y, X), O.SparseIterate(40), O.ProxL1(0.06)
    initialize(f, x)
    O.initialize_(fo, xo)
    N_RESIDUAL_COPIES[0] = 0
    pushes0 = N_ITERATE_PUSHES[0]
    for _ in range(6):                                   # six full passes, visit by visit
        maxh = 0.0
        for k in range(1, 41):
            maxh = max(maxh, abs(descendCoordinate(f, 0.06, None, x, k)))
        mo = O.cdPass_(xo, fo, go, range(1, 41))
        np.testing.assert_allclose(maxh, mo, rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-12)
    assert N_RESIDUAL_COPIES[0] == 0 and N_ITERATE_PUSHES[0] == pushes0
    pull_residual(f)
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-10)
    Xh.close()


def test_lasso_path_hipmatrix_method_keeps_the_carried_residual():
    """solve_resident! does not push an iterate the handle already holds (it just pulled it): after the first lambda no
    cdh_set_iterate is issued, so r stays consistent and `cdh_set_reuse_residual` really skips the rebuilds."""
    rng, X, Y = _problem(18, 3000, 120, 10)
    Xh = HipMatrix(X)
    pushes0 = N_ITERATE_PUSHES[0]
    lams = [0.3, 0.2, 0.1, 0.05]
    _, path = LassoPath_hip(Xh, Y, lams, OPT, standardizeX=False)
    assert N_ITERATE_PUSHES[0] - pushes0 <= 1          # the empty x of the first lambda at most
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(**OPT), standardizeX=False)
    for got, want in zip(path, bo):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)
    Xh.close()


def test_screening_init_with_100_columns_and_with_collinear_ones():
    """`_findInitResiduals!` takes any s and solves Xs \\ y by QR (utils.jl:65-77).  (1) sinit = 100: the Gram block comes
    in pairs of 32-column groups.  (2) Two screening columns correlated to 1 - 1e-8: the normal equations alone would
    square a condition number of ~1e8; eigen-cutoff + refinement gives sigma_init as numpy.linalg.lstsq does."""
    rng, X, Y = _problem(19, 4000, 300, 120, noise=2.0)
    Xh = HipMatrix(X)
    storage = np.zeros(4000)
    got = findInitResiduals(Xh, Y, 100, storage).copy()
    c = np.abs(X.T @ Y)
    S = c >= np.sort(c)[::-1][99]
    coef, *_ = np.linalg.lstsq(X[:, S], Y, rcond=None)
    np.testing.assert_allclose(got, Y - X[:, S] @ coef, rtol=0, atol=1e-9)
    Xh.close()
    # collinear pair inside the screening set
    X2 = X.copy()
    X2[:, 1] = X2[:, 0] + np.sqrt(2e-8) * rng.standard_normal(4000) * np.linalg.norm(X2[:, 0]) / np.sqrt(4000)   # 1 - corr ~ 1e-8
    Y2 = X2[:, :8] @ (3.0 + rng.standard_normal(8)) + rng.standard_normal(4000)
    corr = X2[:, 0] @ X2[:, 1] / np.linalg.norm(X2[:, 0]) / np.linalg.norm(X2[:, 1])
    assert 0.3e-8 < 1 - corr < 3e-8
    Xh = HipMatrix(np.asfortranarray(X2))
    c = np.abs(X2.T @ Y2)
    S = c >= np.sort(c)[::-1][9]
    assert S[0] and S[1]
    res = findInitResiduals(Xh, Y2, 10, np.zeros(4000)).copy()
    coef, *_ = np.linalg.lstsq(X2[:, S], Y2, rcond=None)
    want = Y2 - X2[:, S] @ coef
    np.testing.assert_allclose(np.std(res, ddof=1), np.std(want, ddof=1), rtol=1e-8)
    np.testing.assert_allclose(res, want, rtol=0, atol=1e-6 * np.std(want))
    Xh.close()


def test_resid_std_is_two_pass():
    """std(f.r) with a mean large against the spread: the one-pass (ss - s^2/n)/(n-1) form loses every digit; cdh_resid_std
    centres on the device first, as Statistics.std does."""
    rng = np.random.default_rng(20)
    n = 20000
    X = np.asfortranarray(rng.standard_normal((n, 4)))
    Y = 1e8 + 1e-3 * rng.standard_normal(n)
    Xh = HipMatrix(X)
    f, x = Loss(CDH_LS, Y, Xh), SparseIterate(4)
    initialize(f, x)                                     # r = y
    np.testing.assert_allclose(resid_std(Xh), np.std(Y, ddof=1), rtol=1e-6)
    Xh.close()
