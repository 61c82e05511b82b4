"""Pins the CPU oracle (oracle/cd_oracle.c) against everything the reference's own
tests hold for the hot path: the one analytic known-answer test and every
RNG-agnostic property, plus scikit-learn's Lasso (same objective)
and the numpy twin.  The reference draws its inputs from Julia's RNG, which cannot
be regenerated here, so each case re-draws data of the same shape with numpy.
"""
import numpy as np
import pytest

import oracle as O
from oracle import np_twin


def _problem(rng, n, p, s, noise=1.0, beta=None):
    X = rng.standard_normal((n, p))
    b = rng.standard_normal(s) if beta is None else beta
    Y = X[:, :s] @ b + noise * rng.standard_normal(n)
    return np.asfortranarray(X), Y


def _sprand(rng, p, density):
    v = rng.random(p)
    return np.where(rng.random(p) < density, v, 0.0)


# -- reference test/coordinate_descent.jl:13-25 (the only analytic KAT) -----------
def test_small_proxl1_known_answer():
    Y = np.array([1.0, 1.5])
    g = O.ProxL1(1.2)
    f = O.CDQuadraticLoss(np.eye(2), -Y)
    x = O.SparseIterate(2)
    O.coordinateDescent_(x, f, g, O.CDOptions(maxIter=100, optTol=1e-8, warmStart=True,
                                              randomize=False))
    np.testing.assert_allclose(x.dense(), [0.0, 0.3], rtol=0, atol=1e-15)


# -- reference test/coordinate_descent.jl:29-63 ------------------------------------
@pytest.mark.parametrize("weighted", [False, True])
def test_four_option_agreement_ls(weighted):
    rng = np.random.default_rng(11 + weighted)
    n, p, s = 500, 50, 10 if weighted else 5
    X, Y = _problem(rng, n, p, s)
    lam = 0.01 if weighted else 0.02
    tol = 1e-8 if weighted else 1e-12
    g = O.ProxL1(lam, rng.random(p) if weighted else None)  # :65-99 uses ω = rand(p)
    f = O.CDLeastSquaresLoss(Y, X)
    sols = []
    for warm in (True, False):
        for rand in (False, True):
            x = O.SparseIterate(p, _sprand(rng, p, 0.6))
            st = O.coordinateDescent_(x, f, g, O.CDOptions(maxIter=5000, optTol=tol,
                                                           warmStart=warm, randomize=rand, seed=3))
            assert st["converged"]
            sols.append(x.dense())
    for b in sols[1:]:
        np.testing.assert_allclose(b, sols[0], rtol=0, atol=1e-5)


# -- reference test/lasso.jl:23-34 ----------------------------------------------------
def test_lasso_zero_above_lambda_max():
    rng = np.random.default_rng(1)
    n, p = 100, 10
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X @ np.ones(p) + 0.1 * rng.standard_normal(n)
    lam = np.max(np.abs(X.T @ Y / n)) + 0.1
    out = O.lasso(X, Y, lam)
    assert out.x.nnz == 0
    assert np.all(out.x.dense() == 0.0)


# -- reference test/lasso.jl:36-56 ----------------------------------------------------
def test_lasso_nonzero_ls_equals_covariance_form_and_kkt():
    rng = np.random.default_rng(2)
    n, p, s = 100, 10, 5
    X, Y = _problem(rng, n, p, s, noise=0.1, beta=np.ones(s))
    lam = np.full(p, 0.3)
    beta = O.lasso(X, Y, 1.0, lam, O.CDOptions(optTol=1e-12))
    f = O.CDQuadraticLoss(X.T @ X / n, -X.T @ Y / n)
    x1 = O.SparseIterate(p)
    O.coordinateDescent_(x1, f, O.ProxL1(1.0, lam), O.CDOptions(optTol=1e-12))
    np.testing.assert_allclose(beta.x.dense(), x1.dense(), rtol=0, atol=1e-5)
    kkt = np.max(np.abs(X.T @ (Y - X @ beta.x.dense()) / n))
    assert abs((kkt - 0.3) / 0.3) < 1e-5


# -- reference test/lasso.jl:58-72 ----------------------------------------------------
def test_lasso_interfaces_unweighted_equals_unit_weights():
    rng = np.random.default_rng(3)
    n = p = 500
    X, Y = _problem(rng, n, p, 50)
    x1 = O.lasso(X, Y, 0.1, options=O.CDOptions(seed=1))
    x2 = O.lasso(X, Y, 0.1, np.ones(p), options=O.CDOptions(seed=2))
    np.testing.assert_allclose(x1.x.dense(), x2.x.dense(), rtol=0, atol=1e-5)


# -- reference test/lasso.jl:76-101 + sklearn + numpy twin ----------------------------
def test_cd_lasso_vs_covariance_kkt_sklearn_and_twin():
    from sklearn.linear_model import Lasso

    rng = np.random.default_rng(4)
    n, p, s = 200, 50, 10
    X, Y = _problem(rng, n, p, s, noise=0.1)
    g = O.ProxL1(0.2)
    x1, x2 = O.SparseIterate(p), O.SparseIterate(p)
    O.coordinateDescent_(x1, O.CDQuadraticLoss(X.T @ X / n, -X.T @ Y / n), g,
                         O.CDOptions(optTol=1e-12))
    f2 = O.CDLeastSquaresLoss(Y, X)
    st = O.coordinateDescent_(x2, f2, g, O.CDOptions(optTol=1e-12, randomize=False))
    assert np.max(np.abs(x1.dense() - x2.dense())) < 1e-5
    for x in (x1, x2):
        kkt = np.max(np.abs(X.T @ (Y - X @ x.dense()) / n))
        assert abs((kkt - 0.2) / 0.2) < 1e-5
    # independent solver, identical objective ||y-Xb||^2/(2n) + lam*||b||_1
    sk = Lasso(alpha=0.2, fit_intercept=False, tol=1e-15, max_iter=100000).fit(X, Y)
    np.testing.assert_allclose(x2.dense(), sk.coef_, rtol=0, atol=1e-12)
    # numpy twin follows the same trajectory (ordered iterator)
    bt, rt, passes, trace = np_twin.solve("ls", X, Y, 0.2, optTol=1e-12)
    np.testing.assert_allclose(x2.dense(), bt, rtol=0, atol=1e-13)
    assert passes == st["passes"]
    np.testing.assert_allclose(f2.r, rt, rtol=0, atol=1e-12)
    np.testing.assert_allclose(f2.r, Y - X @ x2.dense(), rtol=0, atol=1e-12)


# -- reference test/lasso.jl:106-125 --------------------------------------------------
def test_sqrt_lasso_kkt():
    rng = np.random.default_rng(5)
    n, p, s = 100, 50, 5
    X, Y = _problem(rng, n, p, s)
    lam = 2.8
    x1 = O.SparseIterate(p)
    f = O.CDSqrtLassoLoss(Y, X)
    O.coordinateDescent_(x1, f, O.ProxL1(lam), O.CDOptions(maxIter=5000, optTol=1e-8))
    r = Y - X @ x1.dense()
    assert max(0.0, np.max(np.abs(X.T @ r / np.linalg.norm(r))) - lam) / lam < 1e-3
    assert not f.domain_error


# -- reference test/lasso.jl:127-181 --------------------------------------------------
def test_sqrt_lasso_interfaces():
    rng = np.random.default_rng(6)
    n = p = 500
    X, Y = _problem(rng, n, p, 50)
    lam = 1.5
    opts = [O.CDOptions(maxIter=5000, optTol=1e-10, warmStart=w, randomize=r, seed=9)
            for w in (True, False) for r in (False, True)]
    f = O.CDSqrtLassoLoss(Y, X)
    ref = None
    for o in opts:
        x = O.SparseIterate(p, _sprand(rng, p, 0.6))
        O.coordinateDescent_(x, f, O.ProxL1(lam), o)
        ref = x.dense() if ref is None else ref
        np.testing.assert_allclose(x.dense(), ref, rtol=0, atol=1e-4)
        y1 = O.sqrtLasso(X, Y, lam, options=o, standardizeX=False)
        np.testing.assert_allclose(y1.x.dense(), ref, rtol=0, atol=1e-4)
        z1 = O.sqrtLasso(X, Y, lam, np.ones(p), o)
        np.testing.assert_allclose(z1.x.dense(), ref, rtol=0, atol=1e-4)
    # sqrt-lasso fixed point == lasso at lambda_L = lam*||r||/n
    r = Y - X @ ref
    lamL = lam * np.linalg.norm(r) / n
    xl = O.lasso(X, Y, lamL, options=O.CDOptions(maxIter=20000, optTol=1e-12, randomize=False))
    np.testing.assert_allclose(xl.x.dense(), ref, rtol=0, atol=1e-6)
    # numpy twin agreement on the sqrt update itself
    bt, _, _, _ = np_twin.solve("sqrt", X, Y, lam, maxIter=5000, optTol=1e-10)
    np.testing.assert_allclose(bt, ref, rtol=0, atol=1e-4)


# -- reference test/lasso.jl:186-216 --------------------------------------------------
def test_scaled_lasso():
    rng = np.random.default_rng(7)
    n, p, s = 1000, 500, 50
    X, Y = _problem(rng, n, p, s)
    lam = 0.12
    cd = O.CDOptions(maxIter=5000, optTol=1e-8, seed=4)
    o1 = O.IterLassoOptions(maxIter=100, optTol=1e-8, optionsCD=cd)
    o2 = O.IterLassoOptions(maxIter=100, optTol=1e-8, initProcedure="InitStd", sigmainit=2.0,
                            optionsCD=cd)
    x1, x2 = O.SparseIterate(p), O.SparseIterate(p)
    s1 = O.scaledLasso_(x1, X, Y, lam, np.ones(p), o1)
    s2 = O.scaledLasso_(x2, X, Y, lam, np.ones(p), o2)
    for x, sol in ((x1, s1), (x2, s2)):
        kkt = np.max(np.abs(X.T @ (Y - X @ x.dense()) / n))
        assert max(kkt - lam * sol.sigma, 0.0) / (sol.sigma * lam) < 1e-4
    np.testing.assert_allclose(x1.dense(), x2.dense(), rtol=0, atol=1e-4)


# -- reference test/lasso.jl:220-288 --------------------------------------------------
@pytest.mark.parametrize("standardize", [False, True])
def test_lasso_path(standardize):
    rng = np.random.default_rng(8)
    n, p, s = 1000, 500, 50
    X, Y = _problem(rng, n, p, s)
    opt = O.CDOptions(maxIter=5000, optTol=1e-8, seed=5)
    load = O.stdX(X) if standardize else None
    np.testing.assert_allclose(O.stdX(X), np.sqrt((X ** 2).sum(0) / n), rtol=1e-14)
    x1 = O.lasso(X, Y, 0.3, load, opt)
    x2 = O.lasso(X, Y, 0.1, load, opt)
    lams, betas = O.LassoPath(X, Y, [0.3, 0.1], opt, standardizeX=standardize)
    assert lams == [0.3, 0.1]
    np.testing.assert_allclose(betas[0], x1.x.dense(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(betas[1], x2.x.dense(), rtol=0, atol=1e-5)
    # max_hat_s early stop (src/lasso.jl:253-256)
    lams2, betas2 = O.LassoPath(X, Y, [0.3, 0.1], opt, max_hat_s=1, standardizeX=standardize)
    assert lams2 == [0.3] and len(betas2) == 1


# -- reference test/atom_iterator.jl:11-48 ---------------------------------------------
def test_ordered_iterator():
    x = O.SparseIterate(5)
    x[2] = 1.0
    x[1] = 2.0
    it = O.Iterator(x, randomize=False)
    assert it.collect().tolist() == [1, 2, 3, 4, 5]
    it.reset(True)
    assert it.collect().tolist() == [1, 2, 3, 4, 5]
    it.reset(False)
    assert it.collect().tolist() == [2, 1]  # support is in insertion order
    assert x.nzval2ind.tolist() == [2, 1]


# -- reference test/atom_iterator.jl:50-85 ---------------------------------------------
def test_random_iterator():
    rng = np.random.default_rng(9)
    p, s = 50, 10
    x = O.SparseIterate(p)
    for _ in range(s):
        x[int(rng.integers(1, p + 1))] = float(rng.standard_normal())
    it = O.Iterator(x, randomize=True, seed=123)
    assert it.collect().tolist() == list(range(1, p + 1))  # fresh iterator: identity order
    it.reset(True)
    full = it.collect()
    assert sorted(full.tolist()) == list(range(1, p + 1)) and full.tolist() != list(range(1, p + 1))
    it.reset(False)
    act = it.collect()
    assert len(act) == x.nnz and sorted(act.tolist()) == sorted(x.nzval2ind.tolist())
    # same seed, same stream
    it2 = O.Iterator(x, randomize=True, seed=123)
    it2.reset(True)
    assert it2.collect().tolist() == full.tolist()


def test_sparse_iterate_bookkeeping():
    x = O.SparseIterate(6)
    x[4] = 1.0
    x[2] = 3.0
    x[6] = -1.0
    x[3] = 0.0  # zero to an unstored coordinate: nothing stored
    assert x.nnz == 3 and x.nzval2ind.tolist() == [4, 2, 6]
    x[4] = 0.0  # zero to a stored coordinate keeps the slot until dropzeros!
    assert x.nnz == 3
    x.dropzeros()
    assert x.nnz == 2 and sorted(x.nzval2ind.tolist()) == [2, 6]
    np.testing.assert_array_equal(x.dense(), [0, 3.0, 0, 0, 0, -1.0])
    y = x.copy()
    x.fill_zero()
    assert x.nnz == 0 and y.nnz == 2


def test_dimension_mismatch_and_errors():
    rng = np.random.default_rng(10)
    X = rng.standard_normal((20, 5))
    with pytest.raises(ValueError):  # cd_differentiable_function.jl:53
        O.CDLeastSquaresLoss(np.zeros(19), X)
    f = O.CDLeastSquaresLoss(np.zeros(20), X)
    with pytest.raises(ValueError):  # coordinate_descent.jl:13
        O.coordinateDescent_(O.SparseIterate(4), f, O.ProxL1(0.1))
    with pytest.raises(ValueError):  # coordinate_descent.jl:14-16
        O.coordinateDescent_(O.SparseIterate(5), f, O.ProxL1(0.1, np.ones(4)))
    with pytest.raises(ValueError):  # cd_differentiable_function.jl:306
        O.CDQuadraticLoss(np.array([[1.0, 2.0], [0.0, 1.0]]), np.zeros(2))


def test_weighted_ls_matches_row_scaled_ls():
    """CDWeightedLSLoss (cd_differentiable_function.jl:118-194): with weights w the
    fixed point equals plain LS on rows scaled by sqrt(w)."""
    rng = np.random.default_rng(12)
    n, p = 300, 20
    X, Y = _problem(rng, n, p, 5)
    w = rng.random(n) + 0.5
    xw = O.SparseIterate(p)
    O.coordinateDescent_(xw, O.CDWeightedLSLoss(Y, X, w), O.ProxL1(0.05),
                         O.CDOptions(optTol=1e-12, randomize=False))
    sw = np.sqrt(w)
    xs = O.SparseIterate(p)
    O.coordinateDescent_(xs, O.CDLeastSquaresLoss(Y * sw, np.asfortranarray(X * sw[:, None])),
                         O.ProxL1(0.05), O.CDOptions(optTol=1e-12, randomize=False))
    np.testing.assert_allclose(xw.dense(), xs.dense(), rtol=0, atol=1e-10)


def test_orthogonal_design_closed_form():
    """Independent analytic check (not from the reference's tests): with X'X = n I the Lasso
    solution is beta_j = S(X_j'y / n, lambda * omega_j) in one pass, whatever the visit order."""
    rng = np.random.default_rng(21)
    n, p = 256, 40
    Q, _ = np.linalg.qr(rng.standard_normal((n, p)))
    X = np.asfortranarray(Q * np.sqrt(n))
    Y = X[:, :5] @ np.array([3.0, -2.0, 1.0, 0.5, -0.2]) + 0.3 * rng.standard_normal(n)
    om = rng.random(p) + 0.5
    lam = 0.25
    z = X.T @ Y / n
    want = np.sign(z) * np.maximum(np.abs(z) - lam * om, 0.0)
    for rand in (False, True):
        x = O.SparseIterate(p)
        st = O.coordinateDescent_(x, O.CDLeastSquaresLoss(Y, X), O.ProxL1(lam, om),
                                  O.CDOptions(optTol=1e-13, randomize=rand, seed=2))
        np.testing.assert_allclose(x.dense(), want, rtol=0, atol=1e-13)
        # the state machine of coordinate_descent.jl:65-92: full pass (moves), active pass
        # (nothing moves -> converged), full pass (still converged -> stop)
        assert st["passes"] == 3 and st["full_passes"] == 2


# -- optimality certificates, independent of any solver ---------------------------------------------
# The reference holds no golden vector for the sqrt-lasso closed form (cd_differentiable_function.jl:
# 271-283) or for the weighted losses: its tests assert one aggregate KKT number (test/lasso.jl:123, 54).
# The FULL KKT system of a convex problem is a certificate of optimality on its own: a point that meets
# it is a minimiser (the minimiser, X having full column rank), whatever produced it.  So the oracle's
# fixed points are pinned here coordinate by coordinate: gradient = lambda * omega_k * sign(beta_k) on
# the support, |gradient| <= lambda * omega_k off it.
@pytest.mark.parametrize("loss", ["ls", "wls", "sqrt"])
@pytest.mark.parametrize("weighted", [False, True], ids=["l1", "weighted_l1"])
def test_fixed_points_satisfy_the_full_kkt_system(loss, weighted):
    rng = np.random.default_rng(71 + 3 * weighted)
    n, p, s = 600, 80, 9
    X, Y = _problem(rng, n, p, s)
    X = np.asfortranarray(X * rng.uniform(0.4, 2.5, size=p))
    om = rng.uniform(0.5, 2.0, size=p) if weighted else np.ones(p)
    w = rng.uniform(0.3, 2.0, size=n)
    lam = {"ls": 0.08, "wls": 0.08, "sqrt": 2.2}[loss]
    f = {"ls": lambda: O.CDLeastSquaresLoss(Y, X), "wls": lambda: O.CDWeightedLSLoss(Y, X, w),
         "sqrt": lambda: O.CDSqrtLassoLoss(Y, X)}[loss]()
    x = O.SparseIterate(p)
    st = O.coordinateDescent_(x, f, O.ProxL1(lam, om if weighted else None),
                              O.CDOptions(maxIter=20000, optTol=1e-13, randomize=False))
    assert st["converged"]
    b = x.dense()
    r = Y - X @ b
    np.testing.assert_allclose(f.r, r, rtol=0, atol=1e-10)          # the carried residual is y - X beta
    if loss == "ls":
        grad = X.T @ r / n                      # -grad f; f = |r|^2 / (2n)     (:38-42)
    elif loss == "wls":
        grad = X.T @ (w * r) / n                # f = sum w r^2 / (2n)          (:113-117)
    else:
        grad = X.T @ r / np.linalg.norm(r)      # f = |r|_2                     (:234-235, test/lasso.jl:123)
    act = b != 0
    assert 3 <= act.sum() < p
    np.testing.assert_allclose(grad[act], lam * om[act] * np.sign(b[act]), rtol=0, atol=1e-9)
    assert np.all(np.abs(grad[~act]) <= lam * om[~act] * (1 + 1e-9))
    assert np.linalg.matrix_rank(X[:, act]) == act.sum()            # hence THE minimiser
