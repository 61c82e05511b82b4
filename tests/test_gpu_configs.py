"""BASELINE.json's configurations 3, 4 and 5 under `-m gpu`: each one at a size the fp64 oracle
covers (parity, tolerance stated in the test) and at its full one-GPU size through size-independent
properties (KKT conditions, sigma fixed point, carried residual == rebuilt residual, early stop).
cfg4 (320 GB) and cfg5 (640 GB) are 8-GPU problems; their full-size cases here are ONE rank's shard
(5e6 / 1e7 rows), which is what a single MI355X holds in the sharded run.  cfg2's full size is in
test_gpu_parity.py::test_full_size_properties_cfg2.  Reference: src/lasso.jl:62-98 (sqrtLasso),
:107-144 (scaledLasso!), :229-260 (LassoPath); benchmark/cd_bench.jl:8-20 for the shapes.
"""
import ctypes as C

import numpy as np
import pytest

import coordinatedescent_jl_amd as cd
import oracle as O

pytestmark = pytest.mark.gpu


def _xtr(f):
    out = np.zeros(f.p)
    cd._lib.check(f._L.cdh_xt_r(f._h, out.ctypes.data), f._h)
    return out


def _moments(f):
    s, ss = C.c_double(), C.c_double()
    cd._lib.check(f._L.cdh_resid_moments(f._h, C.byref(s), C.byref(ss)), f._h)
    return s.value, ss.value


# ---- cfg5: weighted-l1 scaled lasso, fp32 storage, hipGraph-captured sweeps -------------------------
def test_cfg5_workload_at_oracle_size_fp32_omega_scaled_lasso_graph():
    """The combination north_star names for cfg5 -- fp32 X / y, omega = _stdX!, scaledLasso! with the
    :Screening init, default B = 32, every pass replayed from a hipGraph -- against the fp64 oracle on
    the same numbers.  Tolerance (declared): fp32 storage of X, y and r gives ~1e-7 relative per
    element; 3e-4 absolute on beta (values O(1)), 1e-4 relative on sigma.  Graph replay must not
    change a bit relative to node-by-node launches."""
    rng = np.random.default_rng(55)
    n, p, s = 20_000, 400, 25
    X = np.asfortranarray(rng.standard_normal((n, p)) * rng.uniform(0.5, 2.0, size=p))
    Y = X[:, :s] @ (rng.standard_normal(s) * (1 + rng.random(s))) + 3.0 * rng.standard_normal(n)
    lam = float(np.sqrt(2 * np.log(p) / n))                       # benchmark/cd_bench.jl:20
    X32, Y32 = X.astype(np.float32), Y.astype(np.float32)
    cdo = dict(randomize=False, optTol=1e-7, maxIter=2000)
    # sigma loop run to its fixed point (1e-6): a loop stopped at the default 1e-2 could end one iteration
    # apart in fp32 and fp64 when a step lands near the threshold, which is not a storage-precision effect
    io = dict(maxIter=50, optTol=1e-6)
    outs = []
    for graph in (False, True):
        f = cd.CDLeastSquaresLoss(Y32, X32)
        f.set_use_graph(graph)                                    # default sweep: block, B = 32
        om = cd.stdX(f)
        x = cd.SparseIterate(p)
        sol = cd.scaledLasso_(x, f, None, lam, om, cd.IterLassoOptions(optionsCD=cd.CDOptions(**cdo), **io))
        outs.append((x.dense(), sol.sigma, f.r, om))
        f.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])         # graph replay: same bits
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    assert outs[0][1] == outs[1][1]
    beta, sigma, r, om = outs[1]
    np.testing.assert_allclose(om, np.sqrt((X * X).sum(0) / n), rtol=2e-7)         # _stdX! of the fp32 matrix
    xo = O.SparseIterate(p)
    so = O.scaledLasso_(xo, X, Y, lam, O.stdX(X), O.IterLassoOptions(optionsCD=O.CDOptions(**cdo), **io))
    np.testing.assert_allclose(sigma, so.sigma, rtol=1e-4)
    np.testing.assert_allclose(beta, xo.dense(), rtol=0, atol=3e-4)
    assert 0 < np.count_nonzero(beta) < p and r.dtype == np.float32
    # the default sigma tolerance (1e-2, utils.jl:34) as well: it must stop, and where it stops is a fixed
    # point to that tolerance
    f = cd.CDLeastSquaresLoss(Y32, X32)
    f.set_use_graph(True)
    x = cd.SparseIterate(p)
    sol = cd.scaledLasso_(x, f, None, lam, om, cd.IterLassoOptions(optionsCD=cd.CDOptions(**cdo)))
    assert abs(sol.sigma - so.sigma) / so.sigma < 2e-2
    f.close()


def test_cfg5_one_rank_shard_full_size_properties():
    """n = 1e7 rows x p = 2000, fp32 (80 GB: one rank's shard of cfg5), generated in HBM; lambda =
    sqrt(2 log p / n), omega = _stdX!, :Screening init, default sigma tolerance 1e-2, graph replay on.
    Properties: the sigma loop stopped at a fixed point (one more solve moves sigma by < 1e-2), KKT
    holds at lambda * sigma_last * omega, and the carried residual equals a from-scratch rebuild."""
    n, p, s = 10_000_000, 2000, 100
    f, bstar = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=s, noise=6.0, dtype=np.float32)
    try:
        f.set_use_graph(True)
        om = cd.stdX(f)
        assert np.all(np.abs(om - 1.0) < 5e-3)                    # unit-variance columns
        lam = float(np.sqrt(2 * np.log(p) / n))
        x = cd.SparseIterate(p)
        cdo = cd.CDOptions(randomize=False)                       # optTol 1e-7, maxIter 2000: the defaults
        sol = cd.scaledLasso_(x, f, None, lam, om, cd.IterLassoOptions(optionsCD=cdo))
        assert f.last_stats["converged"]
        s1, ss1 = _moments(f)
        sigma_hat = np.sqrt(ss1 / n)
        assert abs(sigma_hat - 6.0) < 0.05                        # the planted noise level
        lam_last = sol.penalty.lambda0                            # lambda * sigma of the last solve
        assert abs(sigma_hat - lam_last / lam) / (lam_last / lam) < 1e-2          # the loop's own stopping rule
        # KKT at lambda * sigma * omega (test/lasso.jl:210-214, there with 1e-4): fp32 residual storage
        # limits how exactly X_k'r can sit on the threshold -- 5e-4 relative on the support, same slack off it
        grad = np.abs(_xtr(f)) / n
        beta = x.dense()
        act = beta != 0
        thr = lam_last * om
        assert act.sum() >= 50 and np.max(np.abs(grad[act] - thr[act]) / thr[act]) < 5e-4
        assert np.all(grad[~act] <= thr[~act] * (1 + 5e-4))
        assert np.max(np.abs(beta[:s] - bstar)) < 0.02            # n = 1e7: the planted coefficients are recovered
        cd.initialize_(f, x)                                      # rebuild r = y - X beta from scratch
        s2, ss2 = _moments(f)
        assert abs(ss1 - ss2) <= 1e-6 * ss2                       # fp32 r carried through every update
        # one more sigma iteration from here moves sigma by less than the tolerance
        cd.coordinateDescent_(x, f, cd.ProxL1(lam * sigma_hat, om), cdo)
        _, ss3 = _moments(f)
        assert abs(np.sqrt(ss3 / n) - sigma_hat) / sigma_hat < 1e-2
        # round 3: the gradient cache serves fp32 storage.  The same sigma loop from x = 0 with the cache off and with
        # it engaged from the first full pass: same sigma (1e-6 relative), same beta (2e-5: the cache's g follows the
        # fp64 problem, the streamed sweep the rounding of its fp32 residual), the same number of passes in the last solve.
        res = {}
        for mode in (0, 2):
            f.set_gradient_cache(mode)
            before = f.cache_stats()
            xm = cd.SparseIterate(p)
            sm = cd.scaledLasso_(xm, f, None, lam, om, cd.IterLassoOptions(optionsCD=cdo))
            after = f.cache_stats()
            res[mode] = (sm.sigma, xm.dense(), f.last_stats["passes"], {k: after[k] - before[k] for k in after})
        assert abs(res[2][0] - res[0][0]) / res[0][0] < 1e-6 and res[2][2] == res[0][2], (res[0][0], res[2][0], res[0][2], res[2][2])
        np.testing.assert_allclose(res[2][1], res[0][1], rtol=0, atol=2e-5)
        assert res[0][3]["passes"] == 0 and res[2][3]["passes"] >= 2 and res[2][3]["device_passes"] >= 2, res[2][3]
        assert res[2][3]["covariance_visits"] > 100 and res[2][3]["gram_columns"] >= int(act.sum()), res[2][3]
    finally:
        f.close()


# ---- cfg3: warm-started lambda path with the active-set iterator ------------------------------------
def test_cfg3_workload_at_oracle_size():
    """100 log-spaced lambdas from lambda_max to 1e-2 lambda_max, omega = _stdX!, ordered sweeps, optTol
    1e-7, screened full passes, the carried residual reused -- against the oracle's LassoPath at every
    lambda (1e-6: two optTol-1e-7 solutions of the same problem) and against a screening-off run
    (same pass counts: screening settles visits, it never changes one)."""
    rng = np.random.default_rng(33)
    n, p, s = 3000, 600, 30
    X = np.asfortranarray(rng.standard_normal((n, p)) * rng.uniform(0.5, 2.0, size=p))
    Y = X[:, :s] @ (rng.standard_normal(s) * (1 + rng.random(s))) + 2.0 * rng.standard_normal(n)
    f = cd.CDLeastSquaresLoss(Y, X)
    x0 = cd.SparseIterate(p)
    cd.initialize_(f, x0)
    om = cd.stdX(f)
    lmax = cd.findLambdaMax(x0, f, cd.ProxL1(1.0, om))
    lams = np.exp(np.linspace(np.log(lmax), np.log(1e-2 * lmax), 100))
    opt = dict(optTol=1e-7, randomize=False)
    path = cd.LassoPath(f, None, lams, cd.CDOptions(**opt))
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(**opt))
    nnz = [b.nnz for b in path.betapath]
    assert nnz[0] <= 1 and nnz[-1] > s and len(path.betapath) == 100
    for i in range(100):
        np.testing.assert_allclose(path.betapath[i].dense(), bo[i], rtol=0, atol=1e-6)
    f2 = cd.CDLeastSquaresLoss(Y, X)
    f2.set_screening(0)
    passes = []
    for g in (f, f2):
        x = cd.SparseIterate(p)
        tot = 0
        for lam in lams[:40]:
            cd.coordinateDescent_(x, g, cd.ProxL1(lam, om), cd.CDOptions(**opt))
            tot += g.last_stats["passes"]
        passes.append((tot, x.dense()))
    assert passes[0][0] == passes[1][0]
    np.testing.assert_allclose(passes[0][1], passes[1][1], rtol=0, atol=1e-12)
    short = cd.LassoPath(f, None, lams, cd.CDOptions(**opt), max_hat_s=10)
    k = len(short.betapath)
    assert k < 100 and short.betapath[-1].nnz > 10 and all(b.nnz <= 10 for b in short.betapath[:-1])
    assert len(short.lambdapath) == k


def test_cfg3_full_size_path_properties():
    """n = 2e6, p = 5000 (80 GB), 100 lambdas, screening on: every path entry satisfies the KKT conditions
    of its own lambda (checked from X'r on the device right after its solve), the supports grow along the
    path, three sampled entries equal unscreened cold solves at their lambda, and max_hat_s stops early
    (lasso.jl:253-256)."""
    n, p, s, nlam = 2_000_000, 5000, 100, 100
    f, bstar = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=s, noise=6.0)
    try:
        x = cd.SparseIterate(p)
        cd.initialize_(f, x)
        om = cd.stdX(f)
        lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0, om))
        lams = np.exp(np.linspace(np.log(lmax), np.log(1e-2 * lmax), nlam))
        opt = cd.CDOptions(optTol=1e-7, randomize=False)
        cd._lib.check(f._L.cdh_set_reuse_residual(f._h, 1), f._h)      # what LassoPath does (api.py)
        keep, nnz, worst_on, worst_off = {}, [], 0.0, 0.0
        for i, lam in enumerate(lams):
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), opt)       # lasso.jl:250-252
            assert f.last_stats["converged"]
            beta = x.dense()
            grad = np.abs(_xtr(f)) / n
            act, thr = beta != 0, lam * om
            if act.any():
                worst_on = max(worst_on, float(np.max(np.abs(grad[act] - thr[act]) / thr[act])))
            worst_off = max(worst_off, float(np.max(grad[~act] / thr[~act])))
            nnz.append(int(act.sum()))
            if i in (30, 60, 99):
                keep[i] = beta.copy()
        cd._lib.check(f._L.cdh_set_reuse_residual(f._h, 0), f._h)
        # the gradient cache served this path (engaged by its default rent-or-buy policy), mostly in whole passes on
        # the device, without a rollback; its carried gradient, taken afresh from X right now, had drifted by far less
        # than the certificates' 1e-9 relative margin ("bound and report the cached gradient's drift")
        cs = f.cache_stats()
        assert cs["passes"] > 100 and cs["device_passes"] > 100 and cs["rollbacks"] == 0 and cs["covariance_visits"] > 10_000, cs
        drift = f.cache_drift(rereference_now=True)
        assert drift["measured"] >= 1 and drift["max"] < 1e-10, drift
        # optTol bounds |h| by 1e-7, i.e. the gradient of a visited coordinate by 1e-7 * ||X_k||^2 / n
        assert worst_on < 1e-5 and worst_off < 1 + 1e-5, (worst_on, worst_off)
        # (a few of the 100 planted coefficients are smaller than the last threshold: 95 survive on this seed)
        assert nnz[0] <= 1 and 80 <= nnz[-1] <= 400 and all(b >= a - 3 for a, b in zip(nnz, nnz[1:]))
        assert np.max(np.abs(keep[99][:s] - bstar)) < 0.5
        f.set_screening(0)                                             # three entries against unscreened solves
        for i, want in keep.items():
            xc = cd.SparseIterate(p)
            cd.coordinateDescent_(xc, f, cd.ProxL1(lams[i], om), opt)
            np.testing.assert_allclose(xc.dense(), want, rtol=0, atol=2e-6)
        f.set_screening(1)
        short = cd.LassoPath(f, None, lams[:60], opt, max_hat_s=20)
        assert len(short.betapath) < 60 and short.betapath[-1].nnz > 20
        assert all(b.nnz <= 20 for b in short.betapath[:-1]) and len(short.lambdapath) == len(short.betapath)
    finally:
        f.close()


# ---- cfg4: sqrt-lasso on a row shard ------------------------------------------------------------------
def test_cfg4_workload_at_oracle_size():
    """sqrt-lasso, lambda = 1.1 * Phi^-1(1 - 0.05 / (2p)) (the pivotal level SURVEY 8d derives; 4.4612 at
    p = 1000), ordered, optTol 1e-8, default B = 32: beta within 1e-10 of the oracle, the reference's own
    KKT assertion (test/lasso.jl:123) far inside its 1e-3."""
    from scipy.stats import norm
    rng = np.random.default_rng(44)
    n, p, s = 20_000, 300, 20
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ (rng.standard_normal(s) * (1 + rng.random(s))) + rng.standard_normal(n)
    lam = float(1.1 * norm.ppf(1 - 0.05 / (2 * p)))
    o = dict(optTol=1e-8, randomize=False, maxIter=2000)
    sol = cd.sqrtLasso(X, Y, lam, options=cd.CDOptions(**o), standardizeX=False)
    so = O.sqrtLasso(X, Y, lam, options=O.CDOptions(**o), standardizeX=False)
    np.testing.assert_allclose(sol.x.dense(), so.x.dense(), rtol=0, atol=1e-10)
    r = Y - X @ sol.x.dense()
    assert max(0.0, np.max(np.abs(X.T @ r / np.linalg.norm(r))) - lam) / lam < 1e-6
    np.testing.assert_allclose(sol.residuals, r, rtol=0, atol=1e-10)


def test_cfg4_one_rank_shard_full_size_properties():
    """n = 5e6 rows x p = 1000 fp64 (40 GB: one rank's shard of cfg4), lambda = 4.4612, omega = 1, ordered,
    optTol 1e-8: the sqrt-lasso KKT conditions at the solution (|X_k'r| / ||r|| = lambda on the support,
    <= lambda off it; test/lasso.jl:123 asserts the aggregate to 1e-3), carried residual == rebuilt, and the
    per-coordinate sweep reaches the same point as the default blocked one."""
    n, p, s, lam = 5_000_000, 1000, 100, 4.4612
    f, bstar = cd.CDSqrtLassoLoss.generate(n, p, seed=123, s=s, noise=1.0)
    try:
        x = cd.SparseIterate(p)
        o = cd.CDOptions(optTol=1e-8, randomize=False)
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), o)
        assert f.last_stats["converged"] and not f.last_stats["domain_error"]
        s1, ss1 = _moments(f)
        g = np.abs(_xtr(f)) / np.sqrt(ss1)
        beta = x.dense()
        act = beta != 0
        # optTol 1e-8 on |h| bounds the gradient slack by a_k h / ||r|| = 5e6 * 1e-8 / 2236 = 2.2e-5
        assert max(0.0, g.max() - lam) / lam < 2e-5               # the reference's assertion, 50x tighter
        assert act.sum() >= 50 and np.max(np.abs(g[act] - lam)) / lam < 2e-5
        assert np.max(np.abs(beta[:s][np.abs(bstar) > 0.1] - bstar[np.abs(bstar) > 0.1])) < 0.05
        assert abs(np.sqrt(ss1 / n) - 1.0) < 0.01                 # residual scale = the planted noise
        cd.initialize_(f, x)
        _, ss2 = _moments(f)
        assert abs(ss1 - ss2) <= 1e-12 * ss2
        f.set_sweep_mode("coord")
        xc = cd.SparseIterate(p)
        cd.coordinateDescent_(xc, f, cd.ProxL1(lam), o)
        np.testing.assert_allclose(xc.dense(), beta, rtol=0, atol=1e-9)
    finally:
        f.close()
