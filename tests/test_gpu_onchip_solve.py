"""The one-launch solve of problems that fit on chip (csrc/small_solve.hpp; cdh_set_onchip_solve, on by default): the
whole of _coordinateDescent! (src/coordinate_descent.jl:65-92) -- passes, active-set logic, SparseIterate bookkeeping,
the numSteps + 1 solves of a cold start (:24-37) -- inside one kernel, visits in covariance form on the full Gram matrix.
These are the reference's own shapes (test/lasso.jl:76-101, benchmark/cd_bench.jl:8-14; BASELINE.json configs[0]).

Parity bar: beta within 1e-10 of the oracle, and -- because the kernel visits one coordinate at a time in the
scheduler's order -- the SAME pass counts and support ORDER as the oracle's per-coordinate sweep, ordered and shuffled.
The rest of the GPU suite runs with CDH_SMALL_PATH=0 (tests/conftest.py) so that the streamed kernels keep the coverage
they had; this module turns the path on, which is what a new handle does by default."""
import glob
import os

import numpy as np
import pytest

import coordinatedescent_jl_amd as cd
import oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SOLVES = sorted(os.path.basename(c)[:-4] for c in glob.glob(os.path.join(GOLD, "*.npz")) if "traj" not in c)
BETA_TOL = 1e-10


@pytest.fixture(autouse=True)
def _onchip(monkeypatch):
    monkeypatch.setenv("CDH_SMALL_PATH", "1")


def _problem(seed, n, p, s, noise=1.0):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ rng.standard_normal(s) + noise * rng.standard_normal(n)
    return rng, X, Y


@pytest.mark.parametrize("name", SOLVES)
def test_golden_solves_in_one_launch(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    X, Y, lam = d["X"], d["Y"], float(d["lam"])
    omega = d["omega"] if "omega" in d else None
    cls = cd.CDSqrtLassoLoss if name.startswith("sqrt") else cd.CDLeastSquaresLoss
    f, g = cls(Y, X), cd.ProxL1(lam, omega)
    p = X.shape[1]
    x = cd.SparseIterate(p, d["x0"] if "x0" in d else None)
    warm = "warm0" not in name
    cd.coordinateDescent_(x, f, g, cd.CDOptions(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=False))
    assert f.onchip_stats() == {"solves": 1 if warm else 51, "gram_matrices": 1}
    assert f.last_stats["converged"]
    np.testing.assert_allclose(x.dense(), d["beta"], rtol=0, atol=BETA_TOL)
    assert x.nzval2ind.tolist() == d["support"].tolist()          # the reference state machine's visit order
    assert f.last_stats["passes"] == int(d["passes"])
    np.testing.assert_allclose(cd.objective(f), float(d["objective"]), rtol=1e-12)
    np.testing.assert_allclose(f.r, d["resid"], rtol=0, atol=1e-9)    # the residual caught up with the moves when asked for
    # the same solve on the streamed kernels: same point, same counts
    f.set_onchip_solve(False)
    f.set_sweep_mode("coord")
    x2 = cd.SparseIterate(p, d["x0"] if "x0" in d else None)
    cd.coordinateDescent_(x2, f, g, cd.CDOptions(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=False))
    assert f.onchip_stats()["solves"] == (1 if warm else 51)
    np.testing.assert_allclose(x2.dense(), x.dense(), rtol=0, atol=BETA_TOL)
    assert x2.nzval2ind.tolist() == x.nzval2ind.tolist()
    f.close()


@pytest.mark.parametrize("weighted", [False, True])
def test_four_options_warm_cold_ordered_shuffled(weighted):
    """test/coordinate_descent.jl:29-99: warm / cold start x ordered / shuffled sweeps agree; here also pass for pass and
    slot for slot with the oracle (same seeded substitute for Julia's global RNG)."""
    rng, X, Y = _problem(11 + weighted, 500, 50, 10 if weighted else 5)
    lam = 0.01 if weighted else 0.02
    om = rng.random(50) if weighted else None
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    sols = []
    for warm in (True, False):
        for rand in (False, True):
            x0 = np.where(rng.random(50) < 0.6, rng.random(50), 0.0)
            x, xo = cd.SparseIterate(50, x0), O.SparseIterate(50, x0)
            o = dict(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=rand, seed=5)
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
            st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
            np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
            assert f.last_stats["passes"] == st["passes"] and f.last_stats["visits"] == st["visits"]
            assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()
            sols.append(x.dense())
    assert f.onchip_stats() == {"solves": 2 + 2 * 51, "gram_matrices": 1}
    for b in sols[1:]:
        np.testing.assert_allclose(b, sols[0], rtol=0, atol=1e-5)
    f.close()


@pytest.mark.parametrize("rand", [False, True], ids=["ordered", "random"])
def test_sqrt_lasso_and_weighted_ls_losses(rand):
    rng, X, Y = _problem(71, 800, 120, 8)
    o = dict(maxIter=5000, optTol=1e-12, randomize=rand, seed=9)
    f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
    x, xo = cd.SparseIterate(120), O.SparseIterate(120)
    for lam in (3.0, 2.2, 1.6):                         # a warm-started path on one handle: r'r carried from solve to solve
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert f.last_stats["passes"] == st["passes"] and x.nzval2ind.tolist() == xo.nzval2ind.tolist()
    rr = Y - X @ x.dense()
    assert max(0.0, np.max(np.abs(X.T @ rr / np.linalg.norm(rr))) - 1.6) / 1.6 < 1e-6         # test/lasso.jl:123
    np.testing.assert_allclose(f.r, rr, rtol=0, atol=1e-9)
    assert f.onchip_stats()["solves"] == 3
    f.close()
    w = rng.random(800) + 0.5
    om = rng.random(120) + 0.5
    f, fo = cd.CDWeightedLSLoss(Y, X, w), O.CDWeightedLSLoss(Y, X, w)
    x, xo = cd.SparseIterate(120), O.SparseIterate(120)
    for lam in (0.2, 0.08):
        cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert f.last_stats["passes"] == st["passes"] and x.nzval2ind.tolist() == xo.nzval2ind.tolist()
    assert f.onchip_stats() == {"solves": 2, "gram_matrices": 1}
    f.close()


def test_front_ends_lasso_sqrtlasso_scaledlasso_path():
    rng, X, Y = _problem(72, 1000, 200, 10)
    X *= rng.uniform(0.5, 2.0, size=200)
    o = dict(maxIter=5000, optTol=1e-11, randomize=False)
    sol, so = cd.lasso(X, Y, 0.1, options=cd.CDOptions(**o)), O.lasso(X, Y, 0.1, None, O.CDOptions(**o))
    np.testing.assert_allclose(sol.x.dense(), so.x.dense(), rtol=0, atol=BETA_TOL)
    np.testing.assert_allclose(sol.sigma, so.sigma, rtol=1e-10)
    np.testing.assert_allclose(sol.residuals, Y - X @ sol.x.dense(), rtol=0, atol=1e-9)
    s2, so2 = cd.sqrtLasso(X, Y, 2.5, options=cd.CDOptions(**o)), O.sqrtLasso(X, Y, 2.5, options=O.CDOptions(**o))
    np.testing.assert_allclose(s2.x.dense(), so2.x.dense(), rtol=0, atol=BETA_TOL)
    om = rng.random(200) + 0.5
    x, xo = cd.SparseIterate(200), O.SparseIterate(200)
    io = dict(maxIter=50, optTol=1e-8)
    s3 = cd.scaledLasso_(x, X, Y, 0.1, om, cd.IterLassoOptions(optionsCD=cd.CDOptions(**o), **io))
    so3 = O.scaledLasso_(xo, X, Y, 0.1, om, O.IterLassoOptions(optionsCD=O.CDOptions(**o), **io))
    np.testing.assert_allclose(s3.sigma, so3.sigma, rtol=1e-9)
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-9)
    lams = [0.3, 0.2, 0.1, 0.05, 0.02]
    f = cd.CDLeastSquaresLoss(Y, X)
    path = cd.LassoPath(f, None, lams, cd.CDOptions(**o))
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(**o))
    for got, want in zip(path.betapath, bo):
        np.testing.assert_allclose(got.dense(), want, rtol=0, atol=BETA_TOL)
    assert f.onchip_stats() == {"solves": 5, "gram_matrices": 1}
    f.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("CDH_FUZZ_ONCHIP", "16"))))
def test_random_configurations(seed):
    rng = np.random.default_rng(7000 + seed)
    p = int(rng.integers(3, 400))
    n = int(rng.integers(max(2, p // 3), 4 * p + 50))
    s = int(rng.integers(1, min(12, p) + 1))
    X = np.asfortranarray(rng.standard_normal((n, p)) * rng.uniform(0.3, 3.0, size=p))
    Y = X[:, :s] @ rng.standard_normal(s) + rng.uniform(0.3, 2.0) * rng.standard_normal(n)
    sqrt = bool(seed % 3 == 2) and n > p
    om = rng.uniform(0.5, 2.0, size=p) if rng.integers(0, 2) else None
    o = dict(maxIter=3000, optTol=1e-11, randomize=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)),
             warmStart=bool(rng.integers(0, 4) > 0), numSteps=int(rng.integers(3, 30)))
    if sqrt:
        f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
        top = 0.9 * float(np.max(np.abs(X.T @ Y) / (om if om is not None else 1.0)) / np.linalg.norm(Y))
        top = min(top, 0.5 * float(np.sqrt(np.min(np.sum(X * X, axis=0)))))
        lams = top * np.array([1.0, 0.85, 0.7])
    else:
        f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
        top = 0.8 * float(np.max(np.abs(X.T @ Y) / (om if om is not None else 1.0)) / n)
        lams = top * np.array([1.0, 0.5, 0.2])
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    for lam in lams:
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
        cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
        if not st["converged"]:
            assert not f.last_stats["converged"] and f.last_stats["passes"] == st["passes"]     # maxIter exhaustion is silent in both
            break
        scale = max(1.0, float(np.max(np.abs(xo.dense()))))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL * scale,
                                   err_msg=f"n={n} p={p} sqrt={sqrt} o={o} lam={lam}")
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"], (n, p, o, lam)
    assert f.onchip_stats()["solves"] > 0
    f.close()


def test_streamed_calls_after_an_onchip_solve_see_the_current_residual():
    """cdh_pass / cdh_descend / gradient after a one-launch solve: the residual's deferred catch-up runs before anything
    reads r, so mixing the two paths gives the oracle's trajectory."""
    rng, X, Y = _problem(73, 600, 90, 6)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    x, xo = cd.SparseIterate(90), O.SparseIterate(90)
    o = dict(maxIter=3, optTol=1e-12, randomize=False)          # stopped early on purpose: not at the fixed point
    cd.coordinateDescent_(x, f, cd.ProxL1(0.05), cd.CDOptions(**o))
    st = O.coordinateDescent_(xo, fo, O.ProxL1(0.05), O.CDOptions(**o))
    assert not f.last_stats["converged"] and f.last_stats["passes"] == st["passes"] == 3
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    k = int(x.nzval2ind[0])
    np.testing.assert_allclose(cd.gradient(f, x, k), O.gradient(fo, xo, k), rtol=1e-9, atol=1e-12)
    visit = np.arange(1, 91)
    mh = cd.cdPass_(x, f, cd.ProxL1(0.05), visit)
    mho = O.cdPass_(xo, fo, O.ProxL1(0.05), visit)
    np.testing.assert_allclose(mh, mho, rtol=1e-7, atol=1e-13)
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.05), cd.CDOptions(maxIter=5000, optTol=1e-12, randomize=False))
    O.coordinateDescent_(xo, fo, O.ProxL1(0.05), O.CDOptions(maxIter=5000, optTol=1e-12, randomize=False))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    f.close()


def test_new_columns_new_y_and_the_size_limits():
    rng, X, Y = _problem(74, 700, 150, 7)
    o = dict(maxIter=5000, optTol=1e-12, randomize=False)
    f = cd.CDLeastSquaresLoss(Y, X)
    x = cd.SparseIterate(150)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.08), cd.CDOptions(**o))
    np.testing.assert_allclose(x.dense(), O.lasso(X, Y, 0.08, None, O.CDOptions(**o)).x.dense(), rtol=0, atol=BETA_TOL)
    Y2 = X[:, 50:55] @ rng.standard_normal(5) + rng.standard_normal(700)            # a new y: the Gram matrix stays
    y2 = np.ascontiguousarray(Y2)
    cd._lib.check(f._L.cdh_set_y(f._h, y2.ctypes.data), f._h)
    x = cd.SparseIterate(150)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.08), cd.CDOptions(**o))
    np.testing.assert_allclose(x.dense(), O.lasso(X, Y2, 0.08, None, O.CDOptions(**o)).x.dense(), rtol=0, atol=BETA_TOL)
    assert f.onchip_stats() == {"solves": 2, "gram_matrices": 1}
    Xn = np.asfortranarray(X.copy())                                               # new columns: it is rebuilt
    Xn[:, :30] = rng.standard_normal((700, 30))
    blk = np.asfortranarray(Xn[:, :30])
    cd._lib.check(f._L.cdh_set_X_cols(f._h, 0, 30, blk.ctypes.data, 700), f._h)
    x = cd.SparseIterate(150)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.08), cd.CDOptions(**o))
    np.testing.assert_allclose(x.dense(), O.lasso(Xn, Y2, 0.08, None, O.CDOptions(**o)).x.dense(), rtol=0, atol=BETA_TOL)
    assert f.onchip_stats() == {"solves": 3, "gram_matrices": 2}
    f.close()
    # beyond the limits (p > 1024; more than 16 MB of X: until the streamed solves have cost what building G would) a solve
    # stays on the streamed kernels
    for n, p in ((300, 1100), (9000, 300)):
        rng, X, Y = _problem(75, n, p, 5)
        f = cd.CDLeastSquaresLoss(Y, X)
        x = cd.SparseIterate(p)
        cd.coordinateDescent_(x, f, cd.ProxL1(0.2), cd.CDOptions(**o))
        assert f.onchip_stats() == {"solves": 0, "gram_matrices": 0}
        np.testing.assert_allclose(x.dense(), O.lasso(X, Y, 0.2, None, O.CDOptions(**o)).x.dense(), rtol=0, atol=BETA_TOL)
        f.close()


def test_rent_or_buy_beyond_the_always_limit():
    """More than 16 MB of X: the Gram matrix is built once the solves run on the streamed kernels have cost as much as
    building it would (small_applicable, csrc/small_solve.hpp) -- here within a few solves -- and a cold start, being
    numSteps + 1 solves at once, buys a cheap build outright.  Same iterates as the oracle on either side of the switch."""
    rng, X, Y = _problem(77, 9000, 300, 8)                                         # 21.6 MB
    o = dict(maxIter=5000, optTol=1e-12, randomize=False)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    x, xo = cd.SparseIterate(300), O.SparseIterate(300)
    seen = []
    for lam in (0.30, 0.24, 0.19, 0.15, 0.12, 0.09, 0.07, 0.05):
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"]
        seen.append(f.onchip_stats()["solves"])
    assert seen[0] == 0                                   # the first solve pays rent
    assert seen[-1] > 0 and f.onchip_stats()["gram_matrices"] == 1
    first = next(i for i, v in enumerate(seen) if v > 0)
    assert seen[first:] == list(range(1, len(seen) - first + 1))      # once built, every solve is one launch
    np.testing.assert_allclose(f.r, Y - X @ x.dense(), rtol=0, atol=1e-9)
    f.close()
    f = cd.CDLeastSquaresLoss(Y, X)                        # a cold start on a new handle: 51 solves, one launch
    x, xo = cd.SparseIterate(300), O.SparseIterate(300)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.05), cd.CDOptions(warmStart=False, **o))
    O.coordinateDescent_(xo, fo, O.ProxL1(0.05), O.CDOptions(warmStart=False, **o))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    assert f.onchip_stats() == {"solves": 51, "gram_matrices": 1}
    f.close()


def test_zero_column_fp32_storage_and_maxiter():
    rng, X, Y = _problem(76, 400, 120, 5)
    X[:, 70] = 0.0                                     # 0 / 0 in x[k] += b / a: soft-thresholded to 0 by oracle and product alike
    o = dict(maxIter=60, optTol=1e-12, randomize=False)
    f = cd.CDLeastSquaresLoss(Y, X)
    x, xo = cd.SparseIterate(120), O.SparseIterate(120)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.15), cd.CDOptions(**o))
    st = O.coordinateDescent_(xo, O.CDLeastSquaresLoss(Y, X), O.ProxL1(0.15), O.CDOptions(**o))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    assert f.last_stats["passes"] == st["passes"] and x[71] == 0.0 and f.onchip_stats()["solves"] == 1
    f.close()
    X[:, 70] = rng.standard_normal(400)
    f32 = cd.CDLeastSquaresLoss(Y.astype(np.float32), X.astype(np.float32))
    x = cd.SparseIterate(120)
    cd.coordinateDescent_(x, f32, cd.ProxL1(0.1), cd.CDOptions(maxIter=2000, optTol=1e-7, randomize=False))
    want = O.lasso(X, Y, 0.1, None, O.CDOptions(maxIter=2000, optTol=1e-10, randomize=False)).x.dense()
    np.testing.assert_allclose(x.dense(), want, rtol=0, atol=3e-4)      # fp32 storage against the fp64 oracle
    assert f32.onchip_stats()["solves"] == 1 and f32.r.dtype == np.float32
    f32.close()



@pytest.mark.parametrize("kind", ["ls", "sqrt"])
@pytest.mark.parametrize("noise", [1e-2, 1e-3])
def test_high_snr_warm_starts_do_not_lose_the_residual_norm(kind, noise):
    """||r|| << ||y|| (near-noiseless y, small lambda): the one-launch solve forms g = X'y - G beta and
    r'r = y'y - sum beta_s (c_s + g_s) by subtraction (small_solve.hpp), which loses ~||y||^2 / ||r||^2 in relative accuracy --
    and the sqrt-lasso's threshold lambda omega ||r|| and its closed form hang on that r'r.  The kernel gives up
    once r'r has fallen below 1e-6 of the value its recurrence started from (kSmallQGuard) and the call runs on the streamed
    kernels, which sum r'r from r itself: noise 1e-2 stays in the kernel (r'r / y'y ~ 2e-5), noise 1e-3 (~2e-7) does not.
    Either way beta is the oracle's to 1e-10.  (At noise 1e-6 the REFERENCE's own closed form, sqrt(rsqr - s^2 / xsqr),
    cancels twelve digits and neither side reaches optTol 1e-12 in 20 000 passes: nothing to compare.)"""
    rng, X, Y = _problem(31, 400, 40, 6, noise=noise)
    lams = [3.0, 2.0, 1.0] if kind == "sqrt" else [1e-2, 1e-3, 1e-4]
    o = dict(maxIter=20000, optTol=1e-12, randomize=False)
    f, fo = (cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)) if kind == "sqrt" else (cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X))
    x, xo = cd.SparseIterate(40), O.SparseIterate(40)
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        assert f.last_stats["converged"] and st["converged"]
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert sorted(x.nzval2ind.tolist()) == sorted(xo.nzval2ind.tolist())
    in_kernel = f.onchip_stats()["solves"]
    assert (in_kernel >= len(lams)) if (kind == "ls" or noise >= 1e-2) else (in_kernel == 0), (kind, noise, f.onchip_stats())
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
    # and a cold start (51 continuation solves inside one launch) from the same data
    x, xo = cd.SparseIterate(40), O.SparseIterate(40)
    cd.coordinateDescent_(x, f, cd.ProxL1(lams[-1]), cd.CDOptions(warmStart=False, **o))
    O.coordinateDescent_(xo, fo, O.ProxL1(lams[-1]), O.CDOptions(warmStart=False, **o))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
