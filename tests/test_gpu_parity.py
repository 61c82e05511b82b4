"""GPU parity tests: the HIP path, called through the C-ABI (include/cdhip.h) via the
host mirror, against the CPU oracle on the same seeded inputs and against the committed
golden vectors.  Tolerances (fp64): |beta_gpu - beta_oracle| <= 1e-10 and relative
objective gap <= 1e-12 at tight optTol, as BASELINE.json's north_star states; bitwise
parity is not meaningful because the reference's own `@simd` sums are unordered.
Shapes follow the reference's tests.
"""
import glob
import os

import numpy as np
import pytest

import coordinatedescent_jl_amd as cd
import oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BETA_TOL = 1e-10
MODES = [("coord", 0), ("block", 8), ("block", 4), ("block", 2), ("block", 16), ("block", 32), ("block", 64)]


def _problem(seed, n, p, s, noise=1.0):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ rng.standard_normal(s) + noise * rng.standard_normal(n)
    return rng, X, Y


def _free_port():
    """A port the OS says is free right now, for torch.distributed.run's rendezvous on 127.0.0.1."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _set_mode(f, mode):
    if mode[0] == "block":
        f.set_sweep_mode("block", mode[1])
    else:
        f.set_sweep_mode("coord")


# ---- single visits ---------------------------------------------------------------------
@pytest.mark.parametrize("loss", ["ls", "sqrt", "wls"])
def test_descend_coordinate_matches_oracle(loss):
    rng, X, Y = _problem(1, 300, 12, 4)
    w = rng.random(300) + 0.5
    lam = {"ls": 0.05, "sqrt": 1.2, "wls": 0.05}[loss]
    om = rng.random(12) + 0.5
    if loss == "ls":
        f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    elif loss == "sqrt":
        f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
    else:
        f, fo = cd.CDWeightedLSLoss(Y, X, w), O.CDWeightedLSLoss(Y, X, w)
    g, go = cd.ProxL1(lam, om), O.ProxL1(lam, om)
    x0 = np.where(rng.random(12) < 0.5, rng.standard_normal(12), 0.0)
    x, xo = cd.SparseIterate(12, x0), O.SparseIterate(12, x0)
    cd.initialize_(f, x)
    O.initialize_(fo, xo)
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-12)
    for k in [3, 1, 12, 7, 3, 3, 5]:
        np.testing.assert_allclose(cd.gradient(f, x, k), O.gradient(fo, xo, k), rtol=1e-12, atol=1e-14)
        h, ho = cd.descendCoordinate_(f, g, x, k), O.descendCoordinate_(fo, go, xo, k)
        np.testing.assert_allclose(h, ho, rtol=0, atol=1e-12)
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-12)
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-11)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()


# ---- fixed number of passes: the trajectory, not just the fixed point --------------------
@pytest.mark.parametrize("mode", MODES, ids=lambda m: f"{m[0]}{m[1]}")
@pytest.mark.parametrize("name", ["ls_n200_p50_traj", "sqrt_n100_p50_traj"])
def test_pass_trajectory_matches_golden(name, mode):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    X, Y, lam = d["X"], d["Y"], float(d["lam"])
    cls = cd.CDSqrtLassoLoss if name.startswith("sqrt") else cd.CDLeastSquaresLoss
    f, g = cls(Y, X), cd.ProxL1(lam)
    _set_mode(f, mode)
    p = X.shape[1]
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    for t in range(d["betas"].shape[0]):
        mh = cd.cdPass_(x, f, g, range(1, p + 1))
        np.testing.assert_allclose(x.dense(), d["betas"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(mh, d["maxhs"][t], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(f.r, d["resid"], rtol=0, atol=1e-11)


# ---- whole solves against the golden vectors ------------------------------------------------
SOLVES = sorted(os.path.basename(c)[:-4] for c in glob.glob(os.path.join(GOLD, "*.npz")) if "traj" not in c)


@pytest.mark.parametrize("mode", [MODES[0], MODES[1], MODES[4], MODES[5], MODES[6]], ids=lambda m: f"{m[0]}{m[1]}")
@pytest.mark.parametrize("name", SOLVES)
def test_solve_matches_golden(name, mode):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    X, Y, lam = d["X"], d["Y"], float(d["lam"])
    omega = d["omega"] if "omega" in d else None
    cls = cd.CDSqrtLassoLoss if name.startswith("sqrt") else cd.CDLeastSquaresLoss
    f, g = cls(Y, X), cd.ProxL1(lam, omega)
    _set_mode(f, mode)
    p = X.shape[1]
    x = cd.SparseIterate(p, d["x0"] if "x0" in d else None)
    warm = "warm0" not in name
    cd.coordinateDescent_(x, f, g, cd.CDOptions(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=False))
    assert f.last_stats["converged"]
    np.testing.assert_allclose(x.dense(), d["beta"], rtol=0, atol=BETA_TOL)
    assert sorted(x.nzval2ind.tolist()) == sorted(d["support"].tolist())
    np.testing.assert_allclose(cd.objective(f), float(d["objective"]), rtol=1e-12)
    np.testing.assert_allclose(f.r, d["resid"], rtol=0, atol=1e-9)
    if mode[0] == "coord":  # same visit order and pass count as the reference state machine
        assert x.nzval2ind.tolist() == d["support"].tolist()
        assert f.last_stats["passes"] == int(d["passes"])


# ---- reference test/coordinate_descent.jl:29-99: warm/cold x ordered/random ---------------
@pytest.mark.parametrize("weighted", [False, True])
def test_four_option_agreement_and_oracle_parity(weighted):
    rng, X, Y = _problem(11 + weighted, 500, 50, 10 if weighted else 5)
    lam = 0.01 if weighted else 0.02
    om = rng.random(50) if weighted else None
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    sols = []
    for warm in (True, False):
        for rand in (False, True):
            x0 = np.where(rng.random(50) < 0.6, rng.random(50), 0.0)
            x, xo = cd.SparseIterate(50, x0), O.SparseIterate(50, x0)
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om),
                                  cd.CDOptions(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=rand, seed=5))
            st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om),
                                      O.CDOptions(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=rand, seed=5))
            np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
            assert f.last_stats["passes"] == st["passes"]  # same seeded visit orders
            assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()
            sols.append(x.dense())
    for b in sols[1:]:
        np.testing.assert_allclose(b, sols[0], rtol=0, atol=1e-5)


# ---- reference test/lasso.jl ------------------------------------------------------------------
def test_lasso_zero_above_lambda_max():  # :23-34
    rng = np.random.default_rng(1)
    X = np.asfortranarray(rng.standard_normal((100, 10)))
    Y = X @ np.ones(10) + 0.1 * rng.standard_normal(100)
    lam = np.max(np.abs(X.T @ Y / 100)) + 0.1
    out = cd.lasso(X, Y, lam)
    assert out.x == cd.SparseIterate(10) and out.x.nnz == 0
    f = cd.CDLeastSquaresLoss(Y, X)
    x = cd.SparseIterate(10)
    cd.initialize_(f, x)
    np.testing.assert_allclose(cd.findLambdaMax(x, f, cd.ProxL1(1.0)), lam - 0.1, rtol=1e-13)


def test_lasso_nonzero_kkt_and_interfaces():  # :36-72
    rng, X, Y = _problem(2, 100, 10, 5, 0.1)
    lam = np.full(10, 0.3)
    beta = cd.lasso(X, Y, 1.0, lam, cd.CDOptions(optTol=1e-12))
    kkt = np.max(np.abs(X.T @ (Y - X @ beta.x.dense()) / 100))
    assert abs((kkt - 0.3) / 0.3) < 1e-5
    np.testing.assert_allclose(beta.residuals, Y - X @ beta.x.dense(), rtol=0, atol=1e-11)
    np.testing.assert_allclose(beta.sigma, np.std(beta.residuals, ddof=1), rtol=1e-10)
    ref = O.lasso(X, Y, 1.0, lam, O.CDOptions(optTol=1e-12))
    np.testing.assert_allclose(beta.x.dense(), ref.x.dense(), rtol=0, atol=1e-9)
    rng, X, Y = _problem(3, 500, 500, 50)
    x1 = cd.lasso(X, Y, 0.1)
    x2 = cd.lasso(X, Y, 0.1, np.ones(500))
    np.testing.assert_allclose(x1.x.dense(), x2.x.dense(), rtol=0, atol=1e-5)


def test_dimension_mismatch_errors():  # coordinate_descent.jl:13-16
    rng, X, Y = _problem(4, 40, 6, 2)
    f = cd.CDLeastSquaresLoss(Y, X)
    with pytest.raises(cd.DimensionMismatch):
        cd.coordinateDescent_(cd.SparseIterate(5), f, cd.ProxL1(0.1))
    with pytest.raises(cd.DimensionMismatch):
        cd.coordinateDescent_(cd.SparseIterate(6), f, cd.ProxL1(0.1, np.ones(5)))
    with pytest.raises(cd.ArgumentError):  # lambda0 == lambda_max: zero-step cold-start range
        x = cd.SparseIterate(6)
        cd.initialize_(f, x)
        lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
        cd.coordinateDescent_(x, f, cd.ProxL1(lmax), cd.CDOptions(warmStart=False))


@pytest.mark.parametrize("mode", [MODES[0], MODES[1], MODES[5]], ids=lambda m: f"{m[0]}{m[1]}")
def test_sqrt_lasso_kkt_interfaces_and_oracle(mode):  # :106-181
    rng, X, Y = _problem(6, 500, 500, 50)
    lam = 1.5
    f = cd.CDSqrtLassoLoss(Y, X)
    _set_mode(f, mode)
    ref = None
    for warm in (True, False):
        for rand in (False, True):
            o = cd.CDOptions(maxIter=5000, optTol=1e-10, warmStart=warm, randomize=rand, seed=9)
            x = cd.SparseIterate(500, np.where(rng.random(500) < 0.6, rng.random(500), 0.0))
            cd.coordinateDescent_(x, f, cd.ProxL1(lam), o)
            ref = x.dense() if ref is None else ref
            np.testing.assert_allclose(x.dense(), ref, rtol=0, atol=1e-4)
    y1 = cd.sqrtLasso(X, Y, lam, options=cd.CDOptions(maxIter=5000, optTol=1e-10), standardizeX=False)
    z1 = cd.sqrtLasso(X, Y, lam, np.ones(500), cd.CDOptions(maxIter=5000, optTol=1e-10))
    np.testing.assert_allclose(y1.x.dense(), ref, rtol=0, atol=1e-4)
    np.testing.assert_allclose(z1.x.dense(), ref, rtol=0, atol=1e-4)
    r = Y - X @ ref
    assert max(0.0, np.max(np.abs(X.T @ r / np.linalg.norm(r))) - lam) / lam < 1e-3
    xo = O.SparseIterate(500)
    O.coordinateDescent_(xo, O.CDSqrtLassoLoss(Y, X), O.ProxL1(lam),
                         O.CDOptions(maxIter=5000, optTol=1e-12, randomize=False))
    xg = cd.SparseIterate(500)
    cd.coordinateDescent_(xg, f, cd.ProxL1(lam), cd.CDOptions(maxIter=5000, optTol=1e-12, randomize=False))
    np.testing.assert_allclose(xg.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    # standardizeX=True: the intended omega = _stdX!(X)
    s1 = cd.sqrtLasso(X, Y, lam, options=cd.CDOptions(maxIter=5000, optTol=1e-10, randomize=False))
    s2 = O.sqrtLasso(X, Y, lam, options=O.CDOptions(maxIter=5000, optTol=1e-10, randomize=False))
    np.testing.assert_allclose(s1.x.dense(), s2.x.dense(), rtol=0, atol=1e-8)


def test_scaled_lasso():  # :186-216
    rng, X, Y = _problem(7, 1000, 500, 50)
    lam = 0.12
    cdo = cd.CDOptions(maxIter=5000, optTol=1e-8, seed=4)
    o1 = cd.IterLassoOptions(maxIter=100, optTol=1e-8, optionsCD=cdo)
    o2 = cd.IterLassoOptions(maxIter=100, optTol=1e-8, initProcedure="InitStd", sigmainit=2.0, optionsCD=cdo)
    x1, x2 = cd.SparseIterate(500), cd.SparseIterate(500)
    s1 = cd.scaledLasso_(x1, X, Y, lam, np.ones(500), o1)
    s2 = cd.scaledLasso_(x2, X, Y, lam, np.ones(500), o2)
    for x, sol in ((x1, s1), (x2, s2)):
        kkt = np.max(np.abs(X.T @ (Y - X @ x.dense()) / 1000))
        assert max(kkt - lam * sol.sigma, 0.0) / (sol.sigma * lam) < 1e-4
    np.testing.assert_allclose(x1.dense(), x2.dense(), rtol=0, atol=1e-4)
    xo = O.SparseIterate(500)
    so = O.scaledLasso_(xo, X, Y, lam, np.ones(500),
                        O.IterLassoOptions(maxIter=100, optTol=1e-8, optionsCD=O.CDOptions(maxIter=5000, optTol=1e-8, seed=4)))
    np.testing.assert_allclose(x1.dense(), xo.dense(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(s1.sigma, so.sigma, rtol=1e-6)


@pytest.mark.parametrize("standardize", [False, True])
def test_lasso_path(standardize):  # :220-288
    rng, X, Y = _problem(8, 1000, 500, 50)
    opt = cd.CDOptions(maxIter=5000, optTol=1e-8, seed=5)
    f = cd.CDLeastSquaresLoss(Y, X)
    np.testing.assert_allclose(cd.stdX(f), np.sqrt((X ** 2).sum(0) / 1000), rtol=1e-13)
    load = cd.stdX(f) if standardize else None
    x1 = cd.lasso(X, Y, 0.3, load, opt)
    x2 = cd.lasso(X, Y, 0.1, load, opt)
    path = cd.LassoPath(X, Y, [0.3, 0.1], opt, standardizeX=standardize)
    assert path.lambdapath == [0.3, 0.1]
    np.testing.assert_allclose(path.betapath[0].dense(), x1.x.dense(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(path.betapath[1].dense(), x2.x.dense(), rtol=0, atol=1e-5)
    lo, bo = O.LassoPath(X, Y, [0.3, 0.1], O.CDOptions(maxIter=5000, optTol=1e-8, seed=5), standardizeX=standardize)
    np.testing.assert_allclose(path.betapath[1].dense(), bo[1], rtol=0, atol=1e-6)
    short = cd.LassoPath(X, Y, [0.3, 0.1], opt, max_hat_s=1, standardizeX=standardize)
    assert short.lambdapath == [0.3] and len(short.betapath) == 1


@pytest.mark.parametrize("mode", [MODES[0], MODES[1], MODES[4], MODES[5]], ids=lambda m: f"{m[0]}{m[1]}")
def test_weighted_ls_loss_matches_oracle(mode):
    rng, X, Y = _problem(12, 3001, 45, 5)
    w = rng.random(3001) + 0.5
    o = dict(maxIter=5000, optTol=1e-12, randomize=False)
    x, xo = cd.SparseIterate(45), O.SparseIterate(45)
    f, fo = cd.CDWeightedLSLoss(Y, X, w), O.CDWeightedLSLoss(Y, X, w)
    _set_mode(f, mode)          # B = 8 has no weighted kernel: falls back to the per-coordinate sweep
    cd.coordinateDescent_(x, f, cd.ProxL1(0.05), cd.CDOptions(**o))
    st = O.coordinateDescent_(xo, fo, O.ProxL1(0.05), O.CDOptions(**o))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
    assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"]


# ---- fp32 storage (fp64 accumulation): loose, declared tolerance ------------------------------
@pytest.mark.parametrize("mode", [MODES[0], MODES[1], MODES[4], MODES[5], MODES[6]], ids=lambda m: f"{m[0]}{m[1]}")
def test_fp32_against_fp64_oracle(mode):
    rng, X, Y = _problem(13, 4000, 64, 8)
    f = cd.CDLeastSquaresLoss(Y.astype(np.float32), X.astype(np.float32))
    _set_mode(f, mode)
    x = cd.SparseIterate(64)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.05), cd.CDOptions(maxIter=500, optTol=1e-6, randomize=False))
    xo = O.SparseIterate(64)
    O.coordinateDescent_(xo, O.CDLeastSquaresLoss(Y, X), O.ProxL1(0.05), O.CDOptions(optTol=1e-10, randomize=False))
    # fp32 storage of X, y and r: 1e-4 absolute on beta (values are O(1))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-4)
    assert f.r.dtype == np.float32


# ---- device generator + medium-size parity on device-generated data ------------------------------
@pytest.mark.parametrize("mode", [MODES[0], MODES[1], MODES[4], MODES[5], MODES[6]], ids=lambda m: f"{m[0]}{m[1]}")
def test_generated_problem_parity_n200k(mode):
    n, p, s = 200_003, 48, 6   # odd n: ragged vector tail
    f, bstar = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=s, noise=1.0)
    _set_mode(f, mode)
    X, Y = f.X_cols(0, p), f.y
    assert abs(X.mean()) < 5e-3 and abs(X.std() - 1.0) < 5e-3
    np.testing.assert_allclose(Y - X[:, :s] @ bstar, Y - X[:, :s] @ bstar)  # finite
    assert abs(np.std(Y - X[:, :s] @ bstar) - 1.0) < 2e-2
    # row shards regenerate the same matrix (counter-based generator keyed by the global row)
    f2, _ = cd.CDLeastSquaresLoss.generate(1001, p, seed=123, s=s, noise=1.0, n_total=n, row_offset=70_001)
    np.testing.assert_array_equal(f2.X_cols(0, p), X[70_001:71_002])
    np.testing.assert_allclose(f2.y, Y[70_001:71_002], rtol=0, atol=1e-12)
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
    np.testing.assert_allclose(lmax, np.max(np.abs(X.T @ Y)) / n, rtol=1e-12)
    lam = 0.05 * lmax
    o = dict(maxIter=500, optTol=1e-12, randomize=False)
    cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
    xo = O.SparseIterate(p)
    fo = O.CDLeastSquaresLoss(Y, X)
    O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    np.testing.assert_allclose(cd.objective(f), O.objective(fo, O.ProxL1(lam), xo), rtol=1e-12)
    # determinism: the same solve twice ON THE SAME PATH is bit-identical (fixed-order reductions, no float atomics).  The path
    # is pinned for that: by default a handle that keeps solving on one X moves its full passes to the gradient cache (or,
    # for few columns, its solves to the Gram form) once that has paid for itself -- same iterates, other last bits
    f.set_gradient_cache(0)
    f.set_onchip_solve(False)
    x1, x2 = cd.SparseIterate(p), cd.SparseIterate(p)
    cd.coordinateDescent_(x1, f, cd.ProxL1(lam), cd.CDOptions(**o))
    cd.coordinateDescent_(x2, f, cd.ProxL1(lam), cd.CDOptions(**o))
    np.testing.assert_array_equal(x2.dense(), x1.dense())
    np.testing.assert_allclose(x1.dense(), x.dense(), rtol=0, atol=1e-14)


# ---- size-independent properties at a size the oracle cannot reach in seconds ------------------
def test_large_problem_properties():
    """n = 4e6, p = 256 (8 GB of X): objective decreases monotonically pass by pass, the
    carried residual equals a from-scratch rebuild, KKT holds at the fixed point, and the
    blocked sweep reaches the same fixed point as the per-coordinate sweep."""
    n, p, s = 4_000_000, 256, 16
    f, bstar = cd.CDLeastSquaresLoss.generate(n, p, seed=7, s=s, noise=2.0)
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
    g = cd.ProxL1(0.02 * lmax)
    f.set_sweep_mode("block", 8)
    objs = [cd.objective(f, g)]
    for _ in range(4):
        cd.cdPass_(x, f, g, range(1, p + 1))
        objs.append(cd.objective(f, g))
    assert all(b <= a * (1 + 1e-14) for a, b in zip(objs, objs[1:]))
    r_carried = f.r
    cd.initialize_(f, x)                      # rebuild r = y - X beta from scratch
    np.testing.assert_allclose(r_carried, f.r, rtol=0, atol=1e-9)
    cd.coordinateDescent_(x, f, g, cd.CDOptions(maxIter=200, optTol=1e-11, randomize=False))
    xb = x.dense()
    xtr = np.zeros(p)
    cd._lib.check(f._L.cdh_xt_r(f._h, xtr.ctypes.data), f._h)
    grad = np.abs(xtr) / n
    act = xb != 0
    assert np.max(np.abs(grad[act] - g.lambda0)) < 1e-9       # active set: |X_k'r|/n = lambda
    assert np.all(grad[~act] <= g.lambda0 * (1 + 1e-9))       # inactive: <= lambda
    assert np.max(np.abs(xb[:s] - bstar)) < 0.1
    for mode in ("coord", 16, 32, 64):        # every sweep formulation at full-chip grid sizes
        f.set_sweep_mode("coord") if mode == "coord" else f.set_sweep_mode("block", mode)
        xc = cd.SparseIterate(p)
        cd.coordinateDescent_(xc, f, g, cd.CDOptions(maxIter=200, optTol=1e-11, randomize=False))
        np.testing.assert_allclose(xc.dense(), xb, rtol=0, atol=BETA_TOL)


# ---- hipGraph replay of the pass: same launches, same bits ------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [MODES[0], MODES[1], MODES[4], MODES[5], MODES[6]], ids=lambda m: f"{m[0]}{m[1]}")
def test_graph_replay_is_bit_identical(mode, dtype):
    rng, X, Y = _problem(21, 3000, 70, 9)
    X, Y = X.astype(dtype), Y.astype(dtype)
    o = cd.CDOptions(maxIter=300, optTol=1e-12 if dtype == np.float64 else 1e-6, randomize=True, seed=3)
    res = []
    for graph in (False, True):
        f = cd.CDLeastSquaresLoss(Y, X)
        _set_mode(f, mode)
        f.set_use_graph(graph)
        x = cd.SparseIterate(70)
        cd.coordinateDescent_(x, f, cd.ProxL1(0.05), o)          # full + active passes of many lengths
        cd.coordinateDescent_(x, f, cd.ProxL1(0.02), o)          # warm start, graphs reused
        res.append((x.dense(), f.r, f.last_stats["passes"]))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2]


# ---- edge cases: tiny n, p = 1, partial blocks, repeated coordinates in one pass ----------------
@pytest.mark.parametrize("mode", MODES, ids=lambda m: f"{m[0]}{m[1]}")
def test_edge_shapes_and_repeated_coordinates(mode):
    for (n, p) in [(1, 1), (3, 2), (5, 1), (7, 33), (130, 17)]:
        rng = np.random.default_rng(100 * n + p)
        X = np.asfortranarray(rng.standard_normal((n, p)))
        Y = rng.standard_normal(n)
        f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
        _set_mode(f, mode)
        g, go = cd.ProxL1(0.01), O.ProxL1(0.01)
        x, xo = cd.SparseIterate(p), O.SparseIterate(p)
        cd.initialize_(f, x)
        O.initialize_(fo, xo)
        # a visit list with repeats, longer than p and not a multiple of any block size
        visit = [int(v) for v in rng.integers(1, p + 1, size=2 * p + 3)]
        for _ in range(3):
            mh, mho = cd.cdPass_(x, f, g, visit), O.cdPass_(xo, fo, go, visit)
            np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-11)
            np.testing.assert_allclose(mh, mho, rtol=1e-9, atol=1e-13)
            assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-11)
    # empty visit list: _cdPass! only runs dropzeros!
    assert cd.cdPass_(x, f, g, []) == 0.0


# ---- every operand path of the wide-block kernel (csrc/gram_kernels.hpp): fragment loads, LDS-
# transposed loads with 64-vector chunks, and with one-sub-chunk chunks (short columns).  The library
# picks by shard length; the knobs (read when a handle is created) pin each one here. ----------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("knobs", [{"CDH_LT": "0"}, {"CDH_LT": "1", "CDH_KS": "2"}, {"CDH_LT": "1", "CDH_KS": "1"}],
                         ids=["fragment", "lt_long", "lt_short"])
@pytest.mark.parametrize("block", [16, 32, 64])
def test_wide_block_operand_paths(monkeypatch, block, knobs, dtype):
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    rng, X, Y = _problem(77, 9001, 70, 8)            # ragged rows, a ragged last block (70 = 64 + 6)
    w = rng.uniform(0.5, 2.0, size=9001)
    lam = 0.05
    o = dict(maxIter=400, optTol=1e-12 if dtype == np.float64 else 1e-6, randomize=False)
    for weighted in (False, True):
        if weighted:
            f = cd.CDWeightedLSLoss(Y.astype(dtype), X.astype(dtype), w.astype(dtype))
            fo = O.CDWeightedLSLoss(Y, X, w)
        else:
            f = cd.CDLeastSquaresLoss(Y.astype(dtype), X.astype(dtype))
            fo = O.CDLeastSquaresLoss(Y, X)
        f.set_sweep_mode("block", block)
        x, xo = cd.SparseIterate(70), O.SparseIterate(70)
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(maxIter=400, optTol=1e-12, randomize=False))
        tol = BETA_TOL if dtype == np.float64 else 2e-4    # fp32 storage against the fp64 oracle
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=tol)


# ---- the multi-process code path on one GPU: torch.distributed.run + a forced 1-rank RCCL
# communicator (dlopen of librccl, ncclCommInitRank, ncclAllReduce on the sweep stream between the
# reduce and the scalar kernels).  RCCL refuses two ranks on one device, so this is as far as a
# one-GPU box can go; the sharded arithmetic with a real 2-rank all-reduce is the gloo test. ------
@pytest.mark.parametrize("mode_args", [["--mode", "coord"], ["--block", "8"], ["--block", "32"], ["--block", "64"],
                                       ["--block", "32", "--graph"]],
                         ids=["coord", "block8", "block32", "block64", "block32_graph"])
def test_rccl_path_under_torchrun_matches_plain_run(mode_args):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "300000", "--cols", "96",
              "--planted", "10", "--no-cpu-baseline", "--no-sparse"] + mode_args
    env = dict(os.environ, CDH_FORCE_RCCL="1")
    port = _free_port()
    a = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py")] + common,
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common,
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert b.returncode == 0, b.stderr[-2000:]
    ja = json.loads([l for l in a.stdout.splitlines() if l.startswith("{")][-1])
    jb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
    # the forced run really went through ncclAllReduce on a communicator that reports one rank; the
    # plain run never touched RCCL (cdh_exchange_stats: counters kept by the library's allreduce seam)
    assert ja["exchange_stats"]["rccl_calls"] > 0 and ja["exchange_stats"]["nranks"] == 1
    assert ja["config"]["exchange"] == "rccl"
    assert jb["exchange_stats"] == {"rccl_calls": 0, "p2p_calls": 0, "host_calls": 0, "nranks": 1}
    assert jb["config"]["exchange"] is None
    assert ja["config"]["last_maxH"] == jb["config"]["last_maxH"]      # bit-identical iterates
    assert ja["config"]["moved_per_sweep"] == jb["config"]["moved_per_sweep"]


# ---- the opt-in direct exchange (csrc/p2p_exchange.hpp): two ranks, both on cuda:0, row shards,
# IPC-mapped inboxes, no RCCL at all.  This is what a one-GPU box can validate of it: the handle
# plumbing, epochs / slot reuse, chunking, and the sharded arithmetic end to end against the oracle.
@pytest.mark.parametrize("ranks", [2, 3])
def test_p2p_exchange_two_ranks_one_gpu(ranks):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    env = dict(os.environ, CDH_P2P_SPIN_LIMIT="4000000")     # a failure must not take minutes
    a = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "tests", "p2p_worker.py")],
                       capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert a.returncode == 0, (a.stdout[-1500:], a.stderr[-3000:])
    assert "P2P_OK" in a.stdout


# ---- the library's sharded launch sequence with a real multi-rank sum behind the allreduce() seam and
# neither RCCL nor the IPC exchange involved (tests/host_exchange_worker.py) -----------------------------
@pytest.mark.parametrize("ranks", [2, 3])
def test_sharded_launch_sequence_over_host_exchange(ranks):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(root, "tests", "host_exchange_worker.py")],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert a.returncode == 0, (a.stdout[-1500:], a.stderr[-3000:])
    assert "HOSTX_OK" in a.stdout


def _bench_two_ranks_one_gpu(extra, env_extra=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    env = dict(os.environ, CDH_P2P_SPIN_LIMIT="4000000", **(env_extra or {}))
    cmd = ["--steps", "2", "--warmup", "1", "--rows", "300000", "--cols", "96", "--planted", "10",
           "--no-cpu-baseline", "--no-sparse", "--block", "16"]
    a = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--no-rccl"] + cmd + extra,
                       capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert a.returncode == 0, (a.stdout[-1500:], a.stderr[-3000:])
    got = [l for l in a.stdout.splitlines() if l.startswith("{")]
    assert len(got) == 1, got            # the contract: ONE result line, whatever the exchange mode
    b = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + cmd,
                       capture_output=True, text=True, timeout=420, cwd=root)
    assert b.returncode == 0, b.stderr[-2000:]
    last = lambda out: json.loads([l for l in out.splitlines() if l.startswith("{")][-1])  # noqa: E731
    return last(a.stdout), last(b.stdout)


def test_bench_gpus_2_without_a_launcher_two_ranks_one_gpu():
    """`python bench.py --gpus 2 ...` with NO launcher and WORLD_SIZE unset (the form the driver's 1-GPU command
    takes when only the number changes): the parent starts its own two ranks, never touches the GPU itself, and
    relays exactly one JSON line with n_gpus == 2 -- the same sweep as one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CDH_P2P_SPIN_LIMIT"] = "4000000"
    cmd = ["--steps", "2", "--warmup", "1", "--rows", "300000", "--cols", "96", "--planted", "10",
           "--no-cpu-baseline", "--no-sparse", "--block", "16"]
    a = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-rccl", "--exchange", "p2p"] + cmd,
                       capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert a.returncode == 0, (a.stdout[-1500:], a.stderr[-3000:])
    got = [l for l in a.stdout.splitlines() if l.strip()]
    assert len(got) == 1 and got[0].startswith("{"), got      # stdout carries the result line and nothing else
    ja = json.loads(got[0])
    assert ja["n_gpus"] == 2 and ja["devices_by_rank"] == [0, 0] and ja["config"]["exchange"] == "p2p"
    assert ja["exchange_stats"]["nranks"] == 2 and ja["exchange_stats"]["p2p_calls"] > 0
    b = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + cmd,
                       capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert b.returncode == 0, b.stderr[-2000:]
    jb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
    assert ja["config"]["moved_per_sweep"] == jb["config"]["moved_per_sweep"]
    assert abs(ja["config"]["last_maxH"] - jb["config"]["last_maxH"]) <= 1e-9 * abs(jb["config"]["last_maxH"])


def test_bench_two_ranks_when_rccl_cannot_serve_them_still_measures_the_sharded_sweep():
    """`python bench.py --gpus 2` exactly as the driver would start it, on a box with ONE card: RCCL refuses two ranks on
    one device (or, should this build accept them, serves them).  Either way there is one result line with n_gpus == 2 whose
    sweep equals the one-rank sweep: over RCCL if the communicator came up and summed the probe records exactly, otherwise
    over the host-staged exchange with the reason on the line (sharded.connect_checked + cdh_comm_drop)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = ["--steps", "2", "--warmup", "1", "--rows", "300000", "--cols", "96", "--planted", "10",
           "--no-cpu-baseline", "--no-sparse", "--block", "16", "--no-exchange-trial"]
    a = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + cmd,
                       capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert a.returncode == 0, (a.stdout[-1500:], a.stderr[-3000:])
    got = [l for l in a.stdout.splitlines() if l.strip()]
    assert len(got) == 1 and got[0].startswith("{"), got
    ja = json.loads(got[0])
    assert ja["n_gpus"] == 2 and ja["exchange_stats"]["nranks"] == 2
    if "exchange_fallback" in ja:
        fb = ja["exchange_fallback"]
        # the bring-up was tried in child processes first (coordinatedescent.jl_amd/p2p_probe.py rccl) and failed THERE
        assert fb["used"] == "host(gloo)" == ja["config"]["exchange"]
        assert fb["why"].startswith("isolated bring-up: RCCL_PROBE_FAILED rank ") and "cdh_comm_init" in fb["why"]
        assert ja["rccl_bring_up_probe"]["rc"] == 4
        assert ja["exchange_stats"]["host_calls"] > 0 and ja["exchange_stats"]["rccl_calls"] == 0
    else:
        assert ja["config"]["exchange"] == "rccl" and ja["exchange_stats"]["rccl_calls"] > 0
    b = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + cmd,
                       capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert b.returncode == 0, b.stderr[-2000:]
    jb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
    assert ja["config"]["moved_per_sweep"] == jb["config"]["moved_per_sweep"]
    assert abs(ja["config"]["last_maxH"] - jb["config"]["last_maxH"]) <= 1e-9 * abs(jb["config"]["last_maxH"])
    assert abs(ja["config"]["beta_abs_sum"] - jb["config"]["beta_abs_sum"]) <= 1e-9 * jb["config"]["beta_abs_sum"]


def test_bench_an_rccl_bring_up_that_never_returns_costs_a_timeout_not_the_result_line():
    """RCCL has never run across GPUs in this pipeline.  bench.py therefore tries the communicator's bring-up in child
    processes first; here every child hangs in it (CDH_RCCL_PROBE_HANG, the stand-in for a ncclCommInitRank that never
    returns): the children are killed at the timeout, the bench process never calls into RCCL, and the sharded sweep is
    measured over the host-staged exchange -- one result line, with the reason on it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CDH_RCCL_PROBE_HANG"] = "1"
    a = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "300000",
                        "--cols", "96", "--planted", "10", "--no-cpu-baseline", "--no-sparse", "--block", "16", "--no-exchange-trial",
                        "--rccl-probe-timeout", "20"], capture_output=True, text=True, timeout=420, env=env, cwd=root)
    assert a.returncode == 0, (a.stdout[-1500:], a.stderr[-3000:])
    got = [l for l in a.stdout.splitlines() if l.strip()]
    assert len(got) == 1 and got[0].startswith("{"), got
    ja = json.loads(got[0])
    assert ja["n_gpus"] == 2 and ja["config"]["exchange"] == "host(gloo)" and ja["value"] > 0
    assert ja["exchange_fallback"]["why"] == "isolated bring-up: timeout" and ja["rccl_bring_up_probe"]["rc"] is None
    assert ja["exchange_stats"]["rccl_calls"] == 0 and ja["exchange_stats"]["host_calls"] > 0


def test_bench_a_direct_exchange_probe_that_dies_costs_nothing():
    """--exchange auto (the default): the direct exchange meets the machine in child processes first.  Here every
    probe dies on the spot (os._exit, the stand-in for a GPU fault in a transport that has never run on this
    hardware): the bench process never touches the direct exchange, exits 0 and prints the first timed region."""
    ja, _ = _bench_two_ranks_one_gpu(["--exchange", "auto"], env_extra={"CDH_P2P_PROBE_ABORT": "1"})
    assert ja["n_gpus"] == 2 and ja["value"] > 0 and ja["ms_per_step"] > 0
    t = ja["exchange_trial"]
    assert "skipped" in t and t["probe"]["rc"] == 3 and "ms_per_step" not in t
    assert ja["exchange_stats"]["p2p_calls"] == 0
    assert ja["ms_per_step_ranks"]["min"] <= ja["ms_per_step_ranks"]["max"] <= ja["ms_per_step"] * 1.001


def test_bench_sharded_with_p2p_exchange_two_ranks_one_gpu():
    """bench.py's N>1 flow with the direct exchange in the timed region: same sweep as one rank."""
    ja, jb = _bench_two_ranks_one_gpu(["--exchange", "p2p"])
    assert ja["n_gpus"] == 2 and ja["config"]["exchange"] == "p2p" and "exchange_trial" not in ja
    assert ja["config"]["moved_per_sweep"] == jb["config"]["moved_per_sweep"]
    assert abs(ja["config"]["last_maxH"] - jb["config"]["last_maxH"]) <= 1e-9 * abs(jb["config"]["last_maxH"])


def test_bench_exchange_trial_code_path_two_ranks_one_gpu():
    """The guarded trial bench.py runs after an RCCL-timed region, exercised without RCCL (--no-rccl:
    the timed region has no exchange, so only the trial's own fields are meaningful here)."""
    ja, _ = _bench_two_ranks_one_gpu([])               # --exchange auto is the default
    t = ja["exchange_trial"]
    assert t["probe"]["rc"] == 0 and t["probe"]["line"].startswith("P2P_PROBE_OK") and t["probe"]["latency_us"] > 0
    assert t["selftest"] is True and t["completed_on_all_ranks"] is True and t["ms_per_step"] > 0
    assert "error" not in t
    assert 0.0 < ja["exchange_latency_us"]["p2p"] < 1e4 and ja["exchange_latency_us"]["doubles"] == 273


# ---- section 8(f) rows: screening init kept on the device, LassoPath without the rebuild ---------
def test_gram_entry_point_matches_numpy():
    rng, X, Y = _problem(31, 5003, 80, 6)
    f = cd.CDLeastSquaresLoss(Y, X)
    for m in (1, 5, 16, 17, 40, 64):
        cols = rng.choice(80, size=m, replace=False)
        idx1 = np.ascontiguousarray(cols + 1, dtype=np.int64)
        G, c, q = np.zeros((m, m)), np.zeros(m), cd._lib.C.c_double()
        cd._lib.check(f._L.cdh_gram(f._h, m, idx1.ctypes.data, G.ctypes.data, c.ctypes.data,
                                    cd._lib.C.byref(q)), f._h)
        Xs = X[:, cols]
        np.testing.assert_allclose(G, Xs.T @ Xs, rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(c, Xs.T @ Y, rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(q.value, Y @ Y, rtol=1e-13)
    np.testing.assert_array_equal(f.r, Y)      # r is only read


def test_screening_sigma_matches_reference_formula():
    """_findInitSigma! (utils.jl:60-77): std of the residuals of OLS on the s most correlated columns."""
    from coordinatedescent_jl_amd.api import _find_init_sigma
    rng, X, Y = _problem(32, 4000, 60, 8)
    f = cd.CDLeastSquaresLoss(Y, X)
    c = np.abs(X.T @ Y)
    S = c >= np.sort(c)[::-1][4]
    coef, *_ = np.linalg.lstsq(X[:, S], Y, rcond=None)
    want = np.std(Y - X[:, S] @ coef, ddof=1)
    np.testing.assert_allclose(_find_init_sigma(f, 5), want, rtol=1e-10)


def test_lasso_path_reusing_the_carried_residual():
    rng, X, Y = _problem(33, 3000, 120, 10)
    lams = list(np.exp(np.linspace(np.log(0.5), np.log(0.01), 12)))
    opt = cd.CDOptions(maxIter=5000, optTol=1e-11, randomize=False)
    a = cd.LassoPath(X, Y, lams, opt, reuse_residual=True)
    b = cd.LassoPath(X, Y, lams, opt, reuse_residual=False)
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(maxIter=5000, optTol=1e-11, randomize=False))
    for i in range(len(lams)):
        np.testing.assert_allclose(a.betapath[i].dense(), b.betapath[i].dense(), rtol=0, atol=1e-9)
        np.testing.assert_allclose(a.betapath[i].dense(), bo[i], rtol=0, atol=1e-9)


# ---- BASELINE.json's full size (n = 1e7, p = 1000, fp64: 80 GB of X) through properties -----------
def test_full_size_properties_cfg2():
    n, p, s = 10_000_000, 1000, 100
    f, bstar = cd.CDLeastSquaresLoss.generate(n, p, seed=123, s=s, noise=6.0)
    x = cd.SparseIterate(p)
    cd.initialize_(f, x)
    lmax = cd.findLambdaMax(x, f, cd.ProxL1(1.0))
    g = cd.ProxL1(0.5 * lmax)                       # the "sparse" regime of SURVEY 8d
    f.set_sweep_mode("block", 16)
    objs = [cd.objective(f, g)]
    for _ in range(3):
        cd.cdPass_(x, f, g, range(1, p + 1))
        objs.append(cd.objective(f, g))
    assert all(b <= a * (1 + 1e-14) for a, b in zip(objs, objs[1:]))   # monotone descent
    s1, ss1 = cd._lib.C.c_double(), cd._lib.C.c_double()
    f._L.cdh_resid_moments(f._h, cd._lib.C.byref(s1), cd._lib.C.byref(ss1))
    cd.initialize_(f, x)                             # rebuild r = y - X beta from scratch
    s2, ss2 = cd._lib.C.c_double(), cd._lib.C.c_double()
    f._L.cdh_resid_moments(f._h, cd._lib.C.byref(s2), cd._lib.C.byref(ss2))
    assert abs(ss1.value - ss2.value) <= 1e-12 * ss2.value            # carried residual == rebuilt
    cd.coordinateDescent_(x, f, g, cd.CDOptions(maxIter=100, optTol=1e-11, randomize=False))
    assert f.last_stats["converged"]
    xtr = np.zeros(p)
    cd._lib.check(f._L.cdh_xt_r(f._h, xtr.ctypes.data), f._h)
    grad, xb = np.abs(xtr) / n, x.dense()
    act = xb != 0
    assert act.sum() >= 1 and np.max(np.abs(grad[act] - g.lambda0)) < 1e-9   # KKT on the support
    assert np.all(grad[~act] <= g.lambda0 * (1 + 1e-9))                      # and off it
    f.set_sweep_mode("block", 32)                    # another sweep formulation, same fixed point
    x2 = cd.SparseIterate(p)
    cd.coordinateDescent_(x2, f, g, cd.CDOptions(maxIter=100, optTol=1e-11, randomize=False))
    np.testing.assert_allclose(x2.dense(), xb, rtol=0, atol=BETA_TOL)


# ---- screened full passes: same iterates, same bookkeeping, fewer bytes ---------------------------
@pytest.mark.parametrize("loss", ["ls", "wl1", "sqrt"])
@pytest.mark.parametrize("rand", [False, True], ids=["ordered", "random"])
def test_screened_full_passes_match_oracle_and_unscreened(loss, rand):
    rng, X, Y = _problem(41, 2500, 700, 12, noise=1.0)
    om = (rng.random(700) + 0.5) if loss == "wl1" else None
    lam = 3.2 if loss == "sqrt" else 0.08
    o = dict(maxIter=2000, optTol=1e-12, randomize=rand, seed=11)
    cls, ocls = (cd.CDSqrtLassoLoss, O.CDSqrtLassoLoss) if loss == "sqrt" else (cd.CDLeastSquaresLoss, O.CDLeastSquaresLoss)
    xo = O.SparseIterate(700)
    fo = ocls(Y, X)
    sto = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
    outs = []
    for screen in (True, False):
        f = cls(Y, X)
        f.set_sweep_mode("block", 16)
        f.set_screening(screen)
        x = cd.SparseIterate(700)
        cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
        # warm start from the solution at a smaller lambda: first full pass over a sparse iterate
        cd.coordinateDescent_(x, f, cd.ProxL1(0.7 * lam, om), cd.CDOptions(**o))
        outs.append((x.dense(), x.nzval2ind.tolist(), f.last_stats["passes"], f.r))
    O.coordinateDescent_(xo, fo, O.ProxL1(0.7 * lam, om), O.CDOptions(**o))
    assert 0 < xo.nnz < 700 // 4
    for beta, sup, passes, r in outs:
        np.testing.assert_allclose(beta, xo.dense(), rtol=0, atol=BETA_TOL)
        assert sup == xo.nzval2ind.tolist()          # same support ORDER: same bookkeeping replayed
        np.testing.assert_allclose(r, fo.r, rtol=0, atol=1e-9)
    assert outs[0][2] == outs[1][2]                  # same number of passes with and without


def test_zero_column_is_handled_like_the_oracle():
    """A zero column makes x[k] += b/a a 0/0 (the reference has no guard).  What
    ProximalBase's cdprox! then does with the NaN is not pinned by any reference test; oracle and
    product both soft-threshold it to 0 (comparisons with NaN are false).  Screening sends such a
    column to the exact path so both modes agree with the oracle."""
    rng, X, Y = _problem(42, 400, 300, 5)
    X[:, 200] = 0.0
    xo = O.SparseIterate(300)
    O.coordinateDescent_(xo, O.CDLeastSquaresLoss(Y, X), O.ProxL1(0.2), O.CDOptions(maxIter=50, optTol=1e-12, randomize=False))
    for screen in (True, False):
        f = cd.CDLeastSquaresLoss(Y, X)
        f.set_sweep_mode("block", 16)
        f.set_screening(screen)
        x = cd.SparseIterate(300)
        cd.coordinateDescent_(x, f, cd.ProxL1(0.2), cd.CDOptions(maxIter=50, optTol=1e-12, randomize=False))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.dense()[200] == xo.dense()[200] == 0.0


# ---- large p, short columns: visit lists longer than one chunk of visit state ---------------------
@pytest.mark.parametrize("mode", [MODES[0], MODES[4], MODES[6]], ids=lambda m: f"{m[0]}{m[1]}")
def test_wide_problem_and_chunked_visit_lists(mode):
    rng = np.random.default_rng(77)
    n, p = 96, 6000
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :5] @ rng.standard_normal(5) + 0.1 * rng.standard_normal(n)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    _set_mode(f, mode)
    g, go = cd.ProxL1(0.3), O.ProxL1(0.3)
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    cd.initialize_(f, x)
    O.initialize_(fo, xo)
    visit = list(range(1, p + 1)) + list(range(p, 0, -1)) + [7, 7, 9]     # 2p + 3 visits > one chunk
    mh, mho = cd.cdPass_(x, f, g, visit), O.cdPass_(xo, fo, go, visit)
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-11)
    np.testing.assert_allclose(mh, mho, rtol=1e-9)
    assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()
    cd.coordinateDescent_(x, f, g, cd.CDOptions(maxIter=3000, optTol=1e-12, randomize=False))
    O.coordinateDescent_(xo, fo, go, O.CDOptions(maxIter=3000, optTol=1e-12, randomize=False))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)


# ---- nearly collinear columns: the Gram recurrence b_i = c_i - sum h_j G_ji must not lose the fixed point
@pytest.mark.parametrize("mode", [MODES[0], MODES[1], MODES[4], MODES[5], MODES[6]], ids=lambda m: f"{m[0]}{m[1]}")
@pytest.mark.parametrize("rho", [0.9, 0.99])
def test_correlated_design(mode, rho):
    rng = np.random.default_rng(91)
    n, p = 4000, 64
    z = rng.standard_normal((n, 1))
    X = np.asfortranarray(np.sqrt(rho) * z + np.sqrt(1 - rho) * rng.standard_normal((n, p)))
    Y = X[:, :6] @ rng.standard_normal(6) + rng.standard_normal(n)
    o = dict(maxIter=40000, optTol=1e-12, randomize=False)
    xo = O.SparseIterate(p)
    fo = O.CDLeastSquaresLoss(Y, X)
    sto = O.coordinateDescent_(xo, fo, O.ProxL1(0.01), O.CDOptions(**o))
    f = cd.CDLeastSquaresLoss(Y, X)
    _set_mode(f, mode)
    x = cd.SparseIterate(p)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.01), cd.CDOptions(**o))
    assert sto["converged"] and f.last_stats["converged"]
    # slow convergence amplifies rounding: 1e-9 on beta here, objective to 1e-12
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(cd.objective(f), O.objective(fo, O.ProxL1(0.01), xo), rtol=1e-12)


# ---- analytic known answer, independent of the oracle: orthogonal design ------------------------
@pytest.mark.parametrize("mode", MODES, ids=lambda m: f"{m[0]}{m[1]}")
def test_orthogonal_design_closed_form(mode):
    rng = np.random.default_rng(21)
    n, p = 256, 40
    Q, _ = np.linalg.qr(rng.standard_normal((n, p)))
    X = np.asfortranarray(Q * np.sqrt(n))
    Y = X[:, :5] @ np.array([3.0, -2.0, 1.0, 0.5, -0.2]) + 0.3 * rng.standard_normal(n)
    om = rng.random(p) + 0.5
    lam = 0.25
    z = X.T @ Y / n
    want = np.sign(z) * np.maximum(np.abs(z) - lam * om, 0.0)   # beta_j = S(X_j'y/n, lambda w_j)
    f = cd.CDLeastSquaresLoss(Y, X)
    _set_mode(f, mode)
    x = cd.SparseIterate(p)
    cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(optTol=1e-13, randomize=True, seed=2))
    np.testing.assert_allclose(x.dense(), want, rtol=0, atol=1e-12)
    assert f.last_stats["passes"] == 3 and f.last_stats["full_passes"] == 2   # full, active, full


# ---- randomized shapes / losses / penalties / sweep modes against the oracle ---------------------
@pytest.mark.parametrize("seed", range(int(os.environ.get("CDH_FUZZ", "24"))))
def test_random_configurations_match_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.integers(1, 140))
    n = int(rng.integers(max(30, 2 * p), 3000))      # n >= 2p: a unique minimiser, as in the reference's tests
    if seed % 4 == 3:
        n *= int(os.environ.get("CDH_FUZZ_ROWS", "40"))   # every fourth draw has long columns (several chunks per wave)
    s = int(rng.integers(0, min(p, 10) + 1))
    X = np.asfortranarray(rng.standard_normal((n, p)) * rng.uniform(0.2, 3.0, size=p))   # uneven column scales
    Y = X[:, :s] @ rng.standard_normal(s) + rng.uniform(0.1, 2.0) * rng.standard_normal(n)
    loss = ["ls", "sqrt", "wls"][seed % 3]
    mode = MODES[int(rng.integers(0, len(MODES)))]
    weighted = bool(rng.integers(0, 2))
    randomize = bool(rng.integers(0, 2))
    om = rng.uniform(0.5, 2.0, size=p) if weighted else None
    o = dict(maxIter=3000, optTol=1e-12, randomize=randomize, seed=int(rng.integers(1, 1 << 30)))
    if loss == "ls":
        f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
        lam = float(rng.uniform(0.02, 0.6)) * float(np.max(np.abs(X.T @ Y)) / n)
    elif loss == "wls":
        w = rng.uniform(0.2, 2.0, size=n)
        f, fo = cd.CDWeightedLSLoss(Y, X, w), O.CDWeightedLSLoss(Y, X, w)
        lam = float(rng.uniform(0.02, 0.6)) * float(np.max(np.abs(X.T @ (w * Y))) / n)
    else:
        f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
        lam = float(rng.uniform(0.3, 0.9)) * float(np.max(np.abs(X.T @ Y)) / np.linalg.norm(Y))
        lam = min(lam, 0.5 * float(np.sqrt(np.min(np.sum(X * X, axis=0)))))   # lambda^2 < xsqr (:280-282)
    _set_mode(f, mode)
    g, go = (cd.ProxL1(lam, om), O.ProxL1(lam, om)) if weighted else (cd.ProxL1(lam), O.ProxL1(lam))
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    cd.coordinateDescent_(x, f, g, cd.CDOptions(**o))
    st = O.coordinateDescent_(xo, fo, go, O.CDOptions(**o))
    if not st["converged"]:
        pytest.skip("oracle did not converge at this draw")
    scale = max(1.0, float(np.max(np.abs(xo.dense()))))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL * scale,
                               err_msg=f"n={n} p={p} loss={loss} mode={mode} weighted={weighted} randomize={randomize}")
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-8 * max(1.0, float(np.max(np.abs(Y)))))


# ---- long columns: the variants the library picks by shard length (B = 16 transposed loads from 2e6
# rows, 64-vector chunks from 16 rounds per launch) at parity level, not only in the timing tools ------
_LONG = {}


def _long_problem():
    if not _LONG:
        n, p, s = 4_300_003, 24, 4
        f, _ = cd.CDLeastSquaresLoss.generate(n, p, seed=7, s=s, noise=2.0)
        X, Y = f.X_cols(0, p), f.y
        x = cd.SparseIterate(p)
        cd.initialize_(f, x)
        lam = 0.02 * cd.findLambdaMax(x, f, cd.ProxL1(1.0))
        o = dict(maxIter=200, optTol=1e-12, randomize=False)
        xo = O.SparseIterate(p)
        fo = O.CDLeastSquaresLoss(Y, X)
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        _LONG.update(f=f, lam=lam, o=o, want=xo.dense().copy(), obj=O.objective(fo, O.ProxL1(lam), xo), passes=st["passes"])
    return _LONG


@pytest.mark.parametrize("block", [16, 32, 64])
def test_long_column_variants_match_oracle(block):
    L = _long_problem()
    f = L["f"]
    f.set_sweep_mode("block", block)
    x = cd.SparseIterate(24)
    cd.coordinateDescent_(x, f, cd.ProxL1(L["lam"]), cd.CDOptions(**L["o"]))
    np.testing.assert_allclose(x.dense(), L["want"], rtol=0, atol=BETA_TOL)
    np.testing.assert_allclose(cd.objective(f), L["obj"], rtol=1e-12)
    assert f.last_stats["passes"] == L["passes"]


# ---- the reference's own benchmark shape (benchmark/cd_bench.jl:8-38): n = 3000, p = 5000 (p > n),
# s = 100, noise 6; scaledLasso! with lambda = sqrt(2 log p / n) and stdX weights, then plain
# coordinateDescent! at lambda = 0.001 (thousands of non-zeros; compared after a fixed number of passes,
# since neither side converges within the reference's own budget there) --------------------------------
def test_reference_benchmark_shape():
    rng = np.random.default_rng(123)
    n, p, s = 3000, 5000, 100
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ (rng.standard_normal(s) * (1.0 + rng.random(s))) + 6.0 * rng.standard_normal(n)
    f = cd.CDLeastSquaresLoss(Y, X)
    sx = cd.stdX(f)
    np.testing.assert_allclose(sx, np.sqrt((X * X).sum(0) / n), rtol=1e-13)
    lam = float(np.sqrt(2.0 * np.log(p) / n))
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    cdo = dict(randomize=False, optTol=1e-9)
    sol = cd.scaledLasso_(x, f, None, lam, sx, cd.IterLassoOptions(optTol=1e-3, maxIter=50, optionsCD=cd.CDOptions(**cdo)))
    so = O.scaledLasso_(xo, X, Y, lam, sx, O.IterLassoOptions(optTol=1e-3, maxIter=50, optionsCD=O.CDOptions(**cdo)))
    np.testing.assert_allclose(sol.sigma, so.sigma, rtol=1e-9)
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-8)
    assert x.nnz == xo.nnz
    # coordinateDescent! at lambda = 0.001, warm start from zero, 40 passes on both sides
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    o = dict(randomize=False, optTol=1e-7, maxIter=40)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.001), cd.CDOptions(**o))
    fo = O.CDLeastSquaresLoss(Y, X)
    st = O.coordinateDescent_(xo, fo, O.ProxL1(0.001), O.CDOptions(**o))
    assert f.last_stats["passes"] == st["passes"] == 40 and not st["converged"]
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-8)
    np.testing.assert_allclose(cd.objective(f), O.objective(fo, O.ProxL1(0.001), xo), rtol=1e-10)


# ---- gradient cache: full passes served from cached X'r + Gram columns of the coordinates that moved ---------
@pytest.mark.parametrize("loss", ["ls", "wl1", "sqrt"])
@pytest.mark.parametrize("rand", [False, True], ids=["ordered", "random"])
def test_gradient_cache_path_matches_oracle_and_plain_passes(loss, rand):
    """A 14-lambda warm-started path on one handle, three ways: gradient cache from the first full pass
    (mode 3: unconditional), the default (mode 1: rent-or-buy, and only on tall problems), and off (mode 0:
    dots-only screens).  Same beta as the oracle at every lambda (1e-10), same support ORDER and the same number of
    passes in all three: the cache skips a visit only when the exact path would have left the coordinate
    at zero."""
    rng, X, Y = _problem(51, 3000, 640, 14, noise=1.0)
    X *= rng.uniform(0.5, 2.0, size=640)
    om = (rng.random(640) + 0.5) if loss == "wl1" else None
    top = 3.6 if loss == "sqrt" else 0.3
    lams = np.exp(np.linspace(np.log(top), np.log((0.6 if loss == "sqrt" else 0.12) * top), 14))
    # optTol 1e-10: a pass count is only comparable when the last maxH does not sit within rounding of the
    # tolerance (the sqrt-lasso closed form carries ~1e-13 of cancellation noise in h)
    o = dict(maxIter=3000, optTol=1e-10, randomize=rand, seed=13)
    cls, ocls = (cd.CDSqrtLassoLoss, O.CDSqrtLassoLoss) if loss == "sqrt" else (cd.CDLeastSquaresLoss, O.CDLeastSquaresLoss)
    xo, fo, want = O.SparseIterate(640), ocls(Y, X), []
    for lam in lams:
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
        want.append((xo.dense().copy(), xo.nzval2ind.tolist(), st["passes"]))
    assert 10 < xo.nnz < 160
    stats = {}
    for mode in (3, 1, 0):
        f = cls(Y, X)
        f.set_sweep_mode("block", 16)
        f.set_gradient_cache(mode)
        x = cd.SparseIterate(640)
        for lam, (beta, sup, passes) in zip(lams, want):
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
            np.testing.assert_allclose(x.dense(), beta, rtol=0, atol=BETA_TOL)
            assert x.nzval2ind.tolist() == sup and f.last_stats["passes"] == passes, (mode, lam)
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
        stats[mode] = f.cache_stats()
        f.close()
    assert stats[0]["passes"] == 0 and stats[0]["gram_columns"] == 0
    for mode in (3,):
        s = stats[mode]
        assert s["passes"] >= 10 and s["settled_visits"] > 5 * s["exact_visits"] > 0, s
        assert s["gram_columns"] >= xo.nnz and s["reference_passes"] <= 3, s


@pytest.mark.parametrize("rand", [False, True], ids=["ordered", "random"])
def test_gradient_cache_serves_weighted_ls_loss(rand):
    """CDWeightedLSLoss (cd_differentiable_function.jl:165-194) through the gradient cache (round 3): g = X'Wr,
    Gram columns X'WX_j (k_cross with the weight on the B operand), covariance-form visits, whole passes on the
    device.  Same beta as the oracle at every lambda (1e-10), same support ORDER and pass counts as the cache-off run."""
    rng, X, Y = _problem(61, 4000, 520, 12, noise=1.0)
    X *= rng.uniform(0.5, 2.0, size=520)
    w = rng.random(4000) + 0.5
    lams = np.exp(np.linspace(np.log(0.35), np.log(0.04), 12))
    o = dict(maxIter=3000, optTol=1e-10, randomize=rand, seed=17)
    xo, fo, want = O.SparseIterate(520), O.CDWeightedLSLoss(Y, X, w), []
    for lam in lams:
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        want.append((xo.dense().copy(), xo.nzval2ind.tolist(), st["passes"]))
    assert 8 < xo.nnz < 130
    stats = {}
    for mode in (3, 0):
        f = cd.CDWeightedLSLoss(Y, X, w)
        f.set_gradient_cache(mode)                      # default sweep: block, B = 32
        x = cd.SparseIterate(520)
        for lam, (beta, sup, passes) in zip(lams, want):
            cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
            np.testing.assert_allclose(x.dense(), beta, rtol=0, atol=BETA_TOL)
            assert x.nzval2ind.tolist() == sup and f.last_stats["passes"] == passes, (mode, lam)
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
        stats[mode] = f.cache_stats()
        f.close()
    s3 = stats[3]
    assert stats[0]["passes"] == 0 and s3["passes"] >= 10 and s3["covariance_visits"] > 0 and s3["device_passes"] > 0, s3
    assert s3["gram_columns"] >= xo.nnz and s3["settled_visits"] > 5 * s3["exact_visits"] > 0, s3


def test_gradient_cache_serves_fp32_storage():
    """fp32 storage of X, y, r through the gradient cache (round 3).  The certificates allow for the rounding of the
    fp32 residual (Ctrl.cert_abs: 64 * 2^-24 * sqrt((U + 1) y'y / n) * sqrt(a_k), grad_cache.hpp); the cache's g is
    carried in fp64 from Gram entries summed in fp64, so it follows the fp64 problem rather than the rounding of the
    streamed fp32 sweep.  Declared tolerances: beta within 3e-4 of the fp64 oracle (the tolerance of every fp32 test
    here), within 2e-5 of the cache-off fp32 run, the same number of passes at every lambda."""
    rng, X, Y = _problem(62, 6000, 500, 12, noise=1.0)
    X *= rng.uniform(0.5, 2.0, size=500)
    lams = np.exp(np.linspace(np.log(0.3), np.log(0.04), 10))
    o = dict(maxIter=3000, optTol=1e-6, randomize=False)
    lo, bo = O.LassoPath(X, Y, lams, O.CDOptions(**dict(o, optTol=1e-10)), standardizeX=False)
    X32, Y32 = X.astype(np.float32), Y.astype(np.float32)
    out = {}
    for mode in (3, 0):
        f = cd.CDLeastSquaresLoss(Y32, X32)
        f.set_gradient_cache(mode)
        x = cd.SparseIterate(500)
        betas, passes = [], []
        for lam in lams:
            cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
            betas.append(x.dense().copy())
            passes.append(f.last_stats["passes"])
        out[mode] = (betas, passes, f.cache_stats(), f.r)
        f.close()
    for i in range(len(lams)):
        np.testing.assert_allclose(out[3][0][i], bo[i], rtol=0, atol=3e-4)
        np.testing.assert_allclose(out[3][0][i], out[0][0][i], rtol=0, atol=2e-5)
    assert out[3][1] == out[0][1], (out[3][1], out[0][1])
    s3 = out[3][2]
    assert s3["passes"] >= 8 and s3["covariance_visits"] > 0 and s3["device_passes"] > 0 and s3["gram_columns"] >= np.count_nonzero(bo[-1]), s3
    assert out[3][3].dtype == np.float32
    np.testing.assert_allclose(out[3][3], Y - X @ out[3][0][-1], rtol=0, atol=2e-3)    # the caught-up residual


@pytest.mark.parametrize("n", [3000, 200_000])
def test_fp32_gram_columns_carry_the_declared_error(n):
    """fp32 storage: k_cross takes the products on the fp32 matrix pipe (256-row partial sums in fp32, folded into fp64:
    round 4).  The Gram columns the cache holds are compared with fp64 sums of the SAME fp32 data: the error of an entry,
    in units of sqrt(a_i a_j), must stay under what cdh_cache_gram_column declares (2^-24 * 512 / sqrt(n)) -- the term the
    certificates of skipped visits allow for.  Includes a pair of fully correlated columns (the worst case of the bound)."""
    rng = np.random.default_rng(64)
    p = 96
    X = rng.standard_normal((n, p))
    X[:, 5] = X[:, 4] * 1.5                      # fully correlated: every chunk's partial sum has the same sign
    X = np.asfortranarray(X.astype(np.float32))
    Y = (X[:, :6].astype(np.float64) @ rng.standard_normal(6) + rng.standard_normal(n)).astype(np.float32)
    f = cd.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    f.set_onchip_solve(False)
    x = cd.SparseIterate(p)
    cd.coordinateDescent_(x, f, cd.ProxL1(0.05), cd.CDOptions(maxIter=50, optTol=1e-6, randomize=False))
    assert f.cache_stats()["gram_columns"] >= x.nnz > 0
    X64 = X.astype(np.float64)
    a = np.einsum("ij,ij->j", X64, X64)
    worst, checked = 0.0, 0
    for k in x.nzval2ind.tolist():
        col, eps = f.cache_gram_column(k)
        exact = X64.T @ X64[:, k - 1]
        err = np.max(np.abs(col - exact) / np.sqrt(a * a[k - 1]))
        assert err <= eps, (n, k, err, eps)
        worst, checked = max(worst, err), checked + 1
    assert checked > 0 and eps == pytest.approx(2.0 ** -24 * 512 / np.sqrt(n))
    f.close()


def test_gradient_cache_keeps_g_on_the_device_and_reports_its_drift():
    """Whole full passes run on the device (scan -> covariance-form blocks with the certificate re-check in the g
    update): the stats say so, the iterates are the oracle's, and cdh_cache_drift -- which takes the carried gradient
    afresh from X and measures max_k |g_carried - X_k'r| / thr_k -- stays orders of magnitude inside the certificates'
    1e-9 margin after a few thousand covariance-form updates."""
    rng, X, Y = _problem(63, 5000, 700, 15, noise=1.0)
    lams = np.exp(np.linspace(np.log(0.3), np.log(0.03), 25))
    o = dict(maxIter=3000, optTol=1e-10, randomize=False)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    x, xo = cd.SparseIterate(700), O.SparseIterate(700)
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"]
    cs = f.cache_stats()
    assert cs["device_passes"] >= 20 and cs["device_passes"] >= cs["passes"] - 3 and cs["covariance_visits"] > 1000, cs
    assert f.cache_drift()["measured"] == 0
    d = f.cache_drift(rereference_now=True)
    assert d["measured"] == 1 and 0.0 <= d["last"] == d["max"] < 1e-11, d
    assert f.cache_stats()["reference_passes"] == cs["reference_passes"] + 1
    cd.coordinateDescent_(x, f, cd.ProxL1(0.025), cd.CDOptions(**o))                 # and the cache goes on from the fresh g
    O.coordinateDescent_(xo, fo, O.ProxL1(0.025), O.CDOptions(**o))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    f.close()


@pytest.mark.parametrize("loss", ["ls", "wl1", "sqrt", "wls"])
@pytest.mark.parametrize("every", [1, 3])
def test_device_passes_that_are_undone_leave_no_trace(monkeypatch, loss, every):
    """A device-side pass whose re-check fails is undone (g and beta from the scan's snapshot) and walked again in
    windows.  CDH_GC_INJECT_ROLLBACK declares every N-th device pass failed AFTER it ran -- every one of them, or one in
    three -- so the undo is exercised wherever the cache works, not only where a certificate happens to break: same beta as
    the oracle at every lambda, same support order, same pass counts; with N = 1 no pass completes on the device at all."""
    monkeypatch.setenv("CDH_GC_INJECT_ROLLBACK", str(every))
    rng, X, Y = _problem(64, 4000, 560, 12, noise=1.0)
    X *= rng.uniform(0.5, 2.0, size=560)
    om = (rng.random(560) + 0.5) if loss == "wl1" else None
    w = rng.random(4000) + 0.5
    top = 3.4 if loss == "sqrt" else 0.3
    lams = np.exp(np.linspace(np.log(top), np.log((0.6 if loss == "sqrt" else 0.1) * top), 10))
    o = dict(maxIter=3000, optTol=1e-10, randomize=(every == 3), seed=21)
    if loss == "sqrt":
        f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
    elif loss == "wls":
        f, fo = cd.CDWeightedLSLoss(Y, X, w), O.CDWeightedLSLoss(Y, X, w)
    else:
        f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    x, xo = cd.SparseIterate(560), O.SparseIterate(560)
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"], (loss, every, lam)
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
    cs = f.cache_stats()
    assert cs["passes"] >= 8 and cs["rollbacks"] >= (cs["passes"] - 2 if every == 1 else 2), cs
    assert (cs["device_passes"] == 0) if every == 1 else (cs["device_passes"] >= 3), cs
    f.close()


def test_gradient_cache_follows_new_iterates_new_y_and_cold_starts():
    """What invalidates or shifts the cached gradient: a warm start from a DIFFERENT x (the difference is
    folded in as moves), a cold start (x zeroed, 51 continuation solves -- all served from the cache), a
    new y on the same X (gradient re-referenced, Gram columns kept), a new X (everything dropped)."""
    rng, X, Y = _problem(52, 2500, 520, 10)
    Y2 = X[:, 100:108] @ rng.standard_normal(8) + rng.standard_normal(2500)
    o = dict(maxIter=3000, optTol=1e-12, randomize=False)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    x, xo = cd.SparseIterate(520), O.SparseIterate(520)

    def both(lam, **kw):
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**dict(o, **kw)))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**dict(o, **kw)))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"]

    both(0.2)
    both(0.1)
    x0 = np.where(rng.random(520) < 0.02, rng.standard_normal(520), 0.0)     # a different starting point
    x, xo = cd.SparseIterate(520, x0), O.SparseIterate(520, x0)
    both(0.1)
    both(0.07)
    before = f.cache_stats()
    both(0.05, warmStart=False)                                              # cold start: 51 solves
    after = f.cache_stats()
    assert after["passes"] - before["passes"] >= 51 and after["reference_passes"] - before["reference_passes"] <= 1
    cols = after["gram_columns"]
    # a new y on the same matrix (what the Julia binding does between two lasso() calls on one HipMatrix)
    y2 = np.ascontiguousarray(Y2)
    cd._lib.check(f._L.cdh_set_y(f._h, y2.ctypes.data), f._h)
    fo = O.CDLeastSquaresLoss(Y2, X)
    x, xo = cd.SparseIterate(520), O.SparseIterate(520)
    both(0.2)
    both(0.08)
    assert f.cache_stats()["gram_columns"] >= cols                           # the columns survived the new y
    # new columns in X: the cache must forget everything
    Xn = np.asfortranarray(X.copy())
    Xn[:, :40] = rng.standard_normal((2500, 40))
    blk = np.asfortranarray(Xn[:, :40])
    cd._lib.check(f._L.cdh_set_X_cols(f._h, 0, 40, blk.ctypes.data, 2500), f._h)
    fo = O.CDLeastSquaresLoss(Y2, Xn)
    x, xo = cd.SparseIterate(520), O.SparseIterate(520)
    both(0.2)
    both(0.08)
    f.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("CDH_FUZZ_CACHE", "12"))))
def test_gradient_cache_random_configurations_match_oracle(seed):
    """Random shapes / losses / penalties / orders / sweep modes, a short warm-started path each, the gradient
    cache from the first full pass: beta at every lambda against the oracle; duplicates in a caller-made visit
    list go through the same certificates (cdh_pass with screening level 2)."""
    rng = np.random.default_rng(5000 + seed)
    p = int(rng.integers(130, 420))
    n = int(rng.integers(2 * p, 5 * p))
    s = int(rng.integers(1, 12))
    X = np.asfortranarray(rng.standard_normal((n, p)) * rng.uniform(0.3, 3.0, size=p))
    Y = X[:, :s] @ rng.standard_normal(s) + rng.uniform(0.3, 2.0) * rng.standard_normal(n)
    sqrt = bool(seed % 3 == 2)
    mode = MODES[int(rng.integers(0, len(MODES)))]
    om = rng.uniform(0.5, 2.0, size=p) if rng.integers(0, 2) else None
    o = dict(maxIter=5000, optTol=1e-11, randomize=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
    if sqrt:
        f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
        top = 0.9 * float(np.max(np.abs(X.T @ Y) / (om if om is not None else 1.0)) / np.linalg.norm(Y))
        top = min(top, 0.5 * float(np.sqrt(np.min(np.sum(X * X, axis=0)))))
        lams = top * np.array([1.0, 0.9, 0.8, 0.72, 0.65])
    else:
        f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
        top = 0.8 * float(np.max(np.abs(X.T @ Y) / (om if om is not None else 1.0)) / n)
        lams = top * np.array([1.0, 0.6, 0.35, 0.2, 0.12])
    _set_mode(f, mode)
    f.set_gradient_cache(3)
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
        if not st["converged"]:
            pytest.skip("oracle did not converge at this draw")
        if xo.nnz * 4 > p:
            break                                       # dense iterate: full passes are no longer screened
        scale = max(1.0, float(np.max(np.abs(xo.dense()))))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL * scale,
                                   err_msg=f"n={n} p={p} sqrt={sqrt} mode={mode} lam={lam}")
        assert sorted(x.nzval2ind.tolist()) == sorted(xo.nzval2ind.tolist())
    assert f.cache_stats()["passes"] > 0
    # a caller-made visit list with repeats, through cdh_pass with screening level 2
    f.set_screening(2)
    visit = [int(v) for v in rng.integers(1, p + 1, size=2 * p + 5)]
    lam = float(lams[1])
    for _ in range(2):
        mh, mho = cd.cdPass_(x, f, cd.ProxL1(lam, om), visit), O.cdPass_(xo, fo, O.ProxL1(lam, om), visit)
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-9)
        np.testing.assert_allclose(mh, mho, rtol=1e-6, atol=1e-12)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-8 * max(1.0, float(np.max(np.abs(Y)))))


def test_gradient_cache_default_mode_engages_on_tall_problems_only(monkeypatch):
    """The default (mode 1): rent-or-buy, and on LONG columns only while the problem has enough rows per non-zero -- n >= 32 nnz
    now that the fold and the certificate re-check run on the device (round 3; n >= 400 nnz while they were p host flops per
    mover).  A tall path engages it and matches the oracle; once the support has outgrown n / (rows per non-zero) the cache
    stands aside.  SHORT columns (under 1 MB: a streamed visit is launch-bound there) keep it whatever n / nnz is."""
    rng, X, Y = _problem(61, 40_000, 256, 6, noise=1.0)
    lams = 0.5 * np.exp(np.linspace(0.0, np.log(0.05), 8))
    o = dict(maxIter=2000, optTol=1e-10, randomize=False)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(1)                      # the default, whatever CDH_GRADIENT_CACHE says
    f.set_onchip_solve(False)                    # (256 columns: left on, the Gram form would take the path over after a few solves)
    x, xo = cd.SparseIterate(256), O.SparseIterate(256)
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"]
    cs = f.cache_stats()
    assert xo.nnz * 32 < 40_000 and cs["passes"] >= 5 and cs["reference_passes"] == 1 and cs["gram_columns"] >= xo.nnz, cs
    f.close()

    def path(n, p, seed, lams_, s=10):
        rng, X, Y = _problem(seed, n, p, s)
        f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
        f.set_gradient_cache(1)
        f.set_onchip_solve(False)
        x, xo = cd.SparseIterate(p), O.SparseIterate(p)
        served = []
        for lam in lams_:
            before = f.cache_stats()["passes"]
            cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
            st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
            np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
            assert f.last_stats["passes"] == st["passes"]
            served.append((x.nnz, f.cache_stats()["passes"] - before))
        f.close()
        return served
    # long columns (1.1 MB), the rule tightened to 16 000 rows per non-zero so that a small support outgrows it
    monkeypatch.setenv("CDH_GC_ROWS_PER_NNZ", "16000")
    served = path(140_000, 48, 62, (0.5, 0.3, 0.2, 0.1, 0.05, 0.03))
    assert served[-1][0] * 16000 > 140_000 and served[-1][1] == 0, served     # the support outgrew the rule: no pass from the cache
    assert any(sv > 0 for nz, sv in served if nz * 16000 <= 140_000), served  # ... which served the head of the path
    # short columns: the same rule would have switched the cache off at nnz > 300 / 32; it stays on
    monkeypatch.delenv("CDH_GC_ROWS_PER_NNZ")
    served = path(300, 900, 62, (0.5, 0.3, 0.2, 0.15, 0.1, 0.08))
    assert served[-1][0] * 32 > 300 and served[-1][1] > 0, served


@pytest.mark.parametrize("loss", ["ls", "sqrt"])
def test_covariance_chunks_roll_back_when_a_skipped_certificate_breaks(loss):
    """Inside one covariance-form chunk the settled positions between two visits are skipped on the strength of
    certificates read BEFORE the chunk's moves.  On a design whose neighbouring columns are 0.97-correlated, a
    move flips its neighbour's certificate within the same chunk: the library must notice (it re-checks every
    skipped position against the gradient as it stood at its turn), undo the pass and run it again with the
    coordinates whose certificates broke on the visit list (a few such rounds; then the windowed walk).
    Pass by pass against the oracle, from iterates away from the optimum; at least one pass must have been undone,
    or this test exercises nothing."""
    if os.environ.get("CDH_GC_COV") == "0":
        pytest.skip("covariance-form visits are switched off by the environment")
    rollbacks = 0
    for seed in range(6):
        rng = np.random.default_rng(900 + seed)
        n, p = 1500, 260
        z = rng.standard_normal((n, p))
        X = z.copy()
        for j in range(1, p):                      # a chain: column j is 0.97-correlated with column j - 1
            X[:, j] = 0.97 * X[:, j - 1] + np.sqrt(1 - 0.97 ** 2) * z[:, j]
        X = np.asfortranarray(X)
        Y = X[:, [20, 90, 91, 180]] @ np.array([2.0, -1.5, 1.0, 0.8]) + 0.5 * rng.standard_normal(n)
        lam = 0.25 if loss == "ls" else 7.0         # sqrt-lasso: the threshold is lambda ||r||, with ||r|| ~ sqrt(n) / 2 here
        if loss == "ls":
            f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
        else:
            f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
        f.set_sweep_mode("block", 16)
        f.set_gradient_cache(3)
        f.set_screening(2)
        g, go = cd.ProxL1(lam), O.ProxL1(lam)
        x0 = np.zeros(p)
        x0[[20, 90, 180]] = [3.5, -3.0, 2.5]       # far from the optimum: the first passes move a lot
        x, xo = cd.SparseIterate(p, x0), O.SparseIterate(p, x0)
        cd.initialize_(f, x)
        O.initialize_(fo, xo)
        visit = list(range(1, p + 1))
        for _ in range(12):
            mh, mho = cd.cdPass_(x, f, g, visit), O.cdPass_(xo, fo, go, visit)
            np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=1e-10)
            np.testing.assert_allclose(mh, mho, rtol=1e-8, atol=1e-13)
            assert x.nzval2ind.tolist() == xo.nzval2ind.tolist()
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
        cs = f.cache_stats()
        assert cs["covariance_visits"] > 0
        rollbacks += cs["rollbacks"] + f.device_loop_stats()["forced_rounds"]["host_pass"]
        f.close()
    assert rollbacks > 0


def test_handles_with_every_optional_buffer_are_created_and_destroyed_repeatedly():
    """Graph replay + gradient cache (Gram columns, device mirrors, pinned staging) + observation weights +
    a host exchange on the same process, 40 handles in a row: destruction releases everything (a leak or a
    double free shows up as a HIP error or a crash long before 40), and the 40th solve equals the first."""
    rng, X, Y = _problem(81, 2000, 300, 8)
    w = rng.uniform(0.5, 2.0, size=2000)
    o = cd.CDOptions(maxIter=2000, optTol=1e-11, randomize=False)
    first = None
    for it in range(40):
        f = cd.CDLeastSquaresLoss(Y, X)
        f.set_use_graph(True)
        f.set_gradient_cache(3)
        if it % 4 == 3:
            f.set_host_exchange(lambda buf: None, 0, 1)          # a one-rank "transport": the seam with nothing behind it
        x = cd.SparseIterate(300)
        for lam in (0.3, 0.15, 0.08):
            cd.coordinateDescent_(x, f, cd.ProxL1(lam), o)
        assert f.cache_stats()["gram_columns"] > 0
        first = x.dense().copy() if first is None else first
        np.testing.assert_allclose(x.dense(), first, rtol=0, atol=1e-12)
        f.close()
        fw = cd.CDWeightedLSLoss(Y, X, w)
        xw = cd.SparseIterate(300)
        cd.coordinateDescent_(xw, fw, cd.ProxL1(0.1), o)
        fw.close()
    xo = O.SparseIterate(300)
    fo = O.CDLeastSquaresLoss(Y, X)
    for lam in (0.3, 0.15, 0.08):
        O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(maxIter=2000, optTol=1e-11, randomize=False))
    np.testing.assert_allclose(first, xo.dense(), rtol=0, atol=BETA_TOL)


@pytest.mark.parametrize("loss", ["ls", "sqrt"])
def test_gradient_cache_rereferences_from_x_without_losing_track(monkeypatch, loss):
    """After enough covariance-form visits the cached gradient is re-read from X (rounding only ever
    accumulates in it).  With the interval cut to 40 visits a 10-lambda path re-references many times: same
    beta, support order and pass counts as the oracle throughout, and warm starts from the SAME x afterwards
    (the Julia binding's call pattern: set_iterate + rebuild) must not cost another reference pass each."""
    monkeypatch.setenv("CDH_GC_REFRESH", "40")
    rng, X, Y = _problem(71, 3000, 400, 10)
    top = 3.4 if loss == "sqrt" else 0.3
    lams = np.exp(np.linspace(np.log(top), np.log((0.65 if loss == "sqrt" else 0.15) * top), 10))
    o = dict(maxIter=3000, optTol=1e-10, randomize=False)
    cls, ocls = (cd.CDSqrtLassoLoss, O.CDSqrtLassoLoss) if loss == "sqrt" else (cd.CDLeastSquaresLoss, O.CDLeastSquaresLoss)
    f, fo = cls(Y, X), ocls(Y, X)
    f.set_gradient_cache(3)
    x, xo = cd.SparseIterate(400), O.SparseIterate(400)
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))      # x is pushed back each time: set_iterate + rebuild
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"]
    cs = f.cache_stats()
    assert cs["covariance_visits"] > 400 and 3 <= cs["reference_passes"] <= 2 + cs["covariance_visits"] // 40, cs
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
    f.close()
