"""The device-resident pass loop of cache-served solves (csrc/cov_solve.hpp): `_coordinateDescent!`
(coordinate_descent.jl:65-92) with ONE host round trip per solve while the gradient cache serves the passes.

Bar: beta within 1e-10 of the oracle at every lambda, and the SAME discrete outcomes -- pass counts, visit counts,
support ORDER (ProximalBase's SparseIterate replayed on the device: the order of the support is the visit order of the
next active pass) -- as the oracle's per-coordinate sweep, ordered and shuffled, for every loss; the loop on and off
agree bit for bit in what they report.  The bound that lets a full pass skip the p x moves Gram update
(|g_k(t)| <= |g_k| + M_k TV(t)) is exercised where it is loose: strongly correlated columns.
"""
import numpy as np
import pytest

import coordinatedescent_jl_amd as cd
import oracle as O

pytestmark = pytest.mark.gpu
BETA_TOL = 1e-10


def _problem(seed, n, p, s, noise=1.0, rho=0.0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    if rho:
        for j in range(1, p):                       # an AR(1) chain of columns: corr(X_j, X_{j+1}) = rho
            X[:, j] = rho * X[:, j - 1] + np.sqrt(1 - rho * rho) * X[:, j]
    X = np.asfortranarray(X)
    Y = X[:, :s] @ rng.standard_normal(s) + noise * rng.standard_normal(n)
    return rng, X, Y


def _losses(kind, Y, X, w):
    if kind == "sqrt":
        return cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
    if kind == "wls":
        return cd.CDWeightedLSLoss(Y, X, w), O.CDWeightedLSLoss(Y, X, w)
    return cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)


@pytest.mark.parametrize("kind", ["ls", "wl1", "sqrt", "wls"])
@pytest.mark.parametrize("rand", [False, True], ids=["ordered", "random"])
def test_device_loop_path_matches_oracle(kind, rand):
    """A 16-lambda warm-started path (lasso.jl:250-252): every lambda against the oracle, loop on; then the same
    path with the loop off must report the same iterates, orders and counts."""
    rng, X, Y = _problem(71, 2500, 600, 12)
    X *= rng.uniform(0.5, 2.0, size=600)
    w = rng.random(2500) + 0.5
    om = (rng.random(600) + 0.5) if kind == "wl1" else None
    top = 3.4 if kind == "sqrt" else 0.3
    lams = np.exp(np.linspace(np.log(top), np.log((0.6 if kind == "sqrt" else 0.1) * top), 16))
    o = dict(maxIter=3000, optTol=1e-10, randomize=rand, seed=29)
    f, fo = _losses(kind, Y, X, w)
    xo, want = O.SparseIterate(600), []
    for lam in lams:
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
        want.append((xo.dense().copy(), xo.nzval2ind.tolist(), st["passes"], st["visits"]))
    assert 8 < xo.nnz < 200
    got = {}
    for loop in (True, False):
        f, _ = _losses(kind, Y, X, w)
        f.set_gradient_cache(3)
        f.set_onchip_solve(False)
        f.set_device_loop(loop)
        x = cd.SparseIterate(600)
        rec = []
        for lam, (beta, sup, passes, visits) in zip(lams, want):
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
            np.testing.assert_allclose(x.dense(), beta, rtol=0, atol=BETA_TOL)
            assert x.nzval2ind.tolist() == sup, (loop, lam)
            assert (f.last_stats["passes"], f.last_stats["visits"]) == (passes, visits), (loop, lam)
            assert f.last_stats["converged"]
            rec.append(x.dense().copy())
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)          # the residual catches up with the moves when read
        ls, cs = f.device_loop_stats(), f.cache_stats()
        if loop:
            assert ls["launches"] >= len(lams) and ls["passes"] > 2 * len(lams), ls
            assert cs["device_passes"] >= len(lams) and cs["settled_visits"] > 5 * cs["exact_visits"], cs
            # about one launch per solve plus one per batch of entering coordinates: not one per pass
            assert ls["launches"] <= len(lams) + 2 * cs["gram_batches"] + 4, (ls, cs)
        else:
            assert ls["launches"] == 0
        got[loop] = rec
        f.close()
    for a, b_ in zip(got[True], got[False]):
        np.testing.assert_allclose(a, b_, rtol=0, atol=1e-12)


@pytest.mark.parametrize("rho", [0.9, 0.99])
def test_device_loop_on_correlated_columns_where_the_bound_is_loose(rho):
    """corr(X_j, X_{j+1}) = rho: M_k is of the order of a_k, the bound fails for whole neighbourhoods of the support,
    which then get their exact gradients or a fold -- and the answer is still the oracle's, counts and order included."""
    rng, X, Y = _problem(73, 1500, 400, 10, rho=rho)
    lams = np.exp(np.linspace(np.log(0.5), np.log(0.02), 12))
    o = dict(maxIter=20000, optTol=1e-10, randomize=False)
    fo, xo = O.CDLeastSquaresLoss(Y, X), O.SparseIterate(400)
    f = cd.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    f.set_onchip_solve(False)
    x = cd.SparseIterate(400)
    for lam in lams:
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"], lam
    ls = f.device_loop_stats()
    assert ls["launches"] > 0 and ls["passes"] > 0, ls
    f.close()


def test_device_loop_cold_start_and_maxiter():
    """The 51 continuation solves of a cold start (coordinate_descent.jl:24-37) each run through the loop; a solve cut
    off by maxIter stops at the same pass with the same iterate."""
    rng, X, Y = _problem(79, 2000, 300, 8)
    fo, f = O.CDLeastSquaresLoss(Y, X), cd.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    f.set_onchip_solve(False)
    for opts in (dict(maxIter=2000, optTol=1e-10, randomize=False, warmStart=False),
                 dict(maxIter=2000, optTol=1e-10, randomize=True, seed=3, warmStart=False),
                 dict(maxIter=3, optTol=1e-12, randomize=False, warmStart=True)):
        x, xo = cd.SparseIterate(300), O.SparseIterate(300)
        cd.coordinateDescent_(x, f, cd.ProxL1(0.03), cd.CDOptions(**opts))
        st = O.coordinateDescent_(xo, fo, O.ProxL1(0.03), O.CDOptions(**opts))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert f.last_stats["passes"] == st["passes"] and f.last_stats["visits"] == st["visits"], opts
        assert bool(f.last_stats["converged"]) == bool(st["converged"]), opts
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist(), opts
    assert f.device_loop_stats()["launches"] >= 51
    f.close()


def test_device_loop_undone_passes_leave_no_trace(monkeypatch):
    """CDH_GC_INJECT_ROLLBACK=2: every second full pass of the loop is declared failed after it ran; the kernel puts
    beta back, the host walks that pass the careful way, and the loop takes over again."""
    monkeypatch.setenv("CDH_GC_INJECT_ROLLBACK", "2")
    rng, X, Y = _problem(83, 2200, 500, 10)
    lams = np.exp(np.linspace(np.log(0.3), np.log(0.03), 10))
    for rand in (False, True):
        o = dict(maxIter=3000, optTol=1e-10, randomize=rand, seed=5)
        fo, xo = O.CDLeastSquaresLoss(Y, X), O.SparseIterate(500)
        f = cd.CDLeastSquaresLoss(Y, X)
        f.set_gradient_cache(3)
        f.set_onchip_solve(False)
        x = cd.SparseIterate(500)
        for lam in lams:
            st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
            cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
            np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
            assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"], (rand, lam)
        assert f.cache_stats()["rollbacks"] > 0 and f.device_loop_stats()["passes"] > 0
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-9)
        f.close()


@pytest.mark.parametrize("loop", [True, False], ids=["device_loop", "host_loop"])
def test_sqrt_lasso_near_noiseless_keeps_its_residual_norm(loop):
    """||r|| << ||y||: the covariance-form visits carry r'r by a recurrence that is only good to ~1e-16 of the value it
    started from; once it has fallen to 1e-4 of the last value summed from r itself it is summed afresh (gc_q_guard, and
    kCsNeedQ inside the device loop).  A warm-started sqrt-lasso path on near-noiseless data against the oracle."""
    rng, X, Y = _problem(91, 3000, 300, 8, noise=1e-5)
    lams = [6.0, 4.0, 3.0, 2.5, 2.0]
    o = dict(maxIter=20000, optTol=1e-12, randomize=False)
    f, fo = cd.CDSqrtLassoLoss(Y, X), O.CDSqrtLassoLoss(Y, X)
    f.set_gradient_cache(3)
    f.set_onchip_solve(False)
    f.set_device_loop(loop)
    x, xo = cd.SparseIterate(300), O.SparseIterate(300)
    for lam in lams:
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert sorted(x.nzval2ind.tolist()) == sorted(xo.nzval2ind.tolist())
    assert f.cache_stats()["covariance_visits"] > 0
    f.close()


@pytest.mark.parametrize("kind,rand", [("ls", False), ("ls", True), ("sqrt", False), ("wl1", True), ("wls", False)],
                         ids=["ls-ordered", "ls-random", "sqrt-ordered", "wl1-random", "wls-ordered"])
def test_device_loop_table_mode_on_large_supports(kind, rand):
    """Visit lists beyond the LDS-sized Gram block (~150 non-zeros; benchmark/cd_bench.jl's path ends at 774).  Active passes:
    the loop reads X_j'X_k from its Gram table in device memory and carries the exact gradient of every coordinate the table
    holds.  Full passes: with helper workgroups in the launch (the default) they keep g current for all p and re-check the
    skipped coordinates while workgroup 0 visits; without, the host's device pass runs them.
    A warm-started path whose support grows from ~180 to ~370 non-zeros (and shrinks again at the end: the last lambda is
    larger), every lambda against the oracle with the same passes, visits and support order -- with helpers, with one
    workgroup, with the loop off."""
    rng, X, Y = _problem(97, 1600, 1400, 300, noise=2.0)
    om = (rng.random(1400) + 0.5) if kind == "wl1" else None
    w = rng.random(1600) + 0.5                                   # (observation weights of the weighted-LS variant)
    if kind == "sqrt":
        lams = list(np.exp(np.linspace(np.log(3.2), np.log(1.6), 14))) + [2.4]
    else:
        lams = list(np.exp(np.linspace(np.log(0.5), np.log(0.1), 14))) + [0.3]
    o = dict(maxIter=5000, optTol=1e-9, randomize=rand, seed=41)
    fo = _losses(kind, Y, X, w)[1]
    xo, want = O.SparseIterate(1400), []
    for lam in lams:
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
        assert st["converged"]
        want.append((xo.dense().copy(), xo.nzval2ind.tolist(), st["passes"], st["visits"]))
    sizes = [len(w[1]) for w in want]
    assert max(sizes) > 300 and min(sizes) < 250, sizes
    got = {}
    for mode in ("helpers", "one workgroup", "host loop"):
        f = _losses(kind, Y, X, w)[0]
        f.set_gradient_cache(3)
        f.set_onchip_solve(False)
        if mode == "helpers":
            f.set_device_loop(True)                   # the default: launches that expect large lists bring 31 helper workgroups
        elif mode == "one workgroup":
            f.set_device_loop(True, helpers=0)        # table mode only; full passes of large supports go to the host's device pass
        else:
            f.set_device_loop(False)
        x, rec = cd.SparseIterate(1400), []
        for lam, (beta, sup, passes, visits) in zip(lams, want):
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
            np.testing.assert_allclose(x.dense(), beta, rtol=0, atol=BETA_TOL)
            assert x.nzval2ind.tolist() == sup, (mode, lam)
            assert (f.last_stats["passes"], f.last_stats["visits"]) == (passes, visits), (mode, lam)
            rec.append(x.dense().copy())
        np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-8)
        ls = f.device_loop_stats()
        if mode == "helpers":
            assert ls["table"]["passes"] > 30 and ls["crew"]["passes"] > 5 and ls["crew"]["jobs"] > 3 * ls["crew"]["passes"], ls
        elif mode == "one workgroup":
            assert ls["table"]["passes"] > 30 and ls["table"]["rows_filled"] > 200 and ls["crew"]["passes"] == 0, ls
            assert ls["table"]["coordinates"] <= ls["table"]["capacity"]
        else:
            assert ls["launches"] == 0 and ls["table"]["passes"] == 0 and ls["crew"]["passes"] == 0
        got[mode] = rec
        f.close()
    for other in ("one workgroup", "host loop"):
        for a, b_ in zip(got["helpers"], got[other]):
            np.testing.assert_allclose(a, b_, rtol=0, atol=1e-11)


def test_device_loop_table_survives_a_new_X_and_an_overflow(monkeypatch):
    """The table holds X_j'X_k: set_X_cols voids it (the loop fills it afresh); a table that runs full starts over with
    the current visit list.  Same answers as the oracle throughout."""
    rng, X, Y = _problem(101, 1500, 2600, 200, noise=1.5)
    o = dict(maxIter=5000, optTol=1e-9, randomize=False)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    f.set_onchip_solve(False)
    x, xo = cd.SparseIterate(2600), O.SparseIterate(2600)
    for lam in np.exp(np.linspace(np.log(0.4), np.log(0.15), 10)):
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
    t0 = f.device_loop_stats()["table"]
    assert t0["passes"] > 0 and t0["coordinates"] >= x.nnz > 160, (t0, x.nnz)
    # new columns under the same handle: the table must not serve stale products
    X2 = np.asfortranarray(X.copy())
    X2[:, :300] = rng.standard_normal((1500, 300))
    blk = np.asfortranarray(X2[:, :300])
    cd._lib.check(f._L.cdh_set_X_cols(f._h, 0, 300, blk.ctypes.data, 1500), f._h)
    fo2 = O.CDLeastSquaresLoss(Y, X2)
    x, xo = cd.SparseIterate(2600), O.SparseIterate(2600)
    for lam in np.exp(np.linspace(np.log(0.4), np.log(0.2), 8)):
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        st = O.coordinateDescent_(xo, fo2, O.ProxL1(lam), O.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"], lam
    assert f.device_loop_stats()["table"]["passes"] > t0["passes"]
    f.close()


@pytest.mark.parametrize("inject", [0, 2], ids=["as it comes", "every second full pass undone"])
def test_device_loop_helpers_undo_a_full_pass_and_restore_the_gradient(inject, monkeypatch):
    """A full pass run with the helpers moves g itself along; when it has to be undone -- coordinates crossed their threshold
    through the pass's own moves (correlated columns: corr(X_j, X_j+1) = 0.9), or the test says so (CDH_GC_INJECT_ROLLBACK) --
    the helpers put g back as the pass found it (the snapshot job at its start), and the pass runs again with the crossing
    coordinates visited, or is left to the host's careful walk.  Supports of 120-320 non-zeros; the oracle's iterates, support
    order and pass counts at every lambda."""
    if inject:
        monkeypatch.setenv("CDH_GC_INJECT_ROLLBACK", str(inject))
    rng, X, Y = _problem(103, 1500, 1300, 260, noise=1.5, rho=0.9)
    lams = np.exp(np.linspace(np.log(0.25), np.log(0.04), 10))
    o = dict(maxIter=20000, optTol=1e-9, randomize=False)
    f, fo = cd.CDLeastSquaresLoss(Y, X), O.CDLeastSquaresLoss(Y, X)
    f.set_gradient_cache(3)
    f.set_onchip_solve(False)
    x, xo = cd.SparseIterate(1300), O.SparseIterate(1300)
    sizes = []
    for lam in lams:
        st = O.coordinateDescent_(xo, fo, O.ProxL1(lam), O.CDOptions(**o))
        cd.coordinateDescent_(x, f, cd.ProxL1(lam), cd.CDOptions(**o))
        np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL)
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"], lam
        sizes.append(x.nnz)
    assert max(sizes) > 200, sizes
    ls, cs = f.device_loop_stats(), f.cache_stats()
    assert ls["crew"]["passes"] > 0 and ls["crew"]["jobs"] > 0, ls
    if inject:
        assert cs["rollbacks"] > 0, cs
    np.testing.assert_allclose(f.r, fo.r, rtol=0, atol=1e-8)
    f.close()


def test_device_loop_helpers_leave_no_trace_of_their_number():
    """The helpers keep g current coordinate by coordinate, each for a fixed slice, the jobs in order: what a solve returns
    must not depend on how many there are, bit for bit -- nor, beyond rounding, on whether there are any.  The reference's
    own benchmark shape (benchmark/cd_bench.jl:10-14), 24 lambdas down to ~290 non-zeros."""
    rng = np.random.default_rng(123)
    n, p, s = 3000, 5000, 100
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ (rng.standard_normal(s) * (1.0 + rng.random(s))) + 6.0 * rng.standard_normal(n)
    lmax = float(np.max(np.abs(X.T @ Y) / np.sqrt((X * X).mean(axis=0)))) / n
    lams = lmax * np.exp(np.linspace(np.log(0.5), np.log(0.05), 24))
    o = dict(maxIter=2000, optTol=1e-7, randomize=False)
    got = {}
    for helpers in (None, 5, 0):
        f = cd.CDLeastSquaresLoss(Y, X)
        f.set_device_loop(True, helpers=helpers)
        path = cd.LassoPath(f, None, lams, cd.CDOptions(**o))
        got[helpers] = np.stack([b_.dense() for b_ in path.betapath])
        ls = f.device_loop_stats()
        assert (ls["crew"]["passes"] > 0) == (helpers != 0) and ls["table"]["passes"] > 0, (helpers, ls)
        assert path.betapath[-1].nnz > 250
        f.close()
    assert np.array_equal(got[None], got[5])
    np.testing.assert_allclose(got[0], got[None], rtol=0, atol=1e-11)
