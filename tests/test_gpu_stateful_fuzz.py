"""Stateful fuzz of ONE handle against the oracle: random sequences of the calls a caller of the reference makes on one loss
object -- warm and cold coordinateDescent! (src/coordinate_descent.jl:7-39), single passes over arbitrary visit lists
(_cdPass!, :94-110), initialize! from another iterate (src/cd_differentiable_function.jl:59-72), gradient / _findLambdaMax,
new penalty weights, a new y, replaced columns, new observation weights -- interleaved with switches of everything the
library decides per handle (sweep mode and width, dots-only screens, gradient-cache mode, the one-launch solve, graph replay,
forced re-references and injected rollbacks of the device-side cache passes).

The single-call parity tests prove each path against the oracle from a fresh handle; what they cannot see is state carried
ACROSS calls (the deferred residual, the cached gradient and its Gram columns, the on-chip Gram matrix, the lazily rebuilt
residual, captured graphs) going stale when the data or the path changes underneath it.  Here every call is checked:
beta within 1e-10 (scaled) of the oracle's, same pass counts and the same support order, maxH of a pass, residual
r = y - X beta on request.

CDH_FUZZ_STATEFUL sets the number of sequences (default 48)."""
import os

import numpy as np
import pytest

import coordinatedescent_jl_amd as cd
import oracle as O

pytestmark = pytest.mark.gpu

BETA_TOL = 1e-10
N_SEQ = int(os.environ.get("CDH_FUZZ_STATEFUL", "48"))
REACHED = {}                 # what the sequences of this run went through, summed (checked by the last test of the module)


class Whole:
    """How a sequence reaches its data: here the whole problem in one process; tests/stateful_fuzz_worker.py substitutes a
    row shard per rank (same calls, same checks: beta, the scalars and the support are replicated on every rank)."""
    sharded = False

    def rows(self, n):
        return slice(0, n)

    def make(self, cls, n, y, X, *extra):
        return cls(y, X, *extra)

    def check_replicas(self, x, log):
        pass


def _switch_paths(rng, f, log):
    """A random change of the per-handle decisions; none of them may change an iterate."""
    what = int(rng.integers(0, 6))
    if what == 0:
        mode = ["coord", "block"][int(rng.integers(0, 2))]
        width = int(rng.choice([2, 4, 8, 16, 32, 64]))
        f.set_sweep_mode(mode, width)
        log.append(f"sweep {mode}{width if mode == 'block' else ''}")
    elif what == 1:
        s = int(rng.integers(0, 3))
        f.set_screening(s)
        log.append(f"screening {s}")
    elif what == 2:
        m = int(rng.choice([0, 1, 2, 3, 3]))
        f.set_gradient_cache(m)
        log.append(f"cache {m}")
    elif what == 3:
        on = bool(rng.integers(0, 2))
        f.set_onchip_solve(on)
        log.append(f"onchip {on}")
    elif what == 4:
        on = bool(rng.integers(0, 2))
        f.set_use_graph(on)
        log.append(f"graph {on}")
    else:
        how = int(rng.integers(0, 4))                  # off one time in four (the pass loop on the host); else with helper
        if how == 0:                                   # workgroups for large visit lists (the default), without, or with few
            f.set_device_loop(False)
        else:
            f.set_device_loop(True, helpers=[None, 0, 5][how - 1])
        log.append(f"device loop {how}")


def _check_iterates(x, xo, log, strict_order=True):
    scale = max(1.0, float(np.max(np.abs(xo.dense()))))
    np.testing.assert_allclose(x.dense(), xo.dense(), rtol=0, atol=BETA_TOL * scale, err_msg=" | ".join(log))
    if strict_order:
        assert x.nzval2ind.tolist() == xo.nzval2ind.tolist(), " | ".join(log)


@pytest.mark.parametrize("seed", range(N_SEQ))
def test_random_call_sequences_on_one_handle(seed, monkeypatch):
    run_sequence(seed, monkeypatch.setenv, Whole())


def run_sequence(seed, setenv, ctx):
    rng = np.random.default_rng(91000 + seed)
    kind = ["ls", "ls", "sqrt", "wls"][seed % 4]
    tall = bool(rng.integers(0, 2))                    # tall: the gradient cache's guard (n >= 32 nnz) lets its passes run
    p = int(rng.integers(8, 260))
    n = int(rng.integers(40 * p, 60 * p)) if tall and p < 120 else int(rng.integers(max(p // 2, 8), 6 * p + 40))
    if kind == "sqrt":
        n = max(n, 2 * p)                              # (the sqrt-lasso needs lambda^2 < ||X_k||^2: comfortably over-determined)
    s = int(rng.integers(1, min(10, p) + 1))
    # env knobs are read when the handle is created
    setenv("CDH_SMALL_PATH", str(int(rng.integers(0, 2))))
    if rng.integers(0, 3) == 0:
        setenv("CDH_GC_REFRESH", str(int(rng.integers(20, 400))))      # re-references of the cached gradient mid-solve
    if rng.integers(0, 3) == 0:
        setenv("CDH_GC_INJECT_ROLLBACK", str(int(rng.integers(1, 4))))  # device-side cache passes undone
    setenv("CDH_SMALL_ALWAYS_BYTES", ["16777216", "16777216", "0"][int(rng.integers(0, 3))])   # 0: the handle rents before it builds G
    # the device loop's LDS block holds ~170 coordinates: far more than these problems' supports.  One sequence in three caps it at
    # 6 or 12, so that their visit lists run from the loop's Gram table and, in full passes, with its helper workgroups
    setenv("CDH_CS_UCAP", ["0", "0", "6", "12"][int(rng.integers(0, 4))])
    X = np.asfortranarray(rng.standard_normal((n, p)) * rng.uniform(0.4, 2.5, size=p))
    Y = X[:, :s] @ rng.standard_normal(s) + rng.uniform(0.3, 2.0) * rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, size=n) if kind == "wls" else None
    log = [f"seed={seed} kind={kind} n={n} p={p} env={ {k: os.environ.get(k) for k in ('CDH_SMALL_PATH', 'CDH_SMALL_ALWAYS_BYTES', 'CDH_GC_REFRESH', 'CDH_GC_INJECT_ROLLBACK', 'CDH_CS_UCAP')} }"]

    def oracle_loss():
        if kind == "sqrt":
            return O.CDSqrtLassoLoss(Y, X)
        if kind == "wls":
            return O.CDWeightedLSLoss(Y, X, w)
        return O.CDLeastSquaresLoss(Y, X)

    rows = ctx.rows(n)
    if kind == "wls":
        f = ctx.make(cd.CDWeightedLSLoss, n, Y[rows], X[rows], w[rows])
    else:
        f = ctx.make({"ls": cd.CDLeastSquaresLoss, "sqrt": cd.CDSqrtLassoLoss}[kind], n, Y[rows], X[rows])
    fo = oracle_loss()
    f.set_gradient_cache(int(rng.choice([0, 1, 2, 3, 3])))
    f.set_sweep_mode(["coord", "block"][int(rng.integers(0, 2))], int(rng.choice([2, 8, 16, 32, 64])))
    x, xo = cd.SparseIterate(p), O.SparseIterate(p)
    om = None

    def lam_top():
        """A lambda scale from the CURRENT data: lambda_max at beta = 0 of the loss in use."""
        wy = Y if w is None else w * Y
        c = np.abs(X.T @ wy) / (om if om is not None else 1.0)
        if kind == "sqrt":
            return min(float(np.max(c)) / float(np.linalg.norm(Y)), 0.5 * float(np.sqrt(np.min(np.sum(X * X, axis=0)))))
        return float(np.max(c)) / n

    for step in range(int(rng.integers(6, 12))):
        op = int(rng.integers(0, 15))
        if op <= 4:                                    # a solve: warm (mostly) or cold, ordered or shuffled
            lam = lam_top() * float(rng.uniform(0.08, 0.95))
            o = dict(maxIter=4000, optTol=1e-12, randomize=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)),
                     warmStart=bool(rng.integers(0, 5) > 0), numSteps=int(rng.integers(2, 12)))
            log.append(f"solve lam={lam:.5g} {o}")
            st = O.coordinateDescent_(xo, fo, O.ProxL1(lam, om), O.CDOptions(**o))
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(**o))
            REACHED["solves"] = REACHED.get("solves", 0) + 1
            ctx.check_replicas(x, log)
            assert f.last_stats["converged"] == st["converged"], " | ".join(log)
            _check_iterates(x, xo, log, strict_order=False)
            assert sorted(x.nzval2ind.tolist()) == sorted(xo.nzval2ind.tolist()), "\n".join(log)
            same_route = (x.nzval2ind.tolist() == xo.nzval2ind.tolist() and f.last_stats["passes"] == st["passes"]
                          and f.last_stats["full_passes"] == st["full_passes"])
            if not same_route:
                # Same minimiser and support set, another route to it.  Two knife edges exist, in the reference as well:
                #  * cold start: the first of the numSteps + 1 solves runs at exp(log(lambda_max)), where the arg-max
                #    coordinate sits exactly on its threshold; whether it moves by an ulp-sized amount -- and so takes the first
                #    slot of the support -- is decided by the last bits of two differently rounded sums (gradient() and
                #    descendCoordinate! associate w * X * r differently, and @simd reassociates both);
                #  * any solve: a pass whose maxH comes out on the other side of optTol = 1e-12 (the iterates agree to
                #    1e-10, not to 1e-12) is followed by the other kind of pass, and under shuffled sweeps by other draws.
                # The slot order steers the active-set passes, so from there the runs differ in pass counts and order.
                # Counted (the second kind must stay rare: bounded in the last test), then carried on from one common order;
                # the strict pass-count and order parity of the state machine is what the golden tests hold.
                kind_of = "cold_start_ties" if not o["warmStart"] else "threshold_crossings"
                REACHED[kind_of] = REACHED.get(kind_of, 0) + 1
                if o["warmStart"] and not o["randomize"]:     # ordered sweeps: at most one active-set pass more or less
                    assert abs(f.last_stats["passes"] - st["passes"]) <= 1 and f.last_stats["full_passes"] == st["full_passes"], \
                        "\n".join(log) + f"\nproduct {f.last_stats}\noracle  {st}\ncache {f.cache_stats()}"
                dense = xo.dense()
                x, xo = cd.SparseIterate(p, dense), O.SparseIterate(p, dense)
                cd.initialize_(f, x)
                O.initialize_(fo, xo)
                log.append("(another route to the same point: support order re-synchronised)")
        elif op == 5:                                  # one pass over an arbitrary visit list (repeats allowed)
            m = int(rng.integers(1, 2 * p))
            visit = rng.integers(1, p + 1, size=m).astype(np.int64)
            lam = lam_top() * float(rng.uniform(0.1, 0.9))
            log.append(f"pass m={m} lam={lam:.5g}")
            mho = O.cdPass_(xo, fo, O.ProxL1(lam, om), visit)
            mh = cd.cdPass_(x, f, cd.ProxL1(lam, om), visit)
            np.testing.assert_allclose(mh, mho, rtol=1e-7, atol=1e-12, err_msg=" | ".join(log))
            _check_iterates(x, xo, log)
        elif op == 6:                                  # initialize! from another iterate
            x0 = np.where(rng.random(p) < 0.15, rng.standard_normal(p), 0.0)
            x, xo = cd.SparseIterate(p, x0), O.SparseIterate(p, x0)
            cd.initialize_(f, x)
            O.initialize_(fo, xo)
            log.append("initialize! from a new iterate")
        elif op == 7:                                  # gradient / _findLambdaMax / the residual
            k = int(rng.integers(1, p + 1))
            log.append(f"gradient k={k}, lambda_max, residual")
            np.testing.assert_allclose(cd.gradient(f, x, k), O.gradient(fo, xo, k), rtol=1e-8, atol=1e-11, err_msg=" | ".join(log))
            np.testing.assert_allclose(f.r, (Y - X @ x.dense())[rows], rtol=0, atol=1e-9 * max(1.0, float(np.max(np.abs(Y)))),
                                       err_msg=" | ".join(log))
        elif op == 8:                                  # new penalty weights (or none)
            om = rng.uniform(0.5, 2.0, size=p) if rng.integers(0, 3) else None
            log.append("new omega" if om is not None else "omega dropped")
        elif op == 9:                                  # a new y on the same X: the same handle, a fresh oracle loss
            Y = X[:, :s] @ rng.standard_normal(s) + rng.uniform(0.3, 2.0) * rng.standard_normal(n)
            yc = np.ascontiguousarray(Y[rows])
            cd._lib.check(f._L.cdh_set_y(f._h, yc.ctypes.data), f._h)
            fo = oracle_loss()
            cd.initialize_(f, x)                       # r = y - X beta for the iterate carried over
            O.initialize_(fo, xo)
            log.append("new y")
        elif op == 10:                                 # replaced columns
            j0 = int(rng.integers(0, p))
            nc = int(rng.integers(1, min(p - j0, 12) + 1))
            X[:, j0:j0 + nc] = rng.standard_normal((n, nc)) * rng.uniform(0.4, 2.5, size=nc)
            blk = np.asfortranarray(X[rows, j0:j0 + nc])
            cd._lib.check(f._L.cdh_set_X_cols(f._h, j0, nc, blk.ctypes.data, blk.shape[0]), f._h)
            fo = oracle_loss()
            cd.initialize_(f, x)
            O.initialize_(fo, xo)
            log.append(f"columns {j0}..{j0 + nc - 1} replaced")
        elif op == 14:                                 # single visits through the operator interface (descendCoordinate!)
            lam = lam_top() * float(rng.uniform(0.1, 0.9))
            ks = rng.integers(1, p + 1, size=int(rng.integers(1, 6)))
            log.append(f"descendCoordinate! k={ks.tolist()} lam={lam:.5g}")
            for k in ks:
                hg = cd.descendCoordinate_(f, cd.ProxL1(lam, om), x, int(k))
                ho = O.descendCoordinate_(fo, O.ProxL1(lam, om), xo, int(k))
                np.testing.assert_allclose(hg, ho, rtol=1e-7, atol=1e-12, err_msg="\n".join(log))
            _check_iterates(x, xo, log)                            # (no dropzeros! in a single visit: zeros keep their slots)
        elif op >= 12 and kind == "ls":                # the front-ends on the SAME loss object (src/lasso.jl:229-260, 107-144)
            o = dict(maxIter=4000, optTol=1e-12, randomize=False, warmStart=True)
            if op == 12:
                lams = lam_top() * np.exp(np.linspace(np.log(0.9), np.log(float(rng.uniform(0.05, 0.4))), int(rng.integers(2, 7))))
                std = bool(rng.integers(0, 2))
                log.append(f"LassoPath {len(lams)} lambdas standardizeX={std}")
                path = cd.LassoPath(f, None, lams, cd.CDOptions(**o), standardizeX=std)
                _, bo = O.LassoPath(X, Y, lams, O.CDOptions(**o), standardizeX=std)
                for got, want in zip(path.betapath, bo):
                    np.testing.assert_allclose(got.dense(), want, rtol=0, atol=BETA_TOL * max(1.0, float(np.max(np.abs(want)))),
                                               err_msg="\n".join(log))
            else:
                lam = float(np.sqrt(2.0 * np.log(p) / n)) * float(rng.uniform(0.6, 1.5))
                io = dict(maxIter=30, optTol=1e-6, initProcedure=["Screening", "InitStd", "WarmStart"][int(rng.integers(0, 3))],
                          sinit=int(min(5, p)), sigmainit=float(np.std(Y)))
                log.append(f"scaledLasso! lam={lam:.5g} {io}")
                x2, xo2 = x.copy(), xo.copy()
                sol = cd.scaledLasso_(x2, f, None, lam, om, cd.IterLassoOptions(optionsCD=cd.CDOptions(**o), **io))
                so = O.scaledLasso_(xo2, X, Y, lam, om, O.IterLassoOptions(optionsCD=O.CDOptions(**o), **io))
                np.testing.assert_allclose(sol.sigma, so.sigma, rtol=1e-9, err_msg="\n".join(log))
                np.testing.assert_allclose(x2.dense(), xo2.dense(), rtol=0, atol=1e-9 * max(1.0, float(np.max(np.abs(xo2.dense())))),
                                           err_msg="\n".join(log))
            cd.initialize_(f, x)                       # the reference's front-ends build a loss of their own: back to ours
            O.initialize_(fo, xo)
        else:                                          # new observation weights (weighted loss), else a path switch
            if kind == "wls":
                w = rng.uniform(0.5, 1.5, size=n)
                wc = np.ascontiguousarray(w[rows])
                cd._lib.check(f._L.cdh_set_obs_weights(f._h, wc.ctypes.data), f._h)
                fo = oracle_loss()
                cd.initialize_(f, x)
                O.initialize_(fo, xo)
                log.append("new observation weights")
        _switch_paths(rng, f, log)
    ls = f.device_loop_stats()
    for k, v in dict(f.cache_stats(), onchip_solves=f.onchip_stats()["solves"], sequences=1, loop_launches=ls["launches"], loop_passes=ls["passes"],
                     loop_table_passes=ls["table"]["passes"], loop_helper_passes=ls["crew"]["passes"], loop_helper_jobs=ls["crew"]["jobs"],
                     loop_passes_run_again=ls["forced_rounds"]["loop"], host_passes_run_again=ls["forced_rounds"]["host_pass"]).items():
        REACHED[k] = REACHED.get(k, 0) + int(v)
    f.close()


@pytest.mark.parametrize("seed", range(max(8, N_SEQ // 3)))
def test_random_call_sequences_on_one_fp32_handle(seed, monkeypatch):
    """The same idea on fp32 storage (BASELINE.json configs[4]'s dtype): the oracle computes in fp64 on the fp32-rounded data,
    so the bar is the declared fp32 one (3e-4 on beta, as tests/test_gpu_parity.py::test_fp32_against_fp64_oracle) and the
    discrete outcomes (pass counts, slot order) are not compared; what a stale Gram column, gradient or residual would do to
    beta is orders of magnitude above that bar."""
    rng = np.random.default_rng(93000 + seed)
    p = int(rng.integers(8, 200))
    n = int(rng.integers(40 * p, 70 * p)) if rng.integers(0, 2) and p < 100 else int(rng.integers(2 * p, 8 * p + 40))
    s = int(rng.integers(1, min(8, p) + 1))
    monkeypatch.setenv("CDH_SMALL_PATH", str(int(rng.integers(0, 2))))
    if rng.integers(0, 3) == 0:
        monkeypatch.setenv("CDH_GC_REFRESH", str(int(rng.integers(20, 400))))
    if rng.integers(0, 3) == 0:
        monkeypatch.setenv("CDH_GC_INJECT_ROLLBACK", str(int(rng.integers(1, 4))))
    monkeypatch.setenv("CDH_CS_UCAP", ["0", "0", "5", "10"][int(rng.integers(0, 4))])     # (the device loop's table and helpers on small supports)
    X = np.asfortranarray((rng.standard_normal((n, p)) * rng.uniform(0.5, 2.0, size=p)).astype(np.float32))
    Y = (X[:, :s].astype(np.float64) @ rng.standard_normal(s) + rng.uniform(0.5, 2.0) * rng.standard_normal(n)).astype(np.float32)
    log = [f"fp32 seed={seed} n={n} p={p} ucap={os.environ.get('CDH_CS_UCAP')}"]
    f = cd.CDLeastSquaresLoss(Y, X)
    assert f.r.dtype == np.float32
    f.set_gradient_cache(int(rng.choice([0, 1, 2, 3, 3])))
    f.set_sweep_mode(["coord", "block"][int(rng.integers(0, 2))], int(rng.choice([2, 8, 16, 32])))
    x, om = cd.SparseIterate(p), None
    for step in range(int(rng.integers(5, 10))):
        op = int(rng.integers(0, 8))
        X64, Y64 = X.astype(np.float64), Y.astype(np.float64)
        if op <= 3:
            top = float(np.max(np.abs(X64.T @ Y64) / (om if om is not None else 1.0))) / n
            lam = top * float(rng.uniform(0.15, 0.9))
            warm = bool(rng.integers(0, 4) > 0)
            log.append(f"solve lam={lam:.5g} warm={warm}")
            cd.coordinateDescent_(x, f, cd.ProxL1(lam, om), cd.CDOptions(maxIter=3000, optTol=1e-7, randomize=False, warmStart=warm,
                                                                          numSteps=int(rng.integers(2, 8))))
            xo = O.SparseIterate(p)
            O.coordinateDescent_(xo, O.CDLeastSquaresLoss(Y64, np.asfortranarray(X64)), O.ProxL1(lam, om),
                                 O.CDOptions(maxIter=5000, optTol=1e-11, randomize=False))
            want = xo.dense()
            np.testing.assert_allclose(x.dense(), want, rtol=0, atol=3e-4 * max(1.0, float(np.max(np.abs(want)))), err_msg="\n".join(log))
        elif op == 4:
            log.append("residual")
            np.testing.assert_allclose(f.r, Y64 - X64 @ x.dense(), rtol=0, atol=2e-4 * max(1.0, float(np.max(np.abs(Y64)))),
                                       err_msg="\n".join(log))
        elif op == 5:
            om = rng.uniform(0.5, 2.0, size=p) if rng.integers(0, 3) else None
            log.append("new omega" if om is not None else "omega dropped")
        elif op == 6:
            Y = (X[:, :s].astype(np.float64) @ rng.standard_normal(s) + rng.uniform(0.5, 2.0) * rng.standard_normal(n)).astype(np.float32)
            yc = np.ascontiguousarray(Y)
            cd._lib.check(f._L.cdh_set_y(f._h, yc.ctypes.data), f._h)
            cd.initialize_(f, x)
            log.append("new y")
        else:
            j0 = int(rng.integers(0, p))
            nc = int(rng.integers(1, min(p - j0, 10) + 1))
            X[:, j0:j0 + nc] = (rng.standard_normal((n, nc)) * rng.uniform(0.5, 2.0, size=nc)).astype(np.float32)
            blk = np.asfortranarray(X[:, j0:j0 + nc])
            cd._lib.check(f._L.cdh_set_X_cols(f._h, j0, nc, blk.ctypes.data, n), f._h)
            cd.initialize_(f, x)
            log.append(f"columns {j0}..{j0 + nc - 1} replaced")
        _switch_paths(rng, f, log)
    ls = f.device_loop_stats()
    for k, v in dict(f.cache_stats(), onchip_solves=f.onchip_stats()["solves"], loop_table_passes=ls["table"]["passes"],
                     loop_helper_passes=ls["crew"]["passes"]).items():
        REACHED["fp32_" + k] = REACHED.get("fp32_" + k, 0) + int(v)
    f.close()


def test_random_call_sequences_on_row_shards():
    """The same sequences with the loss object a ROW SHARD on each of two ranks (one GPU, host-staged exchange behind the
    library's all-reduce seam: tests/stateful_fuzz_worker.py): the sharded launch sequences, the cache's Gram batches and
    device-side passes over shards, and the state they carry, under the same switches."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    count = max(12, N_SEQ // 2)
    a = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "tests", "stateful_fuzz_worker.py"), "0", str(count)],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert a.returncode == 0, (a.stdout[-1500:], a.stderr[-4000:])
    assert f"FUZZ_SHARDS_OK {count}" in a.stdout


def test_the_sequences_reached_the_paths_that_carry_state():
    """The point of the module is state carried across calls: the run must have gone through the one-launch solve, cache
    passes served from carried gradients (on the device and by the windowed walk), covariance-form visits with the residual
    catching up later, Gram batches, re-references and undone passes -- not only through the plain streamed sweep."""
    if os.environ.get("CDH_FUZZ_REPORT"):
        import json
        with open(os.environ["CDH_FUZZ_REPORT"], "w") as fh:
            json.dump(REACHED, fh)
    if REACHED.get("sequences", 0) < 24:
        pytest.skip("fewer than 24 sequences ran in this process")
    for key in ("onchip_solves", "passes", "device_passes", "settled_visits", "covariance_visits", "residual_catchups",
                "gram_batches", "reference_passes", "rollbacks", "loop_launches", "loop_table_passes", "loop_helper_passes"):
        assert REACHED.get(key, 0) > 0, (key, REACHED)
    for key in ("fp32_onchip_solves", "fp32_passes", "fp32_covariance_visits", "fp32_gram_batches"):
        assert REACHED.get(key, 0) > 0, (key, REACHED)
    assert REACHED.get("threshold_crossings", 0) <= max(1, REACHED["solves"] // 50), REACHED
