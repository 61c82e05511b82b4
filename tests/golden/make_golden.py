"""Generates tests/golden/*.npz: small seeded problems (numpy default_rng) in the shapes
of the reference's tests, with the outputs of the CPU oracle (oracle/cd_oracle.c) for
`randomize=false`.  The reference itself cannot run here (no Julia), so these vectors
are oracle-generated; the oracle is pinned by tests/test_oracle_reference_properties.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def problem(seed, n, p, s, noise):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.standard_normal((n, p)))
    Y = X[:, :s] @ rng.standard_normal(s) + noise * rng.standard_normal(n)
    return rng, X, Y


def solve(kind, X, Y, lam, omega, x0, opt):
    f = O.CDLeastSquaresLoss(Y, X) if kind == "ls" else O.CDSqrtLassoLoss(Y, X)
    g = O.ProxL1(lam, omega)
    x = O.SparseIterate(X.shape[1], x0)
    st = O.coordinateDescent_(x, f, g, opt)
    return dict(beta=x.dense(), support=x.nzval2ind, resid=f.r.copy(), objective=O.objective(f, g, x),
                passes=st["passes"], visits=st["visits"], maxH=st["maxH"])


def trajectory(kind, X, Y, lam, omega, npass):
    """npass full ordered passes from beta = 0: beta and maxH after each."""
    p = X.shape[1]
    f = O.CDLeastSquaresLoss(Y, X) if kind == "ls" else O.CDSqrtLassoLoss(Y, X)
    g = O.ProxL1(lam, omega)
    x = O.SparseIterate(p)
    O.initialize_(f, x)
    betas, maxhs = [], []
    for _ in range(npass):
        maxhs.append(O.cdPass_(x, f, g, np.arange(1, p + 1)))
        betas.append(x.dense())
    return np.array(betas), np.array(maxhs), f.r.copy()


def main():
    cases = {}
    # test/coordinate_descent.jl:29-63 shape, warm + cold, from a non-zero start
    rng, X, Y = problem(101, 500, 50, 5, 1.0)
    x0 = np.where(rng.random(50) < 0.6, rng.random(50), 0.0)
    for warm in (True, False):
        o = O.CDOptions(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=False)
        cases[f"ls_n500_p50_warm{int(warm)}"] = dict(X=X, Y=Y, lam=0.02, x0=x0, **solve("ls", X, Y, 0.02, None, x0, o))
    # test/coordinate_descent.jl:65-99 shape: weighted l1
    rng, X, Y = problem(102, 500, 50, 10, 1.0)
    om = rng.random(50)
    x0 = np.where(rng.random(50) < 0.6, rng.random(50), 0.0)
    for warm in (True, False):
        o = O.CDOptions(maxIter=5000, optTol=1e-12, warmStart=warm, randomize=False)
        cases[f"wl1_n500_p50_warm{int(warm)}"] = dict(X=X, Y=Y, lam=0.01, omega=om, x0=x0,
                                                       **solve("ls", X, Y, 0.01, om, x0, o))
    # test/lasso.jl:76-101 shape
    rng, X, Y = problem(103, 200, 50, 10, 0.1)
    o = O.CDOptions(optTol=1e-12, randomize=False)
    cases["ls_n200_p50"] = dict(X=X, Y=Y, lam=0.2, **solve("ls", X, Y, 0.2, None, None, o))
    bt, mh, rr = trajectory("ls", X, Y, 0.2, None, 6)
    cases["ls_n200_p50_traj"] = dict(X=X, Y=Y, lam=0.2, betas=bt, maxhs=mh, resid=rr)
    # test/lasso.jl:106-125 shape: sqrt-lasso
    rng, X, Y = problem(104, 100, 50, 5, 1.0)
    o = O.CDOptions(maxIter=5000, optTol=1e-12, randomize=False)
    cases["sqrt_n100_p50"] = dict(X=X, Y=Y, lam=2.8, **solve("sqrt", X, Y, 2.8, None, None, o))
    bt, mh, rr = trajectory("sqrt", X, Y, 2.8, None, 6)
    cases["sqrt_n100_p50_traj"] = dict(X=X, Y=Y, lam=2.8, betas=bt, maxhs=mh, resid=rr)
    # config 1 of BASELINE.json: n=1000, p=200, lambda=0.1 (host-generated, plumbing)
    rng, X, Y = problem(123, 1000, 200, 10, 1.0)
    o = O.CDOptions(optTol=1e-12, randomize=False)
    cases["cfg1_n1000_p200"] = dict(X=X, Y=Y, lam=0.1, **solve("ls", X, Y, 0.1, None, None, o))
    # odd n (ragged tail of the 16-byte vector loads) and p not a multiple of the block size
    rng, X, Y = problem(105, 1237, 37, 7, 0.5)
    o = O.CDOptions(optTol=1e-12, randomize=False)
    cases["ls_n1237_p37"] = dict(X=X, Y=Y, lam=0.05, **solve("ls", X, Y, 0.05, None, None, o))
    for name, d in cases.items():
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: np.asarray(v) for k, v in d.items()
                                                                   if v is not None})
        print(name, {k: np.asarray(v).shape for k, v in d.items() if v is not None})


if __name__ == "__main__":
    main()
