"""The CPU oracle reproduces the committed golden vectors (tests/golden/*.npz, made by
tests/golden/make_golden.py) -- guards the oracle against drift between rounds."""
import glob
import os

import numpy as np
import pytest

import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(glob.glob(os.path.join(GOLD, "*.npz")))


def test_golden_present():
    assert len(CASES) >= 10


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(c)[:-4] for c in CASES])
def test_oracle_reproduces_golden(path):
    d = np.load(path)  # allow_pickle defaults to False
    name = os.path.basename(path)
    X, Y, lam = np.asfortranarray(d["X"]), d["Y"], float(d["lam"])
    omega = d["omega"] if "omega" in d else None
    kind = O.CDSqrtLassoLoss if name.startswith("sqrt") else O.CDLeastSquaresLoss
    f, g = kind(Y, X), O.ProxL1(lam, omega)
    p = X.shape[1]
    if "betas" in d:
        x = O.SparseIterate(p)
        O.initialize_(f, x)
        for t in range(d["betas"].shape[0]):
            mh = O.cdPass_(x, f, g, np.arange(1, p + 1))
            np.testing.assert_allclose(x.dense(), d["betas"][t], rtol=0, atol=1e-14)
            np.testing.assert_allclose(mh, d["maxhs"][t], rtol=1e-12)
        return
    warm = "warm0" not in name
    tol = 1e-12
    x = O.SparseIterate(p, d["x0"] if "x0" in d else None)
    st = O.coordinateDescent_(x, f, g, O.CDOptions(maxIter=5000, optTol=tol, warmStart=warm, randomize=False))
    np.testing.assert_allclose(x.dense(), d["beta"], rtol=0, atol=1e-13)
    assert x.nzval2ind.tolist() == d["support"].tolist()
    assert st["passes"] == int(d["passes"])
    np.testing.assert_allclose(O.objective(f, g, x), float(d["objective"]), rtol=1e-13)
