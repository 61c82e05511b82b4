"""Pure numpy/Python twin of oracle/cd_oracle.c -- TEST INFRASTRUCTURE ONLY.

A second, independently written restatement of the same reference functions
(src/coordinate_descent.jl:65-110, src/cd_differentiable_function.jl:83-111,
242-291, 323-348) used to cross-check the C oracle on small cases.  Ordered
iterator only; dense beta plus an explicit insertion-ordered support list
stand in for ProximalBase's SparseIterate.
"""
from __future__ import annotations

import numpy as np


class Iterate:
    def __init__(self, p):
        self.val = np.zeros(p)
        self.support = []  # 0-based, insertion order
        self.stored = np.zeros(p, dtype=bool)

    def set(self, k, v):
        if self.stored[k]:
            self.val[k] = v
        elif v != 0.0:
            self.val[k] = v
            self.stored[k] = True
            self.support.append(k)

    def dropzeros(self):
        i = 0
        while i < len(self.support):
            k = self.support[i]
            if self.val[k] == 0.0:
                self.stored[k] = False
                last = self.support.pop()
                if i < len(self.support):
                    self.support[i] = last
            else:
                i += 1


def _soft(v, t):
    return v - t if v > t else (v + t if v < -t else 0.0)


def descend_ls(X, r, beta, k, lam0, omega):
    n = X.shape[0]
    xk = X[:, k]
    a = float(xk @ xk)
    b = float(r @ xk)
    old = beta.val[k]
    beta.set(k, old + b / a)
    new = _soft(beta.val[k] if beta.stored[k] else 0.0,
                (n / a) * lam0 * (1.0 if omega is None else omega[k]))
    beta.set(k, new)
    h = new - old
    r -= xk * h
    return h


def descend_sqrt(X, r, beta, k, lam0, omega):
    xk = X[:, k]
    old = beta.val[k] if beta.stored[k] else 0.0
    r += xk * old
    xsqr = float(xk @ xk)
    s = float(r @ xk)
    rsqr = float(r @ r)
    lam = lam0 * (1.0 if omega is None else omega[k])
    if abs(s) <= lam * np.sqrt(rsqr):
        new = 0.0
    else:
        c = lam / np.sqrt(1 - lam ** 2 / xsqr) * np.sqrt(rsqr - s ** 2 / xsqr)
        new = (s - c) / xsqr if s > 0 else (s + c) / xsqr
    beta.set(k, new)
    r -= xk * new
    return new - old


def descend_quad(A, b, Ax, beta, k, lam0, omega):
    a = 1.0 / A[k, k]
    g = Ax[k] + b[k]
    old = beta.val[k] if beta.stored[k] else 0.0
    beta.set(k, old - g * a)
    new = _soft(beta.val[k] if beta.stored[k] else 0.0,
                a * lam0 * (1.0 if omega is None else omega[k]))
    beta.set(k, new)
    h = new - old
    Ax += A[:, k] * h
    return h


def solve(kind, X, y, lam0, omega=None, beta0=None, maxIter=2000, optTol=1e-7):
    """Warm-start ordered solve; returns (dense beta, residual-or-Ax, passes, maxH per pass)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    p = X.shape[1]
    beta = Iterate(p)
    if beta0 is not None:
        for k, v in enumerate(beta0):
            beta.set(k, float(v))
    if kind == "quad":
        state = X @ beta.val
    else:
        state = y - X[:, beta.support] @ beta.val[beta.support] if beta.support else y.copy()
    step = {"ls": descend_ls, "sqrt": descend_sqrt}.get(kind)
    prev, conv = False, True
    trace = []
    for it in range(maxIter):
        order = list(range(p)) if conv else list(beta.support)
        maxH = 0.0
        for k in order:
            if kind == "quad":
                h = descend_quad(X, y, state, beta, k, lam0, omega)
            else:
                h = step(X, state, beta, k, lam0, omega)
            maxH = max(maxH, abs(h))
        beta.dropzeros()
        trace.append(maxH)
        prev, conv = conv, maxH < optTol
        if prev and conv:
            break
    return beta.val * beta.stored, state, len(trace), trace
