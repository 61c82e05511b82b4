/*
 * cd_oracle.c -- CPU restatement of the coordinate-descent hot path of
 * mlakolar/CoordinateDescent.jl v0.3.0 (reference paths are relative to the
 * reference's src/ directory).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, the smoke check
 * in __graft_entry__.py and bench.py's `cpu_baseline` leg may load it.  The
 * product (coordinatedescent.jl_amd/csrc) never links, loads or calls it.
 *
 * Parity pinning: the reference is Julia and no Julia toolchain exists in the
 * image, so the reference itself cannot be run.  This restatement is pinned by
 *   - the reference's one analytic known-answer test
 *       (test/coordinate_descent.jl:13-25, answer [0, 0.3]),
 *   - every RNG-agnostic property the reference's tests assert
 *       (test/lasso.jl, test/coordinate_descent.jl, test/atom_iterator.jl),
 *   - scikit-learn's Lasso(fit_intercept=False) (same objective) to 1e-12.
 * See tests/test_oracle_*.py.  The arithmetic of ProximalBase.jl 0.3.0
 * (Manifest.toml:68-72, not vendored) -- cdprox!, SparseIterate, dropzeros!,
 * A_mul_B_row, At_mul_B_row -- is restated from its published behaviour as
 * used at the call sites cited below.  One detail is PARITY UNPINNED: the
 * order dropzeros! leaves the support in (no reference test observes it);
 * swap-with-last compaction is used here.  It changes the visit order of an
 * active pass only, never the fixed point.
 *
 * Everything is fp64.  Coordinates cross this API 1-based, as in the reference.
 * Reductions are plain left-to-right loops the compiler may vectorise
 * (`omp simd reduction`), mirroring the reference's `@simd` loops whose
 * summation order is likewise unspecified.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CDO_LS 0    /* CDLeastSquaresLoss  cd_differentiable_function.jl:43-111  */
#define CDO_SQRT 1  /* CDSqrtLassoLoss     cd_differentiable_function.jl:202-291 */
#define CDO_WLS 2   /* CDWeightedLSLoss    cd_differentiable_function.jl:118-194 */
#define CDO_QUAD 3  /* CDQuadraticLoss     cd_differentiable_function.jl:299-348 */

/* ------------------------------------------------------------------------ */
/* SparseIterate (ProximalBase 0.3.0; contract per SURVEY.md Appendix B)     */
/* ------------------------------------------------------------------------ */
typedef struct cdo_iterate {
    int64_t p;
    int64_t nnz;
    double *nzval;       /* [p]  stored values, slots 0..nnz-1               */
    int64_t *nzval2ind;  /* [p]  slot -> 0-based coordinate                  */
    int64_t *ind2nzval;  /* [p]  coordinate -> slot+1, 0 = not stored        */
} cdo_iterate;

cdo_iterate *cdo_iterate_new(int64_t p) {
    cdo_iterate *x = (cdo_iterate *)calloc(1, sizeof(cdo_iterate));
    x->p = p;
    x->nzval = (double *)calloc((size_t)(p > 0 ? p : 1), sizeof(double));
    x->nzval2ind = (int64_t *)calloc((size_t)(p > 0 ? p : 1), sizeof(int64_t));
    x->ind2nzval = (int64_t *)calloc((size_t)(p > 0 ? p : 1), sizeof(int64_t));
    return x;
}
void cdo_iterate_free(cdo_iterate *x) {
    if (!x) return;
    free(x->nzval); free(x->nzval2ind); free(x->ind2nzval); free(x);
}
int64_t cdo_iterate_length(const cdo_iterate *x) { return x->p; }
int64_t cdo_iterate_nnz(const cdo_iterate *x) { return x->nnz; }

static inline double it_get(const cdo_iterate *x, int64_t k0) {
    int64_t s = x->ind2nzval[k0];
    return s ? x->nzval[s - 1] : 0.0;
}
/* setindex!: a non-zero written to an unstored coordinate appends a slot
 * (insertion order = order of first becoming non-zero, test/atom_iterator.jl:
 * 13-28); a zero written to a stored coordinate keeps its slot until
 * dropzeros!, so nnz is stable during an active pass (atom_iterator.jl:21). */
static inline void it_set(cdo_iterate *x, int64_t k0, double v) {
    int64_t s = x->ind2nzval[k0];
    if (s) { x->nzval[s - 1] = v; return; }
    if (v != 0.0) {
        x->nzval[x->nnz] = v;
        x->nzval2ind[x->nnz] = k0;
        x->nnz += 1;
        x->ind2nzval[k0] = x->nnz;
    }
}
double cdo_iterate_get(const cdo_iterate *x, int64_t k1) { return it_get(x, k1 - 1); }
void cdo_iterate_set(cdo_iterate *x, int64_t k1, double v) { it_set(x, k1 - 1, v); }
/* nzval2ind[1:nnz], 1-based */
void cdo_iterate_support(const cdo_iterate *x, int64_t *out) {
    for (int64_t i = 0; i < x->nnz; ++i) out[i] = x->nzval2ind[i] + 1;
}
void cdo_iterate_dense(const cdo_iterate *x, double *out) {
    memset(out, 0, sizeof(double) * (size_t)x->p);
    for (int64_t i = 0; i < x->nnz; ++i) out[x->nzval2ind[i]] = x->nzval[i];
}
/* fill!(x, 0) empties the iterate (coordinate_descent.jl:25) */
void cdo_iterate_fill_zero(cdo_iterate *x) {
    for (int64_t i = 0; i < x->nnz; ++i) x->ind2nzval[x->nzval2ind[i]] = 0;
    x->nnz = 0;
}
/* dropzeros!(x) (coordinate_descent.jl:108).  PARITY UNPINNED: order after
 * removal; swap-with-last compaction. */
void cdo_iterate_dropzeros(cdo_iterate *x) {
    int64_t i = 0;
    while (i < x->nnz) {
        if (x->nzval[i] == 0.0) {
            x->ind2nzval[x->nzval2ind[i]] = 0;
            int64_t last = x->nnz - 1;
            if (i != last) {
                x->nzval[i] = x->nzval[last];
                x->nzval2ind[i] = x->nzval2ind[last];
                x->ind2nzval[x->nzval2ind[i]] = i + 1;
            }
            x->nnz -= 1;
        } else {
            ++i;
        }
    }
}
void cdo_iterate_copy(cdo_iterate *dst, const cdo_iterate *src) {
    dst->nnz = src->nnz;
    memcpy(dst->nzval, src->nzval, sizeof(double) * (size_t)src->p);
    memcpy(dst->nzval2ind, src->nzval2ind, sizeof(int64_t) * (size_t)src->p);
    memcpy(dst->ind2nzval, src->ind2nzval, sizeof(int64_t) * (size_t)src->p);
}

/* ------------------------------------------------------------------------ */
/* ProxL1 (ProximalBase 0.3.0): lambda0 and optional per-coordinate weights  */
/* ------------------------------------------------------------------------ */
typedef struct cdo_prox {
    double lambda0;
    const double *omega; /* NULL = unweighted (ProxL1{T,Nothing}) */
} cdo_prox;

static inline double soft_threshold(double v, double t) {
    if (v > t) return v - t;
    if (v < -t) return v + t;
    return 0.0;
}
/* cdprox!(g, x, k, gamma): x[k] <- S(x[k], gamma*lambda0*(omega[k] or 1)),
 * returns the new value (call sites cd_differentiable_function.jl:103,336;
 * pinned by test/coordinate_descent.jl:13-25 and test/lasso.jl:36-56). */
static inline double cdprox(const cdo_prox *g, cdo_iterate *x, int64_t k0, double gamma) {
    double t = gamma * g->lambda0 * (g->omega ? g->omega[k0] : 1.0);
    double v = soft_threshold(it_get(x, k0), t);
    it_set(x, k0, v);
    return v;
}

/* ------------------------------------------------------------------------ */
/* Loss operators                                                            */
/* ------------------------------------------------------------------------ */
typedef struct cdo_loss {
    int kind;
    int64_t n, p, ld;
    const double *X; /* column-major n x p (QUAD: A, p x p) -- borrowed      */
    const double *y; /* [n] (QUAD: b, [p]) -- borrowed                       */
    const double *w; /* [n] observation weights (WLS) -- borrowed            */
    double *r;       /* [n] residual (QUAD: Ax, [p]) -- owned                */
    int32_t domain_error; /* sticky: sqrt-lasso sqrt of a negative           */
} cdo_loss;

/* Outer constructors: r = copy(y) (cd_differentiable_function.jl:52-55,
 * 128-131, 211-214); QUAD: Ax = zeros (:304-308). */
cdo_loss *cdo_loss_new(int kind, int64_t n, int64_t p, const double *X, int64_t ld,
                       const double *y, const double *w) {
    cdo_loss *f = (cdo_loss *)calloc(1, sizeof(cdo_loss));
    f->kind = kind; f->n = n; f->p = p; f->ld = ld; f->X = X; f->y = y; f->w = w;
    int64_t m = (kind == CDO_QUAD) ? p : n;
    f->r = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    if (kind != CDO_QUAD) memcpy(f->r, y, sizeof(double) * (size_t)n);
    return f;
}
void cdo_loss_free(cdo_loss *f) { if (f) { free(f->r); free(f); } }
double *cdo_loss_residual(cdo_loss *f) { return f->r; }
int64_t cdo_num_coordinates(const cdo_loss *f) { return f->p; }
int32_t cdo_loss_domain_error(const cdo_loss *f) { return f->domain_error; }

/* initialize!(f, x): r_i = y_i - sum_{j in supp} X[i,j]*x[j], row by row in
 * support order (cd_differentiable_function.jl:59-72,134-147,218-231 via
 * A_mul_B_row); QUAD: Ax_i = sum_j A[i,j]*x[j] (:309-319). */
void cdo_initialize(cdo_loss *f, const cdo_iterate *x) {
    int64_t m = (f->kind == CDO_QUAD) ? f->p : f->n;
    for (int64_t i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int64_t s = 0; s < x->nnz; ++s)
            acc += f->X[i + f->ld * x->nzval2ind[s]] * x->nzval[s];
        f->r[i] = (f->kind == CDO_QUAD) ? acc : f->y[i] - acc;
    }
}

static double dot2(const double *a, const double *b, int64_t n) {
    double s = 0.0;
#pragma omp simd reduction(+ : s)
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* gradient(f, x, k): LS -X_k'r/n (:75-76); SQRT -X_k'r/norm(r) (:234-235);
 * WLS -sum w x r / n (:149-158); QUAD Ax[k]+b[k] (:320-321). */
double cdo_gradient(const cdo_loss *f, const cdo_iterate *x, int64_t k1) {
    (void)x;
    int64_t k = k1 - 1;
    const double *xk = f->X + f->ld * k;
    switch (f->kind) {
    case CDO_LS: return -dot2(xk, f->r, f->n) / (double)f->n;
    case CDO_SQRT: return -dot2(xk, f->r, f->n) / sqrt(dot2(f->r, f->r, f->n));
    case CDO_WLS: {
        double s = 0.0;
#pragma omp simd reduction(+ : s)
        for (int64_t i = 0; i < f->n; ++i) s += f->w[i] * xk[i] * f->r[i];
        return -s / (double)f->n;
    }
    default: return f->r[k] + f->y[k];
    }
}

/* descendCoordinate!(f, g, x, k) -> h, for each loss. */
double cdo_descend(cdo_loss *f, const cdo_prox *g, cdo_iterate *x, int64_t k1) {
    int64_t k = k1 - 1;
    const int64_t n = f->n;
    const double *xk = f->X + f->ld * k;
    double *r = f->r;

    if (f->kind == CDO_LS || f->kind == CDO_WLS) {
        /* cd_differentiable_function.jl:83-111 / :165-194 */
        double a = 0.0, b = 0.0;
        if (f->kind == CDO_LS) {
#pragma omp simd reduction(+ : a, b)
            for (int64_t i = 0; i < n; ++i) { a += xk[i] * xk[i]; b += r[i] * xk[i]; }
        } else {
            const double *w = f->w;
#pragma omp simd reduction(+ : a, b)
            for (int64_t i = 0; i < n; ++i) {
                a += xk[i] * xk[i] * w[i];
                b += r[i] * xk[i] * w[i];
            }
        }
        double oldVal = it_get(x, k);
        it_set(x, k, oldVal + b / a);            /* x[k] += b/a          (:102) */
        double newVal = cdprox(g, x, k, (double)n / a); /*              (:103) */
        double h = newVal - oldVal;
#pragma omp simd
        for (int64_t i = 0; i < n; ++i) r[i] -= xk[i] * h; /* even if h==0 (:107) */
        return h;
    }
    if (f->kind == CDO_SQRT) {
        /* cd_differentiable_function.jl:242-291 */
        double xkv = it_get(x, k);
#pragma omp simd
        for (int64_t i = 0; i < n; ++i) r[i] += xk[i] * xkv; /* add back (:254) */
        double s = 0.0, xsqr = 0.0, rsqr = 0.0;
#pragma omp simd reduction(+ : s, xsqr, rsqr)
        for (int64_t i = 0; i < n; ++i) {
            xsqr += xk[i] * xk[i]; s += r[i] * xk[i]; rsqr += r[i] * r[i];
        }
        double lam = g->lambda0 * (g->omega ? g->omega[k] : 1.0); /* (:271-274) */
        double oldVal = xkv, newVal;
        if (fabs(s) <= lam * sqrt(rsqr)) {
            newVal = 0.0;
        } else {
            double u = 1.0 - lam * lam / xsqr;
            double v = rsqr - s * s / xsqr;
            if (u <= 0.0 || v < 0.0) {  /* Julia would throw DomainError here   */
                f->domain_error = 1;
                if (v < 0.0) v = 0.0;
            }
            double c = lam / sqrt(u) * sqrt(v);
            newVal = (s > 0 ? (s - c) : (s + c)) / xsqr; /* (:278-282) */
        }
        it_set(x, k, newVal);
#pragma omp simd
        for (int64_t i = 0; i < n; ++i) r[i] -= xk[i] * newVal; /* (:286) */
        return newVal - oldVal;
    }
    /* CDO_QUAD: cd_differentiable_function.jl:323-348 */
    {
        const int64_t p = f->p;
        double a = 1.0 / f->X[k + f->ld * k];
        double b = f->r[k] + f->y[k];
        double oldVal = it_get(x, k);
        it_set(x, k, oldVal - b * a);
        double newVal = cdprox(g, x, k, a);
        double h = newVal - oldVal;
#pragma omp simd
        for (int64_t i = 0; i < p; ++i) r[i] += xk[i] * h;
        return h;
    }
}

/* ------------------------------------------------------------------------ */
/* Coordinate iterators (atom_iterator.jl)                                   */
/* ------------------------------------------------------------------------ */
/* The reference's RandomIterator draws from Julia's global MersenneTwister
 * (atom_iterator.jl:60), which cannot be reproduced outside Julia.  The
 * restatement keeps the algorithm (identity refill of the first L slots, then
 * Fisher-Yates with j = rand(i:L)) and substitutes a documented generator:
 * splitmix64, j = i + next() mod (L - i + 1).  The product uses the same
 * definition so seeded random visit orders agree. */
typedef struct cdo_iter {
    int randomize;
    int fullPass;
    int64_t p;
    int64_t *order; /* [p], 0-based */
    uint64_t state;
} cdo_iter;

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
cdo_iter *cdo_iter_new(int64_t p, int randomize, uint64_t seed) {
    cdo_iter *it = (cdo_iter *)calloc(1, sizeof(cdo_iter));
    it->p = p; it->randomize = randomize; it->fullPass = 1; it->state = seed;
    it->order = (int64_t *)calloc((size_t)(p > 0 ? p : 1), sizeof(int64_t));
    for (int64_t i = 0; i < p; ++i) it->order[i] = i; /* collect(1:p) (:51) */
    return it;
}
void cdo_iter_free(cdo_iter *it) { if (it) { free(it->order); free(it); } }
/* reset!(it, fullPass) (atom_iterator.jl:34-37, 53-64) */
void cdo_iter_reset(cdo_iter *it, const cdo_iterate *x, int fullPass) {
    it->fullPass = fullPass;
    if (!it->randomize) return;
    int64_t L = fullPass ? it->p : x->nnz;
    for (int64_t i = 0; i < L; ++i) it->order[i] = i;
    for (int64_t i = 0; i + 1 < L; ++i) {
        int64_t j = i + (int64_t)(splitmix64(&it->state) % (uint64_t)(L - i));
        int64_t t = it->order[i]; it->order[i] = it->order[j]; it->order[j] = t;
    }
}
/* collect(it): 1-based visit list; returns its length (atom_iterator.jl:
 * 18-26, 67-75). */
int64_t cdo_iter_collect(const cdo_iter *it, const cdo_iterate *x, int64_t *out) {
    int64_t L = it->fullPass ? it->p : x->nnz;
    for (int64_t i = 0; i < L; ++i) {
        int64_t o = it->randomize ? it->order[i] : i;
        out[i] = (it->fullPass ? o : x->nzval2ind[o]) + 1;
    }
    return L;
}

/* ------------------------------------------------------------------------ */
/* Driver (coordinate_descent.jl)                                            */
/* ------------------------------------------------------------------------ */
typedef struct cdo_options { /* CDOptions, utils.jl:7-20, + seed for the RNG */
    int64_t maxIter;
    double optTol;
    int32_t randomize;
    int32_t warmStart;
    int64_t numSteps;
    uint64_t seed;
} cdo_options;

typedef struct cdo_stats { /* superset: the reference returns nothing (Q7) */
    int64_t passes;
    int64_t full_passes;
    int64_t visits;
    int32_t converged;
    double last_maxH;
} cdo_stats;

/* _cdPass! over an explicit 1-based visit list (coordinate_descent.jl:94-110) */
double cdo_pass(cdo_loss *f, const cdo_prox *g, cdo_iterate *x, int64_t m,
                const int64_t *idx1) {
    double maxH = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        double h = cdo_descend(f, g, x, idx1[i]);
        if (fabs(h) > maxH) maxH = fabs(h);
    }
    cdo_iterate_dropzeros(x);
    return maxH;
}

/* _coordinateDescent! (coordinate_descent.jl:65-92): first pass is full; a
 * converged pass is followed by a full pass; stops after two converged passes
 * in a row, i.e. only after a FULL pass with maxH < optTol. */
static void solve(cdo_loss *f, const cdo_prox *g, cdo_iterate *x, cdo_iter *it,
                  const cdo_options *o, cdo_stats *st, int64_t *scratch) {
    int prev_converged = 0, converged = 1;
    for (int64_t iter = 0; iter < o->maxIter; ++iter) {
        cdo_iter_reset(it, x, converged);
        int64_t m = cdo_iter_collect(it, x, scratch);
        double maxH = cdo_pass(f, g, x, m, scratch);
        if (st) {
            st->passes += 1; st->visits += m; st->last_maxH = maxH;
            if (it->fullPass) st->full_passes += 1;
        }
        prev_converged = converged;
        converged = maxH < o->optTol;
        if (prev_converged && converged) { if (st) st->converged = 1; return; }
    }
    if (st) st->converged = 0;
}

/* _findLambdaMax (coordinate_descent.jl:118-149) */
double cdo_lambda_max(const cdo_loss *f, const cdo_iterate *x, const cdo_prox *g) {
    double lmax = 0.0;
    for (int64_t k = 1; k <= f->p; ++k) {
        double t = fabs(cdo_gradient(f, x, k));
        if (g->omega) t /= g->omega[k - 1];
        if (t > lmax) lmax = t;
    }
    return lmax;
}

/* coordinateDescent!(x, f, g::ProxL1, options) (coordinate_descent.jl:7-39).
 * Returns 0, or 1 = DimensionMismatch (:13-16), 2 = the cold-start range
 * l1:(l2-l1)/numSteps:l2 has a zero step (Julia ArgumentError). */
int32_t cdo_coordinate_descent(cdo_loss *f, const cdo_prox *g, int64_t n_omega,
                               cdo_iterate *x, const cdo_options *o, cdo_stats *st) {
    if (st) memset(st, 0, sizeof(*st));
    if (x->p != f->p) return 1;
    if (g->omega && n_omega != f->p) return 1;
    cdo_iter *it = cdo_iter_new(f->p, o->randomize, o->seed);
    int64_t *scratch = (int64_t *)malloc(sizeof(int64_t) * (size_t)(f->p > 0 ? f->p : 1));
    int32_t rc = 0;
    if (o->warmStart) {
        cdo_initialize(f, x);
        solve(f, g, x, it, o, st, scratch);
    } else {
        cdo_iterate_fill_zero(x);
        cdo_initialize(f, x);
        double lmax = cdo_lambda_max(f, x, g);
        double l1 = log(lmax), l2 = log(g->lambda0);
        double step = (l2 - l1) / (double)o->numSteps;
        if (step == 0.0 || !(step == step)) {
            rc = 2;
        } else {
            /* Julia's range l1:step:l2 has numSteps+1 points l1 + j*step; the
             * last is (up to rounding) l2, so the final solve runs at
             * exp(log(lambda0)). */
            for (int64_t j = 0; j <= o->numSteps; ++j) {
                double l = (j == o->numSteps) ? l2 : l1 + (double)j * step;
                cdo_prox g1 = { exp(l), g->omega };
                solve(f, &g1, x, it, o, st, scratch);
            }
        }
    }
    free(scratch);
    cdo_iter_free(it);
    return rc;
}

/* _stdX!(out, X): out_j = sqrt(sum_i X_ij^2 / n) (utils.jl:127-138) */
void cdo_std_x(int64_t n, int64_t p, const double *X, int64_t ld, double *out) {
    for (int64_t j = 0; j < p; ++j) {
        const double *xj = X + ld * j;
        out[j] = sqrt(dot2(xj, xj, n) / (double)n);
    }
}

/* Objective values (coordinate_descent.jl:1-3 with the losses' header
 * comments; for SQRT the quantity the update actually minimises, SURVEY a-6):
 *   LS/WLS: sum w r^2/(2n) + lambda0 sum omega|b|;  SQRT: ||r||_2 + ...;
 *   QUAD: x'Ax/2 + x'b + ...   (r must be consistent with x). */
double cdo_objective(const cdo_loss *f, const cdo_prox *g, const cdo_iterate *x) {
    double pen = 0.0;
    for (int64_t s = 0; s < x->nnz; ++s)
        pen += fabs(x->nzval[s]) * (g->omega ? g->omega[x->nzval2ind[s]] : 1.0);
    pen *= g->lambda0;
    if (f->kind == CDO_QUAD) {
        double v = 0.0;
        for (int64_t s = 0; s < x->nnz; ++s) {
            int64_t k = x->nzval2ind[s];
            v += x->nzval[s] * (0.5 * f->r[k] + f->y[k]);
        }
        return v + pen;
    }
    double ss = 0.0;
    for (int64_t i = 0; i < f->n; ++i)
        ss += (f->kind == CDO_WLS ? f->w[i] : 1.0) * f->r[i] * f->r[i];
    if (f->kind == CDO_SQRT) return sqrt(ss) + pen;
    return ss / (2.0 * (double)f->n) + pen;
}

/* ------------------------------------------------------------------------ */
/* CPU baseline timing kernel for bench.py (`cpu_baseline`, kind "port"):    */
/* `visits` least-squares coordinate visits over columns 0..ncol-1 cycled,   */
/* exactly the two loops of cd_differentiable_function.jl:96-99,107-109.     */
/* threads == 1 is the faithful single-threaded reference path; threads > 1  */
/* splits the n-loops with OpenMP (a ceiling the reference does not have).   */
/* Returns the sum of h (keeps the work observable).                         */
/* ------------------------------------------------------------------------ */
double cdo_bench_ls_visits(int64_t n, int64_t ncol, const double *X, int64_t ld, double *r,
                           double *beta, double lambda0, int64_t visits, int32_t threads) {
    double acc = 0.0;
    for (int64_t v = 0; v < visits; ++v) {
        int64_t k = v % ncol;
        const double *xk = X + ld * k;
        double a = 0.0, b = 0.0;
        if (threads <= 1) {
#pragma omp simd reduction(+ : a, b)
            for (int64_t i = 0; i < n; ++i) { a += xk[i] * xk[i]; b += r[i] * xk[i]; }
        } else {
#ifdef _OPENMP
#pragma omp parallel for simd reduction(+ : a, b) num_threads(threads) schedule(static)
#endif
            for (int64_t i = 0; i < n; ++i) { a += xk[i] * xk[i]; b += r[i] * xk[i]; }
        }
        double oldVal = beta[k];
        double newVal = soft_threshold(oldVal + b / a, lambda0 * (double)n / a);
        double h = newVal - oldVal;
        beta[k] = newVal;
        if (threads <= 1) {
#pragma omp simd
            for (int64_t i = 0; i < n; ++i) r[i] -= xk[i] * h;
        } else {
#ifdef _OPENMP
#pragma omp parallel for simd num_threads(threads) schedule(static)
#endif
            for (int64_t i = 0; i < n; ++i) r[i] -= xk[i] * h;
        }
        acc += h;
    }
    return acc;
}

int32_t cdo_max_threads(void) {
#ifdef _OPENMP
    return (int32_t)omp_get_max_threads();
#else
    return 1;
#endif
}
