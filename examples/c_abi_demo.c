/* c_abi_demo.c -- the drop-in boundary used from plain C: no Python, no torch.
 *
 *   gcc -O2 -I include examples/c_abi_demo.c -o examples/c_abi_demo \
 *       -L coordinatedescent.jl_amd/csrc -lcdhip -Wl,-rpath,$PWD/coordinatedescent.jl_amd/csrc -lm
 *
 * Solves a small Lasso (shapes of the reference's test/lasso.jl:76-101: n = 200, p = 50, s = 10,
 * lambda = 0.2) with coordinateDescent!'s C entry point and checks the KKT condition the
 * reference's test asserts: max |X'(y - X beta)| / n == lambda on the support.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "cdhip.h"

static double lcg_uniform(unsigned long long* s) {
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)(*s >> 11) + 0.5) / 9007199254740992.0;
}
static double lcg_normal(unsigned long long* s) {
    double u1 = lcg_uniform(s), u2 = lcg_uniform(s);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

#define OK(call)                                                                         \
    do {                                                                                 \
        int32_t st_ = (call);                                                            \
        if (st_ != CDH_OK) {                                                             \
            fprintf(stderr, "%s -> status %d: %s\n", #call, st_, cdh_last_error(h));     \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main(void) {
    const int64_t n = 200, p = 50, s = 10;
    const double lambda = 0.2;
    unsigned long long seed = 12345;
    double* X = malloc(sizeof(double) * n * p);
    double* y = calloc(n, sizeof(double));
    double* beta = malloc(sizeof(double) * p);
    double* r = malloc(sizeof(double) * n);
    for (int64_t j = 0; j < p; ++j)
        for (int64_t i = 0; i < n; ++i) X[i + n * j] = lcg_normal(&seed);
    for (int64_t j = 0; j < s; ++j) {
        double b = lcg_normal(&seed);
        for (int64_t i = 0; i < n; ++i) y[i] += X[i + n * j] * b;
    }
    for (int64_t i = 0; i < n; ++i) y[i] += 0.1 * lcg_normal(&seed);

    cdh_handle h = NULL;
    OK(cdh_create(&h, CDH_F64, CDH_LS, n, n, 0, p, 0));          /* CDLeastSquaresLoss(y, X) */
    OK(cdh_set_X_cols(h, 0, p, X, n));
    OK(cdh_set_y(h, y));
    OK(cdh_set_penalty(h, lambda, NULL, 0));                      /* ProxL1(0.2)              */
    OK(cdh_set_sweep_mode(h, CDH_SWEEP_BLOCK, 16));
    cdh_options opt = {2000, 1e-12, /*randomize*/ 0, /*warmStart*/ 1, 50, 0};
    cdh_stats st;
    OK(cdh_coordinate_descent(h, &opt, &st));                     /* coordinateDescent!       */
    OK(cdh_get_beta(h, beta));
    OK(cdh_get_residual(h, r));

    double kkt = 0.0;
    int nnz = 0;
    for (int64_t j = 0; j < p; ++j) {
        double g = 0.0;
        for (int64_t i = 0; i < n; ++i) g += X[i + n * j] * r[i];
        g = fabs(g) / (double)n;
        if (g > kkt) kkt = g;
        nnz += beta[j] != 0.0;
    }
    printf("passes=%lld converged=%d nnz=%d  max|X'r|/n=%.12f (lambda=%.1f)\n", (long long)st.passes,
           st.converged, nnz, kkt, lambda);
    cdh_destroy(h);
    free(X); free(y); free(beta); free(r);
    return (st.converged && fabs(kkt - lambda) / lambda < 1e-9) ? 0 : 2;
}
