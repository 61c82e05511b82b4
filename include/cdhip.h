/*
 * cdhip.h -- C ABI of the MI355X-native coordinate-descent sweep.
 *
 * This is the drop-in boundary for ONE path of mlakolar/CoordinateDescent.jl
 * v0.3.0: coordinateDescent! -> _coordinateDescent! -> _cdPass! ->
 * descendCoordinate! for CDLeastSquaresLoss / CDSqrtLassoLoss (/ CDWeightedLSLoss)
 * with a ProxL1 penalty, plus the helpers that path calls (initialize!, gradient,
 * _findLambdaMax, _stdX!).  Each entry point names the reference interface it
 * replaces (paths relative to the reference's src/).  The Julia binding a
 * maintainer would add is in INTEGRATION.md and julia/CoordinateDescentHIP.jl.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns a cdh_status.
 *   - Coordinates cross the ABI 1-based int64 (Julia's Int64), rebased inside.
 *   - Host arrays are borrowed for the duration of the call only.
 *   - X is column-major n x p with leading dimension ld (elements), dtype T;
 *     y, r, w are length-n vectors of T; beta/omega and all scalars are double.
 *   - One process drives one GPU.  With several processes each handle owns the
 *     row shard [row_offset, row_offset + n_local) of an n_total-row problem and
 *     the per-coordinate gradient scalars are summed with an RCCL all-reduce
 *     (cdh_comm_init).  beta, omega and every scalar are replicated.
 *   - A handle is not thread-safe; calls block until results are host-visible.
 *   - Nothing here falls back to a CPU path: without a GPU cdh_create fails
 *     with CDH_HIP_ERROR.
 */
#ifndef CDHIP_H
#define CDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cdh_handle_s *cdh_handle;

/* Julia exceptions the codes map to: DIM_MISMATCH -> DimensionMismatch
 * (coordinate_descent.jl:13,15; cd_differentiable_function.jl:53,129,212),
 * BAD_ARG -> ArgumentError (lasso.jl:128; the zero-step cold-start range),
 * DOMAIN -> DomainError (sqrt of a negative, cd_differentiable_function.jl:280,282). */
typedef enum cdh_status {
    CDH_OK = 0,
    CDH_DIM_MISMATCH = 1,
    CDH_BAD_ARG = 2,
    CDH_DOMAIN = 3,
    CDH_HIP_ERROR = 4,
    CDH_RCCL_ERROR = 5,
    CDH_OOM = 6
} cdh_status;

typedef enum cdh_dtype { CDH_F64 = 0, CDH_F32 = 1 } cdh_dtype;

/* CDLeastSquaresLoss (cd_differentiable_function.jl:43-111), CDSqrtLassoLoss
 * (:202-291), CDWeightedLSLoss (:118-194). */
typedef enum cdh_loss { CDH_LS = 0, CDH_SQRT = 1, CDH_WLS = 2 } cdh_loss;

/* How a pass is executed on the device (no reference counterpart).
 *  CDH_SWEEP_COORD : one fused kernel per coordinate visit (apply the previous
 *                    visit's residual update, then the dots of this column),
 *                    one gradient all-reduce per coordinate.
 *  CDH_SWEEP_BLOCK : visits are taken B at a time; one kernel reads the B
 *                    columns once and returns X_B'r and the B x B Gram block, the
 *                    B scalar updates run on those numbers (algebraically the
 *                    same iterates), then one rank-B residual update.  One
 *                    all-reduce per block.  B in {2, 4, 8, 16, 32, 64}; with
 *                    observation weights (CDH_WLS) B >= 16, narrower widths then
 *                    run the per-coordinate sweep.
 * A new handle sweeps in blocks of B = 32, the fastest width on one GPU for long columns (B = 64 for fp64 columns of
 * fewer than 262 144 rows, where a block of visits costs its three launches whatever it reads). */
typedef enum cdh_sweep_mode { CDH_SWEEP_COORD = 0, CDH_SWEEP_BLOCK = 1 } cdh_sweep_mode;

/* CDOptions (utils.jl:7-20), field for field, + the seed of the substitute RNG
 * (the reference's RandomIterator uses Julia's global RNG, atom_iterator.jl:60). */
typedef struct cdh_options {
    int64_t maxIter;   /* 2000 */
    double optTol;     /* 1e-7 */
    int32_t randomize; /* 1    */
    int32_t warmStart; /* 1    */
    int64_t numSteps;  /* 50   */
    uint64_t seed;
} cdh_options;

/* The reference reports nothing (maxIter exhaustion is silent,
 * coordinate_descent.jl:74-91); this is a strict superset. */
typedef struct cdh_stats {
    int64_t passes;
    int64_t full_passes;
    int64_t visits;
    int32_t converged;
    int32_t domain_error;
    double maxH;       /* of the last pass */
    double lambda_max; /* cold start only  */
} cdh_stats;

/* ---- lifetime -------------------------------------------------------------- */
/* Replaces the CDLeastSquaresLoss / CDSqrtLassoLoss / CDWeightedLSLoss outer
 * constructors (cd_differentiable_function.jl:52-55, 211-214, 128-131): allocates
 * X, y, r (and w) on `device`.  Single process: n_local == n_total, row_offset 0. */
int32_t cdh_create(cdh_handle *out, int32_t dtype, int32_t loss, int64_t n_local,
                   int64_t n_total, int64_t row_offset, int64_t p, int32_t device);
int32_t cdh_destroy(cdh_handle h);
const char *cdh_last_error(cdh_handle h); /* h may be NULL: last create error */
int32_t cdh_device_count(int32_t *out);
int32_t cdh_synchronize(cdh_handle h);

/* ---- data ------------------------------------------------------------------ */
/* Upload columns [j0, j0+ncols) (0-based here: a bulk-transfer helper, not a
 * coordinate) of the local row shard from a host column-major block. */
int32_t cdh_set_X_cols(cdh_handle h, int64_t j0, int64_t ncols, const void *host, int64_t ld);
int32_t cdh_get_X_cols(cdh_handle h, int64_t j0, int64_t ncols, void *host, int64_t ld);
/* y (and r = copy(y), as the loss constructors do). */
int32_t cdh_set_y(cdh_handle h, const void *host_y);
int32_t cdh_get_y(cdh_handle h, void *host_y);
/* Observation weights of CDWeightedLSLoss. */
int32_t cdh_set_obs_weights(cdh_handle h, const void *host_w);
/* Which loss the resident X serves from now on.  The reference builds a new loss object around the
 * same matrix for every front-end call (lasso.jl:33,48,71,93,117,245: CDLeastSquaresLoss(y, X),
 * CDSqrtLassoLoss(y, X), CDWeightedLSLoss(y, X, w)); the binding keeps X in HBM across those objects
 * and switches the handle to the kind of the loss it is called with.  Invalidates the carried
 * residual; CDH_WLS needs cdh_set_obs_weights before the next sweep (weights are zero until then). */
int32_t cdh_set_loss(cdh_handle h, int32_t loss);
/* Synthetic Gaussian problem generated on the device from a counter-based
 * generator keyed (seed, global row, column), shapes of benchmark/cd_bench.jl:
 * 8-14: X_ij ~ N(0,1), beta*_j = z_j (1 + u_j) for j < s, y = X beta* + noise e.
 * out_beta_star (may be NULL) receives the s planted values. */
int32_t cdh_generate(cdh_handle h, uint64_t seed, int64_t s, double noise, double *out_beta_star);

/* ---- penalty: ProxL1(lambda0) / ProxL1(lambda0, omega) (ProximalBase) -------- */
/* omega == NULL means unweighted; otherwise n_omega must equal p
 * (coordinate_descent.jl:14-16 -> CDH_DIM_MISMATCH). */
int32_t cdh_set_penalty(cdh_handle h, double lambda0, const double *omega, int64_t n_omega);

/* ---- the operator interface (cd_differentiable_function.jl:1-35) ------------- */
/* numCoordinates(f) */
int32_t cdh_num_coordinates(cdh_handle h, int64_t *out);
/* initialize!(f, x): load the iterate (support in SparseIterate order: idx1[i],
 * val[i]) and rebuild r = y - X beta (cd_differentiable_function.jl:59-72).
 * x_length is numCoordinates(x), checked against p (coordinate_descent.jl:13). */
int32_t cdh_initialize(cdh_handle h, int64_t x_length, int64_t nnz, const int64_t *idx1,
                       const double *val);
/* Load the iterate WITHOUT touching r: what the binding calls before gradient /
 * descendCoordinate! / _cdPass! when the caller's x changed since the last call
 * (the reference reads x[k] directly, cd_differentiable_function.jl:101,253). */
int32_t cdh_set_iterate(cdh_handle h, int64_t x_length, int64_t nnz, const int64_t *idx1,
                        const double *val);
/* gradient(f, x, k) (cd_differentiable_function.jl:75-76, 234-235, 149-158) */
int32_t cdh_gradient(cdh_handle h, int64_t k1, double *out);
/* descendCoordinate!(f, g, x, k) -> h (cd_differentiable_function.jl:83-111,
 * 242-291, 165-194); r is left consistent with beta on return. */
int32_t cdh_descend(cdh_handle h, int64_t k1, double *out_h);

/* ---- the driver (coordinate_descent.jl) -------------------------------------- */
/* _findLambdaMax (coordinate_descent.jl:118-149), at the current r. */
int32_t cdh_lambda_max(cdh_handle h, double *out);
/* _cdPass! over an explicit visit list (coordinate_descent.jl:94-110):
 * maxH = max |h|, then dropzeros!. */
int32_t cdh_pass(cdh_handle h, int64_t m, const int64_t *idx1, double *out_maxH);
/* _coordinateDescent! (coordinate_descent.jl:65-92): assumes r is initialised. */
int32_t cdh_solve(cdh_handle h, const cdh_options *opt, cdh_stats *out);
/* coordinateDescent!(x, f, g::ProxL1, options) (coordinate_descent.jl:7-39):
 * warm start = initialize! from the handle's current iterate + one solve; cold
 * start = zero, lambda_max, numSteps+1 solves down a log grid. */
int32_t cdh_coordinate_descent(cdh_handle h, const cdh_options *opt, cdh_stats *out);

/* ---- results ------------------------------------------------------------------ */
int32_t cdh_get_beta(cdh_handle h, double *out_p);                       /* Vector(x)   */
int32_t cdh_get_support(cdh_handle h, int64_t *out_idx1, int64_t *out_nnz); /* nzval2ind  */
int32_t cdh_get_residual(cdh_handle h, void *out_n_local);               /* f.r (dtype) */
/* _stdX!(out, X) (utils.jl:127-138): out_j = sqrt(sum_i X_ij^2 / n_total). */
int32_t cdh_col_rms(cdh_handle h, double *out_p);
/* out_j = X_j' r for every column at the current r (At_mul_B_row for all j: the
 * screening scores of _findLargestCorrelations, utils.jl:96-106; KKT checks). */
int32_t cdh_xt_r(cdh_handle h, double *out_p);
/* Gram block of up to 4096 columns against the current r:
 * out_G[i*m+j] = X_i'X_j, out_c[i] = X_i'r, *out_q = r'r (all shards).  With r = y this is
 * what the s-column OLS of _findInitResiduals! needs (utils.jl:65-77: Xs \ y), without moving
 * any n-sized array to the host.  Up to 64 columns in one pass over them; more (round 4: the
 * reference takes any s) in groups of 32, one launch per pair of groups. */
int32_t cdh_gram(cdh_handle h, int64_t m, const int64_t *idx1, double *out_G, double *out_c,
                 double *out_q);
/* out_m[i] = X_idx1[i]' r (X'Wr for the weighted loss) for a LIST of columns at the current r: one pass over those
 * columns only.  The refinement step of the screening init: Xs \ y (utils.jl:70, a QR in the reference) is solved from the
 * Gram block, and the normal equations' residual Xs'(y - Xs b) read off here corrects b -- so near-collinear screening
 * columns do not cost the squared condition number. */
int32_t cdh_xt_r_cols(cdh_handle h, int64_t m, const int64_t *idx1, double *out_m);
/* std(f.r) as Statistics.std computes it (lasso.jl:37,52,81,97,143): two passes -- the mean, then the centred sum of
 * squares, Bessel-corrected -- over all shards.  out_mean may be NULL. */
int32_t cdh_resid_std(cdh_handle h, double *out_std, double *out_mean);
/* sum r, sum r^2 over all shards (sigma of scaledLasso!, lasso.jl:134; std(f.r)). */
int32_t cdh_resid_moments(cdh_handle h, double *out_sum, double *out_sumsq);
/* f(beta) + lambda0 sum omega|beta| at the current state (coordinate_descent.jl:1-3). */
int32_t cdh_objective(cdh_handle h, double *out);

/* ---- execution control (no reference counterpart) ------------------------------- */
/* default: CDH_SWEEP_BLOCK, block = 32; `block` is ignored for CDH_SWEEP_COORD */
int32_t cdh_set_sweep_mode(cdh_handle h, int32_t mode, int32_t block);
/* Warm starts normally rebuild r = y - X beta as the reference's initialize! does
 * (coordinate_descent.jl:21).  With reuse on, a warm start whose iterate is the one the
 * handle already holds keeps the carried residual (LassoPath, lasso.jl:250-252: 100 rebuilds
 * of n x nnz work saved); the difference is rounding only. */
int32_t cdh_set_reuse_residual(cdh_handle h, int32_t on);
/* Full passes of cdh_solve / cdh_coordinate_descent over a sparse iterate are screened by
 * default: runs of visits that provably leave beta and r unchanged (beta_k == 0 and |X_k'r|
 * below the threshold) are settled from one dots-only pass over their columns; the iterates
 * are the same as visiting one by one.  on: 0 = never, 1 = full passes of the solves (default; cdh_pass
 * visits every column it is given), 2 = cdh_pass as well. */
int32_t cdh_set_screening(cdh_handle h, int32_t on);
/* Gradient cache of the screened full passes (exact; every loss, fp64 and fp32 storage).  With the Gram
 * columns X'X_j of the coordinates that move, X_k'r is known for every k without reading X, so a full pass
 * over a sparse iterate costs its real visits only -- what a lambda path (lasso.jl:250-252), the sigma loop
 * of scaledLasso! (:132-141) or a cold start's 51 solves (coordinate_descent.jl:32-36) repeat hundreds of
 * times on one X.  mode: 0 off; 1 (default) engages once the handle has run as many screened full passes on
 * the same data as the Gram columns of its support cost to fetch (at least three; a cold start engages at
 * once); 2 from the first full pass (what a path driver that knows it has 100 lambdas to go asks for).
 * On columns of 1 MB and more, modes 1 and 2 engage only while n_total >= 32 * nnz(x) (round 3: the fold of a move into the
 * gradient and the re-check of skipped certificates run on the device; while they were p host flops per mover the bound was
 * 400): a support that large makes most visits real ones anyway; mode 3 is mode 2 without that guard (tests).  Shorter
 * columns keep the cache whatever n / nnz is (a streamed visit is launch-bound there and loses to the covariance form).
 * Same iterates, support order and pass counts as visiting every coordinate.
 * While the cache is engaged the visits themselves need no read of X either (least squares, sqrt-lasso): the block
 * record the scalar-update kernel consumes -- X_k'r, the Gram entries between the block's coordinates -- is
 * read off the cached gradient and Gram columns ("covariance form"), and r is brought up to date once, before
 * anything reads it (cdh_get_residual, the moments, a streamed visit, ...).
 * cdh_cache_stats: out10 = {passes served, visits settled from the cache, visits made in those passes,
 * dots-only re-reference passes over X, Gram batches (up to 32 columns, one pass over X each),
 * Gram columns held, visits made in covariance form, residual catch-ups, covariance chunks / passes rolled back
 * because a skipped coordinate's certificate did not survive their own moves, passes served entirely on the
 * device (scan, visits and certificate re-check without the gradient leaving HBM)}. */
int32_t cdh_set_gradient_cache(cdh_handle h, int32_t mode);
/* The mode the handle is in now (a path driver upgrades 1 -> 2 for its own duration and puts back what it found:
 * lasso.jl:250-252 runs on a caller's loss object, whose settings are the caller's). */
int32_t cdh_get_gradient_cache(cdh_handle h, int32_t *out_mode);
int32_t cdh_cache_stats(cdh_handle h, int64_t *out10);
/* While the cache serves the passes of a solve, the pass loop of _coordinateDescent! itself (coordinate_descent.jl:74-91:
 * reset!(it, full), _cdPass!, dropzeros!, the convergence test) runs on the device -- one workgroup, one host round trip per
 * solve instead of two per pass; it comes back early only for what the host alone can give (Gram columns of coordinates
 * about to enter, a re-reference, the careful walk after a certificate broke).  Same iterates, support order and pass
 * counts.  On by default (environment CDH_COV_SOLVE=0, or on = 0 here, keeps the pass loop on the host).
 * cdh_device_loop_stats: out12 = {launches of the loop, passes run inside it, in-kernel folds of pending moves into the
 * gradient, certificates re-checked with the exact gradient because the bound did not cover them, and the time the kernel
 * spent per phase in 10 ns ticks: building visit lists, the scan, exact gradients of the visited coordinates, the visits,
 * the re-check, accepting a pass, the SparseIterate bookkeeping, dropzeros! and the rest}. */
int32_t cdh_set_device_loop(cdh_handle h, int32_t on);   /* 0: off; 1: on; 2: on, without helper workgroups; n > 2: on, with n helpers (at most 64) */
int32_t cdh_device_loop_stats(cdh_handle h, int64_t *out12);
/* Visit lists beyond the loop's LDS-sized Gram block (~150 non-zeros) run from a Gram TABLE in device memory: X_j'X_k by
 * table id for every coordinate the loop has visited, filled from the cached columns as coordinates enter and kept from
 * solve to solve, with the exact gradient of every coordinate it holds carried through each block of visits (rows of the
 * table: contiguous).  out8 = {passes run in table mode, table rows filled, coordinates the table holds, its capacity,
 * and how often a full pass whose re-check found coordinates crossing their threshold through the pass's own moves was
 * run again with those coordinates visited (instead of being walked position by position): by the host's device pass, by
 * the loop itself}.
 * Launches that expect such lists bring HELPER workgroups (the crew; CDH_CS_CREW = their number, default 31, 0: none): they
 * keep the gradient current for all p coordinates while workgroup 0 visits, and re-check the skipped coordinates of full
 * passes on the way, so that full passes of large supports stay in the loop too; out8[6..7] = {passes run with the crew,
 * jobs it was given}. */
int32_t cdh_device_loop_table(cdh_handle h, int64_t *out8);
/* How far the carried gradient has been from X'r whenever it was taken afresh from X (after CDH_GC_REFRESH
 * covariance-form visits, or right now with rereference_now != 0: one dots-only pass over X):
 *   drift = max_k |g_carried[k] - X_k'r| / thr_k,   thr_k = lambda0 n omega_k (sqrt-lasso: lambda0 omega_k ||r||)
 * -- the quantity the certificates' relative margin (1e-9 for fp64 storage) has to cover.
 * out3 = {drift at the last re-reference, the largest seen on this handle, re-references measured}. */
int32_t cdh_cache_drift(cdh_handle h, int32_t rereference_now, double *out3);
/* Inspection: the Gram column X'X_k (X'WX_k) the cache holds for coordinate k1 (1-based; CDH_BAD_ARG if it holds none),
 * p doubles, and the relative error its entries are DECLARED to carry, in units of sqrt(a_i a_k): 0 for fp64 storage; for
 * fp32 storage 2^-24 * 512 / sqrt(n_total) -- the fp32 matrix pipe sums 256-row chunks in fp32 before folding them into
 * fp64 -- which the certificates of the skipped visits allow for (no reference counterpart: the reference reads X itself at
 * every visit, cd_differentiable_function.jl:94-99). */
int32_t cdh_cache_gram_column(cdh_handle h, int64_t k1, double *out_p, double *out_eps);
/* Problems whose Gram matrix fits on chip -- p <= 1024 columns, one process -- are solved in ONE launch.  With at most
 * 16 MB of X (the reference's own test and benchmark shapes: test/lasso.jl:76-101, benchmark/cd_bench.jl:8-14) from the
 * first solve; with more, once the solves the handle ran on the streamed kernels have cost what building the matrix
 * would (ceil(p / 32) passes over X; a cold start, being numSteps + 1 solves, buys a cheap build outright).  The handle keeps the
 * full Gram matrix X'X (X'WX) of the resident X, and one wave runs the whole of _coordinateDescent!
 * (coordinate_descent.jl:65-92; a cold start's numSteps + 1 solves, :32-36, in the same launch) in covariance form, with
 * the SparseIterate bookkeeping that fixes the visit order of active passes replayed on the device.  Same iterates as the
 * streamed sweeps up to rounding.  On by default (environment CDH_SMALL_PATH=0, or on = 0 here, keeps every solve on the
 * streamed kernels).  cdh_onchip_stats: out2 = {solves run this way, Gram matrices built}. */
int32_t cdh_set_onchip_solve(cdh_handle h, int32_t on);
int32_t cdh_onchip_stats(cdh_handle h, int64_t *out2);
/* Of the last one-launch solve: out3 = {visit steps taken (a step settles a run of positions and makes at most one
 * move), shader cycles the kernel ran for, the same span in 100 MHz ticks} -- cycles / ticks * 0.1 is the clock in GHz
 * the chip held for a one-wave kernel. */
int32_t cdh_onchip_last(cdh_handle h, int64_t *out3);
/* Replay each pass from a captured hipGraph instead of individual launches (the north_star's
 * "full sweep captured under hipGraph").  Works on row shards too: the direct exchange takes its epoch
 * from device memory, RCCL all-reduces are captured with the kernels around them; only the host-staged
 * exchange (cdh_set_host_exchange) is launched node by node. */
int32_t cdh_set_use_graph(cdh_handle h, int32_t on);
/* Multi-process row sharding: rank 0 calls cdh_comm_unique_id, broadcasts the
 * 128 bytes (any transport), every rank calls cdh_comm_init. */
int32_t cdh_comm_unique_id(void *out_128_bytes);
int32_t cdh_comm_init(cdh_handle h, const void *id_128_bytes, int32_t rank, int32_t nranks);
/* Give the communicator up (ncclCommAbort) so that another exchange can be installed on the handle -- for a
 * caller that found it unusable (wrong sums from cdh_exchange_probe).  The shard stays a shard: until
 * cdh_set_host_exchange / the direct exchange takes over, every exchange on it refuses rather than sum
 * local rows only.  A handle that never had a communicator: no-op. */
int32_t cdh_comm_drop(cdh_handle h);
/* Optional direct exchange for the short per-block records (<= 2688 doubles) of a row-sharded
 * sweep: every rank stores its record into an IPC-mapped inbox on every peer and sums the
 * sources in rank order (one hop over the xGMI mesh, no ring).  OFF unless connected AND
 * enabled; longer vectors keep going through RCCL when a communicator exists.
 *   1. every rank: cdh_p2p_local_handle -> 64 opaque bytes (a HIP IPC memory handle);
 *   2. all-gather the handles in rank order (any transport), every rank: cdh_p2p_connect;
 *   3. every rank: cdh_p2p_enable(h, 1)  (collectively: all ranks or none).
 * A peer that never arrives ends the wait after a bounded spin; that call and every later
 * exchange on the handle return CDH_RCCL_ERROR (the ranks no longer agree on what was exchanged).  cdh_exchange_probe all-reduces `count` (<= 4096) host
 * doubles in place through whatever exchange is active (self-test of the transport). */
int32_t cdh_p2p_local_handle(cdh_handle h, void *out_64_bytes);
int32_t cdh_p2p_connect(cdh_handle h, const void *handles_64_bytes_each, int32_t rank, int32_t nranks);
int32_t cdh_p2p_enable(cdh_handle h, int32_t on);
int32_t cdh_exchange_probe(cdh_handle h, double *inout, int64_t count);
/* Bring-your-own transport for the row-shard exchange (MPI.jl from the Julia side, gloo in this
 * repository's tests): `fn` must sum `count` doubles in place over all ranks and return 0.  The library
 * stages the record through pinned host memory and calls it on the caller's thread in the middle of
 * the sweep, between the reduce and the scalar-update kernels -- the same seam the RCCL all-reduce and
 * the direct exchange sit behind.  Slower than both (a stream round trip per exchange); fn == NULL
 * removes it. */
typedef int32_t (*cdh_host_allreduce_fn)(void *user, double *inout, int64_t count);
int32_t cdh_set_host_exchange(cdh_handle h, cdh_host_allreduce_fn fn, void *user, int32_t rank,
                              int32_t nranks);
/* How many all-reduces this handle has issued through each exchange since it was created, and the
 * rank count the active exchange reports (ncclCommCount for RCCL; 1 when the handle is not sharded).
 * Any out pointer may be NULL. */
int32_t cdh_exchange_stats(cdh_handle h, int64_t *out_rccl_calls, int64_t *out_p2p_calls,
                           int64_t *out_host_calls, int32_t *out_nranks);
/* Average time (microseconds, HIP events on the handle's stream) of `iters` back-to-back all-reduces
 * of `count` (<= 4096) doubles through the active exchange; 0-ish when the handle is not sharded.
 * Collective: every rank calls it with the same arguments. */
int32_t cdh_exchange_latency(cdh_handle h, int64_t count, int32_t iters, double *out_us);
/* HIP-event timing of the sweep kernels on the handle's stream: everything
 * launched by cdh_pass / cdh_solve between begin and end.  out_launches counts
 * the dominant (column-streaming) kernel launches, out_ms the event time they
 * span in total. */
int32_t cdh_profile_begin(cdh_handle h);
int32_t cdh_profile_end(cdh_handle h, double *out_ms, int64_t *out_launches,
                        double *out_algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif /* CDHIP_H */
