// cov_solve.hpp -- a cache-served solve with ONE host round trip: the pass loop of _coordinateDescent!
// (coordinate_descent.jl:65-92; the loop of LassoPath around it, lasso.jl:250-252, calls it once per lambda) on the device
// while the gradient cache (grad_cache.hpp) can serve its passes.  A section of cdhip.hip kept in its own file (included
// once, inside cdhip.hip's anonymous namespace, after grad_cache.hpp).
//
// Round 3 ran such a solve pass by pass from the host: per full pass a scan, a copy and a synchronisation to learn how
// many visits there are, the blocks, a second copy and synchronisation, and a host loop over all m positions replaying the
// reference's SparseIterate bookkeeping; per active pass one round trip -- ~45 of cfg3's 100 ms (476 passes, p = 5000) were
// host latency.  Here ONE workgroup keeps the whole state machine on the device: pass after pass, full or active, ordered
// or shuffled, the visits in covariance form (k_cov_block's gather + gram_scalar_body, the same arithmetic in the same order
// as the streamed blocked sweep), ProximalBase's SparseIterate replayed in device memory (slots appended in visit order by
// block-wide ranked appends, dropzeros!' swap-with-last as a match of holes and fillers: small_solve.hpp's scheme, block-wide),
// the convergence test, the scheduler's generator.  The host hears of it when the solve is over -- or when the kernel needs
// something only the host can give: Gram columns for coordinates about to enter (a pass over X), a re-reference, the careful
// walk after a certificate broke (the pass is undone first).
//
// What a full pass costs.  g lives in d_g for all p coordinates, but only the coordinates a pass really visits (the support
// and the few inactive ones near their threshold) need it exactly at every step.  So d_g is kept as it stood at the last
// FOLD and the moves since then stay pending (moved[], beta - bfold): the visited ("tracked") coordinates get their exact
// gradient at the start of every pass, gx_k = g_k - sum_m pend_m G_mk, and keep it current visit by visit; everybody else is
// certified against a BOUND:   |g_k(t)| <= |g_k| + M_k TV(t),   M_k = max_j |G_jk| over the cached columns j != k,
// TV(t) = the total variation sum |h| of all moves since the fold up to time t (it only grows, and it is recorded per visit).
// A coordinate is skipped as settled when the bound at the start of the pass is under its certificate, and re-checked after
// the pass with the bound AT ITS TURN (TV and, for the sqrt-lasso, r'r as they stood when its turn came); one whose bound no
// longer holds gets its exact gradient at its turn (a gather along the moved coordinates' columns), and only if THAT breaks
// the certificate is the pass undone and handed to the host's careful walk.  On a Gaussian design M_k TV is ~1e-3 of a
// threshold: a full pass reads p numbers, not p x (moves) Gram entries -- the round-3 device pass moved 4 MB through the CUs
// per full pass of cfg3 for the g update alone.  When the bound has grown loose (many inactive coordinates fail it) the
// kernel folds: g -= sum_m pend_m G_m over all p, TV = 0.  Same iterates, support order and pass counts as visiting every
// coordinate (tests: the legs of tests/_legs.py, the stateful fuzz, tests/test_gpu_cov_solve.py).
#pragma once

constexpr int kCsThreads = 512;          // 8 waves: gram_scalar_body<4> holds a 64-entry Gram column per lane (128 VGPRs)
constexpr int kCsWaves = kCsThreads / 64;
constexpr int kCsNearMax = 96;           // inactive coordinates failing the bound beyond which the kernel folds and scans again
constexpr int64_t kCsShuffleMaxP = 12288;   // the shuffle's two p-sized int arrays must fit LDS

enum { kCsConverged = 0, kCsMaxIter = 1, kCsNeedColumns = 2, kCsRollback = 3, kCsBusy = 4, kCsRefresh = 5, kCsOutgrown = 6 };

struct CovSolveCtl {
    // in
    double lambda0, n_total, optTol, cert_abs;
    int64_t max_passes;          // passes this launch may run (maxIter - those already done)
    int64_t cov_budget;          // covariance-form visits this launch may make before g is due to be re-read from X
    int32_t loss, has_omega, randomize, nnz_limit /* support size beyond which the cache stands aside */;
    int32_t busy_limit, inject_every, pad0, pad1;
    // in / out
    uint64_t rng;
    double q;                    // r'r (sqrt-lasso)
    int32_t nnz, prev_conv, conv, inject_count;
    // out
    int32_t status, n_list /* kCsNeedColumns / kCsBusy: coordinates that want a Gram column (out_list) */, n_moved, domain_error;
    int64_t passes, full_passes, visits, cov_visits, settled, folds, exact_rechecks;
    double maxH;
};

struct CovSolveBufs {
    int64_t p;
    double* g; const double* Gcols; const int32_t* slot; const double* a; const double* colmax; const double* omega;
    double* beta;
    double *gx, *bfold, *bsnap, *hs, *newval, *qs, *tv, *pendv;
    int64_t *uk, *poff, *voff;
    int32_t *touched, *s2i, *i2s, *list, *vb, *moved, *holes, *fills;
    uint8_t *setflag, *inmoved;
    const int32_t* in_sup;                   // the support in slot order (pinned host memory, read once)
    int32_t *out_sup_idx, *out_moved_idx, *out_list;   // pinned host memory, written once at the end
    double *out_sup_val, *out_moved_val;
};

// exclusive rank of `flag` among the block's threads (thread order), and the block's total; s_w: kCsWaves ints of LDS
__device__ __forceinline__ int cs_block_rank(bool flag, int& total, int* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(flag);
    if (lane == 0) s_w[wave] = __popcll(mask);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kCsWaves; ++w) { const int c = s_w[w]; tot += c; if (w < wave) before += c; }
    __syncthreads();
    total = tot;
    return before + __popcll(mask & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ int cs_block_count(bool flag, int* s_w) { int t; (void)cs_block_rank(flag, t, s_w); return t; }

__global__ __launch_bounds__(kCsThreads) void k_cov_solve(CovSolveCtl* ctl, CovSolveBufs b) {
    using R = GramRec<4>;
    constexpr int B = R::B;
    extern __shared__ int32_t s_shuffle[];       // randomize: order[p], draw[p]
    __shared__ double s_rec[R::N];
    __shared__ int64_t s_k[B], s_off[B], s_moff[B];
    __shared__ double s_h[B];
    __shared__ Ctrl s_ctrl;
    __shared__ int s_w[kCsWaves];
    __shared__ int s_nmove, s_bad, s_nan;
    __shared__ double s_tvrun;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t p = b.p;
    const int loss = ctl->loss, has_omega = ctl->has_omega, randomize = ctl->randomize;
    const bool sqrt_loss = loss == 1;
    const double lambda0 = ctl->lambda0, n_total = ctl->n_total, optTol = ctl->optTol, cert_abs = ctl->cert_abs;
    const int64_t max_passes = ctl->max_passes, cov_budget = ctl->cov_budget;
    const int nnz_limit = ctl->nnz_limit, busy_limit = ctl->busy_limit, inject_every = ctl->inject_every;
    uint64_t rng = ctl->rng;
    double q = ctl->q;
    int nnz = ctl->nnz, inject_count = ctl->inject_count;
    bool prev_conv = ctl->prev_conv != 0, conv = ctl->conv != 0;
    int32_t* s_order = s_shuffle;
    int32_t* s_draw = s_shuffle + p;

    for (int64_t k = tid; k < p; k += kCsThreads) { b.i2s[k] = 0; b.bfold[k] = b.beta[k]; b.inmoved[k] = 0; }
    if (tid == 0) {
        s_ctrl.lambda0 = lambda0; s_ctrl.n_total = n_total; s_ctrl.maxH = 0.0; s_ctrl.loss = loss; s_ctrl.has_omega = has_omega;
        s_ctrl.domain_error = 0; s_ctrl.pad = 0; s_ctrl.q_carry = q; s_ctrl.cert_abs = cert_abs;
    }
    __syncthreads();
    for (int s = tid; s < nnz; s += kCsThreads) { const int k = b.in_sup[s]; b.s2i[s] = k; b.i2s[k] = s + 1; }
    __syncthreads();

    int nmoved = 0, status = kCsMaxIter, n_list = 0, dom_any = 0;
    double TV0 = 0.0, lastH = 0.0;
    int64_t passes = 0, full_passes = 0, visits = 0, cov_visits = 0, settled_total = 0, folds = 0, exact_rechecks = 0;

    // g <- g - sum_m pend_m G_m over all p; nothing is pending afterwards
    auto fold = [&]() {
        for (int m = tid; m < nmoved; m += kCsThreads) {
            const int km = b.moved[m];
            b.pendv[m] = b.beta[km] - b.bfold[km];
            b.poff[m] = (int64_t)b.slot[km] * p;
        }
        __syncthreads();
        for (int64_t k = tid; k < p; k += kCsThreads) {
            double acc = b.g[k];
            for (int m = 0; m < nmoved; ++m) acc = fma(-b.pendv[m], b.Gcols[b.poff[m] + k], acc);
            b.g[k] = acc;
        }
        for (int m = tid; m < nmoved; m += kCsThreads) { const int km = b.moved[m]; b.bfold[km] = b.beta[km]; b.inmoved[km] = 0; }
        __syncthreads();
        nmoved = 0; TV0 = 0.0; folds += 1;
    };

    for (;;) {
        if (passes >= max_passes) { status = kCsMaxIter; break; }
        if (nnz > nnz_limit) { status = kCsOutgrown; break; }
        if (cov_visits > cov_budget) { status = kCsRefresh; break; }
        const bool full = conv;
        const uint64_t rng_before = rng;
        const int L = full ? (int)p : nnz;
        // ---- reset!(it, full) + collect(it) (atom_iterator.jl:34-37, 53-64; the splitmix64 substitute of sparse_iterate.hpp) ----
        if (randomize) {
            for (int i = tid; i < L; i += kCsThreads) {
                s_order[i] = i;
                if (i + 1 < L) {
                    uint64_t st = rng + (uint64_t)i * 0x9E3779B97F4A7C15ull;
                    s_draw[i] = i + (int)mod64_small(small_rng_next(st), (uint32_t)(L - i));
                }
            }
            if (L > 1) rng += (uint64_t)(L - 1) * 0x9E3779B97F4A7C15ull;
            __syncthreads();
            if (tid == 0)
                for (int i = 0; i + 1 < L; ++i) { const int j = s_draw[i]; const int t = s_order[i]; s_order[i] = s_order[j]; s_order[j] = t; }
            __syncthreads();
            for (int i = tid; i < L; i += kCsThreads) b.list[i] = full ? s_order[i] : b.s2i[s_order[i]];
        } else {
            for (int i = tid; i < L; i += kCsThreads) b.list[i] = full ? i : b.s2i[i];
        }
        __syncthreads();

        // ---- the scan: which positions are visited (cdh: "unsettled"), in visit order ----
        int cnt = 0, nocol = 0, nzero = 0, nsupp = 0;
        for (int attempt = 0;; ++attempt) {
            cnt = 0; nocol = 0; nzero = 0; nsupp = 0;
            const double thr_base = lambda0 * (sqrt_loss ? sqrt(q) : n_total);
            for (int i0 = 0; i0 < L; i0 += kCsThreads) {
                const int i = i0 + tid;
                const bool valid = i < L;
                const int k = valid ? b.list[i] : 0;
                const double gk = b.g[k], bk = b.beta[k];
                bool st = false;
                if (valid && full && bk == 0.0) {
                    const double ak = b.a[k], om = has_omega ? b.omega[k] : 1.0;
                    st = ak > 0.0 && fabs(gk) + b.colmax[k] * TV0 <= thr_base * om * (1.0 - 1e-9) - cert_abs * sqrt(ak);
                }
                const bool uns = valid && !st;
                const bool nc = uns && b.slot[k] < 0;
                int tot_u, tot_c;
                const int j = cnt + cs_block_rank(uns, tot_u, s_w);
                const int jl = nocol + cs_block_rank(nc, tot_c, s_w);
                if (valid) {
                    b.vb[k] = j;                      // visited: its index in the visit list; settled: visits before its turn
                    if (full) b.setflag[k] = st ? 1 : 0;
                    if (uns) b.uk[j] = k;
                    if (nc) b.out_list[jl] = k;
                }
                nzero += cs_block_count(st && gk == 0.0, s_w);
                nsupp += cs_block_count(uns && bk != 0.0, s_w);
                cnt += tot_u; nocol += tot_c;
            }
            // many inactive coordinates fail the bound only because it has grown loose: fold and look again
            if (full && attempt == 0 && nmoved > 0 && cnt - nsupp > kCsNearMax) { fold(); continue; }
            break;
        }
        if (nocol > 0) {                 // coordinates about to be visited without a Gram column: the host fetches them
            status = nocol > busy_limit ? kCsBusy : kCsNeedColumns; n_list = nocol; rng = rng_before;
            break;
        }

        // ---- the exact gradient of the visited coordinates: gx = g - sum_m pend_m G_m; beta as it stands (for an undo) ----
        for (int m = tid; m < nmoved; m += kCsThreads) {
            const int km = b.moved[m];
            b.pendv[m] = b.beta[km] - b.bfold[km];
            b.poff[m] = (int64_t)b.slot[km] * p;
        }
        __syncthreads();
        for (int u = tid; u < cnt; u += kCsThreads) {
            const int64_t k = b.uk[u];
            double acc = b.g[k];
            for (int m = 0; m < nmoved; ++m) acc = fma(-b.pendv[m], b.Gcols[b.poff[m] + k], acc);
            b.gx[k] = acc;
            b.bsnap[u] = b.beta[k];
            b.voff[u] = (int64_t)b.slot[k] * p;
        }
        if (tid == 0) { s_ctrl.maxH = 0.0; s_ctrl.domain_error = 0; s_ctrl.q_carry = q; s_bad = 0; s_nan = 0; s_tvrun = 0.0; }
        const double q_start = q;
        __syncthreads();

        // ---- _cdPass! over the visit list, 64 visits at a time: k_cov_block's gather, gram_scalar_body's B sequential updates ----
        for (int j0 = 0; j0 < cnt; j0 += B) {
            const int nb = min(B, cnt - j0);
            if (tid < B) { const int i = tid < nb ? tid : 0; s_k[tid] = b.uk[j0 + i]; s_off[tid] = b.voff[j0 + i]; }
            __syncthreads();
            for (int e = tid; e < B * B; e += kCsThreads) {
                const int sI = e / B, j = e % B;
                if (sI <= j && j < nb) s_rec[R::g(sI, j)] = b.Gcols[s_off[j] + s_k[sI]];
            }
            if (tid < B) s_rec[R::OFF_C + tid] = (tid < nb) ? b.gx[s_k[tid]] : 0.0;
            if (tid == 0) s_rec[R::OFF_Q] = s_ctrl.q_carry;
            __syncthreads();
            if (tid < 64) {
                gram_scalar_body<4>(s_rec, nb, 0, &s_ctrl, b.beta, b.omega, b.uk, b.hs, b.newval, b.touched, j0, b.qs, tid);
                // the block's moves, compacted (k_cov_gupdate's prologue), and the total variation after each visit
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const double hv = tid < nb ? b.hs[j0 + tid] : 0.0;
                const bool nz = hv != 0.0;                       // (a NaN h counts as a move: it propagates, as in the reference)
                const unsigned long long mask = __ballot(nz);
                if (nz) { const int at = __popcll(mask & ((1ull << tid) - 1ull)); s_h[at] = hv; s_moff[at] = s_off[tid]; }
                if (__ballot(hv != hv)) { if (tid == 0) s_nan = 1; }
                double run = (hv == hv) ? fabs(hv) : 0.0;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(run, off, 64); if (tid >= off) run += o; }
                const double base = s_tvrun;
                if (tid < nb) b.tv[j0 + tid] = base + run;
                if (tid == 0) s_nmove = __popcll(mask);
                const double last = __shfl(run, 63, 64);
                if (tid == 0) s_tvrun = base + last;
            }
            __syncthreads();
            const int nmove = s_nmove;
            if (nmove > 0)                                       // the visits still to come in this pass see the block's moves
                for (int u = j0 + nb + tid; u < cnt; u += kCsThreads) {
                    const int64_t k = b.uk[u];
                    double acc = b.gx[k];
                    for (int i = 0; i < nmove; ++i) acc = fma(-s_h[i], b.Gcols[s_moff[i] + k], acc);
                    b.gx[k] = acc;
                }
            __syncthreads();
        }
        const double tv_pass = s_tvrun, q_end = s_ctrl.q_carry;
        if (s_ctrl.domain_error) dom_any = 1;

        // ---- the settled positions, re-checked with the bound as it stood at their turn; exactly where the bound fails ----
        if (full && cnt < L && (tv_pass > 0.0 || TV0 > 0.0)) {
            int nexact = 0;
            for (int64_t k = tid; k < p; k += kCsThreads) {
                if (!b.setflag[k]) continue;
                const int vbk = b.vb[k];
                const double tvk = TV0 + (vbk > 0 ? b.tv[vbk - 1] : 0.0);
                const double qk = sqrt_loss ? (vbk > 0 ? b.qs[vbk - 1] : q_start) : 0.0;
                const double ak = b.a[k], om = has_omega ? b.omega[k] : 1.0, gk = b.g[k];
                const double cert = lambda0 * (sqrt_loss ? sqrt(qk) : n_total) * om * (1.0 - 1e-9) - cert_abs * sqrt(ak);
                if (fabs(gk) + b.colmax[k] * tvk <= cert) continue;
                double acc = gk;                                  // the exact gradient when its turn came
                for (int m = 0; m < nmoved; ++m) acc = fma(-b.pendv[m], b.Gcols[b.poff[m] + k], acc);
                for (int i = 0; i < vbk; ++i) { const double hv = b.hs[i]; if (hv != 0.0) acc = fma(-hv, b.Gcols[b.voff[i] + k], acc); }
                nexact += 1;
                if (!(fabs(acc) <= cert)) s_bad = 1;
            }
            exact_rechecks += nexact;                             // (this thread's; summed below)
        }
        __syncthreads();
        bool undo = s_bad != 0 || s_nan != 0 || (nzero > 0 && (TV0 > 0.0 || tv_pass > 0.0));
        if (full && cnt > 0 && inject_every > 0) { inject_count += 1; if (inject_count % inject_every == 0) undo = true; }
        if (undo) {                       // the pass never happened: the host walks it the careful way
            for (int u = tid; u < cnt; u += kCsThreads) b.beta[b.uk[u]] = b.bsnap[u];
            status = kCsRollback; rng = rng_before;
            __syncthreads();
            break;
        }

        // ---- accepted: what has moved since the fold, in visit order ----
        for (int u0 = 0; u0 < cnt; u0 += kCsThreads) {
            const int u = u0 + tid;
            const int64_t k = u < cnt ? b.uk[u] : 0;
            const bool neu = u < cnt && b.hs[u] != 0.0 && !b.inmoved[k];
            int tot;
            const int at = nmoved + cs_block_rank(neu, tot, s_w);
            if (neu) { b.moved[at] = (int32_t)k; b.inmoved[k] = 1; }
            nmoved += tot;
        }
        TV0 += tv_pass; q = q_end;

        // ---- ProximalBase's SparseIterate, in visit order: a pre-prox non-zero appends a slot (x[k] += b/a), cdprox! stores
        // the value; a settled visit of the least-squares losses with g_k != 0 leaves a zero in a slot of its own ----
        for (int i0 = 0; i0 < L; i0 += kCsThreads) {
            const int i = i0 + tid;
            const bool valid = i < L;
            const int k = valid ? b.list[i] : 0;
            bool app = false;
            if (valid && b.i2s[k] == 0) {
                if (full && b.setflag[k]) app = !sqrt_loss && b.g[k] != 0.0;
                else { const int j = b.vb[k]; app = b.touched[j] != 0 || b.newval[j] != 0.0; }
            }
            int tot;
            const int sl = nnz + cs_block_rank(app, tot, s_w);
            if (app) { b.s2i[sl] = k; b.i2s[k] = sl + 1; }
            nnz += tot;
        }
        __syncthreads();
        // ---- dropzeros!: swap-with-last (sparse_iterate.hpp) as a match of holes and fillers (small_solve.hpp) ----
        {
            int m = 0;
            for (int s0 = 0; s0 < nnz; s0 += kCsThreads) { const int s = s0 + tid; m += cs_block_count(s < nnz && b.beta[b.s2i[s < nnz ? s : 0]] != 0.0, s_w); }
            if (m != nnz) {
                int nh = 0, nf = 0;
                for (int s0 = 0; s0 < m; s0 += kCsThreads) {
                    const int s = s0 + tid;
                    const bool hole = s < m && b.beta[b.s2i[s < m ? s : 0]] == 0.0;
                    int tot;
                    const int at = nh + cs_block_rank(hole, tot, s_w);
                    if (hole) b.holes[at] = s;
                    nh += tot;
                }
                for (int s1 = nnz; s1 > m; s1 -= kCsThreads) {          // chunks from the end; inside a chunk thread 0 is the last slot
                    const int s = s1 - 1 - tid;
                    const bool fil = s >= m && b.beta[b.s2i[s >= m ? s : m]] != 0.0;
                    int tot;
                    const int at = nf + cs_block_rank(fil, tot, s_w);
                    if (fil) b.fills[at] = s;
                    nf += tot;
                }
                __syncthreads();
                for (int s = tid; s < nnz; s += kCsThreads) { const int ks = b.s2i[s]; if (b.beta[ks] == 0.0) b.i2s[ks] = 0; }
                __syncthreads();
                for (int r = tid; r < nh; r += kCsThreads) { const int kf = b.s2i[b.fills[r]]; b.s2i[b.holes[r]] = kf; b.i2s[kf] = b.holes[r] + 1; }
                __syncthreads();
                nnz = m;
            }
        }
        passes += 1; visits += L; cov_visits += cnt; settled_total += L - cnt; lastH = s_ctrl.maxH;
        if (full) full_passes += 1;
        prev_conv = conv;
        conv = lastH < optTol;
        if (prev_conv && conv) { status = kCsConverged; break; }
    }

    // ---- what the host needs: the support in slot order with its values, the moves still pending on g ----
    for (int s = tid; s < nnz; s += kCsThreads) { const int k = b.s2i[s]; b.out_sup_idx[s] = k; b.out_sup_val[s] = b.beta[k]; }
    for (int m = tid; m < nmoved; m += kCsThreads) { const int km = b.moved[m]; b.out_moved_idx[m] = km; b.out_moved_val[m] = b.beta[km] - b.bfold[km]; }
    {   // exact re-checks were counted per thread
        __shared__ unsigned long long s_sum;
        if (tid == 0) s_sum = 0ull;
        __syncthreads();
        if (exact_rechecks) atomicAdd(&s_sum, (unsigned long long)exact_rechecks);
        __syncthreads();
        exact_rechecks = (int64_t)s_sum;
    }
    if (tid == 0) {
        ctl->rng = rng; ctl->q = q; ctl->nnz = nnz; ctl->prev_conv = prev_conv ? 1 : 0; ctl->conv = conv ? 1 : 0;
        ctl->inject_count = inject_count; ctl->status = status; ctl->n_list = n_list; ctl->n_moved = nmoved;
        ctl->domain_error = dom_any; ctl->passes = passes; ctl->full_passes = full_passes; ctl->visits = visits;
        ctl->cov_visits = cov_visits; ctl->settled = settled_total; ctl->folds = folds; ctl->exact_rechecks = exact_rechecks;
        ctl->maxH = lastH;
    }
}

// M_k = max(M_k, |G_jk|) over one freshly cached column j (k != j): the bound of k_cov_solve's certificates
__global__ __launch_bounds__(256) void k_cov_colmax(double* __restrict__ colmax, const double* __restrict__ col, int64_t j, int64_t p) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p || k == j) return;
    const double v = fabs(col[k]);
    if (v > colmax[k] || v != v) colmax[k] = v;       // (a NaN entry poisons the bound: nothing is certified against it)
}
