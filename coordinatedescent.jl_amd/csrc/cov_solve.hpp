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
// FOLD and the moves since then stay pending (moved[], beta - bfold; those pending when the host launches travel in with
// the launch): the visited ("tracked") coordinates carry their exact gradient -- computed once, gx_k = g_k - sum_m pend_m G_mk,
// then kept current visit by visit and from pass to pass -- together with beta, omega and the packed upper triangle of their
// Gram block in LDS, so the visits themselves touch no global memory; everybody else is certified against a BOUND:
//     |g_k(t)| <= |g_k| + M_k TV(t),   M_k = max_j |G_jk| over the cached columns j != k,
// TV(t) = the total variation sum |h| of all moves since the fold up to time t (it only grows, and it is recorded per visit).
// A coordinate is skipped as settled when the bound at the start of the pass is under its certificate -- never one that has
// itself moved since the fold: M_k leaves out G_kk -- and re-checked after the pass with the bound AT ITS TURN (TV and, for
// the sqrt-lasso, r'r as they stood when its turn came); one whose bound no longer holds gets its exact gradient at its turn
// (a gather along the moved coordinates' columns), and only if THAT breaks the certificate is the pass undone (beta itself
// is written back only when a pass is accepted) and handed to the host's careful walk.  On a Gaussian design M_k TV is ~1e-3
// of a threshold: a full pass reads p numbers, not p x (moves) Gram entries -- the round-3 device pass moved 4 MB through the
// CUs per full pass of cfg3 for the g update alone.  When the bound has grown loose (many inactive coordinates fail it, or
// one without a Gram column does) the kernel folds: g -= sum_m pend_m G_m over all p, TV = 0.
// ProximalBase's bookkeeping without its p transient slots: the final slot order of a pass is assembled from the old slots
// and the visited appenders only (see "analytic" below); the straightforward replay stays for the rare exceptions.
// Visit lists beyond the LDS-sized Gram block (~150 non-zeros; benchmark/cd_bench.jl's path ends at 774): one CU cannot gather
// p x moves Gram entries per block of visits (first attempt: 0.38 s against 0.11 s for the pass-by-pass kernels), so
//   * ACTIVE passes run from a Gram TABLE in device memory: Gc[c][d] = X_k(c)' X_k(d) by table id for every coordinate the loop
//     has visited (filled from the cached columns as coordinates enter, kept from solve to solve, void on a new X) and gxc[c],
//     the exact gradient of every coordinate the table holds -- a block's moves update gxc through 64 contiguous rows of Gc;
//   * FULL passes run with HELPER workgroups of the same launch ("the crew", below): they keep g = X'r itself current for all p
//     coordinates and re-check every skipped coordinate at its turn while workgroup 0 visits (a launch without helpers leaves
//     those passes to the host's device pass: kCsHostFull).
// Both share one pipeline with two blocks of 64 visits in flight (see "Visit lists beyond the LDS block" in the kernel).
// A full pass in which coordinates crossed their threshold through the pass's own moves runs again with those coordinates
// visited (forced[], kCsForcedRounds) before anything is handed to the host.
// Why it looks the way it does (measured, LAB_NOTES.md round 4): on one CU every dependent global access is 0.2-0.5 us, so
// the p-sized loops issue all their loads unconditionally (clamped indices, no short-circuit conditions: a conditional load
// is compiled into a load that is waited for on its own) and rank several flags per pair of barriers.
// Same iterates, support order and pass counts as visiting every coordinate (tests: the legs of tests/_legs.py, the stateful
// fuzz, tests/test_gpu_cov_solve.py).
#pragma once

constexpr int kCsThreads = 256;          // 4 waves, one per SIMD: gram_scalar_body<4> holds a 64-entry Gram column per lane (128 VGPRs) next to the loops' state
constexpr int kCsWaves = kCsThreads / 64;
constexpr int kCsNearMax = 96;           // inactive coordinates failing the bound beyond which the kernel folds and scans again
constexpr int64_t kCsShuffleMaxP = 5600;    // the shuffle's six (p + 1)-sized int arrays must fit the kernel's dynamic LDS
constexpr size_t kCsLdsBudget = (size_t)134 * 1024;   // dynamic LDS next to ~24 KB of static arrays (160 KB per CU)
constexpr int kCsTrackedMargin = 24;     // room in the tracked list for entering and near-threshold coordinates next to the support
constexpr int kCsTableCap = 1536;        // coordinates the Gram table of large visit lists holds (1536^2 doubles = 18.9 MB of device memory)
constexpr int kCsTableMargin = 160;      // ... of which this many are left to entering and near-threshold coordinates next to the support
constexpr size_t kCsTableLds = 2688 + 64 * 64 + 640;    // doubles of dynamic LDS table mode and crew passes use (a GramRec<4> record rounded up, a tile, a block's moves)
constexpr int kCsForcedRounds = 4;       // a full pass whose re-check found coordinates crossing their threshold is run again this often with those visited
constexpr int kCsUcapMax = 176;          // tracked coordinates whose Gram block is kept in LDS: symmetric, upper triangle packed (176 x 177 / 2 doubles = 122 KB)
__host__ __device__ constexpr size_t cs_tri_doubles(size_t u) { return u * (u + 1) / 2; }
// index of G_UU[i][j], i <= j, in the packed upper triangle of a cnt x cnt block (row i starts after rows 0 .. i - 1)
__device__ __forceinline__ int cs_tri(int i, int j, int cnt) { return i * cnt - (i * (i - 1)) / 2 + (j - i); }

// Ranks inside a block, E positions per thread (thread t owns positions base + E t .. base + E t + E - 1: rank order is
// position order).  rank[e] = exclusive rank of flag[e]; returns the block's total.  s_w: 2 * kCsWaves ints of LDS.
template <int E>
__device__ __forceinline__ int cs_rank(const bool (&f)[E], int (&rank)[E], int* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int mine = 0, wtot = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) { const unsigned long long m = __ballot(f[e]); mine += __popcll(m & below); wtot += __popcll(m); }
    if (lane == 0) s_w[wave] = wtot;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kCsWaves; ++w) { const int c = s_w[w]; tot += c; if (w < wave) before += c; }
    __syncthreads();
    int run = before + mine;
#pragma unroll
    for (int e = 0; e < E; ++e) { rank[e] = run; run += f[e] ? 1 : 0; }
    return tot;
}
// two flag sets ranked at once (one pair of barriers)
template <int E>
__device__ __forceinline__ void cs_rank2(const bool (&f)[E], const bool (&g)[E], int (&rf)[E], int (&rg)[E], int& totf, int& totg, int* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int mf = 0, wf = 0, mg = 0, wg = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned long long a = __ballot(f[e]), c = __ballot(g[e]);
        mf += __popcll(a & below); wf += __popcll(a); mg += __popcll(c & below); wg += __popcll(c);
    }
    if (lane == 0) { s_w[wave] = wf; s_w[kCsWaves + wave] = wg; }
    __syncthreads();
    int bf = 0, tf = 0, bg = 0, tg = 0;
#pragma unroll
    for (int w = 0; w < kCsWaves; ++w) {
        const int c = s_w[w], d = s_w[kCsWaves + w];
        tf += c; tg += d;
        if (w < wave) { bf += c; bg += d; }
    }
    __syncthreads();
    int runf = bf + mf, rung = bg + mg;
#pragma unroll
    for (int e = 0; e < E; ++e) { rf[e] = runf; runf += f[e] ? 1 : 0; rg[e] = rung; rung += g[e] ? 1 : 0; }
    totf = tf; totg = tg;
}
// block-wide sum of a per-thread count
__device__ __forceinline__ int cs_block_sum(int v, int* s_w) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int w = 0; w < kCsWaves; ++w) tot += s_w[w];
    __syncthreads();
    return tot;
}


// ---- the crew ----------------------------------------------------------------------------------------------------------
// Beyond the LDS-sized Gram block the work per block of 64 visits that is NOT the visits themselves -- g -= sum_i h_i G_i over
// all p coordinates, and in a full pass the re-check of every skipped coordinate at its turn (k_cov_gupdate_chk's job in the
// host's device pass) -- is p x 64 gathers: microseconds for thirty workgroups, ~1 ms for one.  So a launch that expects such
// lists brings helpers: workgroups 1 .. of the same grid.  Workgroup 0 runs the state machine as before and POSTS one job per
// block (the block's moves: h, column offsets, visit indices, r'r after each); the helpers take the jobs in order, each for
// its slice of the coordinates, and report per helper how many they have finished.  Workgroup 0 does not wait for them inside a
// pass at all: its own visits run from the Gram table (gxc: the exact gradients of the coordinates it visits, kept current by the
// table's rows), the helpers' g is read again only when the next pass scans, and their verdict on the skipped coordinates when
// the pass is over.  (A first version had workgroup 0 read the visited coordinates' gradients from g itself, one job behind:
// 19 us per block of visits against 12 -- an agent-scope acquire per block empties the XCD's L2 under the visits.)
// Memory model: job data and g are ordinary device memory; a post is release (agent scope) after the data, a take is acquire;
// a helper's report is release after its stores, workgroup 0's wait acquire.  Every spin is bounded by the wall clock (100 MHz
// counter): a helper that hears nothing for kCsCrewPatience leaves, workgroup 0 gives up a wait after the same time and the
// solve fails loudly (kCsCrewLost) -- a hang is not possible; all helpers are resident (one workgroup per CU, far fewer than CUs).
constexpr uint64_t kCsCrewPatience = 2000000000ull;      // 20 s of 10 ns ticks

__device__ __forceinline__ uint32_t cs_ld_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t cs_ld_relaxed(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cs_st_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

struct CsCrewView {      // what a helper touches (passed by value: a handful of registers)
    CsCrew* crew; int64_t p; double* g; double* g_snap; const double* Gcols; const int32_t* vb;
    const uint8_t* setflag; const double* omega; const double* a; uint8_t* forced; const double* pendv; const int64_t* poff;
};
__device__ __forceinline__ void cs_crew_helper(const CovSolveCtl* ctl, const CsCrewView b) {
    __shared__ double s_h[64], s_q[64];
    __shared__ int64_t s_off[64];
    __shared__ int s_pos[64];
    __shared__ int s_go;
    const int tid = threadIdx.x, hid = (int)blockIdx.x - 1, nh = (int)gridDim.x - 1;
    CsCrew* cw = b.crew;
    const int64_t p = b.p;
    const int loss = ctl->loss, has_omega = ctl->has_omega;
    const double lambda0 = ctl->lambda0, n_total = ctl->n_total, cert_abs = ctl->cert_abs;
    uint32_t seen = 0;
    for (;;) {
        if (tid == 0) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            int go = 1;
            for (;;) {       // (polled relaxed, a few times per microsecond: the acquire is the fence below)
                if (cs_ld_relaxed(&cw->posted) > seen) break;
                if (cs_ld_relaxed(&cw->exit_flag) != 0) { if (cs_ld_relaxed(&cw->posted) > seen) break; go = 0; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > kCsCrewPatience) { go = 0; break; }
                __builtin_amdgcn_s_sleep(12);
            }
            s_go = go;
        }
        __syncthreads();
        const int go = s_go;
        if (!go) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const CsCrewJob* job = &cw->ring[seen % (uint32_t)kCsCrewRing];
        const int kind = job->kind;
        if (kind == kCrewSnapshot) {
            for (int64_t k = (int64_t)hid * kCsThreads + tid; k < p; k += (int64_t)nh * kCsThreads) b.g_snap[k] = b.g[k];
        } else if (kind == kCrewRestore) {
            for (int64_t k = (int64_t)hid * kCsThreads + tid; k < p; k += (int64_t)nh * kCsThreads) b.g[k] = b.g_snap[k];
        } else if (kind == kCrewFold) {          // g -= sum_m pend_m G_m: the moves the table-mode passes have left pending, in their order
            const int nm = job->nmove;
            for (int64_t k = (int64_t)hid * kCsThreads + tid; k < p; k += (int64_t)nh * kCsThreads) {
                double acc = b.g[k];
                int m = 0;
                for (; m + 16 <= nm; m += 16) {
                    double gv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) gv[r] = b.Gcols[b.poff[m + r] + k];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc = fma(-b.pendv[m + r], gv[r], acc);
                }
                for (; m < nm; ++m) acc = fma(-b.pendv[m], b.Gcols[b.poff[m] + k], acc);
                b.g[k] = acc;
            }
        } else {
            const int nmove = job->nmove, j0 = job->j0, nb = job->nb, chk = job->chk, last_block = job->last_block;
            const double q_start = job->q_start;
            if (tid < 64) { s_h[tid] = job->h[tid]; s_q[tid] = job->q[tid]; s_off[tid] = job->off[tid]; s_pos[tid] = job->pos[tid]; }
            __syncthreads();
            for (int64_t k = (int64_t)hid * kCsThreads + tid; k < p; k += (int64_t)nh * kCsThreads) {
                const int t = b.vb[k];                            // (a skipped coordinate: the visits before its turn)
                bool need = chk != 0 && b.setflag[k] != 0 && t >= j0 && (last_block != 0 || t < j0 + nb);
                if (nmove == 0 && !need) continue;
                double acc = b.g[k], q_run = q_start;
                double cert_scale = 0.0, cert_off = 0.0;
                if (need) { cert_scale = lambda0 * (has_omega ? b.omega[k] : 1.0) * (1.0 - 1e-9); cert_off = cert_abs * sqrt(b.a[k]); }
                auto holds = [&](double gv, double qv) { return fabs(gv) <= cert_scale * (loss == 1 ? sqrt(qv) : n_total) - cert_off; };
                for (int i0 = 0; i0 < nmove; i0 += 32) {
                    double gv[32];
#pragma unroll
                    for (int r = 0; r < 32; ++r) gv[r] = b.Gcols[s_off[min(i0 + r, nmove - 1)] + k];
#pragma unroll
                    for (int r = 0; r < 32; ++r) {
                        const int i = i0 + r;
                        if (i < nmove) {
                            if (need && s_pos[i] >= t) {
                                if (!holds(acc, q_run)) { b.forced[k] = 1; cw->bad = 1; }
                                need = false;
                            }
                            acc = fma(-s_h[i], gv[r], acc);
                            q_run = s_q[i];
                        }
                    }
                }
                if (need && !holds(acc, q_run)) { b.forced[k] = 1; cw->bad = 1; }
                if (nmove) b.g[k] = acc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        seen += 1;
        if (tid == 0) cs_st_release(&cw->done[hid], seen);
    }
}

constexpr int kCsE = 4;      // positions per thread and iteration of the p-sized loops: their loads are issued together,
                             // so those loops are written WITHOUT short-circuit conditions (a conditional load is waited for on its own)

// What the visits of a pass work on, indexed by position u in the pass's visit list ("tracked" coordinates): in LDS while the
// list is at most `ucap` long, else in global scratch -- the same code through generic pointers.
struct CsTracked {
    int64_t *k, *voff, *iota;
    double *beta, *om, *gx, *hs, *nv, *qs, *tv;
    int32_t* tch;
};
constexpr size_t kCsTrackedBytes = 3 * 8 + 7 * 8 + 4;      // per tracked coordinate

// Two instantiations.  BIG = false is the loop as the common case needs it -- visit lists that fit the LDS block: no table, no helpers,
// no call, no scratch (a kernel that asks for scratch makes the runtime map it for every wave slot of the chip on its first launch:
// 14 ms measured, once per process; and the code of the large-list paths is a third of the kernel's 300 KB, all of it streamed through the
// instruction cache once per pass otherwise).  A list that outgrows the block ends such a launch (kCsNeedBig); the host comes back
// with BIG = true, and stays with it on that handle.
template <bool BIG>
__global__ __launch_bounds__(kCsThreads) void k_cov_solve(CovSolveCtl* ctl, CovSolveBufs b, int ucap /* tracked coordinates whose Gram block fits LDS */) {
    if constexpr (BIG) if (blockIdx.x != 0) { cs_crew_helper(ctl, CsCrewView{b.crew, b.p, b.g, b.g_snap, b.Gcols, b.vb, b.setflag, b.omega, b.a, b.forced, b.pendv, b.poff}); return; }        // the crew (above): no barrier of workgroup 0 is theirs
    using R = GramRec<4>;
    constexpr int B = R::B;
    constexpr int E = kCsE;
    extern __shared__ double s_dynamic[];        // [G_UU: upper triangle of ucap x ucap doubles, packed; a shuffle's scratch overlays it][tracked arrays: 84 ucap bytes]
    __shared__ double s_rec[R::N];
    __shared__ int s_mu[B];
    __shared__ int s_mu2[2][B], s_cid2[2][B];      // table mode: two blocks in flight
    __shared__ double s_h2[2][B];
    __shared__ double s_h[B];
    __shared__ Ctrl s_ctrl;
    __shared__ int s_w[2 * kCsWaves];
    __shared__ int s_nmove, s_bad, s_nan, s_same, s_nfail, s_task;
    __shared__ double s_tvrun;
    const int tid = threadIdx.x;
    const int64_t p = b.p;
    const int loss = ctl->loss, has_omega = ctl->has_omega, randomize = ctl->randomize;
    const bool sqrt_loss = loss == 1;
    const double lambda0 = ctl->lambda0, n_total = ctl->n_total, optTol = ctl->optTol, cert_abs = ctl->cert_abs;
    const int64_t max_passes = ctl->max_passes, cov_budget = ctl->cov_budget;
    const int nnz_limit = ctl->nnz_limit, busy_limit = ctl->busy_limit, inject_every = ctl->inject_every;
    uint64_t rng = ctl->rng;
    double q = ctl->q;
    const double q_floor = ctl->q_floor;
    // 100 MHz ticks per phase (thread 0's view): list, scan, exact gradients, visits, re-check, accept, bookkeeping, dropzeros!
    // (the phase clocks and the counters the host reads at the end live in LDS, kept by thread 0: a dozen 64-bit loop-carried values per
    // thread otherwise, which the register allocator of the large-list instantiation pays for in scratch)
    __shared__ unsigned long long s_tph[8];
    __shared__ long long s_cnt[12];
    enum { kFullPasses = 0, kVisits, kCovFull, kSettled, kFolds, kExact, kTablePasses, kTableRows, kForcedRounds, kCrewPasses, kCrewJobs };
    uint64_t tmark = __builtin_amdgcn_s_memrealtime();
    const uint64_t cyc0 = __builtin_amdgcn_s_memtime(), tick0 = tmark;
    auto lap = [&](int ph) { const uint64_t now = __builtin_amdgcn_s_memrealtime(); if (tid == 0) s_tph[ph] += now - tmark; tmark = now; };
    auto count = [&](int which, long long by) { if (tid == 0) s_cnt[which] += by; };
    int nnz = ctl->nnz, inject_count = ctl->inject_count;
    const int tcap = ctl->tcap, fold_limit = ctl->fold_limit, full_cap = ctl->full_cap;
    const int ucap_lists = ctl->ucap_limit > 0 ? min(ucap, ctl->ucap_limit) : ucap;      // the longest visit list that runs from the LDS block
    int ncid = ctl->ncid, tepoch = ctl->tepoch + 1;      // (gradients the table carried belong to the launch that carried them)
    const int nhelp = BIG ? (int)gridDim.x - 1 : 0;
    const bool crew = BIG && nhelp > 0;
    CsCrew* cw = b.crew;
    uint32_t njobs = 0, known_done = 0;
    __shared__ uint32_t s_known;
    __shared__ int s_crew_lost;          // a wait for the helpers ran into its bound (kept in LDS: set in one place, read after barriers)
    // every helper has finished `target` jobs (called by all threads; the spin is bounded by the wall clock)
    auto crew_wait = [&](uint32_t target) {
        if (target <= known_done) return;
        if (tid < 64) {
            uint32_t v = 0xffffffffu;
            if (tid < nhelp) {
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    v = cs_ld_relaxed(&cw->done[tid]);
                    if (v >= target || __builtin_amdgcn_s_memrealtime() - t0 > kCsCrewPatience) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = o < v ? o : v; }
            if (tid == 0) { s_known = v; if (v < target) s_crew_lost = 1; }
        }
        __syncthreads();
        known_done = s_known;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
    };
    int forced_rounds_here = 0;
    bool forced_dirty = false;
    bool prev_conv = ctl->prev_conv != 0, conv = ctl->conv != 0;
    double* s_G = s_dynamic;
    CsTracked lt;                                 // the LDS copy of the tracked arrays
    {
        double* d = s_dynamic + cs_tri_doubles((size_t)ucap);
        lt.k = reinterpret_cast<int64_t*>(d); d += ucap;
        lt.voff = reinterpret_cast<int64_t*>(d); d += ucap;
        lt.iota = reinterpret_cast<int64_t*>(d); d += ucap;
        lt.beta = d; d += ucap; lt.om = d; d += ucap; lt.gx = d; d += ucap; lt.hs = d; d += ucap;
        lt.nv = d; d += ucap; lt.qs = d; d += ucap; lt.tv = d; d += ucap;
        lt.tch = reinterpret_cast<int32_t*>(d);
    }
    const CsTracked gt{b.uk, b.voff, b.iota, b.ubeta, b.uom, b.ugx, b.hs, b.newval, b.qs, b.tv, b.touched};   // ... and the global one

    for (int64_t k = tid; k < p; k += kCsThreads) { b.i2s[k] = 0; b.bfold[k] = b.beta[k]; b.inmoved[k] = 0; b.gxp[k] = -1; b.iota[k] = k; }
    for (int u = tid; u < ucap; u += kCsThreads) lt.iota[u] = u;
    if (tid == 0) {
        s_crew_lost = 0;
        for (int i = 0; i < 8; ++i) s_tph[i] = 0ull;
        for (int i = 0; i < 12; ++i) s_cnt[i] = 0ll;
        s_nfail = 0;
        s_ctrl.lambda0 = lambda0; s_ctrl.n_total = n_total; s_ctrl.maxH = 0.0; s_ctrl.loss = loss; s_ctrl.has_omega = has_omega;
        s_ctrl.domain_error = 0; s_ctrl.pad = 0; s_ctrl.q_carry = q; s_ctrl.cert_abs = cert_abs;
    }
    __syncthreads();
    for (int s = tid; s < nnz; s += kCsThreads) { const int k = b.in_sup[s]; b.s2i[s] = k; b.i2s[k] = s + 1; }
    __syncthreads();

    int nmoved = ctl->n_moved, status = kCsMaxIter, n_list = 0, dom_any = 0;
    double TV0 = 0.0, lastH = 0.0;
    {   // the moves still pending on g when the host launched: d_g is the gradient as of the last fold, beta - bfold the moves since
        double tv_mine = 0.0;
        for (int m = tid; m < nmoved; m += kCsThreads) {
            const int km = b.out_moved_idx[m];
            const double d = b.out_moved_val[m];
            b.moved[m] = km; b.inmoved[km] = 1; b.bfold[km] = b.beta[km] - d;
            tv_mine += fabs(d);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tv_mine += __shfl_xor(tv_mine, off, 64);
        __shared__ double s_tvw[kCsWaves];
        if ((tid & 63) == 0) s_tvw[tid >> 6] = tv_mine;
        __syncthreads();
        for (int wv = 0; wv < kCsWaves; ++wv) TV0 += s_tvw[wv];
    }
    int pass_id = 1;             // gx_k is current iff gxp[k] == pass_id - 1: k was tracked in the pass just before
    int cnt_prev = -1;           // the tracked list G_UU in LDS was filled for (b.uprev[0 .. cnt_prev))
    int64_t passes = 0, cov_visits = 0;      // (these two steer the loop; the other counters: s_cnt)

    auto stage_pending = [&]() {       // pend_m and the column offset of every coordinate moved since the fold
        for (int m = tid; m < nmoved; m += kCsThreads) {
            const int km = b.moved[m];
            b.pendv[m] = b.beta[km] - b.bfold[km];
            b.poff[m] = (int64_t)b.slot[km] * p;
        }
        __syncthreads();
    };
    // g_k - sum_m pend_m G_mk: 8 gathers in flight
    auto exact_g = [&](int64_t k) {
        double acc = b.g[k];
        int m = 0;
        for (; m + 8 <= nmoved; m += 8) {
            double gv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) gv[t] = b.Gcols[b.poff[m + t] + k];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = fma(-b.pendv[m + t], gv[t], acc);
        }
        for (; m < nmoved; ++m) acc = fma(-b.pendv[m], b.Gcols[b.poff[m] + k], acc);
        return acc;
    };
    // g <- g - sum_m pend_m G_m over all p; nothing is pending afterwards
    auto fold = [&]() {
        stage_pending();
        for (int64_t k = tid; k < p; k += kCsThreads) b.g[k] = exact_g(k);
        for (int m = tid; m < nmoved; m += kCsThreads) { const int km = b.moved[m]; b.bfold[km] = b.beta[km]; b.inmoved[km] = 0; }
        __syncthreads();
        nmoved = 0; TV0 = 0.0; count(kFolds, 1);
    };

    for (;;) {
        lap(7);
        if (passes >= max_passes) { status = kCsMaxIter; break; }
        if (nnz > nnz_limit) { status = kCsOutgrown; break; }
        if (cov_visits > cov_budget) { status = kCsRefresh; break; }
        if (sqrt_loss && q < q_floor) { status = kCsNeedQ; break; }      // r'r has run out of digits: the host sums it from r
        const bool full = conv;
        if (full && nnz > full_cap && !crew) { status = BIG ? kCsHostFull : kCsNeedBig; break; }
        if (crew) {
            crew_wait(njobs);                                       // g is as current as the jobs posted so far make it
            if (full && nmoved > 0 && nnz + kCsTrackedMargin > ucap_lists) {
                // a full pass of a large support is a crew pass (below): the moves the table-mode passes have left pending are folded
                // into g first -- p x moves gathers, the helpers' work
                stage_pending();
                if (tid < 64) {
                    CsCrewJob* job = &cw->ring[njobs % (uint32_t)kCsCrewRing];
                    if (tid == 0) { job->kind = kCrewFold; job->nmove = nmoved; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (tid == 0) cs_st_release(&cw->posted, njobs + 1);
                }
                njobs += 1; count(kCrewJobs, 1);
                for (int m = tid; m < nmoved; m += kCsThreads) { const int km = b.moved[m]; b.bfold[km] = b.beta[km]; b.inmoved[km] = 0; }
                crew_wait(njobs);
                nmoved = 0; TV0 = 0.0; count(kFolds, 1); pass_id += 2; tepoch += 1;
            }
            if (s_crew_lost) { status = kCsCrewLost; break; }
        }
        const uint64_t rng_before = rng;
        const int L = full ? (int)p : nnz;
        const int Lm1 = L > 0 ? L - 1 : 0;
        const bool direct = full && !randomize;      // an ordered full pass visits k = i: no list
        // ---- reset!(it, full) + collect(it) (atom_iterator.jl:34-37, 53-64; the splitmix64 substitute of sparse_iterate.hpp) ----
        if (randomize && L <= 64) {
            // one lane per position: Fisher-Yates in the registers of wave 0 (small_solve.hpp's wave_build_list), no scratch
            if (tid < 64) {
                int dr = tid;
                if (tid + 1 < L) {
                    uint64_t st = rng + (uint64_t)tid * 0x9E3779B97F4A7C15ull;
                    dr = tid + (int)mod64_small(small_rng_next(st), (uint32_t)(L - tid));
                }
                int val = tid;
                for (int i = 0; i + 1 < L; ++i) {
                    const int j = __builtin_amdgcn_readlane(dr, i);
                    const int vi = __builtin_amdgcn_readlane(val, i), vj = __builtin_amdgcn_readlane(val, j);
                    val = tid == i ? vj : (tid == j ? vi : val);
                }
                if (tid < L) b.list[tid] = full ? val : b.s2i[val];
            }
            if (L > 1) rng += (uint64_t)(L - 1) * 0x9E3779B97F4A7C15ull;
        } else if (randomize) {
            // the shuffle's six (p + 1)-sized arrays overlay the tracked Gram block (and arrays) in LDS, which are refilled below anyway (a
            // shuffled pass visits its coordinates in a new order); Fisher-Yates itself without the serial swaps: small_solve.hpp
            int32_t* f_draw = reinterpret_cast<int32_t*>(s_dynamic);
            int32_t *f_cnt = f_draw + (p + 1), *f_off = f_cnt + (p + 1), *f_bucket = f_off + (p + 1), *f_par = f_bucket + (p + 1), *f_out = f_par + (p + 1);
            for (int i = tid; i + 1 < L; i += kCsThreads) {
                uint64_t st = rng + (uint64_t)i * 0x9E3779B97F4A7C15ull;
                f_draw[i] = i + (int)mod64_small(small_rng_next(st), (uint32_t)(L - i));
            }
            if (L > 1) rng += (uint64_t)(L - 1) * 0x9E3779B97F4A7C15ull;
            __syncthreads();
            parallel_fisher_yates<kCsThreads>(tid, L, f_draw, f_cnt, f_off, f_bucket, f_par, f_out, s_w, [] { __syncthreads(); });
            for (int i = tid; i < L; i += kCsThreads) b.list[i] = full ? f_out[i] : b.s2i[f_out[i]];
            __syncthreads();
            for (int u = tid; u < ucap; u += kCsThreads) lt.iota[u] = u;      // (the scratch may reach into the tracked arrays: all of
            cnt_prev = -1;                                                    // them are rebuilt per pass but this one)
        } else if (!full) {
            for (int i = tid; i < L; i += kCsThreads) b.list[i] = b.s2i[i];
        }
        __syncthreads();
        lap(0);

        // ---- the scan: which positions are visited ("unsettled"), in visit order ----
        int cnt = 0, nocol = 0, nzero = 0, nsupp = 0, nexc = 0;
        bool scanned = false, want_fold = false;
        if (!full) {                     // an active pass visits its whole list: nothing to classify (unless a column is missing)
            int nc_mine = 0;
            for (int i = tid; i < L; i += kCsThreads) {
                const int k = b.list[i];
                b.uk[i] = k; b.upos[i] = i; b.vb[k] = i;
                nc_mine += b.slot[k] < 0 ? 1 : 0;
            }
            if (cs_block_sum(nc_mine, s_w) == 0) { cnt = L; nsupp = L; scanned = true; }
        }
        for (int attempt = 0; !scanned; ++attempt) {
            cnt = 0; nocol = 0;
            int nz_mine = 0, ns_mine = 0, nx_mine = 0;
            const double thr_base = lambda0 * (sqrt_loss ? sqrt(q) : n_total);
            for (int i0 = 0; i0 < L; i0 += kCsThreads * E) {
                int k[E], sl[E], isl[E], inm[E], frc[E];
                bool valid[E], uns[E], nc[E], st[E];
                double gk[E], bk[E], ak[E], om[E], mk[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = i0 + tid * E + e;
                    valid[e] = i < L;
                    const int ic = min(i, Lm1);
                    k[e] = direct ? ic : b.list[ic];
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {                        // every load unconditional (clamped index): all in flight together
                    gk[e] = b.g[k[e]]; bk[e] = b.beta[k[e]]; sl[e] = b.slot[k[e]]; isl[e] = b.i2s[k[e]]; inm[e] = b.inmoved[k[e]];
                    ak[e] = b.a[k[e]]; mk[e] = b.colmax[k[e]]; om[e] = has_omega ? b.omega[k[e]] : 1.0; frc[e] = b.forced[k[e]];
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    // (a coordinate that has itself moved since the fold -- it left the support -- is never settled by the bound:
                    // M_k leaves out G_kk = a_k, by far the largest entry of its column)
                    st[e] = valid[e] & full & (bk[e] == 0.0) & (ak[e] > 0.0) & (inm[e] == 0) & (frc[e] == 0) &
                            (fabs(gk[e]) + mk[e] * TV0 <= thr_base * om[e] * (1.0 - 1e-9) - cert_abs * sqrt(ak[e]));
                    uns[e] = valid[e] & !st[e];
                    nc[e] = uns[e] & (sl[e] < 0);
                    nz_mine += (st[e] & (gk[e] == 0.0)) ? 1 : 0;
                    ns_mine += (uns[e] & (bk[e] != 0.0)) ? 1 : 0;
                    nx_mine += (st[e] & ((gk[e] == 0.0) | (isl[e] != 0))) ? 1 : 0;
                }
                int ru[E], rc[E], tu, tc;
                cs_rank2<E>(uns, nc, ru, rc, tu, tc, s_w);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (valid[e]) {
                        b.vb[k[e]] = cnt + ru[e];          // visited: its index in the visit list; settled: visits before its turn
                        if (full) b.setflag[k[e]] = st[e] ? 1 : 0;
                        if (uns[e]) { b.uk[cnt + ru[e]] = k[e]; b.upos[cnt + ru[e]] = i0 + tid * E + e; }
                        if (nc[e]) b.out_list[nocol + rc[e]] = k[e];
                    }
                cnt += tu; nocol += tc;
            }
            nzero = cs_block_sum(nz_mine, s_w);
            nsupp = cs_block_sum(ns_mine, s_w);
            nexc = cs_block_sum(nx_mine, s_w);
            // many inactive coordinates fail the bound only because it has grown loose -- or one without a Gram column does,
            // which would send the host for a pass over X: fold (exact g for everybody) and look again
            if (attempt == 0 && nmoved > 0 && ((full && cnt - nsupp > kCsNearMax) || nocol > 0)) {
                if (nmoved > fold_limit) { want_fold = true; break; }      // p x moves gathers: the chip does that, not one CU
                fold(); pass_id += 2; tepoch += 1;
                continue;
            }
            break;
        }
        if (want_fold) { status = kCsNeedFold; rng = rng_before; break; }
        if (nocol > 0) {                 // coordinates about to be visited without a Gram column: the host fetches them
            status = nocol > busy_limit ? kCsBusy : kCsNeedColumns; n_list = nocol; rng = rng_before;
            break;
        }
        lap(1);

        // ---- the visited ("tracked") coordinates: beta, omega, their exact gradient (carried from the pass before where it was
        // kept current there, else g - sum_m pend_m G_m), their Gram block ----
        stage_pending();
        const bool in_lds = cnt <= ucap_lists;
        if (!BIG && !in_lds) { status = kCsNeedBig; rng = rng_before; break; }      // (the instantiation without the large-list paths)
        const bool table = BIG && !in_lds;               // a visit list beyond the LDS-sized Gram block: the Gram TABLE in device memory (see k_cov_solve's header)
        const bool crewp = table && crew && full;        // ... and, in a full pass with helpers present, a job per block of visits for them ("a crew pass")
        if (table && cnt > tcap) { status = kCsOutgrown; rng = rng_before; break; }
        if (crewp && nmoved > 0) { status = kCsNeedFold; rng = rng_before; break; }     // (the helpers keep g itself current: nothing may be pending on it)
        const CsTracked T = in_lds ? lt : gt;
        if (tid == 0) s_same = (in_lds && cnt == cnt_prev) ? 1 : 0;
        __syncthreads();
        int nnew = 0;
        if (table) {                     // every visited coordinate gets a table id; those new to the table are listed (newc: their positions)
            for (int again = 0; again < 2; ++again) {
                nnew = 0;
                for (int u0 = 0; u0 < cnt; u0 += kCsThreads) {
                    const int u = u0 + tid;
                    const bool in = u < cnt;
                    const int cidv = b.cidof[b.uk[in ? u : 0]];
                    const bool neu[1] = {in && cidv < 0};
                    int at[1];
                    const int tot = cs_rank<1>(neu, at, s_w);
                    if (in) b.ucid[u] = neu[0] ? ncid + nnew + at[0] : cidv;
                    if (neu[0]) b.newc[nnew + at[0]] = u;
                    nnew += tot;
                }
                if (ncid + nnew <= tcap) break;
                for (int c = tid; c < ncid; c += kCsThreads) b.cidof[b.cidk[c]] = -1;      // the table is full: it starts over with this list
                __syncthreads();
                ncid = 0; tepoch += 1;
            }
            __syncthreads();
            for (int i = tid; i < nnew; i += kCsThreads) { const int64_t k = b.uk[b.newc[i]]; b.cidof[k] = ncid + i; b.cidk[ncid + i] = k; b.gxe[ncid + i] = tepoch - 1; }
            __syncthreads();
        }
        for (int u = tid; u < cnt; u += kCsThreads) {
            const int64_t k = b.uk[u];
            const int gp = b.gxp[k];
            const double bk = b.beta[k], gxk = b.gx[k], omk = has_omega ? b.omega[k] : 1.0;
            const int64_t vo = (int64_t)b.slot[k] * p, kprev = b.uprev[u];
            T.k[u] = k; T.voff[u] = vo; T.beta[u] = bk; T.om[u] = omk;
            if (table) {                 // the table carries the exact gradient of every coordinate it holds through all the moves of table-mode passes
                const int c = b.ucid[u];
                if (b.gxe[c] != tepoch) { b.gxc[c] = (gp == pass_id - 1) ? gxk : exact_g(k); b.gxe[c] = tepoch; }
            } else {
                T.gx[u] = (gp == pass_id - 1) ? gxk : exact_g(k);
            }
            if (kprev != k) s_same = 0;
        }
        if (tid == 0) { s_ctrl.maxH = 0.0; s_ctrl.domain_error = 0; s_ctrl.q_carry = q; s_bad = 0; s_nan = 0; s_tvrun = 0.0; }
        const double q_start = q;
        __syncthreads();
        if (in_lds && !s_same) {         // G_UU[i][j] = X_ki' X_kj for the tracked coordinates (symmetric; row i contiguous)
            for (int e = tid; e < cnt * cnt; e += kCsThreads) { const int i = e / cnt, j = e - i * cnt; if (i <= j) s_G[cs_tri(i, j, cnt)] = b.Gcols[T.voff[j] + T.k[i]]; }
            for (int u = tid; u < cnt; u += kCsThreads) b.uprev[u] = T.k[u];
            cnt_prev = cnt;
        } else if (table) {
            cnt_prev = -1;
            // rows and columns of the coordinates new to the table: Gc[a][c] = Gc[c][a] = (column of a)[k_c], c <= a.  Four rows at a time
            // (the coordinates of the table's ids staged in the LDS the Gram block of small lists would use): 16 gathers in flight per thread
            const int ntot = ncid + nnew;
            int32_t* s_kc = reinterpret_cast<int32_t*>(s_dynamic);
            const int kc_room = (int)(cs_tri_doubles((size_t)ucap) * 2);
            const bool staged = ntot <= kc_room;
            if (nnew > 0 && staged) for (int c = tid; c < ntot; c += kCsThreads) s_kc[c] = (int32_t)b.cidk[c];
            __syncthreads();
            for (int i0 = 0; i0 < nnew; i0 += 4) {
                int64_t off[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) off[r] = T.voff[b.newc[min(i0 + r, nnew - 1)]];
                for (int c0 = 0; c0 < ntot; c0 += 4 * kCsThreads) {
                    double v[4][4];
                    int kc[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const int c = min(c0 + e * kCsThreads + tid, ntot - 1); kc[e] = staged ? s_kc[c] : (int32_t)b.cidk[c]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[r][e] = b.Gcols[off[r] + kc[e]];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = c0 + e * kCsThreads + tid, a = ncid + i0 + r;
                            if (i0 + r < nnew && c <= a) { b.Gc[(size_t)a * tcap + c] = v[r][e]; b.Gc[(size_t)c * tcap + a] = v[r][e]; }
                        }
                }
            }
            count(kTableRows, nnew); count(kTablePasses, 1);
            ncid = ntot;
            __syncthreads();
            cnt_prev = -1;               // (the staging overwrote the LDS Gram block)
        }
        __syncthreads();
        lap(2);

        // ---- _cdPass! over the visit list, 64 visits at a time: k_cov_block's gather, gram_scalar_body's B sequential updates
        // (on the tracked arrays: `beta` and `omega` indexed by position, the identity as the block's coordinate list) ----
        if (in_lds) for (int j0 = 0; j0 < cnt; j0 += B) {
            const int nb = min(B, cnt - j0);
            for (int e = tid; e < B * B; e += kCsThreads) {
                const int sI = e / B, j = e % B;
                if (sI <= j && j < nb) s_rec[R::g(sI, j)] = s_G[cs_tri(j0 + sI, j0 + j, cnt)];
            }
            if (tid < B) s_rec[R::OFF_C + tid] = (tid < nb) ? T.gx[j0 + tid] : 0.0;
            if (tid == 0) s_rec[R::OFF_Q] = s_ctrl.q_carry;
            __syncthreads();
            if (tid < 64) {
                gram_scalar_body<4>(s_rec, nb, 0, &s_ctrl, T.beta, T.om, T.iota, T.hs, T.nv, T.tch, j0, T.qs, tid);
                // the block's moves, compacted (k_cov_gupdate's prologue), and the total variation after each visit
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const double hv = tid < nb ? T.hs[j0 + tid] : 0.0;
                const bool nz = hv != 0.0;                       // (a NaN h counts as a move: the pass is then undone)
                const unsigned long long mask = __ballot(nz);
                if (nz) { const int at = __popcll(mask & ((1ull << tid) - 1ull)); s_h[at] = hv; s_mu[at] = j0 + tid; }
                if (__ballot(hv != hv)) { if (tid == 0) s_nan = 1; }
                double run = (hv == hv) ? fabs(hv) : 0.0;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(run, off, 64); if (tid >= off) run += o; }
                const double base = s_tvrun;
                if (tid < nb) T.tv[j0 + tid] = base + run;
                if (tid == 0) s_nmove = __popcll(mask);
                const double last = __shfl(run, 63, 64);
                if (tid == 0) s_tvrun = base + last;
            }
            __syncthreads();
            const int nmove = s_nmove;
            if (nmove > 0)               // every tracked coordinate sees the block's moves (the block's own members too: gx stays current)
                for (int u = tid; u < cnt; u += kCsThreads) {
                    double acc = T.gx[u];
                    for (int i = 0; i < nmove; ++i) { const int mv = s_mu[i]; acc = fma(-s_h[i], s_G[cs_tri(min(mv, u), max(mv, u), cnt)], acc); }
                    T.gx[u] = acc;
                }
            __syncthreads();
        }
        if constexpr (BIG) if (table && cnt > 0) {
            // Visit lists beyond the LDS block, two blocks in flight.
            // Table mode (active passes; full passes of a launch without helpers never get here): the update of the table's gradients
            // with a block's moves (64 rows of Gc, ~400 KB at 800 coordinates: the L1 of one CU moves that in ~4 us, as long as the block's
            // 64 visits take) runs on waves 1-3 WHILE wave 0 visits the next block; that block's own 64 gradients get the moves first (a
            // 64 x 64 tile of Gc, summed in the same order: bit for bit what the rows will leave in gxc), and its Gram block was gathered
            // during the block before.
            // A crew pass (full passes of a launch with helpers): g[k] itself is the gradient of every coordinate (nothing pending, no
            // bound).  Same pipeline, the Gram entries straight from the cached columns; the tile's result is written back to g and the
            // helpers are told to leave those 64 coordinates alone for the job of the block before; after its visits workgroup 0 POSTS the
            // block's moves: the helpers apply them to all other coordinates and re-check every skipped coordinate at its turn (marks in
            // forced[], cw->bad) -- one job behind, never waited for inside the pass.
            const int nch = (cnt + B - 1) / B;
            auto recb = [&](int which) -> double* { return which ? s_dynamic : s_rec; };
            static_assert(R::N <= 2688, "the second block record");
            double* s_tile = s_dynamic + 2688;                                       // [64 moves][64 coordinates] of the block about to be visited
            int64_t* s_off2 = reinterpret_cast<int64_t*>(s_dynamic + 6784);           // crew: [2][B] column offsets of a block's moves
            double* s_q2 = reinterpret_cast<double*>(s_off2 + 2 * B);                //       [2][B] r'r after each move
            int* s_pos2 = reinterpret_cast<int*>(s_q2 + 2 * B);                      //       [2][B] visit index of each move
            int64_t* s_v2 = reinterpret_cast<int64_t*>(s_pos2 + 2 * B);              //       [2][B] column offsets of a block's coordinates
            const int lane_t = tid & 63, part = tid >> 6;
            // (the next block's Gram entries are fetched by waves 1-3 only: loads return in order, and wave 0's visits start with loads of their own)
            constexpr int NGV = (B * B + (kCsThreads - 64) - 1) / (kCsThreads - 64);
            static_assert(NGV % 2 == 0, "two rounds");
            auto stage_ids = [&](int which, int jb) {            // the coordinates (crew) / table ids of block jb into LDS, by one wave's lanes
                s_cid2[which][lane_t] = b.ucid[min(jb + lane_t, cnt - 1)];
                if (crewp) s_v2[which * B + lane_t] = T.voff[min(jb + lane_t, cnt - 1)];       // (the jobs name a move by its column)
            };
            // the next block's Gram entries, two rounds of eleven gathers per thread of waves 1-3, each round stored as it lands
            auto fetch_block = [&](int which, double* rec, int nbn) {
                if (tid < 64) return;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    double gv[NGV / 2];
#pragma unroll
                    for (int t = 0; t < NGV / 2; ++t) {
                        const int e = min(tid - 64 + (kCsThreads - 64) * (half * (NGV / 2) + t), B * B - 1), j = e & (B - 1), sI = e / B;
                        gv[t] = b.Gc[(size_t)s_cid2[which][j] * tcap + s_cid2[which][sI]];
                    }
#pragma unroll
                    for (int t = 0; t < NGV / 2; ++t) {
                        const int e = tid - 64 + (kCsThreads - 64) * (half * (NGV / 2) + t), sI = e / B, j = e & (B - 1);
                        if (e < B * B && sI <= j && j < nbn) rec[R::g(sI, j)] = gv[t];
                    }
                }
            };
            auto rows_update = [&](const double* hb, const int* mub, int nmv) {   // table mode: gxc -= sum_i h_i Gc[mu_i][.], moves in visit order
                // 256 ids at a time per wave, handed out by a counter: the waves that only move memory start at once, wave 0 joins after its visits
                for (;;) {
                    int task = 0;
                    if (lane_t == 0) task = atomicAdd(&s_task, 1);
                    task = __builtin_amdgcn_readfirstlane(task);
                    if (task * 256 >= ncid) break;
                    const int c0 = task * 256 + 4 * lane_t;
                    if (c0 >= ncid) continue;
                    double4 acc = *reinterpret_cast<const double4*>(b.gxc + c0);
                    int i = 0;
                    for (; i + 16 <= nmv; i += 16) {
                        double4 rv[16];
#pragma unroll
                        for (int t = 0; t < 16; ++t) rv[t] = *reinterpret_cast<const double4*>(b.Gc + (size_t)mub[i + t] * tcap + c0);
#pragma unroll
                        for (int t = 0; t < 16; ++t) {
                            const double hv = -hb[i + t];
                            acc.x = fma(hv, rv[t].x, acc.x); acc.y = fma(hv, rv[t].y, acc.y); acc.z = fma(hv, rv[t].z, acc.z); acc.w = fma(hv, rv[t].w, acc.w);
                        }
                    }
                    for (; i + 4 <= nmv; i += 4) {
                        double4 rv[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) rv[t] = *reinterpret_cast<const double4*>(b.Gc + (size_t)mub[i + t] * tcap + c0);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const double hv = -hb[i + t];
                            acc.x = fma(hv, rv[t].x, acc.x); acc.y = fma(hv, rv[t].y, acc.y); acc.z = fma(hv, rv[t].z, acc.z); acc.w = fma(hv, rv[t].w, acc.w);
                        }
                    }
                    for (; i < nmv; ++i) {
                        const double4 rv = *reinterpret_cast<const double4*>(b.Gc + (size_t)mub[i] * tcap + c0);
                        const double hv = -hb[i];
                        acc.x = fma(hv, rv.x, acc.x); acc.y = fma(hv, rv.y, acc.y); acc.z = fma(hv, rv.z, acc.z); acc.w = fma(hv, rv.w, acc.w);
                    }
                    *reinterpret_cast<double4*>(b.gxc + c0) = acc;       // (ids beyond ncid up to the next multiple of four: scratch nobody reads)
                }
            };
            // crew: one job, written by wave 3 (wave 0 goes straight on to the next block).  The ring holds a pass's worth of jobs, so the slot
            // written is free unless a pass has more than kCsCrewRing blocks: then, and only then, the helpers are waited for inside a pass
            auto crew_post = [&](int kind, int par, int nmv, int j0, int nb, int chk, int last, double q_start) {
                if (njobs >= (uint32_t)kCsCrewRing) crew_wait(njobs - (uint32_t)kCsCrewRing + 1u);
                if (part == 3) {
                    CsCrewJob* job = &cw->ring[njobs % (uint32_t)kCsCrewRing];
                    if (kind == kCrewUpdate) {
                        job->h[lane_t] = lane_t < nmv ? s_h2[par][lane_t] : 0.0; job->q[lane_t] = s_q2[par * B + lane_t];
                        job->off[lane_t] = s_off2[par * B + lane_t]; job->pos[lane_t] = s_pos2[par * B + lane_t];
                    }
                    if (lane_t == 0) {
                        job->kind = kind; job->nmove = nmv; job->j0 = j0; job->nb = nb; job->chk = chk; job->last_block = last; job->cnt = cnt;
                        job->q_start = q_start;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane_t == 0) cs_st_release(&cw->posted, njobs + 1);
                }
                njobs += 1; count(kCrewJobs, 1);
            };
            if (crewp) {
                if (tid == 0) cw->bad = 0;                       // (every job is finished: nobody else touches it now)
                crew_post(kCrewSnapshot, 0, 0, 0, 0, 0, 0, 0.0);              // g as the pass finds it, should the pass have to be undone
            }
            if (part == 1) stage_ids(0, 0);
            __syncthreads();
            fetch_block(0, recb(0), min(B, cnt));
            __syncthreads();
            int pend_n = 0;                                       // moves of the block before, not yet in gxc / in this block's gradients
            for (int ch = 0; ch < nch; ++ch) {
                const int par = ch & 1, j0 = ch * B, nb = min(B, cnt - j0);
                double* rec = recb(par);
                const double* hprev = s_h2[par ^ 1];
                const bool more = ch + 1 < nch;
                if (more && part == 1) stage_ids(par ^ 1, j0 + B);            // the next block's ids, for its gather during this block's visits
                // (A) this block's gradients: as the rows / the helpers left them, then the moves of the block before, in visit order
                double tv_[B / kCsWaves];
                double v0 = 0.0;                                   // (table mode: the rows of the block before last are in gxc since the barrier; on its way with the tile)
                if (tid < B) v0 = b.gxc[s_cid2[par][tid]];
                if (pend_n > 0) {
#pragma unroll
                    for (int r = 0; r < B / kCsWaves; ++r) {
                        const int i = min(part + kCsWaves * r, pend_n - 1);
                        tv_[r] = b.Gc[(size_t)s_mu2[par ^ 1][i] * tcap + s_cid2[par][lane_t]];
                    }
                }
                const double q_blk = s_ctrl.q_carry;
                if (pend_n > 0) {
#pragma unroll
                    for (int r = 0; r < B / kCsWaves; ++r) s_tile[(part + kCsWaves * r) * B + lane_t] = tv_[r];
                    __syncthreads();
                }
                if (tid < B) {
                    double v = v0;
                    for (int i = 0; i < pend_n; ++i) v = fma(-hprev[i], s_tile[i * B + tid], v);
                    rec[R::OFF_C + tid] = tid < nb ? v : 0.0;
                }
                if (tid == 0) { rec[R::OFF_Q] = q_blk; s_task = 0; }
                __syncthreads();
                // (B) wave 0 visits; table mode: the others bring gxc up to date with the block before; the next block's Gram entries are on the way
                if (more) fetch_block(par ^ 1, recb(par ^ 1), min(B, cnt - j0 - B));
                if (tid < 64) {
                    gram_scalar_body<4>(rec, nb, 0, &s_ctrl, T.beta, T.om, T.iota, T.hs, T.nv, T.tch, j0, T.qs, tid);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const double hv = tid < nb ? T.hs[j0 + tid] : 0.0;
                    const bool nz = hv != 0.0;                       // (a NaN h counts as a move: the pass is then undone)
                    const unsigned long long mask = __ballot(nz);
                    if (nz) {
                        const int at = __popcll(mask & ((1ull << tid) - 1ull));
                        s_h2[par][at] = hv;
                        s_mu2[par][at] = s_cid2[par][tid];
                        if (crewp) { s_off2[par * B + at] = s_v2[par * B + tid]; s_pos2[par * B + at] = j0 + tid; s_q2[par * B + at] = sqrt_loss ? T.qs[j0 + tid] : 0.0; }
                    }
                    if (__ballot(hv != hv)) { if (tid == 0) s_nan = 1; }
                    double run = (hv == hv) ? fabs(hv) : 0.0;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(run, off, 64); if (tid >= off) run += o; }
                    const double base = s_tvrun;
                    if (tid < nb) T.tv[j0 + tid] = base + run;
                    if (tid == 0) s_nmove = __popcll(mask);
                    const double last = __shfl(run, 63, 64);
                    if (tid == 0) s_tvrun = base + last;
                }
                if (pend_n > 0) rows_update(hprev, s_mu2[par ^ 1], pend_n);
                __syncthreads();
                pend_n = s_nmove;
                // (C) crew: the block's moves go to the helpers (a block without moves is posted too: its job re-checks, and every job names
                // the coordinates of the block after it as workgroup 0's own)
                if (crewp) crew_post(kCrewUpdate, par, pend_n, j0, nb, 1, more ? 0 : 1, q_blk);
            }
            if (tid == 0) s_task = 0;
            __syncthreads();
            if (pend_n > 0) rows_update(s_h2[(nch - 1) & 1], s_mu2[(nch - 1) & 1], pend_n);
            if (crewp) {                  // the helpers' verdict on the skipped coordinates (their g is needed again only when the next pass scans)
                crew_wait(njobs);
                if (tid == 0 && cs_ld_acquire(&cw->bad) != 0) s_bad = 1;
                count(kCrewPasses, 1);
            }
            __syncthreads();
        }
        // what the p-sized loops below read by position: in global memory
        if (in_lds)
            for (int u = tid; u < cnt; u += kCsThreads) {
                b.hs[u] = T.hs[u]; b.newval[u] = T.nv[u]; b.touched[u] = T.tch[u]; b.tv[u] = T.tv[u]; b.voff[u] = T.voff[u];
                if (sqrt_loss) b.qs[u] = T.qs[u];
            }
        __syncthreads();
        lap(3);
        const double tv_pass = s_tvrun, q_end = s_ctrl.q_carry;
        if (s_ctrl.domain_error) dom_any = 1;

        // ---- the settled positions, re-checked with the bound as it stood at their turn; exactly where the bound fails ----
        if (crew && s_crew_lost) { status = kCsCrewLost; break; }
        if (full && cnt < L && (tv_pass > 0.0 || TV0 > 0.0) && !crewp) {      // (a crew pass: the helpers have re-checked on the way, with g itself)
            const int cm1 = cnt > 0 ? cnt - 1 : 0;
            for (int64_t k0 = 0; k0 < p; k0 += kCsThreads * E) {
                int64_t k[E];
                bool chk[E];
                int vbk[E], sf[E];
                double gk[E], ak[E], om[E], mk[E], tvb[E], qsb[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { const int64_t kk = k0 + tid + (int64_t)e * kCsThreads; chk[e] = kk < p; k[e] = kk < p ? kk : p - 1; }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    sf[e] = b.setflag[k[e]]; vbk[e] = b.vb[k[e]]; gk[e] = b.g[k[e]]; ak[e] = b.a[k[e]]; mk[e] = b.colmax[k[e]];
                    om[e] = has_omega ? b.omega[k[e]] : 1.0;
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int at = min(max(vbk[e] - 1, 0), cm1);
                    tvb[e] = b.tv[at]; qsb[e] = sqrt_loss ? b.qs[at] : 0.0;
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const double tvk = TV0 + (vbk[e] > 0 ? tvb[e] : 0.0);
                    const double qk = vbk[e] > 0 ? qsb[e] : q_start;
                    const double cert = lambda0 * (sqrt_loss ? sqrt(qk) : n_total) * om[e] * (1.0 - 1e-9) - cert_abs * sqrt(ak[e]);
                    const bool fails = chk[e] & (sf[e] != 0) & !(fabs(gk[e]) + mk[e] * tvk <= cert);
                    if (fails) {                                  // listed: the exact gradient when its turn came is a gather the block shares
                        const int at = atomicAdd(&s_nfail, 1);
                        b.holes[at] = (int32_t)k[e]; b.fills[at] = vbk[e]; b.bsnap[at] = cert;
                    }
                }
            }
            __syncthreads();
            // g_k - sum_(pending) pend_m G_mk - sum_(visits before its turn) h_i G_ik.  Few listed coordinates: one per wave, its terms over the
            // lanes (benchmark/cd_bench.jl's path: ~20 per pass with hundreds of terms each, 4 us apiece when one thread walked them).  Many
            // (scaledLasso!'s sigma steps loosen the bound for thousands at once, with a few dozen terms each): one per THREAD, eight gathers in
            // flight -- 0.04 us apiece with every thread busy against 0.28 us through the waves.
            const int nfail = s_nfail;
            const int lane = tid & 63, wave = tid >> 6;
            if (nfail >= 2 * kCsThreads) {
                for (int f = tid; f < nfail; f += kCsThreads) {
                    const int64_t kf = b.holes[f];
                    const int vf = b.fills[f];
                    double acc = exact_g(kf);
                    int i = 0;
                    for (; i + 8 <= vf; i += 8) {
                        double gv[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) gv[t] = b.Gcols[b.voff[i + t] + kf];
#pragma unroll
                        for (int t = 0; t < 8; ++t) acc = fma(-b.hs[i + t], gv[t], acc);
                    }
                    for (; i < vf; ++i) acc = fma(-b.hs[i], b.Gcols[b.voff[i] + kf], acc);
                    if (!(fabs(acc) <= b.bsnap[f])) { s_bad = 1; b.forced[kf] = 1; }
                }
            } else
            for (int f = wave; f < nfail; f += kCsWaves) {
                const int64_t kf = b.holes[f];
                const int vf = b.fills[f];
                double part = 0.0;
                for (int m0 = lane; m0 < nmoved; m0 += 256) {
                    double gv[4], pv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { const int m = min(m0 + 64 * t, nmoved - 1); gv[t] = b.Gcols[b.poff[m] + kf]; pv[t] = (m0 + 64 * t < nmoved) ? b.pendv[m] : 0.0; }
#pragma unroll
                    for (int t = 0; t < 4; ++t) part = fma(-pv[t], gv[t], part);
                }
                for (int i0 = lane; i0 < vf; i0 += 256) {
                    double gv[4], hv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { const int i = min(i0 + 64 * t, vf - 1); gv[t] = b.Gcols[b.voff[i] + kf]; hv[t] = (i0 + 64 * t < vf) ? b.hs[i] : 0.0; }
#pragma unroll
                    for (int t = 0; t < 4; ++t) part = fma(-hv[t], gv[t], part);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
                const double acc = b.g[kf] + part;
                if (lane == 0 && !(fabs(acc) <= b.bsnap[f])) { s_bad = 1; b.forced[kf] = 1; }
            }
            count(kExact, nfail);
        }
        __syncthreads();
        if (tid == 0) s_nfail = 0;
        const bool crossed = s_bad != 0;
        if (crossed) forced_dirty = true;      // (marks were set, by the re-check above or by the helpers: wiped after the next accepted full pass or at the exit)
        bool undo = crossed || s_nan != 0 || (nzero > 0 && (TV0 > 0.0 || tv_pass > 0.0));
        bool injected = false;
        if (full && cnt > 0 && inject_every > 0) { inject_count += 1; if (inject_count % inject_every == 0) { undo = true; injected = true; } }
        if (undo) {                       // the pass never happened (beta itself was not touched)
            rng = rng_before;
            if (crewp && cnt > 0) {       // ... once g is what it was (the helpers had moved it along)
                if (tid < 64) {
                    CsCrewJob* job = &cw->ring[njobs % (uint32_t)kCsCrewRing];
                    if (tid == 0) { job->kind = kCrewRestore; job->nmove = 0; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (tid == 0) cs_st_release(&cw->posted, njobs + 1);
                }
                njobs += 1; count(kCrewJobs, 1);
            }
            // coordinates that crossed their threshold through the pass's own moves: the same pass again with those on the visit list
            // (visiting more than necessary is always right); after a few such rounds, or for anything else, the host walks it the careful way
            if (crossed && !injected && s_nan == 0 && !(nzero > 0 && (TV0 > 0.0 || tv_pass > 0.0)) && forced_rounds_here < kCsForcedRounds) {
                forced_rounds_here += 1; count(kForcedRounds, 1); forced_dirty = true;
                tepoch += 1;              // (the table's gradients have seen the undone moves)
                __syncthreads();
                continue;
            }
            status = kCsRollback;
            break;
        }
        forced_rounds_here = 0;
        if (forced_dirty && full) {       // the marks have served
            for (int64_t k = tid; k < p; k += kCsThreads) b.forced[k] = 0;
            forced_dirty = false;
        }
        lap(4);

        // ---- accepted: beta, the carried gradients, what has moved since the fold (in visit order) ----
        for (int u0 = 0; u0 < cnt; u0 += kCsThreads) {
            const int u = u0 + tid;
            const bool in = u < cnt;
            const int uc = in ? u : 0;
            const int64_t k = cnt > 0 ? T.k[uc] : 0;
            const double hv = cnt > 0 ? b.hs[uc] : 0.0;
            const bool neu[1] = {in && hv != 0.0 && b.inmoved[k] == 0 && !crewp};      // (a crew pass leaves nothing pending on g)
            int at[1];
            const int tot = cs_rank<1>(neu, at, s_w);
            if (neu[0]) { b.moved[nmoved + at[0]] = (int32_t)k; b.inmoved[k] = 1; }
            if (in) { b.gxp[k] = pass_id; b.gx[k] = table ? b.gxc[b.ucid[u]] : T.gx[u]; }
            if (in) { b.beta[k] = T.beta[u]; if (crewp) b.bfold[k] = T.beta[u]; }
            nmoved += tot;
        }
        pass_id += 1;
        if (!table) tepoch += 1;          // (the table's gradients have not seen this pass's moves)
        if (!crewp) TV0 += tv_pass;
        q = q_end;
        __syncthreads();
        lap(5);

        // ---- ProximalBase's SparseIterate.  The straightforward replay (further down) appends a slot for every settled
        // coordinate of a least-squares pass -- p of them -- only for dropzeros! to sweep them out again.  All of them hold
        // zeros, so none can be a filler, and only those whose slot would lie below the final count m can be a hole: with
        // A-index alpha = (appends before it) that is alpha < m - nnz_old <= (entrants).  And alpha of a VISITED coordinate
        // follows from its position alone when every settled coordinate before it appended (no exception: nexc == 0):
        //   alpha_j = (pos_j - j) + (visited appenders before j).
        // So the final slot order is assembled from the old slots and the visited appenders only -- (nnz + cnt)-sized loops.
        const bool analytic = sqrt_loss || !full || nexc == 0;
        if (analytic) {
            const int nnz_old = nnz;
            // 1. the visited coordinates that append: their A-index; those that stay non-zero ("entrants")
            int nA = 0, nE = 0;
            for (int u0 = 0; u0 < cnt; u0 += kCsThreads) {
                const int u = u0 + tid;
                const bool in = u < cnt;
                const int uc = in ? u : 0;
                const int64_t k = b.uk[uc];
                const double nvv = b.newval[uc];
                const bool app[1] = {in && b.i2s[k] == 0 && (b.touched[uc] != 0 || nvv != 0.0)};
                int r[1];
                const int tot = cs_rank<1>(app, r, s_w);
                const int alpha = ((full && !sqrt_loss) ? b.upos[uc] - u : 0) + nA + r[0];
                if (in) b.aidx[u] = app[0] ? (nvv != 0.0 ? alpha : -2 - alpha) : -1;      // >= 0: entrant; <= -2: a zero of its own; -1: no append
                nE += cs_block_sum((app[0] && nvv != 0.0) ? 1 : 0, s_w);
                nA += tot;
            }
            int nzo_mine = 0;
            for (int s = tid; s < nnz_old; s += kCsThreads) nzo_mine += b.beta[b.s2i[s]] != 0.0 ? 1 : 0;
            const int m = cs_block_sum(nzo_mine, s_w) + nE;
            const int na_total = ((full && !sqrt_loss) ? (L - cnt) : 0) + nA;
            if (na_total != 0 || m != nnz_old) {
                // 2. entrants whose slot lies below m take it; the other slots below m that were appended hold zeros: holes
                const int X = max(m - nnz_old, 0);
                for (int a = tid; a < X; a += kCsThreads) b.occ[a] = 0;
                __syncthreads();
                for (int u = tid; u < cnt; u += kCsThreads) {
                    const int al = b.aidx[u];
                    if (al >= 0 && al < X) { const int64_t k = b.uk[u]; b.occ[al] = 1; b.s2i[nnz_old + al] = (int32_t)k; b.i2s[k] = nnz_old + al + 1; }
                }
                __syncthreads();
                int nh = 0, nf = 0;
                const int lim = min(m, nnz_old);
                for (int s0 = 0; s0 < lim; s0 += kCsThreads) {             // holes among the old slots, ascending
                    const int s = s0 + tid;
                    const bool hole[1] = {s < lim && b.beta[b.s2i[s < lim ? s : 0]] == 0.0};
                    int at[1];
                    const int tot = cs_rank<1>(hole, at, s_w);
                    if (hole[0]) b.holes[nh + at[0]] = s;
                    nh += tot;
                }
                for (int a0 = 0; a0 < X; a0 += kCsThreads) {               // ... then among the appended ones
                    const int a = a0 + tid;
                    const bool hole[1] = {a < X && b.occ[a < X ? a : 0] == 0};
                    int at[1];
                    const int tot = cs_rank<1>(hole, at, s_w);
                    if (hole[0]) b.holes[nh + at[0]] = nnz_old + a;
                    nh += tot;
                }
                for (int u1 = cnt; u1 > 0; u1 -= kCsThreads) {             // fillers, from the last slot down: entrants at or beyond m ...
                    const int u = u1 - 1 - tid;
                    const int al = u >= 0 ? b.aidx[u] : -1;
                    const bool fil[1] = {u >= 0 && al >= X};
                    int at[1];
                    const int tot = cs_rank<1>(fil, at, s_w);
                    if (fil[0]) b.fills[nf + at[0]] = (int32_t)b.uk[u];
                    nf += tot;
                }
                for (int s1 = nnz_old; s1 > m; s1 -= kCsThreads) {         // ... then old slots at or beyond m that hold non-zeros
                    const int s = s1 - 1 - tid;
                    const bool fil[1] = {s >= m && b.beta[b.s2i[s >= m ? s : m]] != 0.0};
                    int at[1];
                    const int tot = cs_rank<1>(fil, at, s_w);
                    if (fil[0]) b.fills[nf + at[0]] = b.s2i[s];
                    nf += tot;
                }
                __syncthreads();
                for (int s = tid; s < nnz_old; s += kCsThreads) { const int ks = b.s2i[s]; if (b.beta[ks] == 0.0) b.i2s[ks] = 0; }
                __syncthreads();
                const int nmatch = min(nh, nf);                            // (equal by construction)
                for (int r = tid; r < nmatch; r += kCsThreads) { const int kf = b.fills[r]; b.s2i[b.holes[r]] = kf; b.i2s[kf] = b.holes[r] + 1; }
                __syncthreads();
                nnz = m;
            }
            lap(6);
        }
        // ---- ProximalBase's SparseIterate, in visit order: a pre-prox non-zero appends a slot (x[k] += b/a), cdprox! stores
        // the value; a settled visit of the least-squares losses with g_k != 0 leaves a zero in a slot of its own ----
        if (!analytic) {
            const int cm1 = cnt > 0 ? cnt - 1 : 0;
            for (int i0 = 0; i0 < L; i0 += kCsThreads * E) {
                int k[E], isl[E], sf[E], jv[E], tch[E];
                bool app[E], valid[E];
                double gk[E], nvv[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { const int i = i0 + tid * E + e; valid[e] = i < L; const int ic = min(i, Lm1); k[e] = direct ? ic : b.list[ic]; }
#pragma unroll
                for (int e = 0; e < E; ++e) { isl[e] = b.i2s[k[e]]; sf[e] = full ? b.setflag[k[e]] : 0; jv[e] = b.vb[k[e]]; gk[e] = b.g[k[e]]; }
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = min(max(jv[e], 0), cm1); tch[e] = b.touched[j]; nvv[e] = b.newval[j]; }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool settled_app = !sqrt_loss & (gk[e] != 0.0);
                    const bool visited_app = (cnt > 0) & ((tch[e] != 0) | (nvv[e] != 0.0));
                    app[e] = valid[e] & (isl[e] == 0) & (sf[e] ? settled_app : visited_app);
                }
                int sl[E];
                const int tot = cs_rank<E>(app, sl, s_w);
#pragma unroll
                for (int e = 0; e < E; ++e) if (app[e]) { b.s2i[nnz + sl[e]] = k[e]; b.i2s[k[e]] = nnz + sl[e] + 1; }
                nnz += tot;
            }
        }
        __syncthreads();
        lap(6);
        // ---- dropzeros!: swap-with-last (sparse_iterate.hpp) as a match of holes and fillers (small_solve.hpp) ----
        if (!analytic) {
            int mine = 0;
            for (int s0 = 0; s0 < nnz; s0 += kCsThreads * E) {
                int ks[E];
#pragma unroll
                for (int e = 0; e < E; ++e) ks[e] = b.s2i[min(s0 + tid + e * kCsThreads, nnz - 1)];
#pragma unroll
                for (int e = 0; e < E; ++e) mine += ((s0 + tid + e * kCsThreads < nnz) & (b.beta[ks[e]] != 0.0)) ? 1 : 0;
            }
            const int m = cs_block_sum(mine, s_w);
            if (m != nnz) {
                int nh = 0, nf = 0;
                for (int s0 = 0; s0 < m; s0 += kCsThreads * E) {
                    bool hole[E];
                    int s[E], at[E], ks[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) { s[e] = s0 + tid * E + e; ks[e] = b.s2i[min(s[e], nnz - 1)]; }
#pragma unroll
                    for (int e = 0; e < E; ++e) hole[e] = (s[e] < m) & (b.beta[ks[e]] == 0.0);
                    const int tot = cs_rank<E>(hole, at, s_w);
#pragma unroll
                    for (int e = 0; e < E; ++e) if (hole[e]) b.holes[nh + at[e]] = s[e];
                    nh += tot;
                }
                for (int s1 = nnz; s1 > m; s1 -= kCsThreads * E) {       // chunks from the end; inside a chunk thread 0's first element is the last slot
                    bool fil[E];
                    int s[E], at[E], ks[E];
#pragma unroll
                    for (int e = 0; e < E; ++e) { s[e] = s1 - 1 - (tid * E + e); ks[e] = b.s2i[max(s[e], 0)]; }
#pragma unroll
                    for (int e = 0; e < E; ++e) fil[e] = (s[e] >= m) & (b.beta[ks[e]] != 0.0);
                    const int tot = cs_rank<E>(fil, at, s_w);
#pragma unroll
                    for (int e = 0; e < E; ++e) if (fil[e]) b.fills[nf + at[e]] = s[e];
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
        passes += 1; cov_visits += cnt; lastH = s_ctrl.maxH;
        count(kVisits, L); count(kSettled, L - cnt);
        if (full) { count(kFullPasses, 1); count(kCovFull, cnt); }
        prev_conv = conv;
        conv = lastH < optTol;
        if (prev_conv && conv) { status = kCsConverged; break; }
    }

    // ---- what the host needs: the support in slot order with its values, the moves still pending on g ----
    __syncthreads();
    if (crew) {                           // the helpers finish what is posted, then leave
        crew_wait(njobs);
        if (s_crew_lost) status = kCsCrewLost;
        if (tid == 0) cs_st_release(&cw->exit_flag, 1u);
    }
    if (forced_dirty) for (int64_t k = tid; k < p; k += kCsThreads) b.forced[k] = 0;
    for (int s = tid; s < nnz; s += kCsThreads) { const int k = b.s2i[s]; b.out_sup_idx[s] = k; b.out_sup_val[s] = b.beta[k]; }
    for (int m = tid; m < nmoved; m += kCsThreads) { const int km = b.moved[m]; b.out_moved_idx[m] = km; b.out_moved_val[m] = b.beta[km] - b.bfold[km]; }
    if (tid == 0) {
        ctl->rng = rng; ctl->q = q; ctl->nnz = nnz; ctl->prev_conv = prev_conv ? 1 : 0; ctl->conv = conv ? 1 : 0;
        ctl->ncid = ncid; ctl->tepoch = tepoch; ctl->table_passes = s_cnt[kTablePasses]; ctl->table_rows = s_cnt[kTableRows];
        ctl->forced_rounds = s_cnt[kForcedRounds]; ctl->crew_passes = s_cnt[kCrewPasses]; ctl->crew_jobs = s_cnt[kCrewJobs];
        ctl->inject_count = inject_count; ctl->status = status; ctl->n_list = n_list; ctl->n_moved = nmoved;
        ctl->domain_error = dom_any; ctl->passes = passes; ctl->full_passes = s_cnt[kFullPasses]; ctl->visits = s_cnt[kVisits];
        ctl->cov_visits = cov_visits; ctl->cov_visits_full = s_cnt[kCovFull]; ctl->settled = s_cnt[kSettled]; ctl->folds = s_cnt[kFolds]; ctl->exact_rechecks = s_cnt[kExact];
        ctl->maxH = lastH;
        lap(7);
        for (int i = 0; i < 8; ++i) ctl->ticks[i] = (int64_t)s_tph[i];
        ctl->cycles = (int64_t)(__builtin_amdgcn_s_memtime() - cyc0); ctl->ticks_total = (int64_t)(__builtin_amdgcn_s_memrealtime() - tick0);
    }
}

// M_k = max(M_k, |G_jk|) over the freshly cached columns (slots s0 .. s1 - 1 of the device store; j != k: a coordinate's own
// column is the one whose slot the map names for it): the bound of k_cov_solve's certificates.  One launch per batch of columns.
__global__ __launch_bounds__(256) void k_cov_colmax(double* __restrict__ colmax, const double* __restrict__ G, const int32_t* __restrict__ slot,
                                                    int64_t s0, int64_t s1, int64_t p) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p) return;
    const int64_t own = slot[k];
    double m = colmax[k];
    bool poisoned = false;
    for (int64_t s_ = s0; s_ < s1; ++s_) {
        if (s_ == own) continue;
        const double v = fabs(G[s_ * p + k]);
        if (v > m) m = v;
        if (v != v) poisoned = true;                  // (a NaN entry poisons the bound: nothing is certified against it)
    }
    colmax[k] = poisoned ? __builtin_nan("") : m;
}

// ---- host side ---------------------------------------------------------------------------------------------------
enum { kCsNotNow = 0, kCsFinished = 1, kCsAgain = 2 };

inline size_t cs_align(size_t v) { return (v + 255) / 256 * 256; }
inline int cs_env_int(const char* nm, int dflt) { const char* v = getenv(nm); return v ? atoi(v) : dflt; }

// scratch of the kernel (122 p bytes of device memory) and the pinned block it reads the support from and writes its
// results into (zero-copy, as the one-launch solve's: nothing is copied around the launch)
int32_t cs_alloc(cdh_handle h) {
    GradCache& c = h->gc;
    if (c.cs_dev || !c.cs_enabled) return CDH_OK;
    const size_t p = (size_t)h->p;
    const size_t tc = (size_t)kCsTableCap;
    const size_t dev_bytes = 11 * cs_align(8 * p) + 5 * cs_align(8 * p) + 12 * cs_align(4 * p) + 3 * cs_align(p) + cs_align(8 * p) /* colmax */ +
                             cs_align(8 * tc * tc) + 2 * cs_align(8 * tc) + cs_align(4 * tc) + 3 * cs_align(4 * p) /* the Gram table */ +
                             cs_align(sizeof(CsCrew)) + cs_align(8 * p) /* the crew: jobs, g's snapshot */;
    const size_t pin_bytes = cs_align(sizeof(CovSolveCtl)) + 4 * cs_align(4 * p) + 2 * cs_align(8 * p);
    void* dev_view = nullptr;
    bool fits = hipMalloc((void**)&c.cs_dev, dev_bytes) == hipSuccess && hipHostMalloc((void**)&c.cs_pin, pin_bytes) == hipSuccess &&
                hipHostGetDevicePointer(&dev_view, c.cs_pin, 0) == hipSuccess;
    if (!fits) (void)hipGetLastError();
    CHK(all_ranks_agree(h, fits, &fits));
    if (!fits) {
        if (c.cs_dev) (void)hipFree(c.cs_dev);
        if (c.cs_pin) (void)hipHostFree(c.cs_pin);
        c.cs_dev = nullptr; c.cs_pin = nullptr; c.cs_enabled = false;
        return CDH_OK;
    }
    c.cs_pin_dev = static_cast<char*>(dev_view);
    CovSolveBufs& b = c.cs_bufs;
    char* d = c.cs_dev;
    auto take = [&](size_t bytes) { char* q = d; d += cs_align(bytes); return q; };
    b.p = h->p;
    b.gx = (double*)take(8 * p); b.bfold = (double*)take(8 * p); b.bsnap = (double*)take(8 * p); b.hs = (double*)take(8 * p);
    b.newval = (double*)take(8 * p); b.qs = (double*)take(8 * p); b.tv = (double*)take(8 * p); b.pendv = (double*)take(8 * p);
    b.ubeta = (double*)take(8 * p); b.uom = (double*)take(8 * p); b.ugx = (double*)take(8 * p);
    b.uk = (int64_t*)take(8 * p); b.poff = (int64_t*)take(8 * p); b.voff = (int64_t*)take(8 * p); b.uprev = (int64_t*)take(8 * p); b.iota = (int64_t*)take(8 * p);
    b.touched = (int32_t*)take(4 * p); b.s2i = (int32_t*)take(4 * p); b.i2s = (int32_t*)take(4 * p); b.list = (int32_t*)take(4 * p);
    b.vb = (int32_t*)take(4 * p); b.moved = (int32_t*)take(4 * p); b.holes = (int32_t*)take(4 * p); b.fills = (int32_t*)take(4 * p);
    b.gxp = (int32_t*)take(4 * p); b.upos = (int32_t*)take(4 * p); b.aidx = (int32_t*)take(4 * p); b.occ = (int32_t*)take(4 * p);
    b.setflag = (uint8_t*)take(p); b.inmoved = (uint8_t*)take(p); b.forced = (uint8_t*)take(p);
    // (on the handle's own stream: the first operation on the NULL stream of a process creates its queue -- 10 ms measured)
    HIPCHK(h, hipMemsetAsync(b.forced, 0, p, h->stream));
    c.d_colmax = (double*)take(8 * p);
    b.colmax = c.d_colmax;
    b.Gc = (double*)take(8 * tc * tc); b.gxc = (double*)take(8 * tc); b.cidk = (int64_t*)take(8 * tc); b.gxe = (int32_t*)take(4 * tc);
    b.cidof = (int32_t*)take(4 * p); b.ucid = (int32_t*)take(4 * p); b.newc = (int32_t*)take(4 * p);
    c.cs_ncid = 0; c.cs_table_reset = true;
    b.crew = (CsCrew*)take(sizeof(CsCrew)); b.g_snap = (double*)take(8 * p);
    // the pinned block, as the host and as the device address it
    size_t o = cs_align(sizeof(CovSolveCtl));
    auto pin = [&](size_t bytes) { const size_t at = o; o += cs_align(bytes); return at; };
    const size_t o_in = pin(4 * p), o_si = pin(4 * p), o_mi = pin(4 * p), o_li = pin(4 * p), o_sv = pin(8 * p), o_mv = pin(8 * p);
    c.cs_ctl = reinterpret_cast<CovSolveCtl*>(c.cs_pin);
    c.cs_in_sup = reinterpret_cast<int32_t*>(c.cs_pin + o_in);
    c.cs_out_sup_idx = reinterpret_cast<int32_t*>(c.cs_pin + o_si); c.cs_out_moved_idx = reinterpret_cast<int32_t*>(c.cs_pin + o_mi);
    c.cs_out_list = reinterpret_cast<int32_t*>(c.cs_pin + o_li);
    c.cs_out_sup_val = reinterpret_cast<double*>(c.cs_pin + o_sv); c.cs_out_moved_val = reinterpret_cast<double*>(c.cs_pin + o_mv);
    b.in_sup = reinterpret_cast<const int32_t*>(c.cs_pin_dev + o_in);
    b.out_sup_idx = reinterpret_cast<int32_t*>(c.cs_pin_dev + o_si); b.out_moved_idx = reinterpret_cast<int32_t*>(c.cs_pin_dev + o_mi);
    b.out_list = reinterpret_cast<int32_t*>(c.cs_pin_dev + o_li);
    b.out_sup_val = reinterpret_cast<double*>(c.cs_pin_dev + o_sv); b.out_moved_val = reinterpret_cast<double*>(c.cs_pin_dev + o_mv);
    c.cs_old.assign(p, 0.0);
    c.colmax_slots = 0;
    // dynamic LDS: the tracked coordinates' Gram block and, for shuffled sweeps, the shuffle's two p-sized arrays
    c.cs_lds_budget = kCsLdsBudget;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cov_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCsLdsBudget) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cov_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCsLdsBudget) != hipSuccess) {
        (void)hipGetLastError();
        c.cs_lds_budget = (size_t)36 * 1024;          // what the default 64 KB leave next to the kernel's static arrays
    }
    c.cs_shuffle_ok = true;
    return CDH_OK;
}

// M_k of the kernel's bound, brought up to date with the columns the device store holds
int32_t cs_update_colmax(cdh_handle h) {
    GradCache& c = h->gc;
    if (c.colmax_slots > c.dev_slots) c.colmax_slots = 0;          // the store was refilled from slot 0
    if (c.colmax_slots == c.dev_slots) return CDH_OK;
    if (c.colmax_slots == 0) HIPCHK(h, hipMemsetAsync(c.d_colmax, 0, sizeof(double) * (size_t)h->p, h->stream));
    hipLaunchKernelGGL(k_cov_colmax, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, h->stream, c.d_colmax, c.d_G, c.d_slot,
                       c.colmax_slots, c.dev_slots, h->p);
    HIPCHK(h, hipGetLastError());
    c.colmax_slots = c.dev_slots;
    return CDH_OK;
}

// The passes of a solve from `*iter` on, on the device, as far as the gradient cache can serve them.
//   kCsFinished  the solve is over (converged, or maxIter passes done): statistics and the iterate are up to date
//   kCsAgain     the kernel stopped for something the host has now supplied (Gram columns, a fresh g): call again
//   kCsNotNow    the next pass runs the round-3 way (solve() below): the cache is not engaged, or the kernel undid a pass
int32_t cov_solve(cdh_handle h, const cdh_options* o, cdh::VisitScheduler& sched, cdh_stats* st, bool* prev_conv, bool* conv,
                  int64_t* iter, int* outcome) {
    GradCache& c = h->gc;
    *outcome = kCsNotNow;
    c.prep_state = 0;
    if (!c.cs_enabled || !c.cov || !gc_applicable(h) || c.mode == 0 || h->p < kScreenMinPass || h->p > ((int64_t)1 << 26)) return CDH_OK;
    if (o->randomize && h->p > kCsShuffleMaxP) return CDH_OK;
    const bool full = *conv;
    double cert_abs = 0.0;
    if (full) {
        // (the gate of run_pass: full passes over dense iterates are not screened at all)
        if (!h->screening || h->x.nnz() * 4 > h->p) return CDH_OK;
        bool go = false;
        CHK(gc_prepare_full(h, &go, &cert_abs, false));             // the moves still pending on g go into the kernel with it
        c.prep_state = go ? 1 : 2; c.prep_cert_abs = cert_abs;      // gc_full_pass, if it comes to that, does not prepare twice
        if (!go) { gc_fold(h); return CDH_OK; }
    } else {
        if (!c.valid || !c.d_G) return CDH_OK;
        if (gc_support_outgrown(h)) { gc_invalidate(h, false); return CDH_OK; }
        if (c.cov_since_ref > c.refresh_after) CHK(gc_rereference(h));
        for (int64_t j : c.moved) if (c.slot[(size_t)j] < 0) return CDH_OK;
        gc_q_guard(h);
        if (h->loss == CDH_SQRT) CHK(gc_ensure_q(h));
        CHK(gc_cert_abs(h, &cert_abs));
    }
    // (from here on a "not now" hands the pass to the round-3 code, which expects the pending moves folded into g)
    auto not_now = [&]() -> int32_t { gc_fold(h); return CDH_OK; };
    if (!c.d_G || !c.d_scan || c.dev_slots != (int64_t)c.G.size() || (int64_t)c.moved.size() > h->p) return not_now();
    for (int64_t s_ = 0; s_ < h->x.nnz(); ++s_) if (c.slot[(size_t)h->x.coord(s_)] < 0) return not_now();
    if (h->x.nnz() > gc_max_support(h)) return not_now();
    // The LDS of workgroup 0 holds the Gram block of ~170 tracked coordinates (ucap); longer visit lists run from the Gram table
    // and with the helpers (see the header).
    const size_t shuffle_bytes = 0;              // (a shuffle's scratch overlays the tracked Gram block)
    int ucap = kCsUcapMax;
    {
        const size_t budget = c.cs_lds_budget ? c.cs_lds_budget : kCsLdsBudget;
        while (ucap > 8 && 8 * cs_tri_doubles((size_t)ucap) + (kCsTrackedBytes + 8) * (size_t)ucap + shuffle_bytes > budget) ucap -= 4;
    }
    // (visit lists beyond that run in the kernel's table mode, up to the table's rows; CDH_CS_TABLE=0: they stay with the host)
    static const bool table_on = cs_env_int("CDH_CS_TABLE", 1) != 0;
    const int64_t support_cap = table_on ? kCsTableCap - kCsTableMargin : ucap - kCsTrackedMargin;
    if (h->x.nnz() > support_cap) return not_now();
    CHK(cs_alloc(h));
    if (!c.cs_enabled) return not_now();
    // Full passes of large supports stay with the host's device pass (gc_pass_device): certifying the inactive coordinates takes g
    // for all p after every block of moves -- p x moves gathers per pass, which twenty workgroups do in microseconds and one does not
    // (the bound of the loop's certificates is useless there: M_k TV exceeds the thresholds themselves once hundreds of coordinates move)
    // ... unless the launch brings helpers (the crew, above): then they do that work beside the visits, inside the loop
    static const int full_env = cs_env_int("CDH_CS_FULL_CAP", 0);
    const int crew_env = c.cs_helpers;
    const int ucap_lists = c.cs_ucap_limit > 0 ? std::min(ucap, c.cs_ucap_limit) : ucap;
    const int lds_margin = std::min(kCsTrackedMargin, ucap_lists / 4);
    const int nhelp = (crew_env > 0 && table_on && h->x.nnz() + lds_margin / 2 > ucap_lists - lds_margin) ? std::min(crew_env, kCsCrewMax) : 0;
    const int32_t full_cap = full_env > 0 ? full_env : (nhelp > 0 ? 0x7fffffff : ucap_lists - lds_margin);
    if (full && h->x.nnz() > full_cap) return not_now();
    if (c.cs_table_reset) {          // a new X: the table's entries are void
        HIPCHK(h, hipMemsetAsync(c.cs_bufs.cidof, 0xff, sizeof(int32_t) * (size_t)h->p, h->stream));
        c.cs_ncid = 0; c.cs_table_reset = false;
    }
    // a fold is p x (pending moves) gathers: beyond a few moves the chip does it (gc_fold's kernel), not the one workgroup of the loop
    static const int fold_env = cs_env_int("CDH_CS_FOLD_LIMIT", 0);
    const int32_t fold_limit = fold_env > 0 ? fold_env : (int32_t)std::max<int64_t>(16, 120000 / h->p);
    if ((int64_t)c.moved.size() > fold_limit) gc_fold(h);
    if (!c.slot_dev_ok) {             // the columns were dropped since the map last went down (a new X): the kernel asks d_slot who has one
        HIPCHK(h, hipMemcpyAsync(c.d_slot, c.slot.data(), sizeof(int32_t) * (size_t)h->p, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        c.slot_dev_ok = true;
    }
    CHK(cs_update_colmax(h));
    CHK(gc_need_dev_g(h));
    c.prep_state = 0;                                               // from here on the state moves: a later pass prepares afresh

    CovSolveCtl& ctl = *c.cs_ctl;
    ctl.lambda0 = h->ctrl.lambda0; ctl.n_total = (double)h->n_total; ctl.optTol = o->optTol; ctl.cert_abs = cert_abs;
    ctl.max_passes = o->maxIter - *iter;
    ctl.cov_budget = std::max<int64_t>(0, c.refresh_after - c.cov_since_ref);
    ctl.loss = h->loss; ctl.has_omega = h->has_omega ? 1 : 0; ctl.randomize = o->randomize ? 1 : 0;
    {
        int64_t lim = gc_max_support(h);
        if (!(c.mode == 3 || gc_short_columns(h))) lim = std::min<int64_t>(lim, h->n_total / gc_rows_per_nnz(h));
        ctl.nnz_limit = (int32_t)std::min<int64_t>({lim, support_cap, (int64_t)0x7fffffff});
    }
    ctl.busy_limit = kGcBusy; ctl.inject_every = c.inject_rollback; ctl.fold_limit = fold_limit;
    ctl.tcap = table_on ? kCsTableCap : 0; ctl.ncid = c.cs_ncid; ctl.tepoch = c.cs_tepoch; ctl.full_cap = full_cap; ctl.ucap_limit = c.cs_ucap_limit;
    ctl.rng = sched.state(); ctl.q = c.q; ctl.q_floor = h->loss == CDH_SQRT ? kGcQGuard * c.q_exact : 0.0;
    ctl.nnz = (int32_t)h->x.nnz(); ctl.prev_conv = *prev_conv ? 1 : 0; ctl.conv = *conv ? 1 : 0; ctl.inject_count = c.inject_count;
    ctl.status = -1; ctl.n_list = 0;
    ctl.n_moved = (int32_t)c.moved.size();      // in: the moves still pending on g (out: those pending when the kernel stops)
    for (size_t m = 0; m < c.moved.size(); ++m) { c.cs_out_moved_idx[m] = (int32_t)c.moved[m]; c.cs_out_moved_val[m] = c.dbeta[(size_t)c.moved[m]]; }
    for (int64_t s_ = 0; s_ < h->x.nnz(); ++s_) c.cs_in_sup[s_] = (int32_t)h->x.coord(s_);
    CovSolveBufs b = c.cs_bufs;
    b.g = c.d_g; b.Gcols = c.d_G; b.slot = c.d_slot; b.a = c.d_a; b.omega = h->omega; b.beta = h->beta;
    // the tracked coordinates' Gram block (8 u^2 bytes) and arrays (kCsTrackedBytes u, rounded up) next to the shuffle's
    ucap = kCsUcapMax;
    while (ucap > 8 && 8 * cs_tri_doubles((size_t)ucap) + (kCsTrackedBytes + 8) * (size_t)ucap + shuffle_bytes > c.cs_lds_budget) ucap -= 4;
    // (table mode keeps a second block record and a 64 x 64 tile where the LDS Gram block of small lists would be: kCsTableLds doubles)
    if (cs_tri_doubles((size_t)ucap) < kCsTableLds) ctl.tcap = 0;
    if (ctl.tcap == 0 && h->x.nnz() > ucap - kCsTrackedMargin) return not_now();     // (the budget the runtime really granted is smaller)
    const unsigned lds = (unsigned)(8 * cs_tri_doubles((size_t)ucap) + (kCsTrackedBytes + 8) * (size_t)ucap + shuffle_bytes);
    if (o->randomize && 24 * ((size_t)h->p + 1) > (size_t)lds) return not_now();   // the shuffle's scratch overlays the dynamic LDS
    const int nh = ctl.tcap > 0 ? nhelp : 0;
    // the instantiation with the large-list paths once a list has outgrown the LDS block on this handle (or is about to: helpers are coming)
    const bool big = ctl.tcap > 0 && (c.cs_big || nh > 0 || h->x.nnz() + lds_margin / 2 > ucap_lists - lds_margin);
    if (nh > 0) HIPCHK(h, hipMemsetAsync(b.crew, 0, offsetof(CsCrew, ring), h->stream));      // the counters and flags (the jobs are written before they are posted)
    if (big) hipLaunchKernelGGL(k_cov_solve<true>, dim3(1 + nh), dim3(kCsThreads), lds, h->stream, reinterpret_cast<CovSolveCtl*>(c.cs_pin_dev), b, ucap);
    else hipLaunchKernelGGL(k_cov_solve<false>, dim3(1), dim3(kCsThreads), lds, h->stream, reinterpret_cast<CovSolveCtl*>(c.cs_pin_dev), b, ucap);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    c.n_cs_launches += 1;
    if (ctl.status < 0) return fail(h, CDH_HIP_ERROR, "the device-resident solve returned no status");

    // ---- the iterate: the support in its new slot order; what changed becomes pending residual updates ----
    const int64_t nnz_old = h->x.nnz();
    std::vector<int64_t> old_idx((size_t)nnz_old);
    for (int64_t s_ = 0; s_ < nnz_old; ++s_) { old_idx[(size_t)s_] = h->x.coord(s_); c.cs_old[(size_t)h->x.coord(s_)] = h->x.slot_value(s_); }
    auto note_move = [&](int64_t k, double d) {
        if (d == 0.0) return;
        h->dots_valid = false;        // the residual the handle stands for moves
        if (!h->r_in_pending[(size_t)k]) { h->r_in_pending[(size_t)k] = 1; h->r_pending_list.push_back(k); }
        h->r_pending[(size_t)k] += d;
        if (c.beta_ok) c.beta_ref[(size_t)k] += d;
    };
    h->x.clear();
    for (int32_t s_ = 0; s_ < ctl.nnz; ++s_) {
        const int64_t k = c.cs_out_sup_idx[s_];
        const double v = c.cs_out_sup_val[s_];
        if (v == 0.0) { h->x.set(k, 1.0); h->x.set(k, 0.0); }     // a stored zero keeps its slot (the caller put it there)
        else h->x.set(k, v);
        note_move(k, v - c.cs_old[(size_t)k]);
        c.cs_old[(size_t)k] = 0.0;
    }
    for (int64_t k : old_idx) { if (c.cs_old[(size_t)k] != 0.0) note_move(k, -c.cs_old[(size_t)k]); c.cs_old[(size_t)k] = 0.0; }
    // ---- the cache: moves still pending on g, r'r, the counters ----
    for (int64_t j : c.moved) { c.dbeta[(size_t)j] = 0.0; c.in_moved[(size_t)j] = 0; }      // what went in is in the kernel's list (or folded)
    c.moved.clear();
    for (int32_t m = 0; m < ctl.n_moved; ++m) {
        const int64_t k = c.cs_out_moved_idx[m];
        const double v = c.cs_out_moved_val[m];
        if (v == 0.0) continue;
        c.dbeta[(size_t)k] = v;
        if (!c.in_moved[(size_t)k]) { c.in_moved[(size_t)k] = 1; c.moved.push_back(k); }
    }
    c.cs_ncid = ctl.ncid; c.cs_tepoch = ctl.tepoch; c.n_cs_table_passes += ctl.table_passes; c.n_cs_table_rows += ctl.table_rows;
    c.n_cs_forced_rounds += ctl.forced_rounds; c.n_cs_crew_passes += ctl.crew_passes; c.n_cs_crew_jobs += ctl.crew_jobs;
    if (ctl.crew_passes > 0) c.g_host_ok = false;       // the helpers have moved d_g along
    if (ctl.folds > 0) c.g_host_ok = false;
    if (h->loss == CDH_SQRT) c.q = ctl.q;
    c.inject_count = ctl.inject_count;
    c.n_passes += ctl.full_passes; c.n_dev_passes += ctl.full_passes; c.n_certified += ctl.settled;
    c.n_cov += ctl.cov_visits; c.cov_since_ref += ctl.cov_visits; c.n_cs_passes += ctl.passes; c.n_cs_folds += ctl.folds; c.n_cs_exact += ctl.exact_rechecks;
    for (int i = 0; i < 8; ++i) c.cs_ticks[i] += ctl.ticks[i];
    c.cs_cycles += ctl.cycles; c.cs_ticks_total += ctl.ticks_total;
    c.n_exact += ctl.cov_visits_full;
    if (ctl.domain_error) h->domain_error = true;
    sched.set_state(ctl.rng);
    *prev_conv = ctl.prev_conv != 0; *conv = ctl.conv != 0; *iter += ctl.passes;
    st->passes += ctl.passes; st->full_passes += ctl.full_passes; st->visits += ctl.visits;
    if (ctl.passes > 0) st->maxH = ctl.maxH;

    switch (ctl.status) {
    case kCsConverged: st->converged = 1; *outcome = kCsFinished; return CDH_OK;
    case kCsMaxIter: *outcome = kCsFinished; return CDH_OK;
    case kCsRollback: c.n_rollbacks += 1; return CDH_OK;            // the host walks this pass the careful way
    case kCsOutgrown: return CDH_OK;                                // gc_prepare_full / gc_ready_for_cov draw the consequences
    case kCsRefresh: CHK(gc_rereference(h)); *outcome = kCsAgain; return CDH_OK;
    case kCsNeedQ: c.q_valid = false; CHK(gc_ensure_q(h)); *outcome = kCsAgain; return CDH_OK;
    case kCsNeedFold: gc_fold(h); *outcome = kCsAgain; return CDH_OK;
    case kCsCrewLost: return fail(h, CDH_HIP_ERROR, "the device-resident solve lost its helper workgroups (a wait ran into its 20 s bound)");
    case kCsNeedBig:                                                // a visit list has outgrown the LDS block: the instantiation that knows what to do
        if (ctl.tcap == 0) return CDH_OK;                           // (no room for its scratch in LDS: the host's passes)
        c.cs_big = true; *outcome = kCsAgain; return CDH_OK;
    case kCsHostFull: return CDH_OK;                                // the next (full) pass runs the pass-by-pass way
    case kCsBusy:     // many inactive coordinates about to move: back off (1, 2, 4 ... 16 plain passes), as gc_pass_device does
        c.cooldown = c.backoff; c.backoff = std::min(16, 2 * c.backoff);
        gc_invalidate(h, false);
        c.prep_state = 2;
        return CDH_OK;
    case kCsNeedColumns: {
        if (ctl.passes == 0 && c.cs_stalled) { c.cs_stalled = false; return CDH_OK; }   // (twice in a row without progress: the old way)
        c.cs_stalled = ctl.passes == 0;
        c.backoff = 1;
        std::vector<int64_t> enter(c.cs_out_list, c.cs_out_list + ctl.n_list);
        for (int64_t j : c.moved) if (c.slot[(size_t)j] < 0) return fail(h, CDH_BAD_ARG, "gradient cache: a moved coordinate has no Gram column");
        gc_fold(h);
        if (!c.valid) return CDH_OK;
        CHK(gc_need_host_g(h));
        GcThresholds T{h, h->ctrl.lambda0, (double)h->n_total, 0.0, cert_abs};
        if (h->loss == CDH_SQRT) { CHK(gc_ensure_q(h)); T.rnorm = std::sqrt(c.q); }
        CHK(gc_fetch_entering(h, enter, [&](int64_t k) { return T.cert(k); }, [&](int64_t k) { return T.ratio(k); }));
        if (c.mode == 0) return CDH_OK;
        *outcome = kCsAgain;
        return CDH_OK;
    }
    default: return fail(h, CDH_HIP_ERROR, "the device-resident solve returned an unknown status");
    }
}
