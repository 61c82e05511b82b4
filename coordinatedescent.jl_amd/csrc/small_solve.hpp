// small_solve.hpp -- the whole of _coordinateDescent! (coordinate_descent.jl:65-92) in ONE launch, for problems whose
// Gram matrix fits on chip (p <= 1024 columns): a section of cdhip.hip kept in its own file (included once, inside cdhip.hip's anonymous
// namespace, after grad_cache.hpp, whose Gram-column scratch it borrows).
//
// Why: the reference's own test and benchmark shapes are n <= 3000 (test/lasso.jl:76-101, benchmark/cd_bench.jl:8-14).
// There a column is 8-24 KB and a pass of the streamed sweep is launch latency and nothing else: BASELINE.json's cfg1
// (n = 1000, p = 200) took 0.71 ms per solve in round 2 -- 8 passes x ~21 launches -- against 0.10 ms for ONE CPU core.
//
// How: for p <= 1024 the handle keeps the full Gram matrix G = X'X (X'WX with observation weights) in HBM -- p / 32
// launches of k_cross, once per X; built at the first solve for up to 16 MB of X, and for longer columns once the streamed
// solves have cost as much (small_applicable: rent or buy) -- and a solve is: g = X'y - G beta from the cached X'y (or one
// dots pass for g = X'r where the caller vouches for r: cdh_solve), then ONE wave runs the reference's state machine on
// (g, G): pass after pass, full or active,
// ordered or shuffled (the documented splitmix64 substitute, sparse_iterate.hpp), the visits in covariance form
//   b = g_k,  beta_k <- S(beta_k + b / a_k, lambda n omega_k / a_k)   (sqrt-lasso: the closed form on (b, a_k, r'r)),
//   g <- g - h G_k,   r'r <- r'r - 2 h b + h^2 a_k,
// with ProximalBase's SparseIterate bookkeeping (support in order of first becoming non-zero, zeros kept until
// dropzeros!'s swap-with-last) replayed on the device, because the ORDER of the support is the visit order of
// the next active pass.  The 51 solves of a cold start (coordinate_descent.jl:24-37) run inside the same launch.
// One host round trip per solve: beta, the support and the statistics come back (written by the kernel into pinned host
// memory); the residual is left to be rebuilt when something reads it (r_lazy), or learns of the moves through the same
// deferred catch-up as the gradient cache's covariance-form visits (sync_r).
// A visit step evaluates 64 consecutive positions of the visit list at once against the current g: every position
// before the first one that moves is settled exactly (g does not change until something moves), so a full pass over a
// sparse iterate is a handful of steps, and an active pass is one step per visit.
// Same iterates as the streamed sweeps up to the rounding of the g recurrence (tests: beta within 1e-10 of the oracle,
// same pass counts and support order as its per-coordinate sweep).
#pragma once

// (SmallCtl and the size limits are declared next to the handle in cdhip.hip)

// x mod d for d <= 65536 without a 64-bit division: (hi 2^32 + lo) mod d from three 32-bit remainders
__device__ __forceinline__ uint32_t mod64_small(uint64_t x, uint32_t d) {
    const uint32_t hi = (uint32_t)(x >> 32) % d, lo = (uint32_t)x % d;
    const uint32_t t = (0xffffffffu % d + 1u) % d;          // 2^32 mod d
    return (hi * t + lo) % d;                               // hi, t < 2^16: no overflow
}
__device__ __forceinline__ uint64_t small_rng_next(uint64_t& state) {
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---- what both one-launch kernels share: the scheduler and dropzeros!, executed by ONE wave on arrays in LDS ---------------
// (LDS operations of a wave are issued and served in order, so lane 0's store is seen by a later load of any lane of the
// same wave; the fences only keep the compiler from moving accesses across these points)
#define CDH_WAVE_SYNC()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// ---- Fisher-Yates without its L serial swaps -----------------------------------------------------------------------------
// The sequential loop `for i < L - 1: swap(order[i], order[j_i])` (j_i in [i, L), order = identity) leaves at position i what
// position j_i held just before step i.  A position x >= t has, before step t, been written only by the steps i' < t that
// targeted it (j_i' = x), each depositing what position i' held before step i':
//     final[i] = w(P(i))  if  P(i) = max{ i' < i : j_i' = j_i } exists, else j_i;
//     w(i)     = w(pred(i))  if  pred(i) = max{ i' < i : j_i' = i } exists, else i      (what position i holds before step i).
// So: bucket the steps by target (counting sort with LDS atomics: a bucket holds ~ln L steps, in any order), read P and pred
// off the buckets, resolve the pred chains (they strictly decrease) by pointer doubling in ceil(log2 L) rounds, and gather.
// Same permutation as the loop, bit for bit (tests: the oracle's visit orders; tests/test_host_cpp.py's scheduler).
// NT threads (one wave, or one workgroup) call it together; `sync` is their barrier.  All arrays in LDS: draw[L] (in: j_i for
// i < L - 1), cnt[L], off[L + 1], bucket[L], par[L], out[L] (the permutation); s_x: max(NT / 64, 3) ints (a workgroup's).
template <int NT, typename Sync>
__device__ __forceinline__ int pfy_thread_prefix(int v, int tid, int* s_x, Sync sync) {     // exclusive prefix of v over the threads
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += up; }
    if constexpr (NT > 64) {
        if ((tid & 63) == 63) s_x[tid >> 6] = inc;
        sync();
        int base = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) base += s_x[wv];
        sync();
        return base + inc - v;
    }
    return inc - v;
}
template <int NT, typename Sync>
__device__ __forceinline__ void parallel_fisher_yates(int tid, int L, int32_t* draw, int32_t* cnt, int32_t* off, int32_t* bucket,
                                                      int32_t* par, int32_t* out, int* s_x, Sync sync) {
    for (int i = tid; i < L; i += NT) { cnt[i] = 0; if (i == L - 1) draw[i] = L - 1; }    // (the last position: a step onto itself)
    sync();
    for (int i = tid; i < L; i += NT) atomicAdd(&cnt[draw[i]], 1);
    sync();
    {
        const int seg = (L + NT - 1) / NT, s0 = min(tid * seg, L), s1 = min(s0 + seg, L);
        int sum = 0;
        for (int x = s0; x < s1; ++x) sum += cnt[x];
        int run = pfy_thread_prefix<NT>(sum, tid, s_x, sync);
        for (int x = s0; x < s1; ++x) { const int c = cnt[x]; off[x] = run; run += c; cnt[x] = 0; }
        if (tid == 0) off[L] = L;
    }
    sync();
    for (int i = tid; i < L; i += NT) { const int x = draw[i]; const int at = atomicAdd(&cnt[x], 1); bucket[off[x] + at] = i; }
    sync();
    for (int i = tid; i < L; i += NT) {
        const int x = draw[i];
        int P = -1, pr = -1;
        for (int e = off[x], e1 = off[x + 1]; e < e1; ++e) { const int s = bucket[e]; if (s < i && s > P) P = s; }
        for (int e = off[i], e1 = off[i + 1]; e < e1; ++e) { const int s = bucket[e]; if (s < i && s > pr) pr = s; }
        out[i] = P;
        par[i] = pr >= 0 ? pr : i;
    }
    if constexpr (NT > 64) { if (tid == 0) { s_x[0] = 0; s_x[1] = 0; s_x[2] = 0; } }
    sync();
    // pointer doubling, double-buffered (cnt is free by now), until nothing moves: the chains are a few links long (steps that
    // share a target), so two or three of the ceil(log2 L) rounds do.  One wave: a ballot says so; a workgroup: three rotating
    // flags in s_x (set in round r, read after its barrier, cleared two rounds later -- no barrier of their own)
    int32_t *cur = par, *nxt = cnt;
    for (int span = 1, r = 0; span < L; span <<= 1, ++r) {
        bool moved = false;
        for (int i = tid; i < L; i += NT) { const int c0 = cur[i], c1 = cur[c0]; nxt[i] = c1; moved |= c1 != c0; }
        bool any;
        if constexpr (NT > 64) {
            if (moved) s_x[r % 3] = 1;
            if (tid == 0) s_x[(r + 1) % 3] = 0;
            sync();
            any = s_x[r % 3] != 0;
        } else {
            any = __ballot(moved) != 0ull;
            sync();
        }
        int32_t* t = cur; cur = nxt; nxt = t;
        if (!any) break;
    }
    for (int i = tid; i < L; i += NT) { const int P = out[i]; out[i] = P >= 0 ? cur[P] : draw[i]; }
    sync();
}

// reset!(it, full) + collect(it) (atom_iterator.jl:34-37, 53-64): list[0 .. L) = the pass's visit list.  Shuffled:
// Fisher-Yates, j_i = i + next() mod (L - i); splitmix64's state is a counter -- draw i is a function of state + (i + 1)
// gamma alone -- so the draws are made by all lanes, and the permutation they define is assembled by all lanes too
// (parallel_fisher_yates above).
__device__ __forceinline__ int wave_build_list(int lane, bool full, int randomize, int p, int nnz, uint64_t& rng, int32_t* order,
                                               int32_t* draw, int32_t* list, const int32_t* slot2ind, int32_t* fy_off, int32_t* fy_bucket,
                                               int32_t* fy_par) {
    const int L = full ? p : nnz;
    if (randomize && L <= 64) {
        // a list that fits one lane per position (every active pass of a sparse iterate): Fisher-Yates itself, in registers -- step i swaps
        // the values of lanes i and j_i, both read with v_readlane: ~20 cycles a step against the ~4 us of the parallel construction below
        int dr = lane;
        if (lane + 1 < L) {
            uint64_t st = rng + (uint64_t)lane * 0x9E3779B97F4A7C15ull;
            dr = lane + (int)mod64_small(small_rng_next(st), (uint32_t)(L - lane));
        }
        if (L > 1) rng += (uint64_t)(L - 1) * 0x9E3779B97F4A7C15ull;
        int val = lane;
        for (int i = 0; i + 1 < L; ++i) {
            const int j = __builtin_amdgcn_readlane(dr, i);
            const int vi = __builtin_amdgcn_readlane(val, i), vj = __builtin_amdgcn_readlane(val, j);
            val = lane == i ? vj : (lane == j ? vi : val);
        }
        if (lane < L) list[lane] = full ? val : slot2ind[val];
    } else if (randomize) {
        for (int i = lane; i < L; i += 64) {
            if (i + 1 < L) {
                uint64_t st = rng + (uint64_t)i * 0x9E3779B97F4A7C15ull;
                draw[i] = i + (int)mod64_small(small_rng_next(st), (uint32_t)(L - i));
            }
        }
        if (L > 1) rng += (uint64_t)(L - 1) * 0x9E3779B97F4A7C15ull;
        CDH_WAVE_SYNC();
        // (round 3 ran the L - 1 swaps on lane 0: a third of a shuffled cfg1 solve; `list` serves as the scratch counter array)
        parallel_fisher_yates<64>(lane, L, draw, list, fy_off, fy_bucket, fy_par, order, (int*)nullptr, [] { CDH_WAVE_SYNC(); });
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane;
            const int v = i < L ? (full ? order[i] : slot2ind[order[i]]) : 0;
            if (i < L) list[i] = v;
        }
    } else {
        for (int i = lane; i < L; i += 64) list[i] = full ? i : slot2ind[i];
    }
    CDH_WAVE_SYNC();
    return L;
}

// dropzeros!(x): swap-with-last (sparse_iterate.hpp), as a match of holes and fillers.  With m slots holding non-zeros, the
// zero slots below m are the holes; scanning forward, each is filled by the LAST slot not yet taken that holds a non-zero --
// the r-th hole (ascending) by the r-th non-zero slot from the end (descending), all of which lie at or beyond m.
// Returns the new number of slots.  `holes` and `fills` are scratch of at least nnz entries.
__device__ __forceinline__ int wave_dropzeros(int lane, int nnz, const double* beta, int32_t* slot2ind, int32_t* ind2slot,
                                              int32_t* holes, int32_t* fills) {
    const unsigned long long below = (1ull << lane) - 1ull;
    int m = 0;
    for (int s0 = 0; s0 < nnz; s0 += 64) {
        const int s = s0 + lane;
        m += __popcll(__ballot(s < nnz && beta[slot2ind[s < nnz ? s : 0]] != 0.0));
    }
    if (m == nnz) return nnz;                                  // nothing to drop: the usual end of an active pass
    int nh = 0, nf = 0;
    for (int s0 = 0; s0 < m; s0 += 64) {
        const int s = s0 + lane;
        const bool hole = s < m && beta[slot2ind[s < m ? s : 0]] == 0.0;
        const unsigned long long hm = __ballot(hole);
        if (hole) holes[nh + __popcll(hm & below)] = s;
        nh += __popcll(hm);
    }
    for (int s1 = nnz; s1 > m; s1 -= 64) {                     // chunks from the end; inside a chunk lane 0 is the last slot
        const int s = s1 - 1 - lane;
        const bool fil = s >= m && beta[slot2ind[s >= m ? s : m]] != 0.0;
        const unsigned long long fm = __ballot(fil);
        if (fil) fills[nf + __popcll(fm & below)] = s;
        nf += __popcll(fm);
    }
    CDH_WAVE_SYNC();
    // every slot that holds a zero loses its coordinate -> slot link; then the fillers move into the holes
    for (int s = lane; s < nnz; s += 64) { const int ks = slot2ind[s]; if (beta[ks] == 0.0) ind2slot[ks] = 0; }
    CDH_WAVE_SYNC();
    for (int r = lane; r < nh; r += 64) { const int kf = slot2ind[fills[r]]; slot2ind[holes[r]] = kf; ind2slot[kf] = holes[r] + 1; }
    CDH_WAVE_SYNC();
    return m;
}

// One wave.  ga: interleaved (g_k, a_k) as k_col_dots leaves them; G: p x p, column k at G + k p; q_in: r'r (sqrt-lasso).
// Dynamic LDS: the p-sized state (g, beta, a, omega; the visit list, the support's slots, the shuffle) and then as many
// Gram COLUMNS as fit (ncache): a coordinate's column is kept from its first move on, so the active passes -- one step
// per visit, the bulk of a solve -- never leave the chip.  What was slow in the first version of this kernel and is
// gone: four dependent global loads per move (now all issued at once, or none), and three walks by lane 0 alone -- the
// slots appended by the settled visits of a step, dropzeros!, the generator of the shuffle -- now done by the wave
// (ballot-ranked appends; the swap-with-last compaction as a parallel hole / filler match; splitmix64 is a counter, so
// its draws are independent).  Only Fisher-Yates' swaps themselves stay serial.
// (A lone wave issues about one instruction per four cycles, so a step costs what it has instructions: the sqrt-lasso
// closed form is compiled into its own kernel, and the loops over a p-vector are unrolled NP = 4 / 8 / 16 times for
// p <= 256 / 512 / 1024 -- unrolled 16 times with a predicate per trip they alone were 1300 cycles of a step at p = 200.)
constexpr double kSmallQGuard = 1e-6;       // r'r below this fraction of the value its recurrence started from: out of digits
constexpr int32_t kSmallPrecisionLost = -77;   // small_solve's return code for that (not a cdh status: the callers take the streamed path)
template <bool SQRT, int NP>
__global__ __launch_bounds__(64) void k_solve_small(SmallCtl* ctl, int p, int ncache, const double* __restrict__ ga,
                                                    const double* __restrict__ G, const double* __restrict__ omega,
                                                    const double* __restrict__ q_in, double* __restrict__ beta,
                                                    int32_t* __restrict__ sup /* slot2ind, in / out */,
                                                    double* __restrict__ beta_out /* a copy of beta next to ctl and sup: one copy back */) {
    extern __shared__ double s_dyn[];
    double* s_g = s_dyn;
    double* s_beta = s_g + p;
    double* s_a = s_beta + p;
    double* s_om = s_a + p;
    double* s_cols = s_om + p;                                   // ncache columns of p doubles
    int32_t* s_list = reinterpret_cast<int32_t*>(s_cols + (size_t)ncache * p);
    int32_t* s_slot2ind = s_list + p;
    int32_t* s_ind2slot = s_slot2ind + p;
    int32_t* s_order = s_ind2slot + p;                           // the shuffle; between passes: scratch of dropzeros!
    int32_t* s_draw = s_order + p;
    int32_t* s_colslot = s_draw + p;                             // coordinate -> cached column, -1 = none
    int32_t* s_fyoff = s_colslot + p;                            // the shuffle's buckets (p + 1), their contents, the chain pointers
    int32_t* s_fybucket = s_fyoff + p + 1;
    int32_t* s_fypar = s_fybucket + p;
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int has_omega = ctl->has_omega, randomize = ctl->randomize, nlam = ctl->nlam;
    const double n_total = ctl->n_total, optTol = ctl->optTol;
    const int64_t maxIter = ctl->maxIter;
    uint64_t rng = ctl->rng;
    int nnz = ctl->nnz_in, ncached = 0;
    const int g_from_c = ctl->g_from_c;
    // (the control block may sit in pinned HOST memory -- small_solve: nothing is copied down or back then -- so it is read
    // once, here: the lambdas of all the launch's solves in one wave load, handed out by v_readlane)
    const double lam_lane = ctl->lambdas[lane];
    static_assert(kSmallMaxLam == 64, "one lambda per lane");
    for (int k = lane; k < p; k += 64) {
        s_g[k] = ga[2 * k]; s_a[k] = ga[2 * k + 1]; s_beta[k] = beta[k];
        s_om[k] = has_omega ? omega[k] : 1.0;
        s_ind2slot[k] = 0; s_colslot[k] = -1;
    }
    __syncthreads();
    for (int s = lane; s < nnz; s += 64) { const int k = sup[s]; s_slot2ind[s] = k; s_ind2slot[k] = s + 1; }
    double q = SQRT ? *q_in : 0.0;
    // The sqrt-lasso's thresholds and closed form hang on r'r, which this kernel only knows by subtraction: y'y - sum beta (c + g)
    // at the start, q - 2 h b + h^2 a per move -- good to ~1e-16 of the value it started from, whatever is left of it.  Once r'r
    // has fallen below kSmallQGuard of that reference (near-noiseless y: ||r|| << ||y||) the digits are gone: the kernel gives
    // up, nothing of its work is used, and the host runs the call on the streamed kernels, which sum r'r from r itself.
    const double q_ref = q;
    int precision_lost = 0;
    if (g_from_c) {
        // `ga` held c = X'y (X'Wy), q_in y'y: the gradient and r'r of the iterate follow from the Gram matrix alone,
        //   g = c - G beta,   r'r = y'y - 2 beta'c + beta'G beta = y'y - sum_s beta_s (c_s + g_s)
        // -- no pass over X or r at all (a warm start from beta = 0 is g = c)
        __syncthreads();
        for (int s = 0; s < nnz; ++s) {
            const int ks = s_slot2ind[s];
            const double bs = s_beta[ks];
            const double* __restrict__ col = G + (int64_t)ks * p;
            for (int j = lane; j < p; j += 64) s_g[j] = fma(-bs, col[j], s_g[j]);
        }
        __syncthreads();
        if constexpr (SQRT) {
            double acc = 0.0;
            for (int s = lane; s < nnz; s += 64) { const int ks = s_slot2ind[s]; acc += s_beta[ks] * (ga[2 * ks] + s_g[ks]); }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
            q = *q_in - acc;
            if (q < 0.0) q = 0.0;
            if (q < kSmallQGuard * q_ref) precision_lost = 1;
        }
    }
    const uint64_t t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
    int64_t passes = 0, full_passes = 0, visits = 0, steps = 0;
    int converged = 0, dom_any = 0;
    double lastH = 0.0;
    __syncthreads();
    for (int il = 0; il < nlam && !precision_lost; ++il) {
        const double lambda0 = __shfl(lam_lane, il, 64);
        bool prev_conv = false, conv = true;
        converged = 0;
        for (int64_t iter = 0; iter < maxIter; ++iter) {
            const bool full = conv;
            const int L = full ? p : nnz;
            (void)wave_build_list(lane, full, randomize, p, nnz, rng, s_order, s_draw, s_list, s_slot2ind, s_fyoff, s_fybucket, s_fypar);
            // ---- _cdPass! (coordinate_descent.jl:94-110), 64 positions of the visit list at a time: lane l owns position
            // c0 + l for the whole chunk and keeps its coordinate's a, omega, beta, slot links AND gradient in registers.
            // A step evaluates every lane's visit against its current g, takes the first lane (from `done` on) that
            // moves -- the lanes before it are settled exactly, g does not change until something moves -- and applies
            // the move: each lane corrects its own g with ONE read of the mover's Gram column (g_k -= h G_k,km), while
            // the update of the whole vector in LDS (for the chunks and passes to come) runs beside it.  The critical
            // path of a step is that one read plus the scalar update. ----
            double maxH = 0.0;
            for (int c0 = 0; c0 < L; c0 += 64) {
                const int idx = c0 + lane;
                const bool valid = idx < L;
                const int k = valid ? s_list[idx] : 0;
                const double a = s_a[k], om = s_om[k];
                double oldv = s_beta[k], gk = s_g[k];
                int islot = s_ind2slot[k], cslot_own = s_colslot[k];
                // least squares: the threshold lambda0 omega_k (n / a_k) of visit_update (kernels.hpp), same expression, once per chunk
                const double thr_ls = lambda0 * om * (n_total / a);
                // ... and 1 / a_k once per chunk: a step then costs an fma where it cost a division (~35 instructions of a
                // lone wave's ~180 per step); beta differs from the divided form in the last bit at most
                const double inv_a = 1.0 / a;
                int done = 0;
                for (;;) {
                    VisitOut o{0.0, 0, 0};
                    if constexpr (SQRT) {
                        o = visit_update(1, lambda0, n_total, a, gk, q, oldv, om);
                    } else {
                        const double v = fma(gk, inv_a, oldv);
                        o.tch = (v != 0.0) ? 1 : 0;
                        o.nv = soft_threshold(v, thr_ls);
                    }
                    const double hh = o.nv - oldv;
                    const bool moves = valid && lane >= done && !(hh == 0.0);  // a NaN step "moves" (it propagates, as in the reference)
                    const unsigned long long mmask = __ballot(moves);
                    const int first = __builtin_amdgcn_readfirstlane(mmask ? (int)__builtin_ctzll(mmask) : 64);   // wave-uniform, in an SGPR
                    const bool has_mover = first < 64;
                    // the mover's Gram column: requested before anything else of the step (from LDS if it has moved before)
                    int km = 0, cslot = -1;
                    double cv[NP], gcol_own = 0.0;
                    if (has_mover) {
                        km = __builtin_amdgcn_readlane(k, first);
                        cslot = __builtin_amdgcn_readlane(cslot_own, first);
                        if (cslot < 0) {
                            const double* __restrict__ col = G + (int64_t)km * p;
                            gcol_own = col[k];
#pragma unroll
                            for (int t = 0; t < NP; ++t) cv[t] = col[min(lane + 64 * t, p - 1)];
                        } else {
                            gcol_own = s_cols[(size_t)cslot * p + k];
                        }
                    }
                    if (__ballot(valid && o.dom != 0 && lane >= done && lane <= first)) dom_any = 1;
                    // SparseIterate bookkeeping of the settled visits [done, first), in order: a pre-prox non-zero appends a
                    // slot (x[k] += b/a), cdprox! then stores the unchanged value.  Ranked by ballot: no two lanes share a coordinate
                    {
                        const bool app = valid && lane >= done && lane < first && o.tch != 0 && islot == 0;
                        const unsigned long long amask = __ballot(app);
                        if (app) { islot = nnz + __popcll(amask & below) + 1; s_slot2ind[islot - 1] = k; s_ind2slot[k] = islot; }
                        nnz += __popcll(amask);
                    }
                    steps += 1;
                    if (!has_mover) break;
                    // ---- the visit that moves: everything about it is broadcast from its lane ----
                    const double nv = readlane_f64(o.nv, first), h = readlane_f64(hh, first), bm = readlane_f64(gk, first),
                                 am = readlane_f64(a, first);
                    const int tch = __builtin_amdgcn_readlane(o.tch, first);
                    const bool appm = __builtin_amdgcn_readlane(islot, first) == 0 && (tch != 0 || nv != 0.0);
                    if (lane == first) {
                        if (appm) { islot = nnz + 1; s_slot2ind[nnz] = km; s_ind2slot[km] = nnz + 1; }
                        oldv = nv;
                        s_beta[km] = nv;
                    }
                    if (appm) nnz += 1;
                    const double ah = fabs(h);
                    if (ah > maxH) maxH = ah;                               // a NaN h never raises maxH (coordinate_descent.jl:104)
                    gk = fma(-h, gcol_own, gk);                             // this lane's own gradient: no trip through s_g
                    if (cslot >= 0) {
                        // all reads first (one LDS round trip), then the arithmetic and the stores
                        const double* colc = s_cols + (size_t)cslot * p;
                        double gv[NP];
#pragma unroll
                        for (int t = 0; t < NP; ++t) { const int j = min(lane + 64 * t, p - 1); cv[t] = colc[j]; gv[t] = s_g[j]; }
#pragma unroll
                        for (int t = 0; t < NP; ++t)
                            if (lane + 64 * t < p) s_g[lane + 64 * t] = fma(-h, cv[t], gv[t]);
                    } else {
                        const bool keep = ncached < ncache;
                        double* colc = s_cols + (size_t)(keep ? ncached : 0) * p;
                        double gv[NP];
#pragma unroll
                        for (int t = 0; t < NP; ++t) gv[t] = s_g[min(lane + 64 * t, p - 1)];
#pragma unroll
                        for (int t = 0; t < NP; ++t) {
                            const int j = lane + 64 * t;
                            if (j < p) { s_g[j] = fma(-h, cv[t], gv[t]); if (keep) colc[j] = cv[t]; }
                        }
                        if (keep) {
                            if (lane == first) { cslot_own = ncached; s_colslot[km] = ncached; }
                            ncached += 1;
                            __syncthreads();                                // the column is in LDS before a later step gathers from it
                        }
                    }
                    if constexpr (SQRT) { q = q - 2.0 * h * bm + h * h * am; if (q < 0.0) q = 0.0; if (q < kSmallQGuard * q_ref) precision_lost = 1; }
                    done = first + 1;
                    if (done >= 64) break;
                }
                __syncthreads();                                            // s_g, s_beta and the slot links as the next chunk / pass reads them
            }
            nnz = wave_dropzeros(lane, nnz, s_beta, s_slot2ind, s_ind2slot, s_order, s_draw);
            __syncthreads();
            passes += 1; visits += L; lastH = maxH;
            if (full) full_passes += 1;
            prev_conv = conv;
            conv = maxH < optTol;
            if (prev_conv && conv) { converged = 1; break; }
            if (precision_lost) break;
        }
    }
    for (int k = lane; k < p; k += 64) { beta[k] = s_beta[k]; beta_out[k] = s_beta[k]; }
    for (int s = lane; s < nnz; s += 64) sup[s] = s_slot2ind[s];
    if (lane == 0) {
        ctl->rng = rng; ctl->passes = passes; ctl->full_passes = full_passes; ctl->visits = visits;
        ctl->converged = converged; ctl->domain_error = dom_any; ctl->precision_lost = precision_lost; ctl->maxH = lastH; ctl->nnz = nnz;
        ctl->steps = steps; ctl->cycles = __builtin_amdgcn_s_memtime() - t0c; ctl->ticks = __builtin_amdgcn_s_memrealtime() - t0r;
    }
}

// G[k][b0 + b] <- the cross products of one k_cross batch (records of 64 x 32, tile-major): the batch's columns of the
// full Gram matrix, without a trip through the host
__global__ __launch_bounds__(256) void k_small_unpack(const double* __restrict__ cross, int64_t p, int b0, int nbc,
                                                      double* __restrict__ Gfull) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p) return;
    const int64_t L = k / kCrossA, i = k % kCrossA;
    for (int b = 0; b < nbc; ++b)
        Gfull[(int64_t)(b0 + b) * p + k] = cross[L * kCrossRec + ((i >> 4) * kCrossTB + (b >> 4)) * 256 + (i & 15) * 16 + (b & 15)];
}

// ---- host side ---------------------------------------------------------------------------------------------------
inline size_t small_sup_off() { return (sizeof(SmallCtl) + 15) / 16 * 16; }
inline size_t small_beta_off(int64_t p) { return small_sup_off() + ((size_t)p * sizeof(int32_t) + 15) / 16 * 16; }
inline size_t small_io_bytes(int64_t p) { return small_beta_off(p) + (size_t)p * sizeof(double); }

// Is the full Gram matrix worth building for this X?  X that fits kSmallAlwaysBytes: always (the build is a fraction of
// a millisecond: the reference's own shapes).  Beyond that it is rent or buy.  A solve on the streamed kernels is cheap where
// its passes are served by the screens and the gradient cache (measured, tools/midsize_solve.py: ~1 ms at 80 .. 400 MB of X,
// 3 ms at 3 GB), and building G costs ceil(p / 32) passes over X (1.5 .. 7 ms there) -- more than ONE such solve, a tenth of
// the 51 solves of a cold start, and every later solve then takes 0.1 .. 0.2 ms instead of 1 .. 10.  So the handle pays rent
// -- runs its solves streamed, pricing their passes -- until the rent paid on this X reaches the price of the build; a cold start
// (numSteps + 1 solves at once) buys a cheap build outright.  Never worse than about twice the better choice, whatever the
// caller does next.
inline double small_build_estimate(const cdh_handle_s* h) {
    const double bytes = (double)h->ld * (double)h->p * (double)h->esz;
    return (double)((h->p + 31) / 32 + 1) * std::max(bytes / 4.0e12, 40e-6);
}
inline bool small_candidate(const cdh_handle_s* h, const cdh_options* o) {      // everything but the rent-or-buy decision
    return h->small.enabled && !sharded(h) && h->p <= kSmallMaxP &&
           h->gc.mode != 3 /* tests force the gradient cache's own path with mode 3 */ && (h->loss != CDH_WLS || h->has_w) &&
           o->numSteps + 1 <= kSmallMaxLam && o->numSteps >= 1;
}
inline bool small_applicable(const cdh_handle_s* h, const cdh_options* o, bool many_solves = false) {
    if (!small_candidate(h, o)) return false;
    const double bytes = (double)h->ld * (double)h->p * (double)h->esz;
    if (h->small.max_bytes >= 0) return bytes <= (double)h->small.max_bytes;
    if (h->small.G_valid || bytes <= (double)h->small.always_bytes) return true;
    const double price = small_build_estimate(h);
    // (a cold start on the streamed kernels takes ~10 ms at any of these sizes -- the gradient cache serves most of its
    // passes -- so it buys outright only a build well under that)
    return (many_solves && price <= 3e-3) || h->small.rent_paid >= price;
}
// A streamed solve on a handle that could have had the Gram form pays rent on this X: what it did, priced by the model the
// build's price comes from -- a host round trip per pass; a read of X per full pass the gradient cache did not serve and
// per re-reference; a read and a half per batch of Gram columns -- rather than by the clock, so that WHICH solve of a
// call sequence builds G (and with it the last bits of the iterates) does not depend on the machine's mood.
struct SmallRent {
    cdh_handle_s* h;
    bool on;
    int64_t cached0, refs0, batches0;
    SmallRent(cdh_handle_s* h_, const cdh_options* o)
        : h(h_), on(small_candidate(h_, o) && !h_->small.G_valid), cached0(h_->gc.n_passes), refs0(h_->gc.n_validate),
          batches0(h_->gc.n_batches) {}
    void pay(const cdh_stats& st) const {
        if (!on) return;
        const double read = (double)h->ld * (double)h->p * (double)h->esz / 6.0e12;
        const int64_t streamed_full = std::max<int64_t>(0, st.full_passes - (h->gc.n_passes - cached0));
        h->small.rent_paid += 40e-6 * (double)st.passes +
                              read * ((double)streamed_full + (double)(h->gc.n_validate - refs0) + 1.5 * (double)(h->gc.n_batches - batches0));
    }
};
// (Measured and dropped, round 3: the same state machine in RESIDUAL form for short columns and many coordinates -- the
// reference's benchmark itself is n = 3000, p = 5000, where no Gram matrix fits -- as ONE workgroup with the residual in
// registers and the next columns prefetched into a ring of registers.  Correct (47 parity cases), but every column has to come
// through one CU, and one CU draws ~25 GB/s from HBM: 24 KB per visit = 1.2 us per visit at best, against 0.65 us per
// visit for the streamed sweep that uses the whole chip (0.88 s against 0.53 s on cd_bench.jl's dense lambda = 0.001 solve;
// 1.9 s with four waves per SIMD repeating the scalar update).  A one-launch solve pays only while what a visit reads stays on
// chip -- which is what the Gram form's limits say.)

int32_t small_prepare(cdh_handle h) {     // buffers, X'y and diag(G) of the current y, the full Gram matrix of the resident X
    SmallPath& sp = h->small;
    GradCache& c = h->gc;
    if (!sp.d_io) {
        CHK(gc_size(h));                  // k_cross's scratch (d_cross, d_cross_part, d_cols) is the gradient cache's
        if (!c.d_cross) { sp.enabled = false; return CDH_OK; }
        bool fits = hipMalloc((void**)&sp.d_G, sizeof(double) * (size_t)h->p * (size_t)h->p) == hipSuccess &&
                    hipMalloc((void**)&sp.d_ca, sizeof(double) * (2 * (size_t)h->p + 1)) == hipSuccess &&
                    // one block for everything that crosses the bus per solve: [SmallCtl][support: p int32][beta: p doubles]
                    hipMalloc((void**)&sp.d_io, small_io_bytes(h->p)) == hipSuccess &&
                    hipHostMalloc((void**)&sp.h_io, small_io_bytes(h->p)) == hipSuccess;
        if (!fits) { (void)hipGetLastError(); sp.enabled = false; return CDH_OK; }
        sp.d_ctl = reinterpret_cast<SmallCtl*>(sp.d_io); sp.h_ctl = reinterpret_cast<SmallCtl*>(sp.h_io);
        void* dev_view = nullptr;          // the pinned block as the device addresses it (zero-copy solves)
        if (hipHostGetDevicePointer(&dev_view, sp.h_io, 0) == hipSuccess) sp.hd_io = static_cast<char*>(dev_view);
        else (void)hipGetLastError();
        sp.d_sup = reinterpret_cast<int32_t*>(sp.d_io + small_sup_off()); sp.h_sup = reinterpret_cast<int32_t*>(sp.h_io + small_sup_off());
        sp.d_beta = reinterpret_cast<double*>(sp.d_io + small_beta_off(h->p)); sp.h_beta = reinterpret_cast<double*>(sp.h_io + small_beta_off(h->p));
        // dynamic LDS of the solve kernel: the p-sized state, then as many Gram columns as the rest of the CU's LDS holds
        // (the whole 160 KB when the runtime grants it, else what fits the default 64 KB)
        const size_t state = (size_t)h->p * (4 * sizeof(double) + 9 * sizeof(int32_t)) + 2 * sizeof(int32_t);
        size_t budget = (size_t)160 * 1024;
        auto widen = [&](auto kernel) {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget) == hipSuccess;
        };
        if (!(widen(&k_solve_small<false, 4>) && widen(&k_solve_small<true, 4>) && widen(&k_solve_small<false, 8>) &&
              widen(&k_solve_small<true, 8>) && widen(&k_solve_small<false, 16>) && widen(&k_solve_small<true, 16>))) {
            (void)hipGetLastError();
            budget = (size_t)64 * 1024;
        }
        sp.ncache = (int)std::min<size_t>(256, (budget - state) / ((size_t)h->p * sizeof(double)));
        sp.lds_bytes = (unsigned)(state + (size_t)sp.ncache * (size_t)h->p * sizeof(double));
    }
    if (!sp.c_valid) {                // c = X'y (X'Wy), a = diag(G), y'y: one dots pass over X with y in r's place, once per y
        CHK(col_dots(h, 0, h->p, h->y, h->has_w));
        HIPCHK(h, hipMemcpyAsync(sp.d_ca, h->d_colout, sizeof(double) * 2 * (size_t)h->p, hipMemcpyDeviceToDevice, h->stream));
        sp.h_c.resize((size_t)(2 * h->p));
        HIPCHK(h, hipMemcpyAsync(sp.h_c.data(), h->d_colout, sizeof(double) * 2 * (size_t)h->p, hipMemcpyDeviceToHost, h->stream));
        CHK(resid_moments_dev(h, h->y));
        HIPCHK(h, hipMemcpyAsync(sp.d_ca + 2 * h->p, h->d_red + 1, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        sp.yy = h->h_red[1];
        sp.c_valid = true;
    }
    if (sp.G_valid) return CDH_OK;
    const int64_t launches = (h->p + kCrossA - 1) / kCrossA;
    if (!sp.d_iota) {                 // the column lists of all batches, uploaded once: the batches then run back to back
        std::vector<int64_t> iota((size_t)h->p);
        for (int64_t k = 0; k < h->p; ++k) iota[(size_t)k] = k;
        if (hipMalloc((void**)&sp.d_iota, sizeof(int64_t) * (size_t)h->p) != hipSuccess) { (void)hipGetLastError(); sp.enabled = false; return CDH_OK; }
        HIPCHK(h, hipMemcpyAsync(sp.d_iota, iota.data(), sizeof(int64_t) * (size_t)h->p, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    for (int64_t b0 = 0; b0 < h->p; b0 += kCrossB) {
        const int nbc = (int)std::min<int64_t>(kCrossB, h->p - b0);
        const int64_t* cols = sp.d_iota + b0;
        CHK(dispatch(h, [&](auto* t) {
            using T = std::remove_pointer_t<decltype(t)>;
            const dim3 grid((unsigned)c.cross_GX, (unsigned)c.cross_J), block(64 * kGramWaves);
            if (h->has_w)
                hipLaunchKernelGGL((k_cross<T, true, 2, 2, cross_occ<T>()>), grid, block, 0, h->stream, (const T*)h->X, h->ld, h->nvec, h->p, cols, nbc,
                                   (const T*)h->w, c.d_cross_part);
            else
                hipLaunchKernelGGL((k_cross<T, false, 2, 2, cross_occ<T>()>), grid, block, 0, h->stream, (const T*)h->X, h->ld, h->nvec, h->p, cols, nbc,
                                   (const T*)nullptr, c.d_cross_part);
            return CDH_OK;
        }));
        hipLaunchKernelGGL(k_cross_reduce, dim3(kCrossRec / 256, (unsigned)launches), dim3(256), 0, h->stream, c.d_cross_part,
                           c.cross_J, c.d_cross);
        hipLaunchKernelGGL(k_small_unpack, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, h->stream, c.d_cross, h->p, (int)b0, nbc,
                           sp.d_G);
        HIPCHK(h, hipGetLastError());
    }
    sp.G_valid = true;
    sp.n_gram += 1;
    return CDH_OK;
}

// the solves lambdas[0 .. nlam) of one coordinateDescent! call, back to back in one launch; r must describe the handle's
// iterate on entry (initialize! has run, or the carried residual is being reused)
// from_c: the residual of the handle's iterate is y - X beta by definition (initialize! semantics), so nothing is read
// from r: g = X'y - G beta inside the kernel, and r is left to be rebuilt lazily (r_lazy).  Otherwise (cdh_solve: "assumes
// r is initialised", whatever it holds) g = X'r is taken from the device's r and the moves become pending updates of it.
int32_t small_solve(cdh_handle h, const cdh_options* o, const double* lambdas, int nlam, uint64_t* rng, cdh_stats* st, bool from_c) {
    SmallPath& sp = h->small;
    const double *ga = sp.d_ca, *qin = sp.d_ca + 2 * h->p;
    if (!from_c) {
        CHK(col_dots(h, 0, h->p, h->r, h->has_w));                  // g = X'r (X'Wr), a = diag(G): d_colout
        if (h->loss == CDH_SQRT) CHK(resid_moments_dev(h));         // r'r: d_red[1]
        ga = h->d_colout; qin = h->d_red + 1;
    }
    SmallCtl& ctl = *sp.h_ctl;
    ctl.g_from_c = from_c ? 1 : 0;
    for (int i = 0; i < nlam; ++i) ctl.lambdas[i] = lambdas[i];
    ctl.nlam = nlam; ctl.randomize = o->randomize ? 1 : 0; ctl.loss = h->loss; ctl.has_omega = h->has_omega ? 1 : 0;
    ctl.maxIter = o->maxIter; ctl.optTol = o->optTol; ctl.n_total = (double)h->n_total; ctl.rng = *rng;
    ctl.nnz_in = (int32_t)h->x.nnz();
    for (int64_t s_ = 0; s_ < h->x.nnz(); ++s_) sp.h_sup[s_] = (int32_t)h->x.coord(s_);
    // What crosses the bus per solve is one block each way -- [control][support] down, [control][support][beta] back.
    // Zero-copy (default): the kernel reads and writes that block in the pinned host buffer itself, a few hundred bytes
    // each way over the bus inside the launch, instead of two copies queued around it (~10 us of a 90 us solve at cfg1).
    const bool zc = sp.zero_copy && sp.hd_io != nullptr;
    if (!zc)
        HIPCHK(h, hipMemcpyAsync(sp.d_io, sp.h_io, small_sup_off() + sizeof(int32_t) * (size_t)h->x.nnz(), hipMemcpyHostToDevice, h->stream));
    SmallCtl* k_ctl = zc ? reinterpret_cast<SmallCtl*>(sp.hd_io) : sp.d_ctl;
    int32_t* k_sup = zc ? reinterpret_cast<int32_t*>(sp.hd_io + small_sup_off()) : sp.d_sup;
    double* k_beta = zc ? reinterpret_cast<double*>(sp.hd_io + small_beta_off(h->p)) : sp.d_beta;
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(1), dim3(64), sp.lds_bytes, h->stream, k_ctl, (int)h->p, sp.ncache, ga, sp.d_G,
                           h->omega, qin, h->beta, k_sup, k_beta);
    };
    const bool sq = h->loss == CDH_SQRT;
    if (h->p <= 256) { if (sq) go(k_solve_small<true, 4>); else go(k_solve_small<false, 4>); }
    else if (h->p <= 512) { if (sq) go(k_solve_small<true, 8>); else go(k_solve_small<false, 8>); }
    else { if (sq) go(k_solve_small<true, 16>); else go(k_solve_small<false, 16>); }
    HIPCHK(h, hipGetLastError());
    if (!zc) HIPCHK(h, hipMemcpyAsync(sp.h_io, sp.d_io, small_io_bytes(h->p), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (ctl.precision_lost) {          // r'r ran out of digits (k_solve_small): nothing of the launch is used
        std::vector<double> dense((size_t)h->p, 0.0);
        for (int64_t s_ = 0; s_ < h->x.nnz(); ++s_) dense[(size_t)h->x.coord(s_)] = h->x.slot_value(s_);
        HIPCHK(h, hipMemcpyAsync(h->beta, dense.data(), sizeof(double) * (size_t)h->p, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        sp.n_precision += 1;
        return kSmallPrecisionLost;
    }
    // what moved becomes pending residual updates (r_actual = r_virtual + X * pending: sync_r applies them before anything
    // reads r) and, for a gradient cache that holds a reference, pending moves like those of any other visit
    GradCache& c = h->gc;
    h->dots_valid = false;             // the residual the handle stands for moves with the iterate
    if (from_c) {                      // r was never read and is not touched: it stands for the new iterate, to be formed on demand
        if (c.valid || c.beta_ok) gc_invalidate(h, false);
        drop_r_pending(h);
    }
    for (int64_t k = 0; k < h->p && !from_c; ++k) {
        const double d = sp.h_beta[k] - h->x.get(k);
        if (d == 0.0) continue;
        if (!h->r_in_pending[(size_t)k]) { h->r_in_pending[(size_t)k] = 1; h->r_pending_list.push_back(k); }
        h->r_pending[(size_t)k] += d;
        if (d != d) { if (c.valid || c.beta_ok) gc_invalidate(h, false); continue; }
        if (c.beta_ok) c.beta_ref[(size_t)k] += d;
        if (c.valid) {
            c.dbeta[(size_t)k] += d;
            if (!c.in_moved[(size_t)k]) { c.in_moved[(size_t)k] = 1; c.moved.push_back(k); }
        }
    }
    c.q_valid = false;
    h->x.clear();
    for (int32_t s_ = 0; s_ < ctl.nnz; ++s_) h->x.set(sp.h_sup[s_], sp.h_beta[sp.h_sup[s_]]);
    if (from_c) { h->r_lazy = true; h->x_lazy = h->x; h->r_consistent = true; }
    *rng = ctl.rng;
    st->passes += ctl.passes; st->full_passes += ctl.full_passes; st->visits += ctl.visits;
    st->converged = ctl.converged; st->maxH = ctl.maxH;
    if (ctl.domain_error) h->domain_error = true;
    st->domain_error = h->domain_error ? 1 : 0;
    sp.n_solves += nlam;
    return CDH_OK;
}
