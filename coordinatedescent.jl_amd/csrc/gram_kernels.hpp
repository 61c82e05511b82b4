// gram_kernels.hpp -- wide-block (B = 16, 32 or 64 visits per launch) variant of the blocked sweep.
//
// Same arithmetic as k_blockstep (kernels.hpp): one pass over r applies the previous block's
// rank-B residual update (cd_differentiable_function.jl:107-109 for B visits) and accumulates
// c = X_B' r, the Gram block G = X_B' X_B and q = r'r for the block's columns (:94-99 for B
// visits); the B scalar updates (:101-104 / :271-283) then run on (c, G, q).
//
// Why a second kernel: with B = 16/32 the B(B+1)/2 per-lane accumulators of the vector-ALU
// formulation no longer fit the register file (153 / 561 fp64 values).  The accumulation is
// moved to the matrix pipe, used here as a register-cheap ACCUMULATOR, not as a throughput
// device: v_mfma_f64_16x16x4_f64 keeps a whole 16x16 fp64 tile in 8 VGPRs, exact fp64 FMA
// chains, and on CDNA4 has the same flop rate as the vector ALU.  The kernel stays HBM-bound:
// B = 32 streams 66 vectors per launch and needs ~40 % of the fp64 matrix rate at HBM speed.
// Bytes per visit fall to (2B+2)/B = 2.06 vectors (B = 32) and exchanges per sweep to p/32.
//
// Lane mapping of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): lane l supplies
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; D[i = (l>>4) + 4 t][j = l&15] sits in
// register t.  For G_IJ = X_I' X_J the K dimension is the ROW index, and any assignment of rows
// to K-slices is valid (it is a sum over all rows), so lane (c = l&15, g = l>>4) simply loads
// 16 bytes of column c at vector index 4u + g: one wave instruction fetches 64 contiguous bytes
// from each of 16 columns, and A == B for the diagonal tile.  X' r uses the same A with
// B = r broadcast over j (every column of D then holds c).
#pragma once
#include <type_traits>

#include "kernels.hpp"

namespace cdk {

typedef double dvec4 __attribute__((ext_vector_type(4)));

template <int NG> struct GramRec {
    static constexpr int B = 16 * NG;
    static constexpr int NT = NG * (NG + 1) / 2;      // upper-triangular tiles, row-major
    static constexpr int OFF_C = NT * 256;
    static constexpr int OFF_Q = OFF_C + B;
    static constexpr int N = OFF_Q + 1;
    // record offset of G[s][j], s <= j (positions inside the block)
    __host__ __device__ static constexpr int g(int s, int j) {
        const int gs = s >> 4, gj = j >> 4;
        return tile(gs, gj) * 256 + (s & 15) * 16 + (j & 15);
    }
    // index of tile (gs, gj), gs <= gj
    __host__ __device__ static constexpr int tile(int gs, int gj) {
        return gs * NG - gs * (gs - 1) / 2 + (gj - gs);
    }
};

constexpr int kGramWaves = 4;  // waves per block

// load-group sizes of k_gramstep (-D overrides build tuning experiments)
#ifndef CDH_PG
#define CDH_PG(NG) ((NG) == 4 ? 16 : 8)          // previous-block columns loaded per round (B=64: 16 measured 3-4 % faster than 4)
#endif
#ifndef CDH_UH
#define CDH_UH(NG) ((NG) == 4 ? 2 : 16 / (NG))   // fragment-load vector rows in flight per group
#endif

// LT: operand columns are loaded fully coalesced and transposed into the MFMA fragment layout
// through a wave-private LDS tile, instead of fragment-shaped loads that take 64 B from each of 16
// columns.  The tile holds all 16*NG columns of the block for a sub-chunk of 64/NG vectors; one load
// instruction covers NG columns x 64/NG consecutive vectors (1 KB, 512 B or 256 B runs: whole
// lines), and the wave's 64-vector chunk is consumed in NG sub-chunks.  Column stride = run + pad
// slots of 16 B; pad 2 makes the ds_read_b128 fragment reads conflict-free, B = 64 takes pad 1 (one
// 2-way pair) to keep a block under 80 KB of LDS = 2 blocks per CU.
// (Measured and dropped: B = 16 with half / quarter tiles -- load instructions covering 2 / 4 columns,
// 3 / 4 blocks per CU instead of 2 -- is no faster at 5e6 .. 1e7 rows; occupancy is not the limit.)
template <int NG> struct LtTile {
    static constexpr int SV = 64 / NG;                    // vectors per sub-chunk
    static constexpr int XS = SV + (NG == 4 ? 1 : 2);     // column stride in 16-B slots
    static constexpr int SLOTS = 16 * NG * XS;
};

//
// KS (LT only): sub-chunks per wave iteration.  KS = NG is the 64-vector chunk; KS = 1 makes the
// chunk ONE sub-chunk (64/NG vectors) for short columns, where a launch is only a few rounds of
// chunks long and a partly filled last round costs a whole round (measured at 1.25e6 rows, B = 64:
// 4.2 .. 5.0 rounds all take the time of 5).  Phase A then runs on a 2-D lane layout: NG/KS
// column groups x chunk vectors, one load instruction covering NG/KS previous columns, the column
// groups summed by a butterfly.
template <typename T, int NG, bool NT_, bool LT = false, int KS = NG>
__global__ __launch_bounds__(64 * kGramWaves, (NG >= 2 || LT) ? 2 : 3) void k_gramstep(
    const T* __restrict__ X, int64_t ld, int64_t nvec, const T* __restrict__ w, T* __restrict__ r,
    const int64_t* __restrict__ idx, const double* __restrict__ hs, int pos0, int nb, int nprev,
    double* __restrict__ partials) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    using R = GramRec<NG>;
    constexpr int B = R::B;
    __shared__ V s_w[kGramWaves][64];            // observation weights of the chunk (CDWeightedLSLoss)
    __shared__ V s_r[kGramWaves][64];            // r' of the wave's current 64-vector chunk
    __shared__ double s_hp[B];
    __shared__ int64_t s_kp[B];
    __shared__ double s_q[kGramWaves];
    using LTT = LtTile<NG>;
    __shared__ V s_x[LT ? kGramWaves : 1][LT ? LTT::SLOTS : 1];
    // end-of-kernel cross-wave reduction, per tile; with LT it reuses the (then idle) operand
    // tiles so the block stays under 80 KB of LDS = 2 blocks per CU
    __shared__ double s_red_own[LT ? 1 : kGramWaves][LT ? 1 : 256];
    double (*s_red)[256] = LT ? reinterpret_cast<double (*)[256]>(&s_x[0][0]) : reinterpret_cast<double (*)[256]>(&s_red_own[0][0]);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;

    // previous block's updates, compacted to those with h != 0 (no per-column branches in the
    // streaming loop: conditional loads would be waited for one by one)
    __shared__ int s_nzp;
    if (threadIdx.x < 64) {   // wave 0: all entries fetched at once, compacted with a ballot
        const int i = threadIdx.x;
        double h = 0.0;
        int64_t kp = 0;
        if (i < nprev) { h = hs[pos0 - nprev + i]; kp = idx[pos0 - nprev + i]; }
        const bool nz = (h != 0.0);
        const unsigned long long mask = __ballot(nz);
        if (nz) {
            const int at = __popcll(mask & ((1ull << i) - 1ull));
            s_hp[at] = h; s_kp[at] = kp;
        }
        if (i == 0) s_nzp = __popcll(mask);
    }
    __syncthreads();
    const int nzp = s_nzp;
    const bool anyp = nzp > 0;

    // this lane's column in each 16-column group (columns beyond nb contribute zeros)
    const V* cv[NG];
    bool act[NG];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        const int i = 16 * grp + c;
        act[grp] = i < nb;
        cv[grp] = reinterpret_cast<const V*>(X + idx[pos0 + (act[grp] ? i : 0)] * ld);
    }
    V* __restrict__ rv = reinterpret_cast<V*>(r);
    // LT: load instruction t of a sub-chunk covers columns t*NG .. t*NG+NG-1; this lane takes
    // column t*NG + lt_cl at vector lt_vl of the sub-chunk
    const int lt_cl = lane / LTT::SV, lt_vl = lane % LTT::SV;
    // B = 64 with long chunks sits at the register limit of 2 waves per SIMD (10 accumulator tiles, 16 operand vectors in
    // flight and their fragments): its 16 column pointers per lane live in LDS and are fetched per sub-chunk (round 4: they
    // cost 32 VGPRs and the kernel spilled 8 dwords to scratch)
    constexpr bool LDSCOL = LT && NG == 4 && KS == NG && sizeof(T) == 8;
    __shared__ const V* s_col[LDSCOL ? 16 * NG : 1];
    const V* lt_col[(LT && !LDSCOL) ? 16 : 1];
    unsigned lt_act = 0;
    if constexpr (LT) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int col = t * NG + lt_cl;
            if (col < nb) lt_act |= 1u << t;
            if constexpr (!LDSCOL) lt_col[t] = reinterpret_cast<const V*>(X + idx[pos0 + (col < nb ? col : 0)] * ld);
        }
        if constexpr (LDSCOL) {
            if (threadIdx.x < 16 * NG) s_col[threadIdx.x] = reinterpret_cast<const V*>(X + idx[pos0 + ((int)threadIdx.x < nb ? (int)threadIdx.x : 0)] * ld);
            __syncthreads();
        }
    }

    // fp32 storage: each chunk (256 rows) is accumulated by v_mfma_f32_16x16x4_f32 from zero and
    // folded into the fp64 running tiles at the end of the chunk -- half the matrix-pipe time of
    // converting to fp64 first, and each fp32 partial sum spans 256 rows only (far tighter than the
    // reference's own Float32 running sums over all n rows).  Its D fragment has rows 4g + q
    // where the fp64 instruction has g + 4q; the running tiles simply live in that layout.
    constexpr bool F32 = sizeof(T) == 4;
    // X_I'r: for B >= 32 it is NOT taken on the matrix pipe -- as an MFMA it broadcasts r over 16
    // columns and wastes 15/16 of the instruction; one vector FMA per element into a per-lane
    // partial (summed over the lane's row group at the end) leaves the pipe to the Gram tiles
    // (5 -> 3 MFMAs per 4 rows at B = 32: measured -4 % at 1.25e6 rows).  At B = 16 the pipe has
    // room and the MFMA form measured ~5 % faster at 1e7 rows, so it keeps one c tile.
    constexpr bool CV = (NG >= 2);
    dvec4 tile[R::NT], ctile[NG];
    double cacc[NG];
#pragma unroll
    for (int t = 0; t < R::NT; ++t) tile[t] = dvec4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NG; ++t) { cacc[t] = 0.0; ctile[t] = dvec4{0.0, 0.0, 0.0, 0.0}; }
    double qacc = 0.0;

    static_assert(KS == NG || (LT && KS == 1), "short chunks need the LDS-transposed operand path");
    constexpr int CVN = LT ? LTT::SV * KS : 64;   // vectors per chunk
    constexpr int CGN = 64 / CVN;                         // phase-A column groups
    const int pa_cg = lane / CVN, pa_v = lane % CVN;
    const int64_t nchunks = (nvec + CVN - 1) / CVN;
    // chunk -> (block, wave): the block index runs fastest, so consecutive chunks go to different
    // blocks and no block gets more than one chunk above any other (a block-major split would
    // hand the last partial round to a few blocks, four chunks each: ~10 % tail at 1e6 rows)
    for (int64_t ch = (int64_t)wave * gridDim.x + blockIdx.x; ch < nchunks;
         ch += (int64_t)gridDim.x * kGramWaves) {
        const int64_t v0 = ch * CVN;
        // ---- phase A (coalesced): r' = r - sum_i h_i X_prev,i on this wave's chunk vectors -----
        const int64_t jv = v0 + pa_v;
        const bool inb = jv < nvec;
        V rr = (inb && pa_cg == 0) ? rv[jv] : vzero((V*)nullptr);
        if (anyp) {
            double re[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) re[e] = (double)rr[e];
            constexpr int PG = (LT && NG == 4) ? 8 : CDH_PG(NG);   // independent loads in flight per group
            for (int i0 = 0; i0 < nzp; i0 += PG * CGN) {
                V xp[PG];
#pragma unroll
                for (int t = 0; t < PG; ++t) {
                    const int at = i0 + t * CGN + pa_cg;
                    const int ii = (at < nzp) ? at : nzp - 1;
                    xp[t] = inb ? ld_stream<NT_>(reinterpret_cast<const V*>(X + s_kp[ii] * ld) + jv)
                                : vzero((V*)nullptr);
                }
#pragma unroll
                for (int t = 0; t < PG; ++t) {
                    const int at = i0 + t * CGN + pa_cg;
                    const double h = (at < nzp) ? s_hp[at] : 0.0;
#pragma unroll
                    for (int e = 0; e < NV; ++e) re[e] = fma(-h, (double)xp[t][e], re[e]);
                }
            }
            if constexpr (CGN > 1) {   // sum the column groups: every lane ends with the full r'
#pragma unroll
                for (int off = CVN; off < 64; off <<= 1)
#pragma unroll
                    for (int e = 0; e < NV; ++e) re[e] += __shfl_xor(re[e], off);
            }
#pragma unroll
            for (int e = 0; e < NV; ++e) rr[e] = (T)re[e];
            if (inb && pa_cg == 0) rv[jv] = rr;
        } else if constexpr (CGN > 1) {
#pragma unroll
            for (int e = 0; e < NV; ++e) rr[e] = (T)__shfl((double)rr[e], pa_v);   // group 0 holds r
        }
        if (pa_cg == 0) {
#pragma unroll
            for (int e = 0; e < NV; ++e) qacc = fma((double)rr[e], (double)rr[e], qacc);
        }
        __builtin_amdgcn_wave_barrier();
        s_r[wave][pa_v] = rr;   // the column groups hold (and write) the same value
        if (w) s_w[wave][pa_v] = inb ? reinterpret_cast<const V*>(w)[jv] : vzero((V*)nullptr);
        __builtin_amdgcn_wave_barrier();
        // ---- phase B: tiles += X_I' X_J, X_I' r' over the chunk's rows, in UH-sized groups of
        // fragment loads (16/NG vector rows at a time so the kernel fits 2 waves per SIMD: the
        // other wave's loads then overlap this wave's MFMA phase) ------------------------------------
        constexpr int UH = LT ? LTT::SV / 4 : CDH_UH(NG);
        fvec4 t32[R::NT], ct32[NG];
        float c32[NG];
        if constexpr (F32) {
#pragma unroll
            for (int t = 0; t < R::NT; ++t) t32[t] = fvec4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NG; ++t) { c32[t] = 0.f; ct32[t] = fvec4{0.f, 0.f, 0.f, 0.f}; }
        }
        // not unrolled: one group's fragment loads in flight at a time keeps the register
        // footprint at 2-3 waves per SIMD (unrolled, the scheduler hoists every group's loads)
#pragma unroll 1
        for (int u0 = 0; u0 < (LT ? UH * KS : 16); u0 += UH) {
            V xf[UH][NG];
            if constexpr (LT) {
                constexpr int XS = LTT::XS, NL = 16;
                V xc[NL];
                const int64_t vsub = v0 + 4 * u0 + lt_vl;       // this lane's vector in the sub-chunk
#pragma unroll
                for (int t = 0; t < NL; ++t) {
                    const V* colp;
                    if constexpr (LDSCOL) colp = s_col[t * NG + lt_cl]; else colp = lt_col[t];
                    xc[t] = (((lt_act >> t) & 1) && vsub < nvec) ? ld_stream<NT_>(colp + vsub) : vzero((V*)nullptr);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < NL; ++t) s_x[wave][(t * NG + lt_cl) * XS + lt_vl] = xc[t];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < UH; ++u)
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp) xf[u][grp] = s_x[wave][(16 * grp + c) * XS + 4 * u + g];
            } else {
#pragma unroll
                for (int u = 0; u < UH; ++u) {
                    const int64_t v = v0 + 4 * (u0 + u) + g;
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp)
                        xf[u][grp] = (act[grp] && v < nvec) ? ld_stream<NT_>(cv[grp] + v) : vzero((V*)nullptr);
                }
            }
#pragma unroll
            for (int u = 0; u < UH; ++u) {
                const V rf = s_r[wave][4 * (u0 + u) + g];
                V wf = rf;
                if (w) wf = s_w[wave][4 * (u0 + u) + g];
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    // A operands carry the observation weight (G = X'WX, c = X'Wr:
                    // cd_differentiable_function.jl:177-182); B operands do not
                    if constexpr (F32) {
                        float a[NG], aw[NG];
                        const float we = w ? (float)wf[e] : 1.f;
#pragma unroll
                        for (int grp = 0; grp < NG; ++grp) {
                            a[grp] = (float)xf[u][grp][e];
                            aw[grp] = a[grp] * we;
                        }
                        const float rb = (float)rf[e];
#pragma unroll
                        for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
                            for (int gj = gi; gj < NG; ++gj)
                                t32[R::tile(gi, gj)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    aw[gi], a[gj], t32[R::tile(gi, gj)], 0, 0, 0);
                            if constexpr (CV) c32[gi] = fmaf(aw[gi], rb, c32[gi]);
                            else ct32[gi] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[gi], rb, ct32[gi], 0, 0, 0);
                        }
                        continue;
                    }
                    double a[NG], aw[NG];
                    const double we = w ? (double)wf[e] : 1.0;
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp) {
                        a[grp] = (double)xf[u][grp][e];
                        aw[grp] = a[grp] * we;
                    }
                    const double rb = (double)rf[e];
#pragma unroll
                    for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
                        for (int gj = gi; gj < NG; ++gj)
                            tile[R::tile(gi, gj)] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                aw[gi], a[gj], tile[R::tile(gi, gj)], 0, 0, 0);
                        if constexpr (CV) cacc[gi] = fma(aw[gi], rb, cacc[gi]);
                        else ctile[gi] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[gi], rb, ctile[gi], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (F32) {
#pragma unroll
            for (int t = 0; t < R::NT; ++t)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) tile[t][q4] += (double)t32[t][q4];
#pragma unroll
            for (int t = 0; t < NG; ++t) {
                cacc[t] += (double)c32[t];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) ctile[t][q4] += (double)ct32[t][q4];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if constexpr (LT) __syncthreads();   // every wave is done with its operand tile before s_red aliases it
    // ---- per-wave record -> LDS -> sum over the block's waves -> block-major partials, one tile
    // at a time (an 8 KB staging buffer instead of the whole record x 4 waves) --------------------
    qacc = wave_sum(qacc);
#pragma unroll
    for (int t = 0; t < R::NT; ++t) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) s_red[wave][(F32 ? 4 * g + q4 : g + 4 * q4) * 16 + c] = tile[t][q4];
        __syncthreads();
        {
            const int v = threadIdx.x;   // 256 threads, 256 tile elements
            double sum = 0.0;
#pragma unroll
            for (int wv = 0; wv < kGramWaves; ++wv) sum += s_red[wv][v];
            partials[(int64_t)blockIdx.x * R::N + t * 256 + v] = sum;   // block-major: 2 KB runs
        }
        __syncthreads();
    }
    // c: lane (c, g) holds the partial of column 16 t + c over its row group (vector form), or
    // column j = 0 of the c tile holds c (matrix form: rows spread over g and the 4 registers;
    // staged as 4 slots per column so both forms share the sum below); q: one per wave
#pragma unroll
    for (int t = 0; t < NG; ++t) {
        if constexpr (CV) {
            s_red[wave][(16 * t + c) * 4 + g] = cacc[t];
        } else {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int row = F32 ? 4 * g + q4 : g + 4 * q4;
                // D[row][col c] == c_row for every c: lane column c < 4 fills slot c of its rows
                if (c < 4) s_red[wave][(16 * t + row) * 4 + c] = (c == 0) ? ctile[t][q4] : 0.0;
            }
        }
    }
    if (lane == 0) s_q[wave] = qacc;
    __syncthreads();
    for (int v = threadIdx.x; v < B + 1; v += blockDim.x) {
        double sum = 0.0;
        if (v < B) {
#pragma unroll
            for (int wv = 0; wv < kGramWaves; ++wv)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) sum += s_red[wv][v * 4 + gg];
        } else {
#pragma unroll
            for (int wv = 0; wv < kGramWaves; ++wv) sum += s_q[wv];
        }
        partials[(int64_t)blockIdx.x * R::N + R::OFF_C + v] = sum;
    }
}

// Stage 1 of the reduction over block-major partials (record b at partials + b * nvals: the
// streaming kernel stores whole 2 KB runs, where value-major stores were one 8-byte store per line and
// cost ~9 us of a 234 us launch).  A block of 16 waves owns 32 consecutive values: half-wave lane =
// value, the 32 half-waves split the records -- group g sums records g, g + 32, ... in that order
// (256-byte loads, 8 in flight) -- and the 32 group sums are added in group order: a fixed order, so
// the result is bit-reproducible.
// (Measured and dropped: letting the thread that holds a reduced value take it straight through the
// direct exchange, p2p_exchange.hpp, instead of a separate exchange launch -- connected to itself the
// fused kernel took 14.9 us against 4.9 + 7.7 us for the two.)
constexpr int kReduceWaves = 16, kReduceVals = 32, kReduceGroups = 2 * kReduceWaves;
__global__ __launch_bounds__(64 * kReduceWaves) void k_gram_reduce(const double* __restrict__ partials,
                                                                   int nparts, int nvals,
                                                                   double* __restrict__ rec) {
    __shared__ double s_part[kReduceGroups][kReduceVals];
    const int lv = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int v = blockIdx.x * kReduceVals + lv;
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0;
    if (v < nvals) {
        const double* col = partials + v;
        int b = grp;
        for (; b + 7 * kReduceGroups < nparts; b += 8 * kReduceGroups) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += col[(int64_t)(b + u * kReduceGroups) * nvals];
        }
        for (; b < nparts; b += kReduceGroups) acc[0] += col[(int64_t)b * nvals];
    }
    s_part[grp][lv] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (threadIdx.x < kReduceVals && v < nvals) {
        double sum = 0.0;
#pragma unroll
        for (int gi = 0; gi < kReduceGroups; ++gi) sum += s_part[gi][lv];
        rec[v] = sum;
    }
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), lane));
}

// Stage 2: the B sequential scalar updates, one wave, lane i owns visit i of the block:
//   b_i = c_i - sum_{s<i} h_s G_si is kept current by a lane-parallel rank-1 update after each
//   visit; q_s = q_{s-1} - 2 h_{s-1} b_{s-1} + h_{s-1}^2 a_{s-1}  (kernels.hpp, same identities).
// The loop is fully unrolled and carries only the dependent chain
//   v = b_s / a_s + beta_s -> S(v, thr_s) -> h_s -> broadcast -> b_j -= h_s G_sj ;
// G's column, 1/a and the threshold are fetched / computed by all lanes before it, and maxH,
// the touched flags and the stores come after it.  `dup` says a coordinate may repeat inside
// the block (only caller-supplied visit lists can do that): later visits then see the new value.
// (the body is shared with k_cov_block below, where the record sits in LDS and wave 0 of a wider block runs it)
template <int NG>
__device__ __forceinline__ void gram_scalar_body(const double* __restrict__ rec_g, int nb, int dup,
                                                 Ctrl* ctrl, double* beta,
                                                 const double* __restrict__ omega,
                                                 const int64_t* __restrict__ idx, double* hs,
                                                 double* newval, int32_t* touched, int pos0,
                                                 double* qs /* sqrt-lasso: r'r after each visit */, const int lane) {
    using R = GramRec<NG>;
    constexpr int B = R::B;
    const bool mine = lane < nb;
    const int li = mine ? lane : 0;
    // independent loads first: this lane's column of G, c, a, the iterate and the weights
    const int64_t k_me = idx[pos0 + li];
    double gcol[B];
#pragma unroll
    for (int s = 0; s < B; ++s)   // G[s][li] for s < li; 0 after the lane's own visit, which freezes b_me there
        gcol[s] = (s < li) ? rec_g[R::g(s < li ? s : li, li)] : 0.0;
    double b_me = rec_g[R::OFF_C + li];
    const double a_me = rec_g[R::g(li, li)];
    double q = rec_g[R::OFF_Q];
    const int loss = ctrl->loss;
    const double lambda0 = ctrl->lambda0, n_total = ctrl->n_total;
    const double maxH0 = ctrl->maxH;
    const double om_me = ctrl->has_omega ? omega[k_me] : 1.0;
    double old_me = beta[k_me];
    const double ia_me = 1.0 / a_me;
    const double thr_me = lambda0 * om_me * (n_total * ia_me);
    int dom = 0;
    double v_me = 0.0, nv_me = 0.0, h_me = 0.0, q_me = 0.0;
    if (loss != 1 && !dup) {
        // least squares, no repeated coordinate (every scheduler-made pass): the chain is fma ->
        // soft-threshold -> subtract -> broadcast -> fma and nothing else.  Lanes outside the block
        // never move (threshold = inf); b stays frozen after a lane's own visit (zeros in gcol), so
        // each lane recomputes its own visit after the loop instead of latching it inside.
        const double inf = __builtin_huge_val();
        const double ia_f = mine ? ia_me : 0.0, thr_f = mine ? thr_me : inf, old_f = mine ? old_me : 0.0;
        double b_f = mine ? b_me : 0.0;
#pragma unroll
        for (int s = 0; s < B; ++s) {
            if (s < nb) {   // wave-uniform
                const double nv = soft_threshold(fma(b_f, ia_f, old_f), thr_f);
                const double h_s = readlane_f64(nv - old_f, s);
                b_f = fma(-h_s, gcol[s], b_f);
            }
        }
        v_me = fma(b_f, ia_f, old_f);
        nv_me = soft_threshold(v_me, thr_f);
        h_me = nv_me - old_f;
    } else
#pragma unroll
    for (int s = 0; s < B; ++s) {
        if (s < nb) {   // wave-uniform
            double v, nv;
            if (loss == 1 /* CDH_SQRT */) {
                const VisitOut o = visit_update_rcp(loss, lambda0, n_total, a_me, ia_me, b_me, q, old_me, om_me);
                v = 0.0; nv = o.nv;
                dom |= __builtin_amdgcn_readlane(o.dom, s);
            } else {
                v = fma(b_me, ia_me, old_me);
                nv = soft_threshold(v, thr_me);
            }
            const double hc = nv - old_me;
            const double h_s = readlane_f64(hc, s);
            if (lane == s) { v_me = v; nv_me = nv; h_me = hc; }
            if (loss == 1) {
                const double b_s = readlane_f64(b_me, s), a_s = readlane_f64(a_me, s);
                q = q - 2.0 * h_s * b_s + h_s * h_s * a_s;
                if (q < 0.0) q = 0.0;
                if (lane == s) q_me = q;
            }
            if (dup) {
                const int64_t k_s = (int64_t)readlane_u64((uint64_t)k_me, s);
                const double nv_s = readlane_f64(nv, s);
                if (lane > s && k_me == k_s) old_me = nv_s;
            }
            b_me = fma(-h_s, gcol[s], b_me);   // only lanes > s still use b_me
        }
    }
    // maxH over the block's visits, then the stores.  A NaN h never raises maxH, as in the reference's
    // `abs(h) > maxH` (coordinate_descent.jl:104-106)
    double ah = mine ? fabs(h_me) : 0.0;
    double m = (ah != ah) ? 0.0 : ah;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if (lane == 0) {
        double mh = maxH0;
        if (m > mh) mh = m;
        ctrl->maxH = mh;
        if (dom) ctrl->domain_error = 1;
        if (loss == 1) ctrl->q_carry = q;   // r'r after the block's visits (every lane carries the same recurrence)
    }
    bool last = true;                                // last writer of a repeated coordinate wins
    if (dup) {
        for (int j = 1; j < nb; ++j) {
            const int64_t k_j = (int64_t)readlane_u64((uint64_t)k_me, j);
            if (j > lane && k_j == k_me) last = false;
        }
    }
    if (mine) {
        hs[pos0 + lane] = h_me;
        newval[pos0 + lane] = nv_me;
        touched[pos0 + lane] = (loss == 1) ? 0 : ((v_me != 0.0) ? 1 : 0);
        if (last) beta[k_me] = nv_me;
        if (qs && loss == 1) qs[pos0 + lane] = q_me;
    }
}
template <int NG>
__global__ __launch_bounds__(64) void k_gram_scalar(const double* __restrict__ rec_g, int nb, int dup,
                                                    Ctrl* ctrl, double* beta,
                                                    const double* __restrict__ omega,
                                                    const int64_t* __restrict__ idx, double* hs,
                                                    double* newval, int32_t* touched, int pos0,
                                                    double* qs = nullptr /* sqrt-lasso: r'r after each visit */) {
    gram_scalar_body<NG>(rec_g, nb, dup, ctrl, beta, omega, idx, hs, newval, touched, pos0, qs, (int)threadIdx.x);
}

// Cross products of ALL p columns with up to 32 columns (B) in ONE launch, one pass over X:
//   out[cg][(2 ta + tb) * 256 + i * 16 + j] = X_{64 cg + 16 ta + i}' X_{B[16 tb + j]}
// The Gram COLUMNS the gradient cache keeps for the coordinates that move (cdhip.hip, GradCache): with
// G_j = X'X_j for every moved j, X_k'r is known for every k without reading X again.  Same use of the
// matrix pipe as k_gramstep -- a 16 x 16 fp64 accumulator tile in 8 registers, rows of X as the K
// dimension -- with separate A and B operands (4 x 2 tiles per wave).  Grid: blockIdx.x = lane of SUPER-groups
// (256 consecutive columns: one group of 64 per wave; the block walks the super-groups x, x + gridDim.x, ... one
// after the other), blockIdx.y = row lane (the block takes the row slabs y, y + gridDim.y, ... of 1024 vectors);
// a record per (column group, row lane), summed by k_cross_reduce.  fp32 storage is widened to fp64 on the way
// into the MFMA (an occasional pass, not the sweep).
//
// How it got here (n = 2e6, p = 5000, fp64: 80 GB; one batch of 32 columns; tools/cross_bench.hip holds the earlier
// forms).  Round 2: fragment-shaped A loads straight from HBM -- a wave instruction takes 64 B from each of 16 columns,
// so every 128-B line is asked for twice and each quarter-wave touches 16 different lines for 16 bytes apiece -- with
// the B chunk through LDS: 19.3-21.3 ms = 3.8-4.1 TB/s of X.  Round 3: BOTH operands by LDS-DMA in whole lines:
//   * one `global_load_lds_dwordx4` moves 8 columns x 8 vectors (8 whole 128-B lines) straight into LDS, no VGPRs on
//     the way; an LDS-DMA writes lane l at (wave-uniform base) + 16 l, so the image cannot be padded: the bank
//     conflicts of the MFMA-fragment reads are removed on the SOURCE side instead -- lane (column cl = l / 8, slot
//     j = l % 8) fetches vector j ^ (cl & 6) of its line, i.e. vector v of column cc sits in slot 8 cc + (v ^ (cc & 6)),
//     and the ds_read_b128 of lane (c, g) -- column 16 t + c, vector 4 u + g -- is conflict-free in each of the
//     instruction's four lane groups (checked exhaustively in tests/test_abi_and_host.py);
//   * a wave's unit of work is a HALF sub-chunk: 32 of its 64 A columns x 8 vectors (16 rows of fp64) = 4 pieces = 4 KB,
//     in a ring of R slots; the block's B sub-chunk (32 columns x 8 vectors) is 4 pieces, ONE PER WAVE, shared by the
//     four waves and riding along with the even halves; while slot s is multiplied the next R - 1 are in flight, and
//     slot s + R is issued as soon as s has been read into registers;
//   * the waits are COUNTED (`s_waitcnt vmcnt(N)`: the loads of the younger slots stay out) and, like the DMAs
//     themselves, written in assembly -- hipcc drains every LDS-DMA it knows of (vmcnt(0)) ahead of an LDS read it
//     cannot prove disjoint, which serialises load and matrix phases; it knows nothing of an asm load.  One block
//     barrier per sub-chunk (the B pieces of all four waves have landed; everyone is done with the B buffer about to
//     be refilled).
// Measured (tools/cross_bench.hip, profiles/r3_cross_*): 15.3 ms with one sub-chunk in flight per wave and two
// blocks per CU -- and the SAME with three slots in flight: not latency.  Loads alone take 13.0 ms (6.16 TB/s), the
// matrix work alone 9.6 ms, each at 2.39 GHz; together the chip holds 1.84-1.87 GHz (in-kernel s_memtime /
// s_memrealtime): the two pipes are paid for out of one power budget, so what helps is whatever lets the pass finish
// in fewer joules -- non-temporal A loads (-3 %), a shallower ring at THREE blocks per CU (R = 2: 44 KB of LDS;
// -6 %), priority for the short read / issue stretches over the partner waves' MFMA phases (-1 %):
// 14.4 ms = 5.57 TB/s of X (0.70 of the 8 TB/s peak), fp32 storage 11.1 ms (matrix-bound: 10.3 ms of MFMA on 40 GB).
// Rows past nvec inside a column are the zero pad (ld is a multiple of 32 elements = whole sub-chunks); columns past
// p and B columns past nbc are clamped to valid ones and their results never read.  Observation weights
// (G = X'WX, CDWeightedLSLoss): the weight of each row multiplies the B fragment on its way into the MFMA (a 128-byte
// piece of w per sub-chunk, fetched by every wave to the same place so that all waves count the same loads).
constexpr int kCrossTA = 4, kCrossTB = 2, kCrossA = 16 * kCrossTA, kCrossB = 16 * kCrossTB,
              kCrossRec = kCrossTA * kCrossTB * 256;
constexpr int kCrossSlab = 1024;         // vectors per row slab (fp64: 2048 rows)
constexpr int kCrossOcc = 3;             // blocks per CU of k_cross (its 44 KB of LDS and 151 registers allow three)
// ... of the fp64 kernel.  The fp32 one keeps fp32 tiles next to the fp64 running tiles (32 more registers): two blocks per CU
template <typename T> constexpr int cross_occ() { return sizeof(T) == 4 ? 2 : kCrossOcc; }
constexpr int kX2SV = 8;                          // vectors per sub-chunk
constexpr int kCrossFold = 8;                     // fp32 storage: sub-chunks (of 32 rows) per fp32 partial sum
constexpr int kX2BSlots = kCrossB * kX2SV;        // 16-byte slots of the block's B sub-chunk
constexpr int kX2SPS = kCrossSlab / kX2SV;        // sub-chunks per row slab
template <int AUX>
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    const unsigned lds_dst_uniform = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst);   // the "s" operand must be an SGPR
    if constexpr (AUX == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_uniform(const void* p) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(const __attribute__((address_space(3))) void*)p);
}
template <typename T, bool HASW, int AUX_A = 2, int R = 2, int OCC = kCrossOcc, bool PRIO = true>
__global__ __launch_bounds__(64 * kGramWaves, OCC) void k_cross(const T* __restrict__ X, int64_t ld, int64_t nvec, int64_t p,
                                                               const int64_t* __restrict__ bcols, int nbc,
                                                               const T* __restrict__ w, double* __restrict__ partials) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int HS = 32 * kX2SV;                    // 16-byte slots of one ring slot (32 columns x 8 vectors)
    constexpr int EV = 5 + (HASW ? 1 : 0), OD = 4;    // DMA instructions a wave issues per even / odd ring slot
    static_assert(R == 2 || R == 4 || R == 6, "ring depth");
    constexpr int NB = R / 2 + 1;                     // B pieces in flight + the one being read
    __shared__ V s_a[kGramWaves][R][HS];
    __shared__ V s_b[NB][kX2BSlots];
    __shared__ V s_w[HASW ? NB : 1][HASW ? kX2SV : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int cl = lane >> 3, sj = (lane & 7) ^ (cl & 6);
    const int64_t nslabs = (nvec + kCrossSlab - 1) / kCrossSlab;
    const int64_t ngroups = (p + kCrossA - 1) / kCrossA, nsuper = (ngroups + kGramWaves - 1) / kGramWaves;
    // Which (super-group lane, row lane) this workgroup takes.  Workgroups go to the 8 XCDs round-robin by their linear id, and
    // the gridDim.x blocks of one row lane read the SAME B sub-chunks (32 columns x their rows) at about the same time: placed
    // on one XCD they share them through its L2 instead of each fetching them over the fabric (round 3 counted 89.8 GB fetched
    // per 80 GB of X: the B pieces, once per 256-column super-group).  Speed only, never correctness.
    int bx = blockIdx.x, by = blockIdx.y;
    if (gridDim.y % 8 == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, j = lin >> 3;
        bx = j % gridDim.x;
        by = (j / gridDim.x) * 8 + xcd;
    }
    const int64_t my_slabs = nslabs > by ? (nslabs - by + gridDim.y - 1) / gridDim.y : 0;
    // sub-chunks this block's row lane really has: only the very last slab of the column can be partial
    int64_t nq = my_slabs * kX2SPS;
    if (my_slabs > 0 && (int64_t)by + (my_slabs - 1) * gridDim.y == nslabs - 1)
        nq -= kX2SPS - ((nvec - (nslabs - 1) * kCrossSlab) + kX2SV - 1) / kX2SV;
    auto sub_v0 = [&](int64_t q) { return ((int64_t)by + (q / kX2SPS) * gridDim.y) * kCrossSlab + (q % kX2SPS) * kX2SV; };
    const int bcol_i = 8 * wave + cl;
    const V* bsrc = reinterpret_cast<const V*>(X + bcols[bcol_i < nbc ? bcol_i : 0] * ld) + sj;
    const unsigned a_base = lds_addr_uniform(&s_a[wave][0][0]), b_base = lds_addr_uniform(&s_b[0][0]) + 1024u * (unsigned)wave,
                   w_base = lds_addr_uniform(&s_w[0][0]);
    int f_slot[2][2];                                  // fragment slots inside a ring slot / the B image: [tile][u]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) f_slot[t][u] = (16 * t + c) * kX2SV + ((4 * u + g) ^ ((16 * t + c) & 6));
    for (int64_t sg = bx; sg < nsuper; sg += gridDim.x) {
        const int64_t cg = sg * kGramWaves + wave;
        const int64_t a0 = cg * kCrossA;
        const V* asrc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t col = a0 + 8 * i + cl;
            if (col >= p) col = p - 1;
            asrc[i] = reinterpret_cast<const V*>(X + col * ld) + sj;
        }
        // half h of sub-chunk q -> ring slot (2 q + h) & 3; with the even halves, the block's B piece of the sub-chunk
        auto issue = [&](int64_t q, int h) {
            const int64_t v0 = sub_v0(q);
            const unsigned dst = a_base + 4096u * (unsigned)((2 * q + h) % R);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_asm<AUX_A>(asrc[4 * h + i] + v0, dst + 1024u * (unsigned)i);
            if (h == 0) {
                glds16_asm<0>(bsrc + v0, b_base + 4096u * (unsigned)(q % NB));
                if constexpr (HASW) {
                    // every wave fetches the same 128 bytes to the same place: the counted waits need one count per wave
                    if (lane < kX2SV) glds16_asm<0>(reinterpret_cast<const V*>(w) + v0 + lane, w_base + 128u * (unsigned)(q % NB));
                }
            }
        };
        dvec4 tile[kCrossTA * kCrossTB];
        // fp32 storage (round 4): the products run on the fp32 matrix pipe -- v_mfma_f32_16x16x4_f32, twice the rate of the fp64
        // instruction the widened operands took (11.1 ms per 40 GB, matrix-bound) -- into fp32 tiles that start from zero every
        // kCrossFold sub-chunks (256 rows) and are folded into the fp64 running tiles, as k_gramstep does.  Its D fragment has
        // rows 4 g + q where the fp64 instruction has g + 4 q: the running tiles live in that layout (the store below knows).
        fvec4 t32[F32 ? kCrossTA * kCrossTB : 1];
#pragma unroll
        for (int t = 0; t < kCrossTA * kCrossTB; ++t) tile[t] = dvec4{0.0, 0.0, 0.0, 0.0};
        if constexpr (F32) {
#pragma unroll
            for (int t = 0; t < kCrossTA * kCrossTB; ++t) t32[t] = fvec4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < R / 2; ++j)
            if (j < nq) { issue(j, 0); issue(j, 1); }
        for (int64_t q = 0; q < nq; ++q) {
            const int bq = (int)(q % NB);
            V af[2][2], bf[2][kCrossTB], wf[2];
            // ---- even half: this wave's slot 2q and B piece have landed (the three younger slots may still fly);
            // the barrier: so have the other waves' B pieces, and everyone is done with B(q - 1) = the buffer of B(q + 2)
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            // (loads per slot: even EV = 4 A pieces + the B piece (+ the weights' piece), odd OD = 4 A pieces)
            if (q + R / 2 - 1 < nq) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(OD + (R / 2 - 1) * (EV + OD)) : "memory");   // younger: 2q+1 .. 2q+R-1
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(OD) : "memory");                          // at least 2q+1 is younger (a stricter wait is safe)
            __builtin_amdgcn_s_barrier();
            {
                const V* ring = &s_a[wave][(2 * q) % R][0];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) af[u][t] = ring[f_slot[t][u]];
#pragma unroll
                    for (int t = 0; t < kCrossTB; ++t) bf[u][t] = s_b[bq][f_slot[t][u]];
                    if constexpr (HASW) wf[u] = s_w[bq][4 * u + g];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the reads have returned: the slot may be refilled
            if (q + R / 2 < nq) issue(q + R / 2, 0);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            using BT = std::conditional_t<F32, float, double>;    // the type the products are taken in
            BT b[2][NV][kCrossTB];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < NV; ++e)
#pragma unroll
                    for (int gb = 0; gb < kCrossTB; ++gb) {
                        b[u][e][gb] = (BT)bf[u][gb][e];
                        if constexpr (HASW) b[u][e][gb] *= (BT)wf[u][e];
                    }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < NV; ++e)
#pragma unroll
                    for (int ga = 0; ga < 2; ++ga) {
                        const BT a = (BT)af[u][ga][e];
#pragma unroll
                        for (int gb = 0; gb < kCrossTB; ++gb) {
                            if constexpr (F32) t32[ga * kCrossTB + gb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[u][e][gb], t32[ga * kCrossTB + gb], 0, 0, 0);
                            else tile[ga * kCrossTB + gb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[u][e][gb], tile[ga * kCrossTB + gb], 0, 0, 0);
                        }
                    }
            // ---- odd half: slot 2q + 1 (younger: 2q+2, 2q+3, and 2q+4, just issued) ----------------------------------
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            if (q + R / 2 < nq) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((R / 2 - 1) * (EV + OD) + EV) : "memory");   // younger: 2q+2 .. 2q+R
            else if (q + R / 2 - 1 < nq) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((R / 2 - 1) * (EV + OD)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            {
                const V* ring = &s_a[wave][(2 * q + 1) % R][0];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t) af[u][t] = ring[f_slot[t][u]];
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (q + R / 2 < nq) issue(q + R / 2, 1);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < NV; ++e)
#pragma unroll
                    for (int ga = 0; ga < 2; ++ga) {
                        const BT a = (BT)af[u][ga][e];
#pragma unroll
                        for (int gb = 0; gb < kCrossTB; ++gb) {
                            if constexpr (F32) t32[(2 + ga) * kCrossTB + gb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[u][e][gb], t32[(2 + ga) * kCrossTB + gb], 0, 0, 0);
                            else tile[(2 + ga) * kCrossTB + gb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[u][e][gb], tile[(2 + ga) * kCrossTB + gb], 0, 0, 0);
                        }
                    }
            if constexpr (F32) {
                if ((q % kCrossFold) == kCrossFold - 1 || q == nq - 1) {
#pragma unroll
                    for (int t = 0; t < kCrossTA * kCrossTB; ++t) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) tile[t][q4] += (double)t32[t][q4];
                        t32[t] = fvec4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        double* s_red = reinterpret_cast<double*>(&s_a[wave][0][0]);      // R x 4 KB per wave
#pragma unroll
        for (int t = 0; t < kCrossTA * kCrossTB; ++t) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) s_red[(F32 ? 4 * g + q4 : g + 4 * q4) * 16 + c] = tile[t][q4];   // D[i = g + 4 q4][j = c] (fp32 pipe: i = 4 g + q4)
            __builtin_amdgcn_wave_barrier();
            if (cg < ngroups) {
                double* __restrict__ out = partials + (cg * gridDim.y + by) * kCrossRec + t * 256;
#pragma unroll
                for (int v = lane; v < 256; v += 64) out[v] = s_red[v];
            }
            __builtin_amdgcn_wave_barrier();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}
// out[cg][v] = sum over the J row-slab records of column group cg, in a fixed order
__global__ __launch_bounds__(256) void k_cross_reduce(const double* __restrict__ partials, int J, double* __restrict__ out) {
    const int v = blockIdx.x * 256 + threadIdx.x;          // < kCrossRec
    const double* src = partials + (int64_t)blockIdx.y * J * kCrossRec + v;
    double s0 = 0.0, s1 = 0.0;
    int j = 0;
    for (; j + 1 < J; j += 2) { s0 += src[(int64_t)j * kCrossRec]; s1 += src[(int64_t)(j + 1) * kCrossRec]; }
    if (j < J) s0 += src[(int64_t)j * kCrossRec];
    out[(int64_t)blockIdx.y * kCrossRec + v] = s0 + s1;
}

// ---- covariance-form visits (cdhip.hip / grad_cache.hpp, "cov chunks") --------------------------------------
// While the gradient cache holds g = X'r and the Gram columns G_j = X'X_j of every coordinate of a chunk,
// the block record k_gram_scalar consumes -- c_i = X_ki'r, G_si = X_ks'X_ki, a_i = G_ii -- can be read off
// the cache instead of being streamed from X: no HBM traffic for the visit at all.  k_cov_block gathers the
// record of visits [pos0, pos0 + nb) and runs the same B sequential updates as in the streamed
// sweep; k_cov_gupdate then applies g -= sum_i h_i G_ki to all p entries.  The residual is brought up to
// date later, once (k_multi_axpy over the coordinates that moved).  Least squares and sqrt-lasso (whose r'r
// travels from block to block in the control block: q <- q - 2 h b + h^2 a, the recurrence the streamed
// blocks use inside a block), fp64.
// k_cov_block: gathers the record of visits [pos0, pos0 + nb) -- G[s][j] for s <= j < nb, c_j = g_kj, the carried r'r -- into
// LDS with the whole block, then wave 0 runs k_gram_scalar's B sequential updates on it (one launch: the covariance-form
// blocks are launch latency and little else; rounds 2-3 had the gather in a kernel of its own, k_cov_record, 2.5 - 7.5 us
// per block).  Same arithmetic in the same order.
template <int NG>
__global__ __launch_bounds__(256) void k_cov_block(const double* __restrict__ g, const double* __restrict__ Gcols,
                                                   const int32_t* __restrict__ slot, int64_t p,
                                                   const int64_t* __restrict__ idx, int pos0, int nb, int dup,
                                                   Ctrl* ctrl, double* beta, const double* __restrict__ omega,
                                                   double* hs, double* newval, int32_t* touched, double* qs,
                                                   double* __restrict__ q_block /* r'r the block starts from: k_cov_gupdate_chk reads it */) {
    using R = GramRec<NG>;
    constexpr int B = R::B;
    __shared__ int64_t s_k[B];
    __shared__ int64_t s_off[B];
    __shared__ double s_rec[R::N];
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const int64_t k = idx[pos0 + (i < nb ? i : 0)];
        s_k[i] = k;
        s_off[i] = (int64_t)slot[k] * p;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < B * B; e += blockDim.x) {
        const int sI = e / B, j = e % B;
        if (sI <= j && j < nb) s_rec[R::g(sI, j)] = Gcols[s_off[j] + s_k[sI]];
    }
    for (int i = threadIdx.x; i < B; i += blockDim.x) s_rec[R::OFF_C + i] = (i < nb) ? g[s_k[i]] : 0.0;
    if (threadIdx.x == 0) { const double q0 = ctrl->q_carry; s_rec[R::OFF_Q] = q0; *q_block = q0; }
    __syncthreads();
    if (threadIdx.x < 64) gram_scalar_body<NG>(s_rec, nb, dup, ctrl, beta, omega, idx, hs, newval, touched, pos0, qs, (int)threadIdx.x);
}

// ---- a whole full pass from the cache, on the device (grad_cache.hpp: gc_pass_device) ------------------------------
// A visit of coordinate k is SETTLED when the exact visit would leave beta_k at zero and r untouched:
//   beta_k == 0, a_k > 0 (a zero column goes to the exact path: the reference turns it into NaN) and
//   |g_k| <= thr_k (1 - 1e-9) - cert_abs sqrt(a_k),   thr_k = lambda0 omega_k n   (sqrt-lasso: lambda0 omega_k ||r||).
__device__ __forceinline__ bool cov_settled(double g, double a, double beta, double thr_base, double om, double cert_abs) {
    return beta == 0.0 && a > 0.0 && fabs(g) <= thr_base * om * (1.0 - 1e-9) - cert_abs * sqrt(a);
}
struct CovScanOut { int32_t count, nzero, bad_pos, pad; };
// One block: classifies the m positions of the pass against the gradient as it stands, compacts the UNSETTLED ones in
// visit order (upos[j] = position, vis[j] = coordinate: the visit list of the covariance-form blocks that follow),
// records for every coordinate its position and whether it was settled (the re-check below needs both), counts the
// settled coordinates whose gradient is exactly zero (the host's SparseIterate bookkeeping differs for those: it then
// takes the pass the slow way), and snapshots g and beta so that a pass whose re-check fails can be undone.
__global__ __launch_bounds__(1024) void k_cov_scan(const double* __restrict__ g, const double* __restrict__ a,
                                                   const double* __restrict__ beta, const double* __restrict__ omega,
                                                   const Ctrl* ctrl, const int64_t* __restrict__ idx, int m, int64_t p,
                                                   int32_t* __restrict__ pos_of, uint8_t* __restrict__ setflag,
                                                   int32_t* __restrict__ upos, int64_t* __restrict__ vis,
                                                   double* __restrict__ g_snap, double* __restrict__ beta_snap,
                                                   CovScanOut* out, const uint8_t* __restrict__ forced /* visited whatever their certificate says */) {
    __shared__ int s_wcnt[16];
    __shared__ int s_base, s_nzero;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int64_t k = tid; k < p; k += 1024) { g_snap[k] = g[k]; beta_snap[k] = beta[k]; pos_of[k] = -1; setflag[k] = 0; }
    if (tid == 0) { s_base = 0; s_nzero = 0; }
    const int loss = ctrl->loss, has_omega = ctrl->has_omega;
    const double thr_base = ctrl->lambda0 * (loss == 1 ? sqrt(ctrl->q_carry) : ctrl->n_total), cert_abs = ctrl->cert_abs;
    __syncthreads();
    for (int i0 = 0; i0 < m; i0 += 1024) {
        const int i = i0 + tid;
        const bool valid = i < m;
        const int64_t k = valid ? idx[i] : 0;
        const double gk = g[k];
        const bool st = valid && forced[k] == 0 && cov_settled(gk, a[k], beta[k], thr_base, has_omega ? omega[k] : 1.0, cert_abs);
        const bool uns = valid && !st;
        if (valid) { pos_of[k] = i; setflag[k] = st ? 1 : 0; }
        if (st && gk == 0.0) atomicAdd(&s_nzero, 1);
        const unsigned long long mask = __ballot(uns);
        if (lane == 0) s_wcnt[wave] = __popcll(mask);
        __syncthreads();
        int before = s_base;
        for (int wv = 0; wv < wave; ++wv) before += s_wcnt[wv];
        if (uns) {
            const int j = before + __popcll(mask & ((1ull << lane) - 1ull));
            upos[j] = i; vis[j] = k;
        }
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int wv = 0; wv < 16; ++wv) tot += s_wcnt[wv]; s_base += tot; }
        __syncthreads();
    }
    if (tid == 0) { out->count = s_base; out->nzero = s_nzero; out->bad_pos = 0x7fffffff; out->pad = 0; }
}
// beta and g back to the snapshot (a pass whose re-check failed never happened)
__global__ __launch_bounds__(256) void k_cov_restore(double* __restrict__ g, double* __restrict__ beta,
                                                     const double* __restrict__ g_snap, const double* __restrict__ beta_snap, int64_t p) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < p) { g[k] = g_snap[k]; beta[k] = beta_snap[k]; }
}
// k_cov_gupdate for a block of a device-side full pass: g -= sum_i h_i G_ki as before, and on the way the RE-CHECK of the
// positions that were skipped as settled: coordinate k's certificate was read before this pass's moves, so it is tested
// again against the gradient as it stands when its turn comes -- after the moves of the visits before position
// pos_of[k], before those after it (sqrt-lasso: with ||r|| as it stands then, qs[] = r'r after each visit).  The block
// answers for the skipped positions in (position of the previous block's last visit, position of its own last visit],
// the last block of the pass also for those after its last visit.  A certificate that no longer holds is reported as
// the smallest such position (out->bad_pos) and its coordinate marked in forced[]: the host undoes the pass and runs it
// again with the marked coordinates on the visit list (visiting more than necessary is always right; everything from the
// first broken position on was computed on a wrong trajectory, so later marks may be spurious: harmless), and only after
// a few such rounds walks it the careful way.
__global__ __launch_bounds__(256) void k_cov_gupdate_chk(double* __restrict__ g, const double* __restrict__ Gcols,
                                                         const int32_t* __restrict__ slot, const double* __restrict__ a,
                                                         const double* __restrict__ omega, const Ctrl* ctrl, int64_t p,
                                                         const int64_t* __restrict__ vis, const double* __restrict__ hs,
                                                         const double* __restrict__ qs, const int32_t* __restrict__ upos,
                                                         const int32_t* __restrict__ pos_of, const uint8_t* __restrict__ setflag,
                                                         int j0, int nb, int m, int last_block, const double* __restrict__ q_start,
                                                         CovScanOut* out, uint8_t* __restrict__ forced) {
    __shared__ double s_h[64], s_q[64];
    __shared__ int64_t s_off[64];
    __shared__ int s_pos[64];
    __shared__ int s_n;
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        double h = 0.0, q = 0.0;
        int64_t off = 0;
        int pos = 0;
        if (i < nb) { h = hs[j0 + i]; off = (int64_t)slot[vis[j0 + i]] * p; pos = upos[j0 + i]; q = qs[j0 + i]; }
        const bool nz = (h != 0.0);
        const unsigned long long mask = __ballot(nz);
        if (nz) {
            const int at = __popcll(mask & ((1ull << i) - 1ull));
            s_h[at] = h; s_off[at] = off; s_pos[at] = pos; s_q[at] = q;
        }
        if (i == 0) s_n = __popcll(mask);
    }
    __syncthreads();
    const int nmove = s_n;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p) return;
    const int p_prev = j0 > 0 ? upos[j0 - 1] : -1, p_last = last_block ? m - 1 : upos[j0 + nb - 1];
    const int t = pos_of[k];
    bool need = setflag[k] != 0 && t > p_prev && t <= p_last;
    if (nmove == 0 && !need) return;
    const int loss = ctrl->loss;
    double acc = g[k], q_run = *q_start;
    double cert_scale = 0.0, cert_off = 0.0;
    if (need) {
        cert_scale = ctrl->lambda0 * (ctrl->has_omega ? omega[k] : 1.0) * (1.0 - 1e-9);
        cert_off = ctrl->cert_abs * sqrt(a[k]);
    }
    const double n_total = ctrl->n_total;
    auto holds = [&](double gv, double qv) { return fabs(gv) <= cert_scale * (loss == 1 ? sqrt(qv) : n_total) - cert_off; };
    for (int i = 0; i < nmove; ++i) {
        if (need && s_pos[i] > t) {
            if (!holds(acc, q_run)) { atomicMin(&out->bad_pos, t); forced[k] = 1; }
            need = false;
        }
        acc = fma(-s_h[i], Gcols[s_off[i] + k], acc);
        q_run = s_q[i];
    }
    if (need && !holds(acc, q_run)) { atomicMin(&out->bad_pos, t); forced[k] = 1; }
    if (nmove) g[k] = acc;
}

// Everything a covariance-form pass sends back, gathered into one block: [Ctrl: 8 doubles][CovScanOut: 2][h: m][newval: m]
// [touched: m int32] -- one copy over the bus instead of five (each costs the host several microseconds to enqueue)
constexpr int kPackHead = 10;
__global__ __launch_bounds__(256) void k_cov_pack(const Ctrl* ctrl, const CovScanOut* scan, const double* __restrict__ hs,
                                                  const double* __restrict__ newval, const int32_t* __restrict__ touched, int m,
                                                  double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        *reinterpret_cast<Ctrl*>(out) = *ctrl;
        if (scan) *reinterpret_cast<CovScanOut*>(out + 8) = *scan;
    }
    if (i < m) {
        out[kPackHead + i] = hs[i];
        out[kPackHead + m + i] = newval[i];
        reinterpret_cast<int32_t*>(out + kPackHead + 2 * (size_t)m)[i] = touched[i];
    }
}
static_assert(sizeof(Ctrl) <= 64 && sizeof(CovScanOut) == 16, "k_cov_pack's header layout");

__global__ void k_scatter_f64(double* __restrict__ dst, const int64_t* __restrict__ idx, const double* __restrict__ val, int m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) dst[idx[i]] = val[i];
}

__global__ __launch_bounds__(256) void k_cov_gupdate(double* __restrict__ g, const double* __restrict__ Gcols,
                                                     const int32_t* __restrict__ slot, int64_t p,
                                                     const int64_t* __restrict__ idx,
                                                     const double* __restrict__ hs, int pos0, int nb) {
    // the block's moves first, fetched side by side (idx -> slot -> column is a dependent chain per visit:
    // walked serially it cost 8 us for 32 visits), compacted to those with h != 0
    __shared__ double s_h[64];
    __shared__ int64_t s_off[64];
    __shared__ int s_n;
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        double h = 0.0;
        int64_t off = 0;
        if (i < nb) { h = hs[pos0 + i]; off = (int64_t)slot[idx[pos0 + i]] * p; }
        const bool nz = (h != 0.0);
        const unsigned long long mask = __ballot(nz);
        if (nz) {
            const int at = __popcll(mask & ((1ull << i) - 1ull));
            s_h[at] = h; s_off[at] = off;
        }
        if (i == 0) s_n = __popcll(mask);
    }
    __syncthreads();
    const int nmove = s_n;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p || nmove == 0) return;
    double acc = g[k];
    for (int i = 0; i < nmove; ++i) acc = fma(-s_h[i], Gcols[s_off[i] + k], acc);
    g[k] = acc;
}

// r -= sum_i hs[pos0+i] X[:, idx[pos0+i]], i < nprev: the trailing update of a pass (any width)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_multi_axpy(const T* __restrict__ X, int64_t ld,
                                                       int64_t nvec, T* __restrict__ r,
                                                       const int64_t* __restrict__ idx,
                                                       const double* __restrict__ hs, int pos0,
                                                       int nprev) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    __shared__ double s_hp[64];
    __shared__ int64_t s_kp[64];
    __shared__ int s_nzp;
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        double h = 0.0;
        int64_t kp = 0;
        if (i < nprev) { h = hs[pos0 + i]; kp = idx[pos0 + i]; }
        const bool nz = (h != 0.0);
        const unsigned long long mask = __ballot(nz);
        if (nz) {
            const int at = __popcll(mask & ((1ull << i) - 1ull));
            s_hp[at] = h; s_kp[at] = kp;
        }
        if (i == 0) s_nzp = __popcll(mask);
    }
    __syncthreads();
    const int nzp = s_nzp;
    if (nzp == 0) return;
    V* __restrict__ rv = reinterpret_cast<V*>(r);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec; j += stride) {
        V rr = rv[j];
        double re[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) re[e] = (double)rr[e];
        for (int i0 = 0; i0 < nzp; i0 += 8) {
            V xp[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int ii = (i0 + t < nzp) ? i0 + t : nzp - 1;
                xp[t] = ld_stream<true>(reinterpret_cast<const V*>(X + s_kp[ii] * ld) + j);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double h = (i0 + t < nzp) ? s_hp[i0 + t] : 0.0;
#pragma unroll
                for (int e = 0; e < NV; ++e) re[e] = fma(-h, (double)xp[t][e], re[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) rr[e] = (T)re[e];
        rv[j] = rr;
    }
}

}  // namespace cdk
