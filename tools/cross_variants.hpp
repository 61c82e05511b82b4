// cross_variants.hpp -- earlier forms of the Gram-column kernel, kept ONLY for tools/cross_bench.hip (A/B timing and
// the loads-only / matrix-only / clock-stamp experiments quoted in DESIGN.md 5a).  Not part of the library.
#pragma once
#include "../coordinatedescent.jl_amd/csrc/gram_kernels.hpp"

namespace cdk {

// ---- round 2's kernel: fragment-shaped A loads from HBM, the B chunk through LDS ----------------------------------
#ifndef CDH_CROSS_UH
#define CDH_CROSS_UH 2
#endif
#ifndef CDH_CROSS_OCC
#define CDH_CROSS_OCC 2
#endif
constexpr int kCrossUH = CDH_CROSS_UH;   // vector rows of fragment loads in flight per group
constexpr int kCrossXS = 64 + 2;          // LDS column stride of the B chunk in 16-byte slots (pad 2: conflict-free fragment reads)
template <typename T>
__global__ __launch_bounds__(64 * kGramWaves, CDH_CROSS_OCC) void k_cross_frag(const T* __restrict__ X, int64_t ld, int64_t nvec,
                                                                        int64_t p, const int64_t* __restrict__ bcols,
                                                                        int nbc, double* __restrict__ partials) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    // the block's B chunk (32 columns x 64 vectors), double-buffered: loaded once per chunk, fully coalesced
    // (a wave instruction = 64 consecutive vectors of one column), and shared by the four waves, each of which
    // works on a DIFFERENT group of 64 X columns -- so the B fragments reach the matrix pipe through LDS and
    // the vector-memory path carries (256 + 32) columns per 256 of X instead of (64 + 32) per 64
    __shared__ V s_b[2][kCrossB * kCrossXS];
    __shared__ double s_red[kGramWaves][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const V* bld[kCrossB / kGramWaves];       // this thread's share of the cooperative B load: columns wave, wave + 4, ...
    bool bact[kCrossB / kGramWaves];
#pragma unroll
    for (int i = 0; i < kCrossB / kGramWaves; ++i) {
        const int col = wave + kGramWaves * i;
        bact[i] = col < nbc;
        bld[i] = reinterpret_cast<const V*>(X + bcols[bact[i] ? col : 0] * ld);
    }
    const int64_t nslabs = (nvec + kCrossSlab - 1) / kCrossSlab;
    const int64_t ngroups = (p + kCrossA - 1) / kCrossA, nsuper = (ngroups + kGramWaves - 1) / kGramWaves;
    constexpr int CPS = kCrossSlab / 64;      // chunks per slab
    // chunk number q of this block's row lane -> first vector
    auto chunk_v0 = [&](int64_t q) { return ((int64_t)blockIdx.y + (q / CPS) * gridDim.y) * kCrossSlab + (q % CPS) * 64; };
    const int64_t my_slabs = nslabs > blockIdx.y ? (nslabs - blockIdx.y + gridDim.y - 1) / gridDim.y : 0;
    const int64_t nq = my_slabs * CPS;
    auto load_b = [&](V (&regs)[kCrossB / kGramWaves], int64_t q) {
        const int64_t v = chunk_v0(q) + lane;
#pragma unroll
        for (int i = 0; i < kCrossB / kGramWaves; ++i)
            regs[i] = (bact[i] && v < nvec) ? ld_stream<false>(bld[i] + v) : vzero((V*)nullptr);
    };
    auto store_b = [&](const V (&regs)[kCrossB / kGramWaves], int buf) {
#pragma unroll
        for (int i = 0; i < kCrossB / kGramWaves; ++i) s_b[buf][(wave + kGramWaves * i) * kCrossXS + lane] = regs[i];
    };
    // only gridDim.x super-groups (4 column groups each, one per wave) are in flight at any time; the block walks
    // its share of them one after the other, a record per (column group, row lane)
    for (int64_t sg = blockIdx.x; sg < nsuper; sg += gridDim.x) {
        const int64_t cg = sg * kGramWaves + wave;
        const int64_t a0 = cg * kCrossA;
        const V* av[kCrossTA];
        bool aact[kCrossTA];
#pragma unroll
        for (int grp = 0; grp < kCrossTA; ++grp) {
            const int64_t col = a0 + 16 * grp + c;
            aact[grp] = cg < ngroups && col < p;
            av[grp] = reinterpret_cast<const V*>(X + (aact[grp] ? col : 0) * ld);
        }
        dvec4 tile[kCrossTA * kCrossTB];
#pragma unroll
        for (int t = 0; t < kCrossTA * kCrossTB; ++t) tile[t] = dvec4{0.0, 0.0, 0.0, 0.0};
        V breg[kCrossB / kGramWaves];
        __syncthreads();                       // the previous super-group's last chunk has been consumed
        if (nq > 0) { load_b(breg, 0); store_b(breg, 0); }
        __syncthreads();
        for (int64_t q = 0; q < nq; ++q) {
            const int buf = (int)(q & 1);
            const int64_t v0 = chunk_v0(q);
            if (q + 1 < nq) load_b(breg, q + 1);          // in flight while this chunk is multiplied
            if (v0 < nvec) {
#pragma unroll 1
                for (int u0 = 0; u0 < 16; u0 += kCrossUH) {
                    V xa[kCrossUH][kCrossTA], xb[kCrossUH][kCrossTB];
#pragma unroll
                    for (int u = 0; u < kCrossUH; ++u) {
                        const int64_t v = v0 + 4 * (u0 + u) + g;
                        const bool in = v < nvec;
#pragma unroll
                        for (int grp = 0; grp < kCrossTA; ++grp)
                            xa[u][grp] = (aact[grp] && in) ? ld_stream<true>(av[grp] + v) : vzero((V*)nullptr);
#pragma unroll
                        for (int grp = 0; grp < kCrossTB; ++grp) xb[u][grp] = s_b[buf][(16 * grp + c) * kCrossXS + 4 * (u0 + u) + g];
                    }
#pragma unroll
                    for (int u = 0; u < kCrossUH; ++u)
#pragma unroll
                        for (int e = 0; e < NV; ++e) {
                            double b[kCrossTB];
#pragma unroll
                            for (int gb = 0; gb < kCrossTB; ++gb) b[gb] = (double)xb[u][gb][e];
#pragma unroll
                            for (int ga = 0; ga < kCrossTA; ++ga) {
                                const double a = (double)xa[u][ga][e];
#pragma unroll
                                for (int gb = 0; gb < kCrossTB; ++gb)
                                    tile[ga * kCrossTB + gb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[gb], tile[ga * kCrossTB + gb], 0, 0, 0);
                            }
                        }
                }
            }
            if (q + 1 < nq) store_b(breg, buf ^ 1);
            __syncthreads();                   // the next chunk is in LDS; this one has been read by every wave
        }
        // this wave's record: its own column group (waves past the last group have nothing to write)
#pragma unroll
        for (int t = 0; t < kCrossTA * kCrossTB; ++t) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) s_red[wave][(g + 4 * q4) * 16 + c] = tile[t][q4];   // D[i = g + 4 q4][j = c]
            __builtin_amdgcn_wave_barrier();
            if (cg < ngroups) {
                double* __restrict__ out = partials + (cg * gridDim.y + blockIdx.y) * kCrossRec + t * 256;
#pragma unroll
                for (int v = lane; v < 256; v += 64) out[v] = s_red[wave][v];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
// ---- k_cross2: the same cross products with BOTH operands brought in by LDS-DMA in whole 128-byte lines ------------
// k_cross's A operands are fragment-shaped loads: a wave instruction takes 64 B from each of 16 columns, so every
// 128-B line is asked for twice and each quarter-wave touches 16 different lines for 16 bytes apiece -- twice the
// texture-addresser work per byte, and the loads sit in VGPRs that a wave cannot refill while it issues its MFMAs
// (21.3 ms per pass over 80 GB: 3.76 TB/s, the sum of the memory time and the matrix time, not their maximum).
// Here nothing goes through registers on the way in:
//   * one `global_load_lds_dwordx4` moves 8 columns x 8 vectors (8 whole 128-B lines) straight into LDS; a wave's
//     sub-chunk of 64 A columns x 8 vectors (16 rows of fp64) is 8 such instructions = 8 KB, double-buffered per
//     wave; the block's B sub-chunk (32 columns x 8 vectors) is 4 instructions, ONE PER WAVE, shared by the four;
//   * an LDS-DMA writes lane l at (wave-uniform base) + 16 l, so the image cannot be padded; the bank conflicts of
//     the MFMA-fragment reads are removed on the SOURCE side instead: lane (column cl = l / 8, slot j = l % 8) loads
//     vector j ^ (cl & 6) of its line, i.e. vector v of column cc sits in slot 8 cc + (v ^ (cc & 6)), and the
//     ds_read_b128 of lane (c, g) -- column 16 t + c, vector 4 u + g -- is conflict-free in each of the
//     instruction's four lane groups (checked exhaustively; the counters agree);
//   * per sub-chunk: wait for the own loads of THIS sub-chunk (issued one MFMA phase ago), one block barrier (the B
//     pieces of all four waves have landed; everyone is done with the other buffers), issue the NEXT sub-chunk's
//     9 loads, then 12 fragment reads and 32 MFMAs -- the loads are in flight for the whole matrix phase.
// LDS: 4 waves x 2 x 8 KB + 2 x 4 KB = 72 KB, two blocks per CU.  Rows past nvec inside a column are the zero pad
// (ld is a multiple of 32 elements = whole sub-chunks); columns past p and B columns past nbc are clamped to valid
// ones and their results never read.  Observation weights (G = X'WX): the weight of each row multiplies the B
// fragment on its way into the MFMA (a 128-byte piece of w per sub-chunk, staged by wave 0 the same way).
constexpr int kX2ASlots = kCrossA * kX2SV;        // 16-byte slots of a wave's A sub-chunk
template <int AUX, typename V>
__device__ __forceinline__ void glds16(const V* gsrc, V* lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, AUX);
}
__device__ unsigned long long g_cross_stamps[4 * 2048];     // EXP & 4: per block (shader cycles, 100 MHz ticks) at start and end
template <typename T, bool HASW, int AUX_A, int EXP = 0>   // EXP (timing experiments): 1 = loads only, 2 = matrix work only, +4 = clock stamps
__global__ __launch_bounds__(64 * kGramWaves, 2) void k_cross2(const T* __restrict__ X, int64_t ld, int64_t nvec, int64_t p,
                                                               const int64_t* __restrict__ bcols, int nbc,
                                                               const T* __restrict__ w, double* __restrict__ partials) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    __shared__ V s_a[2][kGramWaves][kX2ASlots];
    __shared__ V s_b[2][kX2BSlots];
    __shared__ V s_w[2][kX2SV];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int cl = lane >> 3, sj = (lane & 7) ^ (cl & 6);          // this lane's column of a DMA piece, and which vector of the line it fetches
    const int64_t nslabs = (nvec + kCrossSlab - 1) / kCrossSlab;
    const int64_t ngroups = (p + kCrossA - 1) / kCrossA, nsuper = (ngroups + kGramWaves - 1) / kGramWaves;
    const int64_t my_slabs = nslabs > blockIdx.y ? (nslabs - blockIdx.y + gridDim.y - 1) / gridDim.y : 0;
    const int64_t nq = my_slabs * kX2SPS;
    auto sub_v0 = [&](int64_t q) { return ((int64_t)blockIdx.y + (q / kX2SPS) * gridDim.y) * kCrossSlab + (q % kX2SPS) * kX2SV; };
    // the block's B piece this wave brings in: columns 8 wave .. 8 wave + 7 of the batch
    const int bcol_i = 8 * wave + cl;
    const V* bsrc = reinterpret_cast<const V*>(X + bcols[bcol_i < nbc ? bcol_i : 0] * ld) + sj;
    // fragment-read slots: lane (c, g) reads column 16 t + c, vector 4 u + g
    int a_slot[kCrossTA][2], b_slot[kCrossTB][2];
#pragma unroll
    for (int t = 0; t < kCrossTA; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) a_slot[t][u] = (16 * t + c) * kX2SV + ((4 * u + g) ^ ((16 * t + c) & 6));
#pragma unroll
    for (int t = 0; t < kCrossTB; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) b_slot[t][u] = (16 * t + c) * kX2SV + ((4 * u + g) ^ ((16 * t + c) & 6));
    unsigned long long t0c = 0, t0r = 0;
    if constexpr ((EXP & 4) != 0) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    for (int64_t sg = blockIdx.x; sg < nsuper; sg += gridDim.x) {
        const int64_t cg = sg * kGramWaves + wave;
        const int64_t a0 = cg * kCrossA;
        const V* asrc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t col = a0 + 8 * i + cl;
            if (col >= p) col = p - 1;                  // clamped: those rows of the record are never read
            asrc[i] = reinterpret_cast<const V*>(X + col * ld) + sj;
        }
        auto issue = [&](int64_t q, int buf) {
            const int64_t v0 = sub_v0(q);
            if (v0 >= nvec) return;                     // block-uniform
            if constexpr ((EXP & 3) == 2) return;
#pragma unroll
            for (int i = 0; i < 8; ++i) glds16<AUX_A>(asrc[i] + v0, &s_a[buf][wave][64 * i]);
            glds16<0>(bsrc + v0, &s_b[buf][64 * wave]);
            if constexpr (HASW) {
                if (wave == 0 && lane < kX2SV) glds16<0>(reinterpret_cast<const V*>(w) + v0 + lane, &s_w[buf][0]);
            }
        };
        dvec4 tile[kCrossTA * kCrossTB];
#pragma unroll
        for (int t = 0; t < kCrossTA * kCrossTB; ++t) tile[t] = dvec4{0.0, 0.0, 0.0, 0.0};
        if (nq > 0) issue(0, 0);
        for (int64_t q = 0; q < nq; ++q) {
            const int buf = (int)(q & 1);
            // this sub-chunk's pieces (issued one matrix phase ago) have landed; every fragment read of the previous
            // phase has returned; then the barrier: all four waves' B pieces are there, the other buffers are free
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool live = sub_v0(q) < nvec && (EXP & 3) != 1;          // block-uniform
            V af[2][kCrossTA], bf[2][kCrossTB], wf[2];
            if (live) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int t = 0; t < kCrossTA; ++t) af[u][t] = s_a[buf][wave][a_slot[t][u]];
#pragma unroll
                    for (int t = 0; t < kCrossTB; ++t) bf[u][t] = s_b[buf][b_slot[t][u]];
                    if constexpr (HASW) wf[u] = s_w[buf][4 * u + g];
                }
            }
            // the fragment reads are issued BEFORE the next sub-chunk's DMA: hipcc drains every outstanding LDS-DMA
            // (vmcnt(0)) ahead of an LDS read it cannot prove disjoint, which would serialise load and matrix phases
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < nq) issue(q + 1, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if (live) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < NV; ++e) {
                        double b[kCrossTB];
#pragma unroll
                        for (int gb = 0; gb < kCrossTB; ++gb) {
                            b[gb] = (double)bf[u][gb][e];
                            if constexpr (HASW) b[gb] *= (double)wf[u][e];
                        }
#pragma unroll
                        for (int ga = 0; ga < kCrossTA; ++ga) {
                            const double a = (double)af[u][ga][e];
#pragma unroll
                            for (int gb = 0; gb < kCrossTB; ++gb)
                                tile[ga * kCrossTB + gb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[gb], tile[ga * kCrossTB + gb], 0, 0, 0);
                        }
                    }
            }
        }
        // every wave is past its last fragment read before anyone's next super-group refills buffer 0 (B is shared),
        // and before this wave's record goes through its own (now idle) A buffer
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        double* s_red = reinterpret_cast<double*>(&s_a[0][wave][0]);      // 8 KB per wave: room for a 256-value tile
#pragma unroll
        for (int t = 0; t < kCrossTA * kCrossTB; ++t) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) s_red[(g + 4 * q4) * 16 + c] = tile[t][q4];   // D[i = g + 4 q4][j = c]
            __builtin_amdgcn_wave_barrier();
            if (cg < ngroups) {
                double* __restrict__ out = partials + (cg * gridDim.y + blockIdx.y) * kCrossRec + t * 256;
#pragma unroll
                for (int v = lane; v < 256; v += 64) out[v] = s_red[v];
            }
            __builtin_amdgcn_wave_barrier();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // the records have left LDS before the next super-group's DMA overwrites it
    }
    if constexpr ((EXP & 4) != 0) {
        if (threadIdx.x == 0) {
            const int b = blockIdx.y * gridDim.x + blockIdx.x;
            if (b < 2048) {
                g_cross_stamps[4 * b + 0] = t0c; g_cross_stamps[4 * b + 1] = t0r;
                g_cross_stamps[4 * b + 2] = __builtin_amdgcn_s_memtime(); g_cross_stamps[4 * b + 3] = __builtin_amdgcn_s_memrealtime();
            }
        }
    }
}

}  // namespace cdk
