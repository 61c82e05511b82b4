// kernels.hpp -- HIP kernels (gfx950 / CDNA4) of the coordinate-descent sweep.
//
// Reference loops these replace (paths relative to the reference's src/):
//   k_step        descendCoordinate! dots   cd_differentiable_function.jl:94-99 (LS),
//                                           :177-182 (WLS), :262-266 (SQRT), fused with the
//                                           residual update of the PREVIOUS visit (:107-109)
//   k_finalize    the scalar update         :101-104 (+ ProximalBase cdprox!), :271-283
//   k_axpy        the trailing residual update of a pass (:107-109)
//   k_init_resid  initialize!               :59-72 (A_mul_B_row)
//   k_col_dots    gradient / _findLambdaMax / _stdX!   :75-76, coordinate_descent.jl:118-149,
//                                           utils.jl:127-138
//   k_blockstep / k_block_finalize          the same visit arithmetic, B visits per launch
//
// Everything here is HBM-bound BLAS-1/2 work: 16-byte coalesced loads of
// contiguous column streams, fp64 per-lane accumulators, wave64 __shfl_down
// reductions, LDS across the waves of a block, per-block partials summed in a
// FIXED order by a one-block kernel (bit-stable run to run; no float atomics).
// No MFMA: arithmetic intensity is ~0.1 flop/B.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cdk {

constexpr int kBlock = 256;   // 4 waves
constexpr int kUnroll = 4;    // independent 16-B loads per stream per thread
constexpr int kNSum = 4;      // a, b, q (+pad) per partial record

// Device-resident control block: read by the scalar-update kernels so that a
// captured graph stays valid when lambda changes.
struct Ctrl {
    double lambda0;
    double n_total;
    double maxH;
    int32_t loss;        // cdh_loss
    int32_t has_omega;
    int32_t domain_error;
    int32_t pad;
    double q_carry;      // r'r carried from block to block by the covariance-form visits (sqrt-lasso)
    double cert_abs;     // gradient-cache certificates: |g_k| <= thr_k (1 - 1e-9) - cert_abs sqrt(a_k); 0 for fp64 storage,
                         // the rounding of an fp32 residual otherwise (cdhip.hip, gc_cert_abs)
};

// 16-byte native vectors (clang ext_vector_type: element access v[e] stays in registers and
// __builtin_nontemporal_load accepts them).
typedef double dvec2 __attribute__((ext_vector_type(2)));
typedef float fvec4 __attribute__((ext_vector_type(4)));
template <typename T> struct VecOf;
template <> struct VecOf<double> { using V = dvec2; static constexpr int N = 2; };
template <> struct VecOf<float>  { using V = fvec4; static constexpr int N = 4; };

__device__ __forceinline__ dvec2 vzero(dvec2*) { return dvec2{0.0, 0.0}; }
__device__ __forceinline__ fvec4 vzero(fvec4*) { return fvec4{0.f, 0.f, 0.f, 0.f}; }

// X columns are streamed once per launch: a non-temporal load keeps them from evicting the
// residual vector (read AND written by every launch) out of L2 / Infinity Cache.
template <bool NT, typename V> __device__ __forceinline__ V ld_stream(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// r <- r - h*xp ; (a,b,q) += (xc.xc, xc.r, r.r)   [optionally weighted by w]
__device__ __forceinline__ void visit_elem(double xc, double& r, double xp, double h, bool apply,
                                           double wgt, bool hasw, double& a, double& b, double& q) {
    if (apply) r = fma(-h, xp, r);
    if (hasw) {
        double wx = wgt * xc;
        a = fma(wx, xc, a);
        b = fma(wx, r, b);
    } else {
        a = fma(xc, xc, a);
        b = fma(xc, r, b);
    }
    q = fma(r, r, q);
}
__device__ __forceinline__ void visit_vec(const dvec2& xc, dvec2& r, const dvec2& xp, double h,
                                          bool apply, const dvec2& w, bool hasw, double& a,
                                          double& b, double& q) {
    double r0 = r.x, r1 = r.y;
    visit_elem(xc.x, r0, xp.x, h, apply, w.x, hasw, a, b, q);
    visit_elem(xc.y, r1, xp.y, h, apply, w.y, hasw, a, b, q);
    r = dvec2{r0, r1};
}
__device__ __forceinline__ void visit_vec(const fvec4& xc, fvec4& r, const fvec4& xp, double h,
                                          bool apply, const fvec4& w, bool hasw, double& a,
                                          double& b, double& q) {
    double r0 = r.x, r1 = r.y, r2 = r.z, r3 = r.w;
    visit_elem(xc.x, r0, xp.x, h, apply, w.x, hasw, a, b, q);
    visit_elem(xc.y, r1, xp.y, h, apply, w.y, hasw, a, b, q);
    visit_elem(xc.z, r2, xp.z, h, apply, w.z, hasw, a, b, q);
    visit_elem(xc.w, r3, xp.w, h, apply, w.w, hasw, a, b, q);
    r = fvec4{(float)r0, (float)r1, (float)r2, (float)r3};
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}

// Sum NV per-thread values over a kBlock-thread block; result valid in thread 0.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* [NV * 4] */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = wave_sum(v[i]);
        if (lane == 0) lds[i * (kBlock / 64) + wid] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) s += lds[i * (kBlock / 64) + w];
            v[i] = s;
        }
    }
}

// Sum NV per-thread values over the block and store them VALUE-MAJOR: out[v * G + block]
// (so the one-block finalize kernels read each value's G partials as one coalesced run).
template <int NV>
__device__ __forceinline__ void block_sum_store(double (&v)[NV], double* lds /* [NV * 4] */,
                                                double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = wave_sum(v[i]);
        if (lane == 0) lds[i * (kBlock / 64) + wid] = v[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += kBlock) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += lds[i * (kBlock / 64) + w];
        out[(int64_t)i * gridDim.x + blockIdx.x] = s;
    }
}
// One wave sums G partials of one value: lanes stride the run, fixed order.
__device__ __forceinline__ double wave_sum_run(const double* __restrict__ run, int G, int lane) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // 4 independent loads in flight per round
    int i = lane;
    for (; i + 192 < G; i += 256) { s0 += run[i]; s1 += run[i + 64]; s2 += run[i + 128]; s3 += run[i + 192]; }
    for (; i < G; i += 64) s0 += run[i];
    return wave_sum((s0 + s1) + (s2 + s3));
}

// ---------------------------------------------------------------------------------
// k_step: visit number `pos` of a pass.  Applies the pending residual update of visit
// pos-1 (r -= hs[pos-1] * X[:, idx[pos-1]], skipped when that h is 0 or pos == 0) and,
// in the same sweep over r, accumulates for column k = idx[pos]:
//   a = sum w x^2, b = sum w x r, q = sum r^2     (w = 1 unless HASW)
// One partial record per block.  Streams: X_k (+X_prev) read, r read (+written).
// ---------------------------------------------------------------------------------
template <typename T, bool HASW, bool NT>
__global__ __launch_bounds__(kBlock) void k_step(const T* __restrict__ X, int64_t ld, int64_t nvec,
                                                 const T* __restrict__ w, T* __restrict__ r,
                                                 const int64_t* __restrict__ idx,
                                                 const double* __restrict__ hs, int pos,
                                                 double* __restrict__ partials) {
    using V = typename VecOf<T>::V;
    __shared__ double lds[3 * (kBlock / 64)];
    const int64_t k = idx[pos];
    double h = 0.0;
    int64_t kp = k;
    if (pos > 0) { h = hs[pos - 1]; kp = idx[pos - 1]; }
    const bool apply = (h != 0.0);  // NaN compares unequal: it propagates, as in the reference
    const V* __restrict__ cv = reinterpret_cast<const V*>(X + k * ld);
    const V* __restrict__ pv = reinterpret_cast<const V*>(X + kp * ld);
    const V* __restrict__ wv = reinterpret_cast<const V*>(w);
    V* __restrict__ rv = reinterpret_cast<V*>(r);

    double acc[3] = {0.0, 0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock * kUnroll;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * kUnroll + threadIdx.x; base < nvec;
         base += stride) {
        V xc[kUnroll], rr[kUnroll], xp[kUnroll], ww[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t j = base + (int64_t)u * kBlock;
            xc[u] = vzero((V*)nullptr); rr[u] = xc[u]; xp[u] = xc[u]; ww[u] = xc[u];
            if (j < nvec) {
                xc[u] = ld_stream<NT>(cv + j);
                rr[u] = rv[j];
                if (apply) xp[u] = ld_stream<NT>(pv + j);
                if (HASW) ww[u] = wv[j];
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t j = base + (int64_t)u * kBlock;
            visit_vec(xc[u], rr[u], xp[u], h, apply, ww[u], HASW, acc[0], acc[1], acc[2]);
            if (apply && j < nvec) rv[j] = rr[u];
        }
    }
    block_sum_store<3>(acc, lds, partials);
}

// r -= h * X[:, k] for the last visit of a pass (or of a chunk): pos = number of visits done.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_axpy(const T* __restrict__ X, int64_t ld, int64_t nvec,
                                                 T* __restrict__ r, const int64_t* __restrict__ idx,
                                                 const double* __restrict__ hs, int pos) {
    using V = typename VecOf<T>::V;
    const double h = hs[pos - 1];
    if (!(h != 0.0)) return;
    const V* __restrict__ pv = reinterpret_cast<const V*>(X + idx[pos - 1] * ld);
    V* __restrict__ rv = reinterpret_cast<V*>(r);
    const int64_t stride = (int64_t)gridDim.x * kBlock * kUnroll;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * kUnroll + threadIdx.x; base < nvec;
         base += stride) {
        V rr[kUnroll], xp[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t j = base + (int64_t)u * kBlock;
            rr[u] = vzero((V*)nullptr); xp[u] = rr[u];
            if (j < nvec) { rr[u] = rv[j]; xp[u] = pv[j]; }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t j = base + (int64_t)u * kBlock;
            double a = 0, b = 0, q = 0;
            visit_vec(xp[u], rr[u], xp[u], h, true, xp[u], false, a, b, q);
            if (j < nvec) rv[j] = rr[u];
        }
    }
}

// ---------------------------------------------------------------------------------
// The scalar update of one visit from the reduced sums (a, b, q).
//   LS / WLS (cd_differentiable_function.jl:101-104, 184-187 + cdprox!):
//     v = beta_k + b/a ; beta_k <- S(v, lambda0 * omega_k * n / a)
//   SQRT (:253-283) on the un-added-back residual (SURVEY 8 a-6 identities):
//     s = b + beta_k a ; rsqr = q + 2 beta_k b + beta_k^2 a ; closed form
// Records h, the new value and whether the pre-prox value was non-zero (which is
// what makes ProximalBase's SparseIterate append a support slot).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double soft_threshold(double v, double t) {
    return v > t ? v - t : (v < -t ? v + t : 0.0);
}
// Pure function: (sums, old value, weight) -> (new value, pre-prox-non-zero flag, domain flag).
struct VisitOut { double nv; int32_t tch; int32_t dom; };
__device__ __forceinline__ VisitOut visit_update(int loss, double lambda0, double n_total, double a,
                                                 double b, double q, double oldv, double om) {
    VisitOut o{0.0, 0, 0};
    if (loss == 1 /* CDH_SQRT */) {
        const double lam = lambda0 * om;
        const double s = fma(oldv, a, b);
        double rsqr = q + 2.0 * oldv * b + oldv * oldv * a;
        if (rsqr < 0.0) rsqr = 0.0;
        if (fabs(s) <= lam * sqrt(rsqr)) {
            o.nv = 0.0;
        } else {
            double u = 1.0 - lam * lam / a;
            double v = rsqr - s * s / a;
            // Julia throws DomainError on sqrt of a negative; flag it when it is clearly
            // negative, clamp rounding-level negatives (this formulation gets rsqr from an
            // identity and carries a little more cancellation noise than the reference).
            if (u <= 0.0 || v < -1e-12 * rsqr) o.dom = 1;
            if (v < 0.0) v = 0.0;
            if (u <= 0.0) u = 1e-300;
            const double c = lam / sqrt(u) * sqrt(v);
            o.nv = (s > 0.0 ? (s - c) : (s + c)) / a;
        }
    } else {
        const double v = oldv + b / a;
        o.tch = (v != 0.0) ? 1 : 0;
        o.nv = soft_threshold(v, lambda0 * om * (n_total / a));
    }
    return o;
}
// Same update with 1/a supplied by the caller: the wide-block recurrence visits B coordinates
// serially in one wave, and a_i does not change, so the division leaves the serial chain.
__device__ __forceinline__ VisitOut visit_update_rcp(int loss, double lambda0, double n_total, double a,
                                                     double ia, double b, double q, double oldv,
                                                     double om) {
    VisitOut o{0.0, 0, 0};
    if (loss == 1 /* CDH_SQRT */) {
        const double lam = lambda0 * om;
        const double s = fma(oldv, a, b);
        double rsqr = q + 2.0 * oldv * b + oldv * oldv * a;
        if (rsqr < 0.0) rsqr = 0.0;
        if (fabs(s) <= lam * sqrt(rsqr)) {
            o.nv = 0.0;
        } else {
            double u = 1.0 - lam * lam * ia;
            double v = rsqr - s * s * ia;
            if (u <= 0.0 || v < -1e-12 * rsqr) o.dom = 1;
            if (v < 0.0) v = 0.0;
            if (u <= 0.0) u = 1e-300;
            const double c = lam / sqrt(u) * sqrt(v);
            o.nv = (s > 0.0 ? (s - c) : (s + c)) * ia;
        }
    } else {
        const double v = fma(b, ia, oldv);
        o.tch = (v != 0.0) ? 1 : 0;
        o.nv = soft_threshold(v, lambda0 * om * (n_total * ia));
    }
    return o;
}
__device__ inline void scalar_update(double a, double b, double q, Ctrl* ctrl, double* beta,
                                     const double* omega, int64_t k, int pos, double* hs,
                                     double* newval, int32_t* touched) {
    const double om = ctrl->has_omega ? omega[k] : 1.0;
    const double oldv = beta[k];
    const VisitOut o = visit_update(ctrl->loss, ctrl->lambda0, ctrl->n_total, a, b, q, oldv, om);
    if (o.dom) ctrl->domain_error = 1;
    const double h = o.nv - oldv;
    beta[k] = o.nv;
    hs[pos] = h;
    newval[pos] = o.nv;
    touched[pos] = o.tch;
    const double ah = fabs(h);
    if (ah > ctrl->maxH) ctrl->maxH = ah;  // NaN never raises maxH (coordinate_descent.jl:104)
}

// Sum the per-block partials (value-major, G per value) in a fixed order: wave v sums
// value v.  FUSED: also do the scalar update (single process).  Otherwise write (a, b, q)
// to `red` for the all-reduce.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_finalize(const double* __restrict__ partials, int nparts,
                                                     Ctrl* ctrl, double* beta,
                                                     const double* __restrict__ omega,
                                                     const int64_t* __restrict__ idx, double* hs,
                                                     double* newval, int32_t* touched, int pos,
                                                     double* red) {
    __shared__ double lds[3 * (kBlock / 64)];
    // issue the scalar operands' loads before the reduction so their latency overlaps it
    int64_t k = 0;
    double oldv = 0.0, om = 1.0;
    if (FUSED && threadIdx.x == 0) {
        k = idx[pos];
        oldv = beta[k];
        if (ctrl->has_omega) om = omega[k];
    }
    // every thread takes a strided share of each value's run (coalesced, all loads independent),
    // then one fixed-order block reduction
    double sums[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nparts; i += kBlock) {
        sums[0] += partials[i];
        sums[1] += partials[(int64_t)nparts + i];
        sums[2] += partials[2 * (int64_t)nparts + i];
    }
    block_sum<3>(sums, lds);
    if (threadIdx.x == 0) {
        if (FUSED) {
            const VisitOut o = visit_update(ctrl->loss, ctrl->lambda0, ctrl->n_total, sums[0], sums[1],
                                            sums[2], oldv, om);
            if (o.dom) ctrl->domain_error = 1;
            const double h = o.nv - oldv;
            beta[k] = o.nv; hs[pos] = h; newval[pos] = o.nv; touched[pos] = o.tch;
            const double ah = fabs(h);
            if (ah > ctrl->maxH) ctrl->maxH = ah;
        } else {
            red[0] = sums[0]; red[1] = sums[1]; red[2] = sums[2]; red[3] = 0.0;
        }
    }
}
__global__ void k_scalar_update(const double* __restrict__ red, Ctrl* ctrl, double* beta,
                                const double* __restrict__ omega, const int64_t* __restrict__ idx,
                                double* hs, double* newval, int32_t* touched, int pos) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        scalar_update(red[0], red[1], red[2], ctrl, beta, omega, idx[pos], pos, hs, newval, touched);
}

// ---------------------------------------------------------------------------------
// Blocked sweep: visits [pos0, pos0+nb) (nb <= B) in one launch.
//   - applies the rank-(<=B) residual update of the previous block
//       r -= sum_i hs[pos0-B+i] * X[:, idx[pos0-B+i]]        (columns with h == 0 skipped)
//   - in the same sweep over r accumulates, for the block's columns c_0..c_{nb-1}:
//       c_i = X_ci . r,  G_ij = X_ci . X_cj (i <= j),  q = r . r
// The B sequential scalar updates then run on (c, G, q) in k_block_finalize:
//   b_i = c_i - sum_{j<i} h_j G_ji ;   q_i = q - 2 sum_{j<i} h_j c_j' ... (see below)
// Streams per block: B columns once + up to B previous columns + r read/write.
// ---------------------------------------------------------------------------------
template <int B> struct BlockRec {
    static constexpr int NG = B * (B + 1) / 2;
    static constexpr int N = NG + B + 1;  // G (upper, row-major), c, q
};

template <typename T, int B, bool NT>
__global__ __launch_bounds__(kBlock) void k_blockstep(const T* __restrict__ X, int64_t ld,
                                                      int64_t nvec, T* __restrict__ r,
                                                      const int64_t* __restrict__ idx,
                                                      const double* __restrict__ hs, int pos0,
                                                      int nb, int nprev,
                                                      double* __restrict__ partials) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    constexpr int NREC = BlockRec<B>::N;
    __shared__ double lds[NREC * (kBlock / 64)];

    const V* cv[B];
    const V* pv[B];
    double hp[B];
    bool any = false;
#pragma unroll
    for (int i = 0; i < B; ++i) {
        const int64_t kc = idx[pos0 + (i < nb ? i : 0)];
        cv[i] = reinterpret_cast<const V*>(X + kc * ld);
        double h = 0.0;
        int64_t kp = kc;
        if (i < nprev) { h = hs[pos0 - nprev + i]; kp = idx[pos0 - nprev + i]; }
        hp[i] = h;
        pv[i] = reinterpret_cast<const V*>(X + kp * ld);
        any = any || (h != 0.0);
    }
    V* __restrict__ rv = reinterpret_cast<V*>(r);

    double acc[NREC];
#pragma unroll
    for (int i = 0; i < NREC; ++i) acc[i] = 0.0;

    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec; j += stride) {
        V rr = rv[j];
        V xc[B];
#pragma unroll
        for (int i = 0; i < B; ++i) xc[i] = (i < nb) ? ld_stream<NT>(cv[i] + j) : vzero((V*)nullptr);
        double re[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) re[e] = (double)rr[e];
        if (any) {
#pragma unroll
            for (int i = 0; i < B; ++i) {
                if (hp[i] != 0.0) {
                    const V xp = ld_stream<NT>(pv[i] + j);
#pragma unroll
                    for (int e = 0; e < NV; ++e) re[e] = fma(-hp[i], (double)xp[e], re[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < NV; ++e) rr[e] = (T)re[e];
            rv[j] = rr;
#pragma unroll
            for (int e = 0; e < NV; ++e) re[e] = (double)rr[e];  // what is stored is what is used
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            double xe[B];
#pragma unroll
            for (int i = 0; i < B; ++i) xe[i] = (double)xc[i][e];
            int g = 0;
#pragma unroll
            for (int i = 0; i < B; ++i) {
#pragma unroll
                for (int l = i; l < B; ++l) { acc[g] = fma(xe[i], xe[l], acc[g]); ++g; }
            }
#pragma unroll
            for (int i = 0; i < B; ++i)
                acc[BlockRec<B>::NG + i] = fma(xe[i], re[e], acc[BlockRec<B>::NG + i]);
            acc[NREC - 1] = fma(re[e], re[e], acc[NREC - 1]);
        }
    }
    block_sum_store<NREC>(acc, lds, partials);
}

// Rank-(<=B) residual update for the last block of a pass: r -= sum_i hs[pos0+i] X[:, idx[pos0+i]]
template <typename T, int B>
__global__ __launch_bounds__(kBlock) void k_block_axpy(const T* __restrict__ X, int64_t ld,
                                                       int64_t nvec, T* __restrict__ r,
                                                       const int64_t* __restrict__ idx,
                                                       const double* __restrict__ hs, int pos0,
                                                       int nprev) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    const V* pv[B];
    double hp[B];
    bool any = false;
#pragma unroll
    for (int i = 0; i < B; ++i) {
        double h = 0.0;
        int64_t kp = idx[pos0];
        if (i < nprev) { h = hs[pos0 + i]; kp = idx[pos0 + i]; }
        hp[i] = h;
        pv[i] = reinterpret_cast<const V*>(X + kp * ld);
        any = any || (h != 0.0);
    }
    if (!any) return;
    V* __restrict__ rv = reinterpret_cast<V*>(r);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec; j += stride) {
        V rr = rv[j];
        double re[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) re[e] = (double)rr[e];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            if (hp[i] != 0.0) {
                const V xp = ld_stream<true>(pv[i] + j);
#pragma unroll
                for (int e = 0; e < NV; ++e) re[e] = fma(-hp[i], (double)xp[e], re[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) rr[e] = (T)re[e];
        rv[j] = rr;
    }
}

// The B sequential scalar updates of a block from the reduced record (G, c, q):
// before visit i the residual is r_i = r_0 - sum_{j<i} h_j X_cj, hence
//   b_i = c_i - sum_{j<i} h_j G_ji
//   q_i = q_{i-1} - 2 h_{i-1} b_{i-1} + h_{i-1}^2 G_{i-1,i-1}
// where b_{i-1} is taken before the update (X_c(i-1) . r_{i-1}).
// Runs in ONE wave: lanes 0..nb-1 fetch idx / beta / omega in parallel (no dependent
// global-load chain inside the recurrence), lane 0 runs the recurrence on registers
// exchanged through LDS, lanes 0..nb-1 write the results back.
template <int B>
__device__ inline void block_scalar_updates(const double* rec /* LDS or global */, int nb, Ctrl* ctrl,
                                            double* beta, const double* omega, const int64_t* idx,
                                            int pos0, double* hs, double* newval, int32_t* touched,
                                            double* sh /* LDS: [4 * B] doubles + [2 * B] ints */) {
    constexpr int NG = BlockRec<B>::NG;
    const int lane = threadIdx.x & 63;
    double* s_old = sh;            // [B]
    double* s_om = sh + B;         // [B]
    double* s_nv = sh + 2 * B;     // [B]
    double* s_h = sh + 3 * B;      // [B]
    int64_t* s_k = reinterpret_cast<int64_t*>(sh + 4 * B);       // [B]
    int32_t* s_t = reinterpret_cast<int32_t*>(sh + 5 * B);       // [B]
    const int has_omega = ctrl->has_omega;
    if (lane < nb) {
        const int64_t k = idx[pos0 + lane];
        s_k[lane] = k;
        s_old[lane] = beta[k];
        s_om[lane] = has_omega ? omega[k] : 1.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
        const int loss = ctrl->loss;
        const double lambda0 = ctrl->lambda0, n_total = ctrl->n_total;
        double maxH = ctrl->maxH;
        int dom = 0;
        double q = rec[NG + B];
        for (int i = 0; i < nb; ++i) {
            // G index of (j, i), j <= i, row-major upper: off(j) + (i - j), off(j) = j*B - j(j-1)/2
            double b = rec[NG + i];
            for (int j = 0; j < i; ++j) b = fma(-s_h[j], rec[j * B - j * (j - 1) / 2 + (i - j)], b);
            const double a = rec[i * B - i * (i - 1) / 2];
            double oldv = s_old[i];
            for (int j = 0; j < i; ++j)            // the same coordinate earlier in this block
                if (s_k[j] == s_k[i]) oldv = s_nv[j];
            const VisitOut o = visit_update(loss, lambda0, n_total, a, b, q, oldv, s_om[i]);
            dom |= o.dom;
            const double h = o.nv - oldv;
            s_nv[i] = o.nv; s_h[i] = h; s_t[i] = o.tch;
            const double ah = fabs(h);
            if (ah > maxH) maxH = ah;
            q = q - 2.0 * h * b + h * h * a;
            if (q < 0.0) q = 0.0;
        }
        ctrl->maxH = maxH;
        if (dom) ctrl->domain_error = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < nb) {
        hs[pos0 + lane] = s_h[lane];
        newval[pos0 + lane] = s_nv[lane];
        touched[pos0 + lane] = s_t[lane];
        // last writer of a repeated coordinate wins, as sequential visits would have it
        bool last = true;
        for (int j = lane + 1; j < nb; ++j) last = last && (s_k[j] != s_k[lane]);
        if (last) beta[s_k[lane]] = s_nv[lane];
    }
}

template <int B, bool FUSED>
__global__ __launch_bounds__(1024) void k_block_finalize(const double* __restrict__ partials,
                                                         int nparts, int nb, Ctrl* ctrl,
                                                         double* beta,
                                                         const double* __restrict__ omega,
                                                         const int64_t* __restrict__ idx, double* hs,
                                                         double* newval, int32_t* touched, int pos0,
                                                         double* red) {
    constexpr int NREC = BlockRec<B>::N;
    __shared__ double rec[NREC];
    __shared__ double sh[6 * B];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;  // 16 waves
    for (int v = wid; v < NREC; v += 16) {
        const double s = wave_sum_run(partials + (int64_t)v * nparts, nparts, lane);
        if (lane == 0) rec[v] = s;
    }
    __syncthreads();
    if (wid == 0) {
        if (FUSED) {
            block_scalar_updates<B>(rec, nb, ctrl, beta, omega, idx, pos0, hs, newval, touched, sh);
        } else {
            for (int v = lane; v < NREC; v += 64) red[v] = rec[v];
        }
    }
}
template <int B>
__global__ __launch_bounds__(64) void k_block_scalar(const double* __restrict__ red, int nb, Ctrl* ctrl,
                                                     double* beta, const double* __restrict__ omega,
                                                     const int64_t* __restrict__ idx, double* hs,
                                                     double* newval, int32_t* touched, int pos0) {
    constexpr int NREC = BlockRec<B>::N;
    __shared__ double rec[NREC];
    __shared__ double sh[6 * B];
    for (int v = threadIdx.x; v < NREC; v += 64) rec[v] = red[v];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    block_scalar_updates<B>(rec, nb, ctrl, beta, omega, idx, pos0, hs, newval, touched, sh);
}

// ---------------------------------------------------------------------------------
// initialize!: r = y - sum_{s} X[:, sup_idx[s]] * sup_val[s], support order, one sweep.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_init_resid(const T* __restrict__ X, int64_t ld,
                                                       int64_t nvec, const T* __restrict__ y,
                                                       T* __restrict__ r,
                                                       const int64_t* __restrict__ sup_idx,
                                                       const double* __restrict__ sup_val, int nnz) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    const V* yv = reinterpret_cast<const V*>(y);
    V* rv = reinterpret_cast<V*>(r);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec; j += stride) {
        double acc[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[e] = 0.0;
        for (int s = 0; s < nnz; ++s) {
            const V xv = reinterpret_cast<const V*>(X + sup_idx[s] * ld)[j];
            const double bv = sup_val[s];
            const T* xp = reinterpret_cast<const T*>(&xv);
#pragma unroll
            for (int e = 0; e < NV; ++e) acc[e] = fma((double)xp[e], bv, acc[e]);
        }
        V yy = yv[j];
        T* yp = reinterpret_cast<T*>(&yy);
#pragma unroll
        for (int e = 0; e < NV; ++e) yp[e] = (T)((double)yp[e] - acc[e]);
        rv[j] = yy;
    }
}

// ---------------------------------------------------------------------------------
// Batched column dots (GEMV-T shape): block (chunk, group) handles kColGroup consecutive
// columns j0 + group*kColGroup + i (or, with `cols`, the list entries cols[j0 + ...]) over one
// row chunk, reading r (and w) ONCE for the group:
//   out0 = sum w x r,  out1 = sum w x^2      (w = 1 when w == nullptr)
// partials[((group * nchunks + chunk) * kColGroup + i) * 2 + {0,1}], reduced by k_col_dots_reduce.
// ---------------------------------------------------------------------------------
constexpr int kColGroup = 8;

template <typename T>
__global__ __launch_bounds__(kBlock) void k_col_dots(const T* __restrict__ X, int64_t ld,
                                                     int64_t nvec, const T* __restrict__ w,
                                                     const T* __restrict__ r, int64_t j0, int ncols,
                                                     const int64_t* __restrict__ cols,
                                                     double* __restrict__ partials) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    __shared__ double lds[2 * kColGroup * (kBlock / 64)];
    const int c0 = blockIdx.y * kColGroup;
    const V* cv[kColGroup];
#pragma unroll
    for (int i = 0; i < kColGroup; ++i) {  // columns past the end alias the group's first (discarded)
        const int64_t ci = j0 + (c0 + i < ncols ? c0 + i : c0);
        cv[i] = reinterpret_cast<const V*>(X + (cols ? cols[ci] : ci) * ld);   // list entry or column
    }
    const V* rv = reinterpret_cast<const V*>(r);
    const V* wv = reinterpret_cast<const V*>(w);
    double acc[2 * kColGroup];
#pragma unroll
    for (int i = 0; i < 2 * kColGroup; ++i) acc[i] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec; j += stride) {
        V xv[kColGroup];
#pragma unroll
        for (int i = 0; i < kColGroup; ++i) xv[i] = ld_stream<true>(cv[i] + j);
        const V rr = rv[j];
        V wwv = rr;
        if (w) wwv = wv[j];
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const double re = (double)rr[e];
            const double we = w ? (double)wwv[e] : 1.0;
#pragma unroll
            for (int i = 0; i < kColGroup; ++i) {
                const double wx = we * (double)xv[i][e];
                acc[2 * i] = fma(wx, re, acc[2 * i]);
                acc[2 * i + 1] = fma(wx, (double)xv[i][e], acc[2 * i + 1]);
            }
        }
    }
    block_sum<2 * kColGroup>(acc, lds);
    if (threadIdx.x == 0) {
        double* out = partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * kColGroup);
#pragma unroll
        for (int i = 0; i < 2 * kColGroup; ++i) out[i] = acc[i];
    }
}
// one wave per column: out[2*col + {0,1}] = sum over chunks
__global__ __launch_bounds__(64) void k_col_dots_reduce(const double* __restrict__ partials,
                                                        int nchunks, double* __restrict__ out) {
    const int col = blockIdx.x, grp = col / kColGroup, i = col % kColGroup;
    const double* pr = partials + (int64_t)grp * nchunks * (2 * kColGroup) + 2 * i;
    double s0 = 0.0, s1 = 0.0;
    for (int c = threadIdx.x; c < nchunks; c += 64) {
        s0 += pr[(int64_t)c * (2 * kColGroup)];
        s1 += pr[(int64_t)c * (2 * kColGroup) + 1];
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (threadIdx.x == 0) { out[2 * col] = s0; out[2 * col + 1] = s1; }
}

// sum (r - shift), sum (r - shift)^2 (+ sum w (r - shift)^2 when w) -> partials[(block, 4)]; rows beyond n (the zero pad of
// the last vectors) are left out when a shift is given (shift == 0: they contribute nothing anyway)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_resid_moments(int64_t nvec, const T* __restrict__ r,
                                                          const T* __restrict__ w,
                                                          double* __restrict__ partials, double shift = 0.0, int64_t n = 0) {
    using V = typename VecOf<T>::V;
    constexpr int NV = VecOf<T>::N;
    __shared__ double lds[3 * (kBlock / 64)];
    const V* rv = reinterpret_cast<const V*>(r);
    const V* wv = reinterpret_cast<const V*>(w);
    double acc[3] = {0.0, 0.0, 0.0};
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nvec; j += stride) {
        const V rr = rv[j];
        const T* rp = reinterpret_cast<const T*>(&rr);
        V wwv = rr;
        if (w) wwv = wv[j];
        const T* wp = reinterpret_cast<const T*>(&wwv);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            double re = (double)rp[e] - shift;
            if (shift != 0.0 && j * NV + e >= n) re = 0.0;
            acc[0] += re;
            acc[1] = fma(re, re, acc[1]);
            acc[2] = fma(w ? (double)wp[e] * re : re, re, acc[2]);
        }
    }
    block_sum_store<3>(acc, lds, partials);
}
// sum the value-major partials -> red[0..2]
__global__ __launch_bounds__(kBlock) void k_sum_records(const double* __restrict__ partials,
                                                        int nparts, double* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (wid < 3) {
        const double s = wave_sum_run(partials + (int64_t)wid * nparts, nparts, lane);
        if (lane == 0) red[wid] = s;
    }
    if (threadIdx.x == 0) red[3] = 0.0;
}

// ---------------------------------------------------------------------------------
// Synthetic Gaussian problem (benchmark/cd_bench.jl:8-14 shapes): Philox4x32-10 keyed
// by the seed, counter (global row pair, column, stream) + Box-Muller, so any row shard
// of any process regenerates exactly the same X.
// ---------------------------------------------------------------------------------
struct Philox {
    __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
        const uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
    }
    __host__ __device__ static inline void run(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
        for (int rnd = 0; rnd < 10; ++rnd) {
            uint32_t hi0, lo0, hi1, lo1;
            mulhilo(0xD2511F53u, c[0], hi0, lo0);
            mulhilo(0xCD9E8D57u, c[2], hi1, lo1);
            const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
            c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    // two uniforms in (0,1) with 53 random bits each
    __host__ __device__ static inline void uniforms(uint64_t seed, uint64_t ctr, uint32_t col,
                                                    uint32_t stream, double& u1, double& u2) {
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), col, stream};
        run(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double s = 1.0 / 9007199254740992.0;  // 2^-53
        u1 = (((double)(c[0] >> 5)) * 67108864.0 + (double)(c[1] >> 6) + 0.5) * s;
        u2 = (((double)(c[2] >> 5)) * 67108864.0 + (double)(c[3] >> 6) + 0.5) * s;
    }
    // standard normal number `gi` of (column, stream): pair gi/2, member gi&1
    __host__ __device__ static inline double normal(uint64_t seed, uint64_t gi, uint32_t col,
                                                    uint32_t stream) {
        double u1, u2;
        uniforms(seed, gi >> 1, col, stream, u1, u2);
        const double rad = sqrt(-2.0 * log(u1));
        const double th = 6.283185307179586476925286766559 * u2;
        return (gi & 1) ? rad * sin(th) : rad * cos(th);
    }
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_gen_X(T* __restrict__ X, int64_t ld, int64_t n,
                                                  int64_t row0, int64_t j0, uint64_t seed) {
    const int64_t col = j0 + blockIdx.y;
    T* xc = X + col * ld;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    // one thread per row PAIR of the global index space so each Box-Muller draw is used twice
    const int64_t gp0 = row0 >> 1, gp1 = (row0 + n + 1) >> 1;
    for (int64_t gp = gp0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; gp < gp1; gp += stride) {
        double u1, u2;
        Philox::uniforms(seed, (uint64_t)gp, (uint32_t)col, 0u, u1, u2);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
        const int64_t i0 = 2 * gp - row0, i1 = i0 + 1;
        if (i0 >= 0 && i0 < n) xc[i0] = (T)(rad * cs);
        if (i1 >= 0 && i1 < n) xc[i1] = (T)(rad * sn);
    }
}
// y_i = sum_{j<s} X_ij beta*_j + noise * e_i ; r = y
template <typename T>
__global__ __launch_bounds__(kBlock) void k_gen_y(const T* __restrict__ X, int64_t ld, int64_t n,
                                                  int64_t row0, int64_t s,
                                                  const double* __restrict__ bstar, double noise,
                                                  uint64_t seed, T* __restrict__ y,
                                                  T* __restrict__ r) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double acc = 0.0;
        for (int64_t j = 0; j < s; ++j) acc = fma((double)X[j * ld + i], bstar[j], acc);
        const uint64_t gi = (uint64_t)(row0 + i);
        double u1, u2;
        Philox::uniforms(seed, gi >> 1, 0u, 1u, u1, u2);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
        acc += noise * ((gi & 1) ? rad * sn : rad * cs);
        y[i] = (T)acc;
        r[i] = (T)acc;
    }
}

}  // namespace cdk
