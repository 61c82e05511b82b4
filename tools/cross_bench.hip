// cross_bench.hip -- A/B harness for the Gram-column kernels (k_cross vs k_cross2 and its variants) on one GPU:
// same X (generated on the device), same batch of B columns, outputs compared entry by entry, each kernel timed with
// HIP events.  Measurement helper only (tools/README.md); build: tools/build_cross_bench.sh.
//   usage: cross_bench <rows> <cols> [reps] [f32]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cross_variants.hpp"

using namespace cdk;

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

template <typename T> int run(int64_t n, int64_t p, int reps) {
    constexpr int NV = VecOf<T>::N;
    const int64_t ld = (n + 31) / 32 * 32, nvec = (n + NV - 1) / NV;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    T* X; T* w;
    CK(hipMalloc(&X, sizeof(T) * (size_t)ld * (size_t)p));
    CK(hipMalloc(&w, sizeof(T) * (size_t)ld));
    CK(hipMemset(X, 0, sizeof(T) * (size_t)ld * (size_t)p));
    const int64_t pairs = (n + 3) / 2;
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (pairs + kBlock - 1) / kBlock));
    for (int64_t j0 = 0; j0 < p; j0 += 32768) {
        const int64_t nc = std::min<int64_t>(32768, p - j0);
        hipLaunchKernelGGL(k_gen_X<T>, dim3(gx, (unsigned)nc), dim3(kBlock), 0, 0, X, ld, n, (int64_t)0, j0, (uint64_t)123);
    }
    {   // weights in [0.5, 1.5): column 0 of X squashed
        std::vector<T> hw((size_t)ld, (T)0);
        for (int64_t i = 0; i < n; ++i) hw[(size_t)i] = (T)(0.5 + (double)((i * 2654435761u) % 1000) / 1000.0);
        CK(hipMemcpy(w, hw.data(), sizeof(T) * (size_t)ld, hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());
    const int64_t groups = (p + kCrossA - 1) / kCrossA, nslabs = (nvec + kCrossSlab - 1) / kCrossSlab;
    const int64_t nsuper = (groups + kGramWaves - 1) / kGramWaves;
    const char* gxe = getenv("CDH_CROSS_GX");
    const int GX = (int)std::max<int64_t>(1, std::min<int64_t>(nsuper, gxe ? atoi(gxe) : 4));
    const int J = (int)std::max<int64_t>(1, std::min<int64_t>(nslabs, (2 * (int64_t)cus) / GX));
    const int J3 = (int)std::max<int64_t>(1, std::min<int64_t>(nslabs, ((int64_t)kCrossOcc * cus) / GX));   // the library's grid
    double *part, *out0, *out1;
    CK(hipMalloc(&part, sizeof(double) * (size_t)groups * J * kCrossRec));
    double* part3;
    CK(hipMalloc(&part3, sizeof(double) * (size_t)groups * (size_t)(4 * cus / GX + 1) * kCrossRec));
    CK(hipMalloc(&out0, sizeof(double) * (size_t)groups * kCrossRec));
    CK(hipMalloc(&out1, sizeof(double) * (size_t)groups * kCrossRec));
    int64_t hcols[kCrossB];
    for (int b = 0; b < kCrossB; ++b) hcols[b] = (b * 37 + 5) % p;
    int64_t* dcols;
    CK(hipMalloc(&dcols, sizeof hcols));
    CK(hipMemcpy(dcols, hcols, sizeof hcols, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = (double)n * p * sizeof(T) / 1e9;
    printf("n=%lld p=%lld %s  X=%.1f GB  grid (%d,%d)  CUs %d\n", (long long)n, (long long)p, sizeof(T) == 8 ? "f64" : "f32", gb, GX, J, cus);
    std::vector<double> ref((size_t)groups * kCrossRec), got(ref.size());
    auto timeit = [&](const char* name, auto launch, double* out, int nbc) {
        float best = 1e30f, sum = 0.f;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipMemset(part, 0, sizeof(double) * (size_t)groups * J * kCrossRec));
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) { best = std::min(best, ms); sum += ms; }
        }
        hipLaunchKernelGGL(k_cross_reduce, dim3(kCrossRec / 256, (unsigned)groups), dim3(256), 0, 0, part, J, out);
        CK(hipDeviceSynchronize());
        printf("%-34s nbc %2d  best %8.3f ms  mean %8.3f ms  %6.2f TB/s of X\n", name, nbc, best, sum / reps, gb / best);
        fflush(stdout);
    };
    auto compare = [&](const char* name, double* a, double* b, int nbc, bool weighted) {
        CK(hipMemcpy(ref.data(), a, sizeof(double) * ref.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), b, sizeof(double) * got.size(), hipMemcpyDeviceToHost));
        double worst = 0.0, scale = 0.0;
        for (int64_t k = 0; k < p; ++k)
            for (int b2 = 0; b2 < nbc; ++b2) {
                const int64_t L = k / kCrossA, i = k % kCrossA;
                const size_t at = (size_t)(L * kCrossRec + ((i >> 4) * kCrossTB + (b2 >> 4)) * 256 + (i & 15) * 16 + (b2 & 15));
                worst = std::max(worst, std::fabs(ref[at] - got[at]));
                scale = std::max(scale, std::fabs(ref[at]));
            }
        printf("   %-30s max |diff| %.3e  (largest entry %.3e)%s\n", name, worst, scale, weighted ? "  [weighted]" : "");
        return worst <= (sizeof(T) == 4 ? 2e-6 : 1e-9) * scale;   // fp32 storage: the product kernel sums 256-row chunks in fp32 (round 4)
    };
    bool ok = true;
    for (int nbc : {32, 16, 5}) {
        timeit("k_cross_frag (round 2 kernel)", [&] {
            hipLaunchKernelGGL(k_cross_frag<T>, dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, part); }, out0, nbc);
        timeit("k_cross2 (LDS-DMA, plain)", [&] {
            hipLaunchKernelGGL((k_cross2<T, false, 0>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        ok = compare("k_cross2 plain vs k_cross", out0, out1, nbc, false) && ok;
        timeit("k_cross R=4 occ2 plain", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 0, 4, 2, false>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        ok = compare("k_cross R=4 occ2 plain vs round 2", out0, out1, nbc, false) && ok;
        timeit("k_cross R=4 occ2 nt", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 2, 4, 2, false>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        ok = compare("k_cross R=4 occ2 nt vs round 2", out0, out1, nbc, false) && ok;
        timeit("k_cross PRODUCT (R=2 occ3 nt prio)", [&] {
            hipLaunchKernelGGL((k_cross<T, false>), dim3(GX, J3), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part3); }, out1, nbc);
        if (sizeof(T) == 4) {
            const int J2 = (int)std::max<int64_t>(1, std::min<int64_t>(nslabs, ((int64_t)cross_occ<T>() * cus) / GX));
            timeit("k_cross PRODUCT f32 (fp32 pipe, occ2)", [&] {
                hipLaunchKernelGGL((k_cross<T, false, 2, 2, cross_occ<T>()>), dim3(GX, J2), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part3); }, out1, nbc);
        }
        timeit("k_cross2 (LDS-DMA, nt A)", [&] {
            hipLaunchKernelGGL((k_cross2<T, false, 2>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        ok = compare("k_cross2 nt vs k_cross", out0, out1, nbc, false) && ok;
        if (nbc != 32) continue;
        auto clocks = [&](const char* name) {
            std::vector<unsigned long long> st(4 * 2048);
            CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_cross_stamps), sizeof(unsigned long long) * st.size()));
            std::vector<double> ghz;
            for (int b = 0; b < std::min(2048, GX * J); ++b) {
                const double dc = (double)(st[4 * b + 2] - st[4 * b + 0]), dr = (double)(st[4 * b + 3] - st[4 * b + 1]);
                if (dr > 0) ghz.push_back(dc / dr * 0.1);
            }
            std::sort(ghz.begin(), ghz.end());
            if (!ghz.empty()) printf("   %-30s in-kernel clock: median %.3f GHz (min %.3f, max %.3f)\n", name, ghz[ghz.size() / 2], ghz.front(), ghz.back());
        };
        timeit("k_cross2 nt + stamps", [&] {
            hipLaunchKernelGGL((k_cross2<T, false, 2, 4>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        clocks("combined");
        timeit("k_cross2 loads only + stamps", [&] {
            hipLaunchKernelGGL((k_cross2<T, false, 2, 5>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        clocks("loads only");
        timeit("k_cross2 matrix only + stamps", [&] {
            hipLaunchKernelGGL((k_cross2<T, false, 2, 6>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        clocks("matrix only");
        timeit("k_cross R=2 occ3 nt", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 2, 2, 3, false>), dim3(GX, (unsigned)std::min<int64_t>(nslabs, 3 * cus / GX)), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part3); }, out1, nbc);
        timeit("k_cross R=2 occ3 nt prio", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 2, 2, 3, true>), dim3(GX, (unsigned)std::min<int64_t>(nslabs, 3 * cus / GX)), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part3); }, out1, nbc);
        timeit("k_cross R=2 occ3 plain", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 0, 2, 3, false>), dim3(GX, (unsigned)std::min<int64_t>(nslabs, 3 * cus / GX)), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part3); }, out1, nbc);
        timeit("k_cross R=2 occ4 nt", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 2, 2, 4, false>), dim3(GX, (unsigned)std::min<int64_t>(nslabs, 4 * cus / GX)), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part3); }, out1, nbc);
        timeit("k_cross R=2 occ3 nt (2 row lanes/CU grid)", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 2, 2, 3, false>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        timeit("k_cross R=4 occ2 nt prio", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 2, 4, 2, true>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        ok = compare("k_cross R=4 occ2 nt prio vs round 2", out0, out1, nbc, false) && ok;
        timeit("k_cross R=6 occ1 nt", [&] {
            hipLaunchKernelGGL((k_cross<T, false, 2, 6, 1, false>), dim3(GX, (unsigned)std::min<int64_t>(nslabs, cus / GX)), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        timeit("k_cross2 EXP loads only", [&] {
            hipLaunchKernelGGL((k_cross2<T, false, 2, 1>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        timeit("k_cross2 EXP matrix work only", [&] {
            hipLaunchKernelGGL((k_cross2<T, false, 2, 2>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)nullptr, part); }, out1, nbc);
        timeit("k_cross2 (LDS-DMA, weights)", [&] {
            hipLaunchKernelGGL((k_cross2<T, true, 0>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)w, part); }, out0, nbc);
        timeit("k_cross PRODUCT, weights", [&] {
            hipLaunchKernelGGL((k_cross<T, true>), dim3(GX, J), dim3(64 * kGramWaves), 0, 0, X, ld, nvec, p, dcols, nbc, (const T*)w, part); }, out1, nbc);
        ok = compare("k_cross weights vs k_cross2 weights", out0, out1, nbc, true) && ok;
        // weighted check against a host sum on a few entries (small problems only)
        if ((double)n * p < 3e8) {
            std::vector<T> hX((size_t)ld * p), hw((size_t)ld);
            CK(hipMemcpy(hX.data(), X, sizeof(T) * hX.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(hw.data(), w, sizeof(T) * hw.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), out1, sizeof(double) * got.size(), hipMemcpyDeviceToHost));
            double worst = 0.0, scale = 0.0;
            for (int64_t k = 0; k < p; k += std::max<int64_t>(1, p / 97))
                for (int b2 = 0; b2 < kCrossB; b2 += 3) {
                    double sacc = 0.0;
                    for (int64_t i = 0; i < n; ++i) sacc += (double)hX[(size_t)(k * ld + i)] * (double)hw[(size_t)i] * (double)hX[(size_t)(hcols[b2] * ld + i)];
                    const int64_t L = k / kCrossA, i = k % kCrossA;
                    const size_t at = (size_t)(L * kCrossRec + ((i >> 4) * kCrossTB + (b2 >> 4)) * 256 + (i & 15) * 16 + (b2 & 15));
                    worst = std::max(worst, std::fabs(sacc - got[at]));
                    scale = std::max(scale, std::fabs(sacc));
                }
            printf("   weighted k_cross2 vs host sums: max |diff| %.3e (largest %.3e)\n", worst, scale);
            ok = ok && worst <= 1e-9 * scale;
        }
    }
    printf(ok ? "CROSS_BENCH_OK\n" : "CROSS_BENCH_MISMATCH\n");
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 2000000, p = argc > 2 ? atoll(argv[2]) : 5000;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const bool f32 = argc > 4 && argv[4][0] == 'f' && argv[4][1] == '3';
    return f32 ? run<float>(n, p, reps) : run<double>(n, p, reps);
}
