// grad_cache.hpp -- the gradient cache's passes: a section of cdhip.hip kept in its own file (included
// once, inside cdhip.hip's anonymous namespace, after the handle, col_dots, allreduce and the grid helpers
// it uses; run_chunk is defined further down in cdhip.hip).  The bookkeeping half (what invalidates the
// cache, how moves are noted) stays next to the code that triggers it in cdhip.hip.
//
// What it replaces: the n-sized reads of X that the reference's full passes spend on coordinates that do not
// move (coordinate_descent.jl:94-110 over 1..p with descendCoordinate!'s dots, cd_differentiable_function.jl:
// 94-99).  No reference counterpart; same iterates.  Design and measurements: DESIGN.md section 5a.
#pragma once

void gc_fold(cdh_handle h);
int32_t finish_chunk(cdh_handle h, const int64_t* idx0, int m, double* maxH);

// On row shards every rank must take the same turns (a rank that streams a chunk calls the exchange, a rank
// that serves it from its cache does not): whatever depends on a local resource -- here: did the cache's
// buffers fit on THIS device -- is agreed on through the exchange itself.  One rank: the local answer.
int32_t all_ranks_agree(cdh_handle h, bool mine, bool* all) {
    *all = mine;
    if (!sharded(h) || h->nranks <= 1) return CDH_OK;
    h->h_red[0] = mine ? 1.0 : 0.0;
    HIPCHK(h, hipMemcpyAsync(h->d_red, h->h_red, sizeof(double), hipMemcpyHostToDevice, h->stream));
    CHK(allreduce(h, h->d_red, 1));
    HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *all = h->h_red[0] == (double)h->nranks;
    return CDH_OK;
}

// ---- gradient cache: the certified full pass -------------------------------------------------------------
int32_t gc_size(cdh_handle h) {   // first use on this handle
    GradCache& c = h->gc;
    if (!c.g.empty()) return CDH_OK;
    const size_t p = (size_t)h->p;
    c.g.assign(p, 0.0); c.a.assign(p, 0.0); c.dbeta.assign(p, 0.0);
    c.in_moved.assign(p, 0); c.slot.assign(p, -1);
    c.beta_ref.assign(p, 0.0);
    c.beta_ok = h->r_consistent;          // r == y - X * (the handle's iterate) right now?
    if (c.beta_ok)
        for (int64_t s_ = 0; s_ < h->x.nnz(); ++s_) c.beta_ref[(size_t)h->x.coord(s_)] = h->x.slot_value(s_);
    // one k_cross launch per batch: a block per (group of 64 columns, j of J), about two resident blocks per CU
    const int64_t launches = (h->p + kCrossA - 1) / kCrossA;      // column groups
    const int64_t nslabs = (h->nvec + kCrossSlab - 1) / kCrossSlab;
    // gridDim.x = column groups in flight (super-groups of 4 x 64 columns; env CDH_CROSS_GX, default 4), gridDim.y = row lanes; never more blocks
    // than stay resident, kCrossOcc per CU (a partly filled second round would double the time)
    const char* gxe = getenv("CDH_CROSS_GX");
    const int64_t nsuper = (launches + kGramWaves - 1) / kGramWaves;   // a block's four waves take four column groups
    c.cross_GX = (int)std::max<int64_t>(1, std::min<int64_t>(nsuper, gxe ? atoi(gxe) : 4));
    const int64_t occ = h->dtype == CDH_F32 ? cross_occ<float>() : cross_occ<double>();
    c.cross_J = (int)std::max<int64_t>(1, std::min<int64_t>(nslabs, (occ * h->cus) / c.cross_GX));
    // short columns have few row slabs to hand out (n = 3000: two): the resident blocks they leave unused take further
    // column super-groups instead (benchmark/cd_bench.jl's shape: 8 blocks walked 79 column groups, 0.78 ms per batch of a
    // 120 MB X; 40 blocks: one group per wave)
    if (!gxe) c.cross_GX = (int)std::min<int64_t>(nsuper, std::max<int64_t>(c.cross_GX, (occ * h->cus) / c.cross_J));
    // (a handle whose first sizing failed may be asked again: what a failed attempt got was freed below)
    bool fits = hipMalloc((void**)&c.d_cross, sizeof(double) * (size_t)launches * kCrossRec) == hipSuccess &&
                hipMalloc((void**)&c.d_cross_part, sizeof(double) * (size_t)launches * (size_t)c.cross_J * kCrossRec) == hipSuccess &&
                hipMalloc((void**)&c.d_cols, sizeof(int64_t) * kCrossB) == hipSuccess;
    if (!fits) (void)hipGetLastError();
    CHK(all_ranks_agree(h, fits, &fits));
    if (!fits) {                      // no room for the cache's scratch (on some rank): the dots-only screens stay
        if (c.d_cross) (void)hipFree(c.d_cross);
        if (c.d_cross_part) (void)hipFree(c.d_cross_part);
        if (c.d_cols) (void)hipFree(c.d_cols);
        c.d_cross = nullptr; c.d_cross_part = nullptr; c.d_cols = nullptr;
        c.mode = 0;
        c.g.clear(); c.g.shrink_to_fit();
        return CDH_OK;
    }
    c.h_cross.assign((size_t)launches * kCrossRec, 0.0);
    return CDH_OK;
}

// ---- device mirrors for the covariance-form visits ------------------------------------------------------
// The Gram columns the host holds are mirrored slot-major on the device (p doubles each), with the slot map;
// g is sent down before a chunk and comes back after it, so the host copy stays the one truth.
// room on the device for `have` Gram columns and for what the device-side passes need (first use: the buffers; later: the
// store grows 64 columns at a time).  c.cov turns false if it cannot be had (on any rank).
int32_t gc_dev_reserve(cdh_handle h, int64_t have) {
    GradCache& c = h->gc;
    if (!c.cov) return CDH_OK;
    const int64_t p = h->p;
    // no room on the device for the mirrors (on any rank) is not an error: the visits stay in residual form
    if (!c.d_g) {
        bool fits = hipMalloc((void**)&c.d_g, sizeof(double) * (size_t)p) == hipSuccess &&
                    hipMalloc((void**)&c.d_slot, sizeof(int32_t) * (size_t)p) == hipSuccess &&
                    // pinned staging for g on its way down and back (pageable copies are staged by the
                    // runtime, one hidden synchronisation each: two per chunk of visits)
                    hipHostMalloc((void**)&c.h_g_pin, sizeof(double) * 2 * (size_t)p) == hipSuccess;
        // ... and what a whole full pass on the device needs (gc_pass_device): a, the snapshots, the per-coordinate
        // position / settled flag, the compacted visit list, the per-visit r'r, the scan's counters
        fits = fits && hipMalloc((void**)&c.d_a, sizeof(double) * (size_t)p) == hipSuccess &&
               hipMalloc((void**)&c.d_g_snap, sizeof(double) * (size_t)p) == hipSuccess &&
               hipMalloc((void**)&c.d_beta_snap, sizeof(double) * (size_t)p) == hipSuccess &&
               hipMalloc((void**)&c.d_qs, sizeof(double) * (size_t)h->cap) == hipSuccess &&
               hipMalloc((void**)&c.d_pass_idx, sizeof(int64_t) * (size_t)h->cap) == hipSuccess &&
               hipMalloc((void**)&c.d_pos_of, sizeof(int32_t) * (size_t)p) == hipSuccess &&
               hipMalloc((void**)&c.d_scanbuf, sizeof(int32_t) * (size_t)(h->cap + 4)) == hipSuccess &&
               hipMalloc((void**)&c.d_setflag, 2 * (size_t)p) == hipSuccess &&       // [settled flags][forced marks]
               hipMalloc((void**)&c.d_pack, sizeof(double) * (size_t)(kPackHead + 3 * h->cap)) == hipSuccess &&
               hipHostMalloc((void**)&c.h_scanbuf, sizeof(int32_t) * (size_t)(h->cap + 4)) == hipSuccess &&
               hipHostMalloc((void**)&c.h_pack, sizeof(double) * (size_t)(kPackHead + 3 * h->cap)) == hipSuccess;
        if (!fits) (void)hipGetLastError();
        CHK(all_ranks_agree(h, fits, &fits));
        if (!fits) { c.cov = false; return CDH_OK; }
        c.d_scan = reinterpret_cast<cdk::CovScanOut*>(c.d_scanbuf); c.d_upos = c.d_scanbuf + 4;
        c.h_scan = reinterpret_cast<cdk::CovScanOut*>(c.h_scanbuf); c.h_upos = c.h_scanbuf + 4;
        HIPCHK(h, hipMemsetAsync(c.d_qs, 0, sizeof(double) * (size_t)h->cap, h->stream));
        HIPCHK(h, hipMemsetAsync(c.d_setflag, 0, 2 * (size_t)p, h->stream));
        c.d_forced = c.d_setflag + p; c.forced_dirty = false;
        c.g_dev_ok = false; c.a_dev_ok = false;
    }
    if (have > c.dev_slots_cap) {   // grow the store (at least 64 columns more, doubling once it holds 128: a path that ends at 800 columns
                                    // reallocates 5 times, not 14 -- each is a hipMalloc, a copy and a hipFree, milliseconds apiece on a
                                    // fresh handle; at most kGcMaxBytes), keeping what is there
        const int64_t most = (int64_t)(kGcMaxBytes / sizeof(double)) / std::max<int64_t>(p, 1);
        const int64_t cap = std::min<int64_t>(std::max<int64_t>((have + 63) / 64 * 64 + 64, c.dev_slots_cap >= 128 ? 2 * c.dev_slots_cap : 0), most);
        if (cap < have) { c.cov = false; return CDH_OK; }   // does not fit (the same on every rank): residual form
        double* bigger = nullptr;
        bool fits = hipMalloc((void**)&bigger, sizeof(double) * (size_t)cap * (size_t)p) == hipSuccess;
        if (!fits) (void)hipGetLastError();
        CHK(all_ranks_agree(h, fits, &fits));
        if (!fits) { if (bigger) (void)hipFree(bigger); c.cov = false; return CDH_OK; }
        if (c.d_G && c.dev_slots > 0)
            HIPCHK(h, hipMemcpyAsync(bigger, c.d_G, sizeof(double) * (size_t)c.dev_slots * (size_t)p, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        // (the store it replaces is released with the handle: hipFree synchronises the device and, after large allocations have gone,
        // can take tens of milliseconds -- seen as one 66 ms solve in the first path of a process that had freed 80 GB; the stores a
        // handle outgrows add up to less than the one it ends with)
        if (c.d_G) c.d_G_retired.push_back(c.d_G);
        c.d_G = bigger; c.dev_slots_cap = cap;
    }
    return CDH_OK;
}
int32_t gc_dev_upload(cdh_handle h) {
    GradCache& c = h->gc;
    if (!c.cov) return CDH_OK;
    const int64_t p = h->p, have = (int64_t)c.G.size();
    CHK(gc_dev_reserve(h, have));
    if (!c.cov) return CDH_OK;
    for (int64_t s_ = c.dev_slots; s_ < have; ++s_)       // (columns unpacked on the device are there already: gc_fetch)
        HIPCHK(h, hipMemcpyAsync(c.d_G + s_ * p, c.G[(size_t)s_].data(), sizeof(double) * (size_t)p, hipMemcpyHostToDevice, h->stream));
    c.dev_slots = have;
    HIPCHK(h, hipMemcpyAsync(c.d_slot, c.slot.data(), sizeof(int32_t) * (size_t)p, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    c.slot_dev_ok = true;
    return CDH_OK;
}

// The host's copy of Gram column `s`: columns that went straight into the device store (gc_fetch) come over when host
// code first asks for them -- the careful walk's re-check, a fold without device mirrors, cdh_cache_gram_column.
int32_t gc_host_column(cdh_handle h, int64_t s_) {
    GradCache& c = h->gc;
    std::vector<double>& col = c.G[(size_t)s_];
    if (!col.empty()) return CDH_OK;
    if (!c.d_G || s_ >= c.dev_slots) return fail(h, CDH_BAD_ARG, "gradient cache: a Gram column is neither on the host nor on the device");
    col.resize((size_t)h->p);
    HIPCHK(h, hipMemcpyAsync(col.data(), c.d_G + s_ * h->p, sizeof(double) * (size_t)h->p, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

// ---- which copy of g is current ------------------------------------------------------------------------------
// The device-side passes leave g in d_g; host code that reads c.g (the legacy walk, the choice of columns to
// fetch, a re-reference's drift measurement) asks for it first.  The other direction likewise.
int32_t gc_need_host_g(cdh_handle h) {
    GradCache& c = h->gc;
    if (c.g_host_ok) return CDH_OK;
    HIPCHK(h, hipMemcpyAsync(c.h_g_pin, c.d_g, sizeof(double) * (size_t)h->p, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::memcpy(c.g.data(), c.h_g_pin, sizeof(double) * (size_t)h->p);
    c.g_host_ok = true;
    return CDH_OK;
}
int32_t gc_need_dev_g(cdh_handle h) {
    GradCache& c = h->gc;
    if (!c.a_dev_ok) {
        HIPCHK(h, hipStreamSynchronize(h->stream));                 // h_g_pin's second half doubles as staging for a
        std::memcpy(c.h_g_pin + h->p, c.a.data(), sizeof(double) * (size_t)h->p);
        HIPCHK(h, hipMemcpyAsync(c.d_a, c.h_g_pin + h->p, sizeof(double) * (size_t)h->p, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        c.a_dev_ok = true;
    }
    if (c.g_dev_ok) return CDH_OK;
    std::memcpy(c.h_g_pin, c.g.data(), sizeof(double) * (size_t)h->p);
    HIPCHK(h, hipMemcpyAsync(c.d_g, c.h_g_pin, sizeof(double) * (size_t)h->p, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));                     // the staging buffer is free again
    c.g_dev_ok = true;
    return CDH_OK;
}

// What the certificates must allow for besides the 1e-9 relative margin: with fp32 storage the residual is rounded to
// fp32 by every launch that rewrites it, and the streamed kernels' own dots carry fp32 chunk sums -- the cache's g,
// carried in fp64 from Gram entries summed in fp64, describes a slightly DIFFERENT residual than the one the exact
// visit would read.  A rounding of relative size 2^-24 per element and rewrite, U rewrites since the residual was last
// rebuilt, uncorrelated with the column: |X_k'(r_fp32 - r)| ~ 2^-24 sqrt(U a_k r'r / n); with r'r <= y'y along a path
// and a factor 64 of safety:   cert_abs sqrt(a_k),  cert_abs = 64 * 2^-24 * sqrt((U + 1) y'y / n).   fp64 storage: 0.
constexpr double kCrossF32EpsFactor = 512.0;   // eps_G = 2^-24 * this / sqrt(n_total)
int32_t gc_cert_abs(cdh_handle h, double* out) {
    GradCache& c = h->gc;
    *out = 0.0;
    if (h->dtype != CDH_F32) return CDH_OK;
    if (!c.yy_ok) {
        CHK(resid_moments_dev(h, h->y));
        HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        c.yy = h->h_red[1];
        c.yy_ok = true;
    }
    // ... plus (round 4) what the Gram entries themselves carry: k_cross takes fp32 storage on the fp32 matrix pipe, 256-row
    // partial sums in fp32 folded into fp64 tiles, so an entry is good to eps_G sqrt(a_i a_j) with eps_G ~ 2^-24 * 128 / sqrt(n)
    // for fully correlated columns (measured: tests/test_gpu_parity.py::test_fp32_gram_columns_carry_the_declared_error); a
    // carried gradient updated with them is off by at most eps_G sqrt(a_k) sum_j |dbeta_j| sqrt(a_j) <~ eps_G sqrt(a_k y'y).
    // Declared with a factor 4: 2^-24 * 512 / sqrt(n) -- eight times the residual's own term at U = 0.
    *out = 5.9604644775390625e-8 * std::sqrt(c.yy / (double)h->n_total) * (64.0 * std::sqrt((double)(h->r_roundings + 1)) + kCrossF32EpsFactor);
    return CDH_OK;
}

// May visits idx0[0..m) run in covariance form?  Least squares in fp64, a current g with nothing left to
// fold, and a Gram column on the device for every coordinate of the chunk.
bool cov_ok(cdh_handle h, const int64_t* idx0, int64_t m) {
    const GradCache& c = h->gc;
    if (!c.cov || !c.valid || !c.moved.empty() || !gc_applicable(h) || !c.d_G) return false;
    if (h->loss == CDH_SQRT && !c.q_valid) return false;
    if (c.dev_slots != (int64_t)c.G.size()) return false;
    for (int64_t i = 0; i < m; ++i) if (c.slot[(size_t)idx0[i]] < 0) return false;
    return true;
}

// r'r of the residual g describes, for the sqrt-lasso thresholds and updates: carried through the
// covariance-form visits, re-read from r (one pass over r, after it has caught up) when a streamed visit,
// a rebuild or a re-reference has changed r behind its back
int32_t gc_ensure_q(cdh_handle h) {
    GradCache& c = h->gc;
    if (c.q_valid) return CDH_OK;
    CHK(resid_moments_dev(h));
    HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    c.q = h->h_red[1];
    c.q_exact = c.q;
    c.q_valid = true;
    return CDH_OK;
}
// The carried r'r is only good to ~1e-16 of the value its recurrence started from (q <- q - 2 h b + h^2 a, block after block,
// pass after pass): once it has fallen to kGcQGuard of the last value summed from r itself, it is summed afresh before the
// next pass (one pass over r, after it has caught up with the moves) -- near-noiseless problems, ||r|| << ||y||.
constexpr double kGcQGuard = 1e-4;
inline void gc_q_guard(cdh_handle h) {
    GradCache& c = h->gc;
    if (h->loss == CDH_SQRT && c.q_valid && c.q < kGcQGuard * c.q_exact) c.q_valid = false;
}

// fold what can be folded, then ask cov_ok (the active passes of a solve call this before every chunk)
int32_t gc_rereference(cdh_handle h);
bool gc_ready_for_cov(cdh_handle h, const int64_t* idx0, int64_t m) {
    GradCache& c = h->gc;
    if (!c.cov || !c.valid || !gc_applicable(h) || !c.d_G) return false;
    // the same bounds the full passes keep (gc_full_pass): the cache only pays on tall problems, and g is carried
    // through a bounded number of covariance-form updates before it is taken afresh from X
    if (gc_support_outgrown(h)) { gc_invalidate(h, false); return false; }
    if (c.cov_since_ref > c.refresh_after && gc_rereference(h) != CDH_OK) return false;
    if (!c.moved.empty()) {
        for (int64_t j : c.moved) if (c.slot[(size_t)j] < 0) return false;
        gc_fold(h);
    }
    gc_q_guard(h);
    if (h->loss == CDH_SQRT && gc_ensure_q(h) != CDH_OK) return false;
    return cov_ok(h, idx0, m);
}


// Visits idx0[0..m) from the cache alone: record from (g, G) -> the B scalar updates -> g update, block by
// block; nothing n-sized is touched.  r learns about the moves later (sync_r).  With `chk` (a device-side full
// pass: the visit list is the scan's compaction, d_upos its positions) the g update also re-checks the skipped
// positions the block answers for (k_cov_gupdate_chk).
template <int NG> int32_t launch_cov_chunk(cdh_handle h, int m, bool chk = false, int m_pass = 0) {
    using R = GramRec<NG>;
    constexpr int B = R::B;
    GradCache& c = h->gc;
    for (int pos0 = 0; pos0 < m; pos0 += B) {
        const int nb = std::min(B, m - pos0);
        hipLaunchKernelGGL((k_cov_block<NG>), dim3(1), dim3(256), 0, h->stream, c.d_g, c.d_G, c.d_slot, h->p, h->d_idx, pos0, nb,
                           h->chunk_dup ? 1 : 0, h->d_ctrl, h->beta, h->omega, h->d_hs, h->d_newval, h->d_touched, c.d_qs,
                           h->d_red + R::OFF_Q);
        if (chk)
            hipLaunchKernelGGL(k_cov_gupdate_chk, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, h->stream, c.d_g, c.d_G, c.d_slot,
                               c.d_a, h->omega, h->d_ctrl, h->p, h->d_idx, h->d_hs, c.d_qs, c.d_upos, c.d_pos_of, c.d_setflag,
                               pos0, nb, m_pass, pos0 + nb >= m ? 1 : 0, h->d_red + R::OFF_Q, c.d_scan, c.d_forced);
        else
            hipLaunchKernelGGL(k_cov_gupdate, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, h->stream, c.d_g, c.d_G, c.d_slot, h->p,
                               h->d_idx, h->d_hs, pos0, nb);
    }
    return CDH_OK;
}
// the width of the handle's blocked sweep where it has one (same recurrence, same kernel), else 16
int32_t launch_cov_blocks(cdh_handle h, int m, bool chk = false, int m_pass = 0) {
    const int B = (h->mode == CDH_SWEEP_BLOCK && h->blockB >= 16) ? h->blockB : 16;
    if (B == 64) return launch_cov_chunk<4>(h, m, chk, m_pass);
    if (B == 32) return launch_cov_chunk<2>(h, m, chk, m_pass);
    return launch_cov_chunk<1>(h, m, chk, m_pass);
}

// the results of m covariance-form visits (and the control block, and with `scan` the re-check's verdict) back to the
// host's staging arrays: one gather kernel, one copy, one synchronisation
int32_t cov_fetch_results(cdh_handle h, int m, bool scan) {
    GradCache& c = h->gc;
    hipLaunchKernelGGL(k_cov_pack, dim3((unsigned)((std::max(m, 1) + 255) / 256)), dim3(256), 0, h->stream, h->d_ctrl,
                       scan ? c.d_scan : (const cdk::CovScanOut*)nullptr, h->d_hs, h->d_newval, h->d_touched, m, c.d_pack);
    HIPCHK(h, hipGetLastError());
    const size_t bytes = sizeof(double) * (size_t)(kPackHead + 2 * m) + sizeof(int32_t) * (size_t)m;
    HIPCHK(h, hipMemcpyAsync(c.h_pack, c.d_pack, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::memcpy(h->h_ctrl, c.h_pack, sizeof(Ctrl));
    if (scan) std::memcpy(c.h_scan, c.h_pack + 8, sizeof(cdk::CovScanOut));
    std::memcpy(h->h_hs, c.h_pack + kPackHead, sizeof(double) * (size_t)m);
    std::memcpy(h->h_newval, c.h_pack + kPackHead + m, sizeof(double) * (size_t)m);
    std::memcpy(h->h_touched, c.h_pack + kPackHead + 2 * (size_t)m, sizeof(int32_t) * (size_t)m);
    return CDH_OK;
}

// enqueue the visits, bring the results back into staging (h_hs / h_newval / h_touched / h_ctrl) -- nothing on the
// host has changed yet.  g stays on the device; with `want_g` (the legacy walk re-checks skipped certificates on
// the host) the updated g also comes back into c.g_new.
int32_t cov_run(cdh_handle h, const int64_t* idx0, int m, bool want_g) {
    GradCache& c = h->gc;
    CHK(gc_need_dev_g(h));
    std::memcpy(h->h_idx, idx0, sizeof(int64_t) * (size_t)m);
    note_duplicates(h, idx0, m);
    HIPCHK(h, hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    h->ctrl.maxH = 0.0;
    h->ctrl.domain_error = 0;
    h->ctrl.q_carry = c.q;
    CHK(upload_ctrl(h));
    CHK(launch_cov_blocks(h, m));
    HIPCHK(h, hipGetLastError());
    if (want_g) {
        c.g_new.resize((size_t)h->p);
        HIPCHK(h, hipMemcpyAsync(c.h_g_pin + h->p, c.d_g, sizeof(double) * (size_t)h->p, hipMemcpyDeviceToHost, h->stream));
    }
    CHK(cov_fetch_results(h, m, false));
    if (want_g) std::memcpy(c.g_new.data(), c.h_g_pin + h->p, sizeof(double) * (size_t)h->p);
    c.g_host_ok = false;              // d_g has moved on; c.g is the gradient BEFORE the chunk until it is accepted
    return CDH_OK;
}
// the staged results of visit i become real: SparseIterate writes, the move noted for r, maxH
void cov_apply_visit(cdh_handle h, int64_t k, int i) {
    if (h->h_touched[i] && h->x.get(k) == 0.0) h->x.set(k, 1.0);  // pre-prox non-zero: slot appended
    h->x.set(k, h->h_newval[i]);
    const double hv = h->h_hs[i];
    if (hv == 0.0) return;
    h->dots_valid = false;            // the residual the handle stands for moves
    if (!h->r_in_pending[(size_t)k]) { h->r_in_pending[(size_t)k] = 1; h->r_pending_list.push_back(k); }
    h->r_pending[(size_t)k] += hv;
    if (hv == hv && h->gc.beta_ok) h->gc.beta_ref[(size_t)k] += hv;
}
// the chunk is accepted: maxH, r'r, the counters; `have_g_new`: c.g_new is the gradient after the chunk
void cov_accept_tail(cdh_handle h, int m, double* maxH, bool have_g_new) {
    GradCache& c = h->gc;
    const double mh = h->h_ctrl->maxH;
    if (mh > *maxH) *maxH = mh;
    if (h->h_ctrl->domain_error) h->domain_error = true;
    if (h->loss == CDH_SQRT) c.q = h->h_ctrl->q_carry;
    if (have_g_new) { c.g.swap(c.g_new); c.g_host_ok = true; }
    c.n_cov += m;
    c.cov_since_ref += m;
    bool nan = false;
    for (int i = 0; i < m; ++i) nan = nan || (h->h_hs[i] != h->h_hs[i]);
    if (nan) gc_invalidate(h, false);   // r turns NaN at the catch-up, as in the reference; the cache knows nothing any more
}
// a chunk in which every position is visited (the active passes of a solve): g never leaves the device
int32_t cov_chunk(cdh_handle h, const int64_t* idx0, int m, double* maxH) {
    CHK(cov_run(h, idx0, m, false));
    for (int i = 0; i < m; ++i) cov_apply_visit(h, idx0[i], i);
    cov_accept_tail(h, m, maxH, false);
    return CDH_OK;
}
// the visits never happened: beta on the device goes back to what the host-side iterate still says, and the
// device's g (which has seen the chunk's moves) is void -- c.g still holds the gradient from before the chunk
int32_t cov_reject(cdh_handle h, const int64_t* idx0, int m) {
    GradCache& c = h->gc;
    for (int i = 0; i < m; ++i) { h->h_idx[i] = idx0[i]; h->h_hs[i] = h->x.get(idx0[i]); }
    HIPCHK(h, hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_hs, h->h_hs, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_scatter_f64, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, h->beta, h->d_idx, h->d_hs, m);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    c.g_host_ok = true; c.g_dev_ok = false;
    return CDH_OK;
}

// g = X'r and a = diag(X'X) (with observation weights: X'Wr, diag(X'WX)) from one dots-only pass over all of X: the
// new reference point
int32_t gc_validate(cdh_handle h) {
    GradCache& c = h->gc;
    std::vector<double> cd;
    if (h->dots_valid && h->dots_w == h->has_w && (int64_t)h->dots_stash.size() == 2 * h->p) {
        // the dots of all p columns with this very residual were taken a moment ago (_findLambdaMax, coordinate_descent.jl:118-149,
        // or cdh_xt_r, before the first solve of a path): that pass over X is the reference pass
        cd = h->dots_stash;
        h->n_dots_adopted += 1;
    } else {
        CHK(col_dots(h, 0, h->p, h->r, h->has_w));
        cd.resize((size_t)(2 * h->p));
        HIPCHK(h, hipMemcpyAsync(cd.data(), h->d_colout, sizeof(double) * 2 * h->p, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->dots_stash = cd; h->dots_valid = true; h->dots_w = h->has_w;
    }
    for (int64_t k = 0; k < h->p; ++k) { c.g[(size_t)k] = cd[(size_t)(2 * k)]; c.a[(size_t)k] = cd[(size_t)(2 * k + 1)]; }
    for (int64_t j : c.moved) { c.dbeta[(size_t)j] = 0.0; c.in_moved[(size_t)j] = 0; }
    c.moved.clear();
    c.valid = true;
    c.g_host_ok = true; c.g_dev_ok = false; c.a_dev_ok = false;
    c.n_validate += 1;
    c.cov_since_ref = 0;
    return CDH_OK;
}

// The dots of all p columns with the residual as it stands have just been taken for another purpose (_findLambdaMax at the
// head of a cold start, coordinate_descent.jl:29, right after initialize!): that pass over X is a reference pass the cache
// need not repeat.  `cd`: (X_k'W r, X_k'W X_k) pairs as col_dots leaves them.
int32_t gc_adopt_dots(cdh_handle h, const std::vector<double>& cd) {
    GradCache& c = h->gc;
    if (!gc_applicable(h) || c.mode == 0 || (int64_t)cd.size() != 2 * h->p) return CDH_OK;
    CHK(gc_size(h));
    if (c.mode == 0 || c.g.empty()) return CDH_OK;
    const bool beta_known = c.beta_ok;
    std::vector<double> beta_keep;
    if (beta_known) beta_keep = c.beta_ref;
    gc_invalidate(h, false);
    for (int64_t k = 0; k < h->p; ++k) { c.g[(size_t)k] = cd[(size_t)(2 * k)]; c.a[(size_t)k] = cd[(size_t)(2 * k + 1)]; }
    c.valid = true;
    c.g_host_ok = true; c.g_dev_ok = false; c.a_dev_ok = false;
    c.cov_since_ref = 0;
    c.beta_ok = beta_known;
    if (beta_known) c.beta_ref.swap(beta_keep);
    return CDH_OK;
}

// Take g afresh from X (one dots-only pass) and MEASURE how far the carried g had drifted from it:
//   drift = max_k |g_carried[k] - X_k'r| / thr_k   (thr_k: the threshold the certificates compare |g_k| with)
// -- the quantity the certificates' relative margin must cover.  Pending moves are folded first so that both sides
// describe the same residual; what is known about beta is kept (g will describe the same residual, freshly summed).
int32_t gc_rereference(cdh_handle h) {
    GradCache& c = h->gc;
    const bool beta_known = c.beta_ok;
    std::vector<double> beta_keep;
    if (beta_known) beta_keep = c.beta_ref;
    bool comparable = c.valid;
    if (comparable)
        for (int64_t j : c.moved) if (c.slot[(size_t)j] < 0) comparable = false;
    std::vector<double> carried;
    if (comparable) { gc_fold(h); CHK(gc_need_host_g(h)); carried = c.g; }
    gc_invalidate(h, false);
    CHK(gc_validate(h));
    c.beta_ok = beta_known;
    if (beta_known) c.beta_ref.swap(beta_keep);
    if (comparable && h->ctrl.lambda0 > 0.0) {
        double scale = h->ctrl.lambda0 * (double)h->n_total;
        if (h->loss == CDH_SQRT) { CHK(gc_ensure_q(h)); scale = h->ctrl.lambda0 * std::sqrt(c.q); }
        double worst = 0.0;
        for (int64_t k = 0; k < h->p; ++k) {
            const double thr = scale * (h->has_omega ? h->h_omega[(size_t)k] : 1.0);
            if (!(c.a[(size_t)k] > 0.0) || !(thr > 0.0)) continue;
            const double d = std::fabs(carried[(size_t)k] - c.g[(size_t)k]) / thr;
            if (d > worst) worst = d;
        }
        c.drift_last = worst;
        if (worst > c.drift_max) c.drift_max = worst;
        c.n_drift += 1;
    }
    return CDH_OK;
}

// G[(b0 + b) p + k] <- the cross products of one k_cross batch (records of 64 x 32, tile-major): the batch's columns, slot-major
__global__ __launch_bounds__(256) void k_cross_unpack(const double* __restrict__ cross, int64_t p, int b0, int nbc, double* __restrict__ Gcols) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p) return;
    const int64_t L = k / kCrossA, i = k % kCrossA;
    for (int b = 0; b < nbc; ++b)
        Gcols[(int64_t)(b0 + b) * p + k] = cross[L * kCrossRec + ((i >> 4) * kCrossTB + (b >> 4)) * 256 + (i & 15) * 16 + (b & 15)];
}

// Gram columns G_j = X'X_j for the coordinates in `cols` (those not cached yet), up to 32 per pass over X
int32_t gc_fetch(cdh_handle h, const std::vector<int64_t>& cols) {
    GradCache& c = h->gc;
    std::vector<int64_t> todo;
    for (int64_t j : cols) if (c.slot[(size_t)j] < 0 && std::find(todo.begin(), todo.end(), j) == todo.end()) todo.push_back(j);
    if (todo.empty()) return CDH_OK;
    if ((c.G.size() + todo.size()) * (size_t)h->p * sizeof(double) > kGcMaxBytes) {
        gc_invalidate(h, true);
        c.mode = 0;                       // this problem's supports do not fit: plain screens from now on
        return CDH_OK;
    }
    const int64_t launches = (h->p + kCrossA - 1) / kCrossA;      // column groups
    const char* bmax = getenv("CDH_CROSS_BATCH");      // experiments: fewer B columns per launch (16: one tile column)
    const size_t batch = (size_t)std::max(1, std::min(kCrossB, bmax ? atoi(bmax) : kCrossB));
    for (size_t b0 = 0; b0 < todo.size(); b0 += batch) {
        const int nbc = (int)std::min<size_t>(batch, todo.size() - b0);
        HIPCHK(h, hipMemcpyAsync(c.d_cols, todo.data() + b0, sizeof(int64_t) * (size_t)nbc, hipMemcpyHostToDevice, h->stream));
        CHK(dispatch(h, [&](auto* t) {
            using T = std::remove_pointer_t<decltype(t)>;
            const dim3 grid((unsigned)c.cross_GX, (unsigned)c.cross_J), block(64 * kGramWaves);
            if (h->has_w)   // G = X'WX (CDWeightedLSLoss: cd_differentiable_function.jl:177-182)
                hipLaunchKernelGGL((k_cross<T, true, 2, 2, cross_occ<T>()>), grid, block, 0, h->stream, (const T*)h->X, h->ld, h->nvec, h->p, c.d_cols, nbc,
                                   (const T*)h->w, c.d_cross_part);
            else
                hipLaunchKernelGGL((k_cross<T, false, 2, 2, cross_occ<T>()>), grid, block, 0, h->stream, (const T*)h->X, h->ld, h->nvec, h->p, c.d_cols, nbc,
                                   (const T*)nullptr, c.d_cross_part);
            return CDH_OK;
        }));
        hipLaunchKernelGGL(k_cross_reduce, dim3(kCrossRec / 256, (unsigned)launches), dim3(256), 0, h->stream, c.d_cross_part,
                           c.cross_J, c.d_cross);
        HIPCHK(h, hipGetLastError());
        CHK(allreduce(h, c.d_cross, (size_t)launches * kCrossRec));
        c.n_batches += 1; c.n_columns += nbc;
        // Where the device store has room, the batch's columns go into it straight from the cross-product records (round 4):
        // no copy of the records to the host, no host-side unpacking, no 32 pageable uploads -- ~1.5 ms per batch at p = 5000 --
        // and the host does not wait for k_cross at all.  The host's copies are fetched when host code asks (gc_host_column).
        bool on_device = false;
        if (c.cov && c.dev_slots == (int64_t)c.G.size()) {
            CHK(gc_dev_reserve(h, (int64_t)c.G.size() + nbc));
            on_device = c.cov && c.d_G && (int64_t)c.G.size() + nbc <= c.dev_slots_cap;
        }
        if (on_device) {
            hipLaunchKernelGGL(k_cross_unpack, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, h->stream, c.d_cross, h->p, (int)c.dev_slots, nbc, c.d_G);
            HIPCHK(h, hipGetLastError());
            for (int b = 0; b < nbc; ++b) {
                c.G.emplace_back();
                c.slot[(size_t)todo[b0 + (size_t)b]] = (int32_t)(c.G.size() - 1);
            }
            c.dev_slots += nbc;
        } else {
            HIPCHK(h, hipMemcpyAsync(c.h_cross.data(), c.d_cross, sizeof(double) * (size_t)launches * kCrossRec,
                                     hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            for (int b = 0; b < nbc; ++b) {
                c.G.emplace_back((size_t)h->p, 0.0);
                std::vector<double>& col = c.G.back();
                for (int64_t k = 0; k < h->p; ++k) {
                    const int64_t L = k / kCrossA, i = k % kCrossA;
                    col[(size_t)k] = c.h_cross[(size_t)(L * kCrossRec + ((i >> 4) * kCrossTB + (b >> 4)) * 256 + (i & 15) * 16 + (b & 15))];
                }
                c.slot[(size_t)todo[b0 + (size_t)b]] = (int32_t)(c.G.size() - 1);
            }
        }
        CHK(gc_dev_upload(h));
    }
    return CDH_OK;
}

// g <- g - sum_j dbeta_j G_j over the pending moves (all of which must have their columns)
// On the device where the column mirrors live (k_cov_gupdate, 64 moves per launch: the arithmetic of the
// gradient stays on the GPU, the host only compares it with thresholds); on the host only when there are no
// mirrors (the covariance form switched off, or no room for them).
bool gc_fold_device(cdh_handle h) {
    GradCache& c = h->gc;
    if (!c.cov || !c.d_G || !c.h_g_pin || c.dev_slots != (int64_t)c.G.size()) return false;
    const int64_t M = (int64_t)c.moved.size();
    if (M > h->cap) return false;
    if (gc_need_dev_g(h) != CDH_OK) { (void)hipGetLastError(); return false; }
    for (int64_t i = 0; i < M; ++i) { h->h_idx[i] = c.moved[(size_t)i]; h->h_hs[i] = c.dbeta[(size_t)c.moved[(size_t)i]]; }
    bool ok = hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)M, hipMemcpyHostToDevice, h->stream) == hipSuccess &&
              hipMemcpyAsync(h->d_hs, h->h_hs, sizeof(double) * (size_t)M, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    for (int64_t pos0 = 0; ok && pos0 < M; pos0 += 64)
        hipLaunchKernelGGL(k_cov_gupdate, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, h->stream, c.d_g, c.d_G, c.d_slot,
                           h->p, h->d_idx, h->d_hs, (int)pos0, (int)std::min<int64_t>(64, M - pos0));
    ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess;   // the staging arrays are reused
    if (!ok) {                        // a HIP failure mid-fold: the device copy is void
        (void)hipGetLastError();
        if (c.g_host_ok) c.g_dev_ok = false;      // c.g (host) is still the unfolded gradient: the host loop folds it
        else gc_invalidate(h, false);             // no copy left: the next full pass takes g afresh from X
        return false;
    }
    c.g_host_ok = false;              // g stays where the next pass wants it
    return true;
}

void gc_fold(cdh_handle h) {
    GradCache& c = h->gc;
    if (c.moved.empty()) return;
    // (the host copy is needed only for the host-side fallback; when the device folds, it is merely marked stale)
    const bool on_device = gc_fold_device(h);
    if (!on_device) {
        if (!c.valid) return;                     // invalidated by a failed device fold (moved is already empty)
        (void)gc_need_host_g(h);
        c.g_dev_ok = false;
    }
    for (int64_t j : c.moved) {
        const double d = c.dbeta[(size_t)j];
        if (!on_device && d != 0.0) {
            (void)gc_host_column(h, c.slot[(size_t)j]);
            const std::vector<double>& col = c.G[(size_t)c.slot[(size_t)j]];
            for (int64_t k = 0; k < h->p; ++k) c.g[(size_t)k] -= d * col[(size_t)k];
        }
        c.dbeta[(size_t)j] = 0.0; c.in_moved[(size_t)j] = 0;
    }
    c.moved.clear();
}

int32_t run_chunk(cdh_handle h, const int64_t* idx0, int m, double* maxH);

// The inactive coordinates of `enter` are about to move and have no Gram column: fetch their columns now, the batch
// filled with the inactive coordinates nearest their certificate (on a lambda path: the next entrants).  Needs the
// host copy of g.  `cert(k)`: the bound |g_k| is compared with; `ratio(k)`: |g_k| relative to its threshold.
template <typename Cert, typename Ratio>
int32_t gc_fetch_entering(cdh_handle h, std::vector<int64_t>& enter, Cert&& cert, Ratio&& ratio) {
    GradCache& c = h->gc;
    const size_t room = (kCrossB - enter.size() % kCrossB) % kCrossB;
    std::vector<std::pair<double, int64_t>> near;
    for (int64_t k = 0; k < h->p && room > 0; ++k)
        if (c.slot[(size_t)k] < 0 && h->x.get(k) == 0.0 && c.a[(size_t)k] > 0.0 && std::fabs(c.g[(size_t)k]) <= cert(k))
            near.emplace_back(ratio(k), k);
    const size_t take = std::min(room, near.size());
    std::partial_sort(near.begin(), near.begin() + (std::ptrdiff_t)take, near.end(), std::greater<std::pair<double, int64_t>>());
    for (size_t i = 0; i < take; ++i) enter.push_back(near[i].second);
    return gc_fetch(h, enter);
}

// ---- a whole full pass on the device ----------------------------------------------------------------------------
// Round 2 walked the pass on the host in windows of <= 2048 positions: g went down and came back per window, and the
// certificates of the skipped positions were re-checked on the host with p flops per mover -- 2862 copies and ~40 ms
// of host time in cfg3's 140 ms.  Here g stays in d_g: k_cov_scan classifies all m positions and compacts the
// unsettled ones (one small copy back: their number and positions, to check that each has its Gram column); the
// covariance-form blocks then run over that list, their g update re-checking on the way every skipped certificate
// the block answers for (k_cov_gupdate_chk); one more copy brings back the visits' results and the verdict.  Two
// synchronisations per pass, nothing p-sized on the bus.  A failed re-check (never seen in cfg3; forced in
// test_covariance_chunks_roll_back...) restores g and beta from the scan's snapshot and leaves the pass to the
// windowed walk below, which knows how to stop at the offending position.
enum { kDevDone = 0, kDevPlain = 1, kDevWalk = 2 };
constexpr int kGcForcedRounds = 4;     // a device pass whose re-check failed is run again this often with the failing coordinates visited
template <typename Cert, typename Ratio>
int32_t gc_pass_device(cdh_handle h, const int64_t* idx0, int64_t m, double* maxH, double cert_abs, Cert&& cert, Ratio&& ratio,
                       int* outcome) {
    GradCache& c = h->gc;
    *outcome = kDevWalk;
    if (!c.cov || !c.d_G || !c.d_scan || !c.moved.empty() || c.dev_slots != (int64_t)c.G.size() || m > h->cap) return CDH_OK;
    if (h->loss == CDH_SQRT && !c.q_valid) return CDH_OK;
    note_duplicates(h, idx0, (int)m);
    if (h->chunk_dup) return CDH_OK;            // a caller-made list that repeats a coordinate: positions are not unique
    CHK(gc_need_dev_g(h));
    const bool host_ok_before = c.g_host_ok;
    if ((int64_t)c.pass_idx_host.size() != m || std::memcmp(c.pass_idx_host.data(), idx0, sizeof(int64_t) * (size_t)m) != 0) {
        std::memcpy(h->h_idx, idx0, sizeof(int64_t) * (size_t)m);
        HIPCHK(h, hipMemcpyAsync(c.d_pass_idx, h->h_idx, sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice, h->stream));
        c.pass_idx_host.assign(idx0, idx0 + m);
    }
    // (the marks of a pass that was run again are wiped however the pass ends)
    struct ForcedGuard {
        cdh_handle h;
        ~ForcedGuard() {
            GradCache& c = h->gc;
            if (c.forced_dirty) { (void)hipMemsetAsync(c.d_forced, 0, (size_t)h->p, h->stream); c.forced_dirty = false; }
        }
    } forced_guard{h};
    int cnt = 0;
  for (int round = 0;; ++round) {
    for (int attempt = 0;; ++attempt) {
        h->ctrl.maxH = 0.0;
        h->ctrl.domain_error = 0;
        h->ctrl.q_carry = c.q;
        h->ctrl.cert_abs = cert_abs;
        CHK(upload_ctrl(h));
        hipLaunchKernelGGL(k_cov_scan, dim3(1), dim3(1024), 0, h->stream, c.d_g, c.d_a, h->beta, h->omega, h->d_ctrl, c.d_pass_idx, (int)m,
                           h->p, c.d_pos_of, c.d_setflag, c.d_upos, h->d_idx, c.d_g_snap, c.d_beta_snap, c.d_scan, c.d_forced);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(c.h_scanbuf, c.d_scanbuf, sizeof(int32_t) * (size_t)(m + 4), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        cnt = c.h_scan->count;
        if (c.h_scan->nzero > 0) return CDH_OK;     // a settled coordinate with g == 0 exactly: its bookkeeping differs; the walk knows
        std::vector<int64_t> enter;
        for (int j = 0; j < cnt; ++j) {
            const int64_t k = idx0[c.h_upos[j]];
            if (c.slot[(size_t)k] < 0) enter.push_back(k);
        }
        if (enter.empty()) break;
        if (attempt > 0) return CDH_OK;             // cannot happen (the fetch below gave every one of them a column)
        if ((int)enter.size() > kGcBusy) {          // busy: back off (1, 2, 4 ... 16 plain passes) so that dense problems pay next to nothing
            c.cooldown = c.backoff; c.backoff = std::min(16, 2 * c.backoff);
            gc_invalidate(h, false);
            *outcome = kDevPlain;
            return CDH_OK;
        }
        CHK(gc_need_host_g(h));
        CHK(gc_fetch_entering(h, enter, cert, ratio));
        if (c.mode == 0) { *outcome = kDevPlain; return CDH_OK; }
    }
    c.backoff = 1;
    if (cnt > 0) {
        h->chunk_dup = false;
        CHK(launch_cov_blocks(h, cnt, true, (int)m));
        HIPCHK(h, hipGetLastError());
        CHK(cov_fetch_results(h, cnt, true));
        // (tests: CDH_GC_INJECT_ROLLBACK=N declares every N-th device pass failed after the fact, so that the undo and the
        // windowed walk that takes over are exercised on every problem of the suite, not only where a certificate breaks)
        const bool injected = c.inject_rollback > 0 && (++c.inject_count % c.inject_rollback) == 0;
        if ((int64_t)c.h_scan->bad_pos < m) c.forced_dirty = true;     // (marks were set: wiped when this function returns)
        if ((int64_t)c.h_scan->bad_pos < m || injected) {       // a skipped certificate did not survive the pass's own moves: undo
            hipLaunchKernelGGL(k_cov_restore, dim3((unsigned)((h->p + 255) / 256)), dim3(256), 0, h->stream, c.d_g, h->beta, c.d_g_snap,
                               c.d_beta_snap, h->p);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipStreamSynchronize(h->stream));
            c.g_host_ok = host_ok_before;           // d_g is the gradient of the pass's start again
            // coordinates that crossed their threshold through the pass's own moves (the first full pass after lambda has changed,
            // on large supports: benchmark/cd_bench.jl's shape has them in most full passes beyond ~200 non-zeros): the same pass
            // again with those on the visit list -- a device pass, not the windowed walk
            if (!injected && round < kGcForcedRounds) { c.forced_dirty = true; c.n_forced_rounds += 1; continue; }
            c.n_rollbacks += 1;
            return CDH_OK;                          // -> the windowed walk
        }
        c.g_host_ok = false;
    } else {
        *h->h_ctrl = h->ctrl;                       // nothing ran: maxH 0, no domain error, r'r as it was
    }
    break;
  }
    // bookkeeping in visit order: what the reference's SparseIterate would have seen
    int j = 0;
    for (int64_t q = 0; q < m; ++q) {
        const int64_t k = idx0[q];
        if (j < cnt && c.h_upos[j] == (int32_t)q) { cov_apply_visit(h, k, j); ++j; }
        else {                                       // settled, g_k != 0: LS stores x[k] += b/a first, cdprox! zeroes it
            if (h->loss != CDH_SQRT) h->x.set(k, 1.0);
            h->x.set(k, 0.0);
        }
    }
    c.n_exact += cnt; c.n_certified += m - cnt; c.n_dev_passes += 1;
    cov_accept_tail(h, cnt, maxH, false);
    *outcome = kDevDone;
    return CDH_OK;
}

// Everything a cache-served full pass does before it looks at its visit list: the engagement policy, the sizing, a current g
// (re-reference when there is none or the carried one is due; the Gram columns of the support and of what has moved,
// the last batch filled with the inactive coordinates nearest their threshold; the fold) and ||r|| for the sqrt-lasso.
// *go = false: the pass runs the plain way.  Shared by gc_full_pass and the device-resident solve (cov_solve.hpp).
struct GcThresholds {             // the certificates' bounds as the host evaluates them (k_cov_scan's, on the host)
    cdh_handle h; double lam, nt, rnorm, cert_abs;
    double thr_of(int64_t k) const {
        const double w = h->has_omega ? h->h_omega[(size_t)k] : 1.0;
        return (h->loss == CDH_SQRT ? lam * w * rnorm : lam * nt * w) * (1.0 - 1e-9);
    }
    double cert(int64_t k) const { return thr_of(k) - cert_abs * std::sqrt(h->gc.a[(size_t)k]); }
    double ratio(int64_t k) const { return std::fabs(h->gc.g[(size_t)k]) / thr_of(k); }
};
int32_t gc_prepare_full(cdh_handle h, bool* go, double* cert_abs_out, bool fold = true /* false: the caller carries the pending moves itself (cov_solve) */) {
    GradCache& c = h->gc;
    *go = false;
    c.full_seen += 1;
    // mode 1 buys the Gram columns of the support (1.5 passes over X per 32 of them) only after the handle has
    // paid that much in plain full passes on the same data: at most twice the cost of having known in advance
    const int64_t rent = std::max<int64_t>(kGcEngage, 1 + (3 * h->x.nnz()) / 64);
    if (!gc_applicable(h) || (c.mode == 1 && !c.valid && c.full_seen <= rent)) return CDH_OK;
    // the support has outgrown what the cache pays for (gc_support_outgrown, cdhip.hip: long columns only)
    if (gc_support_outgrown(h)) {
        if (c.valid) gc_invalidate(h, false);
        return CDH_OK;
    }
    // g has been carried through this many covariance-form updates without looking at X: it is taken afresh
    // below (rounding only ever accumulates in g; one dots-only pass resets it), or dropped if this pass turns
    // out to run the plain way -- the active passes that follow must not keep carrying it either
    const bool refresh_due = c.valid && c.cov_since_ref > c.refresh_after;
    if (h->x.nnz() > gc_max_support(h)) { if (refresh_due) gc_invalidate(h, false); return CDH_OK; }
    if (c.cooldown > 0) { c.cooldown -= 1; if (c.valid) gc_invalidate(h, false); return CDH_OK; }
    CHK(gc_size(h));
    if (c.mode == 0) return CDH_OK;
    double cert_abs = 0.0;            // fp32 storage: what the certificates allow for the residual's rounding
    CHK(gc_cert_abs(h, &cert_abs));
    *cert_abs_out = cert_abs;
    // 1. a current g: fold the pending moves, fetching the columns that are missing; too many missing (or no
    //    reference yet, or the carried one is due): one dots-only pass over X gives a fresh g instead
    //    (re-referencing keeps what is known about beta: g will describe the same residual, only freshly summed)
    std::vector<int64_t> want;
    if (c.valid && !refresh_due) {
        for (int64_t j : c.moved) if (c.slot[(size_t)j] < 0) want.push_back(j);
        if ((int)want.size() > kGcMaxFetch) { gc_invalidate(h, false); want.clear(); }
    }
    if (!c.valid || refresh_due) CHK(gc_rereference(h));
    for (int64_t s_ = 0; s_ < h->x.nnz(); ++s_)       // the support moves in every pass: its columns first
        if (c.slot[(size_t)h->x.coord(s_)] < 0) want.push_back(h->x.coord(s_));
    if (!want.empty()) {
        // fill the last batch of 32 with the inactive coordinates nearest their threshold: the likeliest to
        // enter the support next (on a lambda path: at one of the next lambdas)
        const size_t room = (kCrossB - want.size() % kCrossB) % kCrossB;
        if (room > 0 && c.moved.empty()) {
            CHK(gc_need_host_g(h));
            GcThresholds T{h, h->ctrl.lambda0, (double)h->n_total, 0.0, cert_abs};
            if (h->loss == CDH_SQRT) { CHK(gc_ensure_q(h)); T.rnorm = std::sqrt(c.q); }
            std::vector<std::pair<double, int64_t>> near;
            for (int64_t k = 0; k < h->p; ++k)
                if (c.slot[(size_t)k] < 0 && h->x.get(k) == 0.0 && c.a[(size_t)k] > 0.0)
                    near.emplace_back(std::fabs(c.g[(size_t)k]) / T.thr_of(k), k);
            const size_t take = std::min(room, near.size());
            std::partial_sort(near.begin(), near.begin() + (std::ptrdiff_t)take, near.end(), std::greater<std::pair<double, int64_t>>());
            for (size_t i = 0; i < take; ++i) want.push_back(near[i].second);
        }
        CHK(gc_fetch(h, want));
        if (c.mode == 0) return CDH_OK;
    }
    for (int64_t j : c.moved) if (c.slot[(size_t)j] < 0) return fail(h, CDH_BAD_ARG, "gradient cache: a moved coordinate has no Gram column");
    if (fold) gc_fold(h);
    if (!c.valid) return CDH_OK;      // (a device fold that failed hard leaves no gradient: the pass runs the plain way)
    gc_q_guard(h);
    if (h->loss == CDH_SQRT) CHK(gc_ensure_q(h));
    *go = true;
    return CDH_OK;
}

// A full pass over idx0[0..m) from the cache.  *handled = false: the caller runs the pass the plain way.
int32_t gc_full_pass(cdh_handle h, const int64_t* idx0, int64_t m, double* maxH, bool* handled) {
    GradCache& c = h->gc;
    *handled = false;
    double cert_abs = 0.0;
    {
        bool go = false;
        if (c.prep_state) { go = c.prep_state == 1; cert_abs = c.prep_cert_abs; c.prep_state = 0; }   // cov_solve has prepared this pass
        else CHK(gc_prepare_full(h, &go, &cert_abs));
        if (!go) return CDH_OK;
    }
    const double lam = h->ctrl.lambda0, nt = (double)h->n_total;
    const std::vector<double>& om = h->h_omega;
    double rnorm = 0.0;
    auto refresh_rnorm = [&]() -> int32_t {
        if (h->loss != CDH_SQRT) return CDH_OK;
        CHK(gc_ensure_q(h));
        rnorm = std::sqrt(c.q);
        return CDH_OK;
    };
    auto thr_of = [&](int64_t k) {
        const double w = h->has_omega ? om[(size_t)k] : 1.0;
        return (h->loss == CDH_SQRT ? lam * w * rnorm : lam * nt * w) * (1.0 - 1e-9);
    };
    // the bound a certificate compares |g_k| with (k_cov_scan's, on the host)
    auto cert = [&](int64_t k) { return thr_of(k) - cert_abs * std::sqrt(c.a[(size_t)k]); };
    auto ratio = [&](int64_t k) { return std::fabs(c.g[(size_t)k]) / thr_of(k); };
    // settled = the exact visit would leave beta_k at zero and r untouched (zero columns take the exact path)
    auto settled = [&](int64_t k) {
        return h->x.get(k) == 0.0 && c.a[(size_t)k] > 0.0 && std::fabs(c.g[(size_t)k]) <= cert(k);
    };
    CHK(refresh_rnorm());
    // 2. the whole pass on the device where it can be (g stays there); else -- and after a failed re-check -- the
    //    windowed walk below, with g on the host
    {
        int outcome = kDevWalk;
        CHK(gc_pass_device(h, idx0, m, maxH, cert_abs, cert, ratio, &outcome));
        if (outcome == kDevDone) { *handled = true; c.n_passes += 1; return CDH_OK; }
        if (outcome == kDevPlain) return CDH_OK;
        if (!c.valid) return CDH_OK;
    }
    CHK(gc_need_host_g(h));
    // 2'. inactive coordinates about to move: many of them means the pass is mostly real visits anyway
    std::vector<int64_t> enter;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t k = idx0[i];
        if (h->x.get(k) == 0.0 && !settled(k) && c.slot[(size_t)k] < 0) enter.push_back(k);
    }
    std::sort(enter.begin(), enter.end());
    enter.erase(std::unique(enter.begin(), enter.end()), enter.end());
    if ((int)enter.size() > kGcBusy) {   // busy: back off (1, 2, 4 ... 16 plain passes) so that dense problems pay next to nothing
        c.cooldown = c.backoff; c.backoff = std::min(16, 2 * c.backoff);
        gc_invalidate(h, false);
        return CDH_OK;
    }
    c.backoff = 1;
    if (!enter.empty()) {
        // their columns now, with the nearest other candidates filling the batch
        CHK(gc_fetch_entering(h, enter, cert, ratio));
        if (c.mode == 0) return CDH_OK;
    }
    *handled = true;
    c.n_passes += 1;
    // 3. the walk: runs of settled visits are skipped (with the bookkeeping the reference's SparseIterate
    //    would have seen); everything else is visited by the ordinary kernels, a few visits per chunk
    const int B = (h->mode == CDH_SWEEP_BLOCK) ? h->blockB : 1;
    const int64_t maxlen = std::min<int64_t>(h->cap, std::max<int64_t>(4 * B, 64));
    int64_t pos = 0;
    while (pos < m) {
        CHK(gc_need_host_g(h));       // (a covariance-form chunk or a device fold below leaves g on the device)
        const int64_t k = idx0[pos];
        if (settled(k)) {
            if (h->loss != CDH_SQRT && c.g[(size_t)k] != 0.0) h->x.set(k, 1.0);   // x[k] += b/a stores a slot ...
            h->x.set(k, 0.0);                                                          // ... cdprox! zeroes it
            c.n_certified += 1;
            ++pos;
            continue;
        }
        // Covariance form (least squares, every unsettled coordinate ahead has its Gram column): the unsettled
        // positions of a window go down as ONE chunk and the settled positions between them are skipped; since
        // a settled coordinate's certificate was read BEFORE the chunk's moves, it is re-checked afterwards
        // against the gradient as it stood when its turn came (the movers' columns are at hand).  A certificate
        // that no longer holds -- rare -- rolls the chunk back and reruns it up to that position.
        if (c.cov && c.moved.empty() && c.d_G && c.dev_slots == (int64_t)c.G.size() && (h->loss != CDH_SQRT || c.q_valid)) {
            std::vector<int64_t> upos;
            int64_t wend = pos;
            bool have_all = true;
            for (; wend < m && (int64_t)upos.size() < maxlen && wend - pos < kGcCovWindow; ++wend)
                if (!settled(idx0[wend])) {
                    if (c.slot[(size_t)idx0[wend]] < 0) { have_all = false; break; }
                    upos.push_back(wend);
                }
            if (have_all || !upos.empty()) {
                if (!have_all) wend = upos.back() + 1;        // stop the window before the column-less coordinate
                for (;;) {
                    std::vector<int64_t> vis(upos.size());
                    for (size_t i = 0; i < upos.size(); ++i) vis[i] = idx0[upos[i]];
                    const int mv = (int)vis.size();
                    CHK(cov_run(h, vis.data(), mv, true));
                    // re-check the skipped positions in visit order, carrying g through the chunk's moves
                    std::vector<double> gv(c.g);
                    double q_run = c.q;                  // sqrt-lasso: ||r|| moves with every visit, and the thresholds with it
                    const double rnorm_before = rnorm;
                    int64_t bad = -1;
                    size_t iu = 0;
                    auto settled_at = [&](int64_t kq) {
                        return h->x.get(kq) == 0.0 && c.a[(size_t)kq] > 0.0 && std::fabs(gv[(size_t)kq]) <= cert(kq);
                    };
                    for (int64_t q = pos; q < wend; ++q) {
                        if (iu < upos.size() && upos[iu] == q) {
                            const double hv = h->h_hs[iu];
                            if (hv != 0.0) {
                                const int64_t kv = vis[iu];
                                if (h->loss == CDH_SQRT) {
                                    q_run = q_run - 2.0 * hv * gv[(size_t)kv] + hv * hv * c.a[(size_t)kv];
                                    if (q_run < 0.0) q_run = 0.0;
                                    rnorm = std::sqrt(q_run);
                                }
                                CHK(gc_host_column(h, c.slot[(size_t)kv]));
                                const std::vector<double>& col = c.G[(size_t)c.slot[(size_t)kv]];
                                for (int64_t kk = 0; kk < h->p; ++kk) gv[(size_t)kk] -= hv * col[(size_t)kk];
                            }
                            ++iu;
                        } else if (!settled_at(idx0[q])) { bad = q; break; }
                    }
                    rnorm = rnorm_before;
                    if (bad < 0) {
                        iu = 0;
                        for (int64_t q = pos; q < wend; ++q) {      // bookkeeping in visit order
                            const int64_t kq = idx0[q];
                            if (iu < upos.size() && upos[iu] == q) { cov_apply_visit(h, kq, (int)iu); ++iu; c.n_exact += 1; }
                            else {   // settled: what the reference's SparseIterate would have seen (LS stores x[k] += b/a first)
                                if (h->loss != CDH_SQRT && c.g[(size_t)kq] != 0.0) h->x.set(kq, 1.0);
                                h->x.set(kq, 0.0);
                                c.n_certified += 1;
                            }
                        }
                        cov_accept_tail(h, mv, maxH, true);
                        if (h->loss == CDH_SQRT) rnorm = std::sqrt(c.q);
                        break;
                    }
                    CHK(cov_reject(h, vis.data(), mv));
                    c.n_rollbacks += 1;
                    wend = bad;                                      // position `bad` starts the next window
                    while (!upos.empty() && upos.back() >= bad) upos.pop_back();
                }
                pos = wend;
                if (!c.valid) { *handled = true; break; }
                continue;
            }
        }
        // streamed visits [pos, end): up to the last unsettled position that is no more than 4 past the previous one
        int64_t end = pos + 1;
        for (int64_t q = pos + 1; q < m && q - end < 4 && end - pos < maxlen; ++q)
            if (!settled(idx0[q])) end = q + 1;
        if (cov_ok(h, idx0 + pos, end - pos)) CHK(cov_chunk(h, idx0 + pos, (int)(end - pos), maxH));
        else CHK(run_chunk(h, idx0 + pos, (int)(end - pos), maxH));
        c.n_exact += end - pos;
        pos = end;
        if (!c.valid) { *handled = true; break; }     // a NaN step: finish below the plain way
        if (!c.moved.empty()) {
            std::vector<int64_t> miss;
            for (int64_t j : c.moved) if (c.slot[(size_t)j] < 0) miss.push_back(j);
            if (!miss.empty()) { CHK(gc_fetch(h, miss)); if (c.mode == 0) break; }
            gc_fold(h);
            CHK(refresh_rnorm());
        }
    }
    if (pos < m) {   // the cache went away mid-pass: the rest of the list the plain way, chunk by chunk
        for (int64_t off = pos; off < m; off += h->cap) {
            const int mm = (int)std::min<int64_t>(h->cap, m - off);
            CHK(run_chunk(h, idx0 + off, mm, maxH));
        }
    }
    return CDH_OK;
}

