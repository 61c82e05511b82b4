// cdhip.hip -- the C-ABI library (include/cdhip.h): handle, device memory, the pass
// / solve state machine of src/coordinate_descent.jl restated on the host, launches
// of the kernels in kernels.hpp, RCCL row-shard all-reduce.  gfx950 only.
#include "../../include/cdhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "gram_kernels.hpp"
#include "p2p_exchange.hpp"
#include "sparse_iterate.hpp"

using namespace cdk;

namespace {

std::string g_create_error;

constexpr int kMaxStepGrid = 2048;   // blocks of k_step (8 per CU on 256 CUs)
constexpr int kBlockGridPerCU = 4;   // upper bound of k_blockstep blocks per CU (partials sizing)
constexpr int kColChunks = 64;       // row chunks per column in k_col_dots
constexpr int kMaxBlockB = 8;
constexpr int kShortRounds = 16;     // below this many rounds of 64-vector chunks per launch: short chunks

// ---- RCCL through dlopen: only multi-process runs need it ------------------------
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    void* CommInitRank = nullptr;  // (ncclComm_t*, int nranks, ncclUniqueId by value, int rank)
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
struct UniqueId { char bytes[128]; };
typedef int (*comm_init_rank_fn)(void**, int, UniqueId, int);
Rccl g_rccl;

bool load_rccl(std::string& err) {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
        g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) { err = std::string("dlopen librccl failed: ") + dlerror(); return false; }
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.CommAbort = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommAbort");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(
        g_rccl.lib, "ncclAllReduce");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(g_rccl.lib, "ncclCommCount");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) {
        err = "librccl is missing ncclGetUniqueId/ncclCommInitRank/ncclAllReduce";
        return false;
    }
    return true;
}
constexpr int kNcclDouble = 8;  // ncclFloat64
constexpr int kNcclSum = 0;

}  // namespace

// Gradient cache of the screened full passes (no reference counterpart; exact).  A visit of a coordinate
// with beta_k == 0 changes nothing unless |X_k'r| exceeds its threshold, and X_k'r is known WITHOUT reading
// X if the Gram columns G_j = X'X_j of every coordinate that moved since a reference point are at hand:
//   X'r = g_ref - sum_j dbeta_j G_j .
// Only the support ever moves, so a handle that keeps solving on the same X (a lambda path, the sigma loop
// of scaledLasso!, the 51 continuation solves of a cold start) pays one pass over X for g_ref, 1.25 passes
// per 32 Gram columns as coordinates enter the support, and after that a full pass costs its exact visits
// only.  Same iterates as visiting one by one: a coordinate is skipped only when the exact path would have
// left it at zero (1e-9 relative margin, as for the dots-only screens), everything else is visited by the
// same kernels.  fp64 storage, no observation weights.
#include "cov_solve_types.hpp"   // CovSolveCtl, CovSolveBufs (the kernel and its host side: cov_solve.hpp, below)
struct GradCache {
    int mode = 1;                   // 0 off, 1 rent-or-buy, 2 from the first full pass (both only where the host-side
                                    // fold is cheaper than reading X), 3 from the first full pass, unconditionally
    bool valid = false;             // g (with the pending dbeta) describes X'r of the device's current r
    bool beta_ok = false;           // r == y - X beta_ref up to rounding
    int64_t full_seen = 0;          // screened full passes since the data last changed
    int cooldown = 0, backoff = 1;  // after a busy pass: this many full passes run the plain way
    std::vector<double> g, a, dbeta, beta_ref;
    std::vector<int64_t> moved;     // coordinates with dbeta != 0, each once
    std::vector<uint8_t> in_moved;
    std::vector<int32_t> slot;      // coordinate -> Gram column, -1 = not cached
    std::vector<std::vector<double>> G;
    double* d_cross = nullptr;      // device: ceil(p / 64) records of 64 x 32 cross products
    double* d_cross_part = nullptr; // device: the same per row-slab block (cross_J of them per column group)
    int cross_J = 1, cross_GX = 1;
    int64_t* d_cols = nullptr;      // device: the B columns of a batch
    std::vector<double> h_cross;
    // covariance-form visits: device mirrors of g, the Gram columns (slot-major, p doubles each) and the slot map
    bool cov = true;                // env CDH_GC_COV
    double *d_g = nullptr, *d_G = nullptr, *h_g_pin = nullptr;
    int32_t* d_slot = nullptr;
    // whole full passes on the device (gc_pass_device): g lives in d_g between passes and comes back only when host
    // code asks for it.  g_host_ok / g_dev_ok say which copies are current (at least one always is while `valid`).
    bool g_host_ok = true, g_dev_ok = false;
    double *d_a = nullptr, *d_g_snap = nullptr, *d_beta_snap = nullptr, *d_qs = nullptr;
    int64_t* d_pass_idx = nullptr;        // the pass's visit list (0-based), as uploaded last
    std::vector<int64_t> pass_idx_host;   // ... and what it holds
    int32_t *d_pos_of = nullptr, *d_upos = nullptr;
    uint8_t *d_setflag = nullptr, *d_forced = nullptr;   // (d_forced: the second half of d_setflag's allocation)
    bool forced_dirty = false;
    int64_t n_forced_rounds = 0, n_cs_forced_rounds = 0, n_cs_crew_passes = 0, n_cs_crew_jobs = 0;
    // the scan's counters sit at the head of the buffer of unsettled positions (one copy brings both back); the results of
    // a pass come back through k_cov_pack's block
    int32_t *d_scanbuf = nullptr, *h_scanbuf = nullptr;   // [CovScanOut: 4 int32][positions: cap]; h_: pinned
    cdk::CovScanOut* d_scan = nullptr;    // = d_scanbuf
    cdk::CovScanOut* h_scan = nullptr;    // = h_scanbuf
    int32_t* h_upos = nullptr;            // = h_scanbuf + 4
    double *d_pack = nullptr, *h_pack = nullptr;          // kPackHead + 2.5 cap doubles; h_: pinned
    bool a_dev_ok = false;                // d_a mirrors c.a
    double yy = 0.0;                      // y'y over all shards (fp32 certificate margin), valid while yy_ok
    bool yy_ok = false;
    int64_t n_dev_passes = 0;
    int inject_rollback = 0, inject_count = 0;   // env CDH_GC_INJECT_ROLLBACK (tests)
    int64_t dev_slots_cap = 0, dev_slots = 0;   // columns the device store can hold / holds
    std::vector<double*> d_G_retired;           // stores outgrown on the way (freed with the handle)
    int64_t cov_since_ref = 0;      // covariance-form visits since g was last taken from X itself
    int64_t refresh_after = 0;      // ... after which it is (kGcCovRefresh; env CDH_GC_REFRESH for tests)
    std::vector<double> g_new;      // g as a covariance-form chunk left it, until the chunk is accepted
    double q = 0.0;                 // r'r of the (virtual) residual g describes: sqrt-lasso thresholds and updates
    double q_exact = 0.0;           // ... as last summed from r itself (the carried value is refreshed once it has fallen far below it)
    bool q_valid = false;
    int64_t n_rollbacks = 0;
    double drift_last = 0.0, drift_max = 0.0;   // max_k |g_carried - X'r| / thr_k at the re-references so far
    int64_t n_drift = 0;
    int64_t n_validate = 0, n_batches = 0, n_columns = 0, n_certified = 0, n_exact = 0, n_passes = 0, n_cov = 0,
            n_reconcile = 0;
    // the device-resident pass loop (cov_solve.hpp): its scratch, the pinned block it reads from and writes into, the bound's M_k
    bool cs_enabled = true;          // env CDH_COV_SOLVE (default 1)
    bool cs_big = false;             // a visit list has outgrown the loop's LDS block on this handle: launches use the instantiation with the table and the helpers
    int cs_ucap_limit = 0;           // env CDH_CS_UCAP (tests): visit lists longer than this leave the LDS block
    int cs_helpers = 31;             // helper workgroups a launch that expects large visit lists brings (env CDH_CS_CREW; 0: none, table mode only)
    bool cs_shuffle_ok = true, cs_stalled = false;
    size_t cs_lds_budget = 0;
    char *cs_dev = nullptr, *cs_pin = nullptr, *cs_pin_dev = nullptr;
    CovSolveBufs cs_bufs{};
    CovSolveCtl* cs_ctl = nullptr;
    int32_t *cs_in_sup = nullptr, *cs_out_sup_idx = nullptr, *cs_out_moved_idx = nullptr, *cs_out_list = nullptr;
    double *cs_out_sup_val = nullptr, *cs_out_moved_val = nullptr, *d_colmax = nullptr;
    int64_t colmax_slots = 0;        // columns of the device store already folded into d_colmax
    bool slot_dev_ok = false;        // d_slot mirrors `slot`
    std::vector<double> cs_old;      // scratch: the iterate's values before a launch, by coordinate (zero between launches)
    int prep_state = 0;              // gc_prepare_full has run for the pass about to be walked: 1 go, 2 no-go (0: not yet)
    double prep_cert_abs = 0.0;
    int64_t n_cs_launches = 0, n_cs_passes = 0, n_cs_folds = 0, n_cs_exact = 0, n_cs_table_passes = 0, n_cs_table_rows = 0;
    int32_t cs_ncid = 0, cs_tepoch = 0;      // the kernel's Gram table: coordinates it holds, the epoch of its carried gradients
    bool cs_table_reset = true;
    int64_t cs_ticks[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs_cycles = 0, cs_ticks_total = 0;
};

// The one-launch solve of problems that fit on chip (small_solve.hpp): the full Gram matrix of the resident X, the
// control block of the solve kernel, pinned staging for what comes back.
constexpr int kSmallMaxP = 1024;
constexpr int kSmallMaxLam = 64;                 // solves per launch (a cold start's numSteps + 1 = 51 by default)
// How much X the Gram form is worth building for is a cost comparison (small_worth_building, small_solve.hpp); X that fits
// kSmallAlwaysBytes takes it regardless (the reference's own test and benchmark shapes).
constexpr size_t kSmallAlwaysBytes = (size_t)16 << 20;

struct SmallCtl {
    double lambdas[kSmallMaxLam];
    int32_t nlam, randomize, loss, has_omega;
    int64_t maxIter;
    double optTol, n_total;
    uint64_t rng;                // splitmix64 state of the substitute RandomIterator: in / out
    int32_t nnz_in, g_from_c;   // g_from_c: `ga` holds (X'y, a) and g = X'y - G beta is formed in the kernel; else ga holds (X'r, a)
    // out
    int64_t passes, full_passes, visits;
    int32_t converged, domain_error;
    double maxH;
    int32_t nnz, precision_lost;
    int64_t steps;               // visit steps taken (each settles a run of positions and makes at most one move)
    uint64_t cycles, ticks;      // shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) the kernel ran for
};

struct SmallPath {
    bool enabled = true;             // env CDH_SMALL_PATH (default 1), cdh_set_small_path
    int64_t max_bytes = -1;          // env CDH_SMALL_MAX_BYTES: a fixed limit on n p sz instead of rent-or-buy (experiments)
    int64_t always_bytes = (int64_t)kSmallAlwaysBytes;   // env CDH_SMALL_ALWAYS_BYTES (tests: 0 makes every handle rent first)
    double rent_paid = 0.0;          // modelled seconds of streamed solves on the current X while G was not built (SmallRent)
    int64_t* d_iota = nullptr;       // 0 .. p-1: the column lists of the Gram build
    bool G_valid = false;
    double* d_G = nullptr;           // p x p
    char *d_io = nullptr, *h_io = nullptr;    // [SmallCtl][support][beta]: what crosses the bus per solve, one block each way
    char* hd_io = nullptr;           // h_io as the device addresses it
    bool zero_copy = true;           // env CDH_SMALL_ZEROCOPY (default 1): the kernel works on h_io itself, nothing is copied
    SmallCtl *d_ctl = nullptr, *h_ctl = nullptr;   // views into d_io / h_io (pinned)
    int32_t *d_sup = nullptr, *h_sup = nullptr;
    double *d_beta = nullptr, *h_beta = nullptr;
    int64_t n_solves = 0, n_gram = 0, n_precision = 0;
    bool c_valid = false;            // d_ca / h_c / yy hold X'y (X'Wy), diag(G) and y'y of the current y
    double* d_ca = nullptr;          // interleaved (c_k, a_k), then y'y at [2p]
    std::vector<double> h_c;         // host copy of c (lambda_max of a cold start needs no device work)
    double yy = 0.0;
    int ncache = 0;                  // Gram columns the solve kernel can keep in LDS
    unsigned lds_bytes = 0;
};

struct cdh_handle_s {
    int dtype = CDH_F64, loss = CDH_LS, device = 0;
    int64_t n = 0, n_total = 0, row0 = 0, p = 0, ld = 0, nvec = 0;
    size_t esz = 8;
    hipStream_t stream = nullptr;
    // device
    void *X = nullptr, *y = nullptr, *r = nullptr, *w = nullptr;
    double *beta = nullptr, *omega = nullptr;
    Ctrl* d_ctrl = nullptr;
    int64_t* d_idx = nullptr;
    double *d_hs = nullptr, *d_newval = nullptr;
    int32_t* d_touched = nullptr;
    double *d_partials = nullptr, *d_red = nullptr, *d_colout = nullptr;
    int64_t* d_sup_idx = nullptr;
    double* d_sup_val = nullptr;
    // pinned host staging
    int64_t* h_idx = nullptr;
    double *h_hs = nullptr, *h_newval = nullptr, *h_red = nullptr;
    int32_t* h_touched = nullptr;
    Ctrl* h_ctrl = nullptr;
    // state
    int64_t cap = 0;          // visits per chunk
    size_t partials_doubles = 0;
    Ctrl ctrl{};
    bool has_omega = false, has_w = false, y_set = false;
    std::vector<double> h_omega;  // host copy of the penalty weights (thresholds, objective)
    bool omega_dev_ok = false;    // the device's omega holds h_omega
    cdh::SupportList x;
    int mode = CDH_SWEEP_BLOCK, blockB = 32;   // the default (cdh_create: 64 on short fp64 columns); cdh_set_sweep_mode changes it
    // The default width is chosen from the rank-LOCAL row count, and near-equal row shards can fall on opposite sides of the
    // cut (n_total = 524287 over two ranks: 262144 and 262143 rows) -- ranks with different B would issue different numbers
    // and sizes of all-reduces per pass.  So a sharded handle that still has its default agrees on it through the exchange
    // before its first streamed chunk (agree_default_width): 64 only if every rank chose 64.
    bool width_default = true, width_agreed = false;
    bool use_graph = false;
    int screening = 1;            // 0 never, 1 the solves' full passes over sparse iterates, 2 cdh_pass too
    bool reuse_residual = false;  // warm starts skip initialize! when r is known to match beta
    bool r_consistent = false;    // r == y - X beta (up to rounding) for the handle's current iterate
    // The one-launch solve never touches r: afterwards the residual on the device is STALE and stands for
    // y - X * x_lazy, rebuilt (one k_init_resid launch) by sync_r when something asks for r -- a residual nobody reads
    // is never formed.  x_lazy is the iterate that residual belongs to: the solve's result, and still that after the
    // caller loads another iterate without initialize! (cdh_set_iterate leaves r alone, as the reference's x[k] = ... does).
    bool r_lazy = false;
    cdh::SupportList x_lazy;
    // What is known about the residual BUFFER: `r_pristine` -- it is bit for bit what initialize! makes of the iterate
    // x_pristine (k_init_resid is deterministic) and nothing has written it since: another initialize! of that very iterate
    // changes nothing and is skipped (lasso.jl:250-252 calls initialize! at every lambda; utils.jl / lasso.jl front-ends call
    // it right after _findLambdaMax).  `dots_valid` -- dots_stash holds (X_k'W r, X_k'W X_k) for all k of the residual the
    // handle stands for as of now (taken by _findLambdaMax / cdh_xt_r): the gradient cache's reference pass adopts them
    // instead of reading X again.  Everything that writes r, queues updates for it, or changes X, y, W clears both.
    bool r_pristine = false, dots_valid = false, dots_w = false;
    cdh::SupportList x_pristine;
    std::vector<double> dots_stash;
    int64_t n_rebuild_skipped = 0, n_dots_adopted = 0;
    int64_t r_roundings = 0;      // launches that have rewritten r (each rounds it to the storage type once) since it was last rebuilt
    bool chunk_dup = false;       // the current chunk's visit list repeats a coordinate
    std::vector<int32_t> stamp;   // duplicate detection scratch, size p
    struct GraphEntry { uint64_t key; hipGraphExec_t exec; unsigned exchanges; unsigned rccl; };
    std::vector<GraphEntry> graphs;  // captured chunk launch sequences
    bool graph_broken = false;       // a capture failed on this handle: launch node by node from now on
    bool domain_error = false;
    int step_grid = 1, block_grid = 1, gram_per_cu = 2, cus = 1, gram32_per_cu = 1;
    int lt_per_cu = 0;  // experiments: blocks per CU of the LDS-transposed variants (0 = by tile size)
    int64_t gram_units = 1;
    bool nt = true;  // non-temporal loads for the X column streams
    int ks = 0;      // chunk length of the LDS-transposed path: 0 by shard length, 1 short, 2 long (env CDH_KS)
    int lt = 2;      // coalesced operand loads transposed through LDS: 0 off, 1 on, 2 by size (default)
    // comm
    void* comm = nullptr;
    int rank = 0, nranks = 1;
    // optional direct exchange of the short records (p2p_exchange.hpp); off unless connected and enabled
    unsigned long long* p2p_inbox = nullptr;
    cdk::P2PPeers p2p_peers{};
    std::vector<void*> p2p_mapped;
    int* p2p_timeout = nullptr;  // pinned host flag written by a kernel whose bounded spin ran out
    unsigned p2p_epoch = 0, p2p_spin_limit = cdk::kP2PSpinLimit;
    int p2p_ranks = 0;
    bool p2p_on = false, p2p_dead = false;
    bool lost_exchange = false;       // a multi-rank shard whose host transport was taken away: allreduce() refuses
    unsigned* d_p2p_base = nullptr;   // epoch base of a replayed graph's exchanges (device memory)
    bool capturing = false;           // run_chunk is recording a graph: exchanges take base + offset epochs
    unsigned cap_exchanges = 0, cap_rccl = 0;   // exchanges recorded in the graph being captured
    // bring-your-own transport (cdh_set_host_exchange): staged through pinned host memory
    cdh_host_allreduce_fn host_fn = nullptr;
    void* host_user = nullptr;
    double* h_xchg = nullptr;
    size_t h_xchg_doubles = 0;
    int64_t n_rccl_calls = 0, n_p2p_calls = 0, n_host_calls = 0;
    // profile
    GradCache gc;
    SmallPath small;
    // beta changes the covariance-form visits have made that r has not seen yet: r_actual = r_virtual + X * pending
    std::vector<double> r_pending;
    std::vector<int64_t> r_pending_list;
    std::vector<uint8_t> r_in_pending;
    bool prof = false;
    double prof_ms = 0.0, prof_bytes = 0.0;
    int64_t prof_launches = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
};

namespace {

#define HIPCHK(h, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            char buf_[512];                                                                 \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                   \
            (h)->err = buf_;                                                                \
            return e_ == hipErrorOutOfMemory ? CDH_OOM : CDH_HIP_ERROR;                     \
        }                                                                                   \
    } while (0)

#define CHK(call)                                   \
    do {                                            \
        int32_t s_ = (call);                        \
        if (s_ != CDH_OK) return s_;                \
    } while (0)

int32_t fail(cdh_handle h, int32_t code, const char* msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

// every export checks what it dereferences: a NULL handle or out-pointer is CDH_BAD_ARG, never a crash
#define NEED_H(h)                                                              \
    do {                                                                       \
        if (!(h)) return fail(nullptr, CDH_BAD_ARG, "handle is NULL");         \
    } while (0)
#define NEED_P(h, ptr)                                                         \
    do {                                                                       \
        if (!(ptr)) return fail((h), CDH_BAD_ARG, #ptr " is NULL");            \
    } while (0)

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Grid size for a grid-stride kernel with `units` work items and at most `gmax` resident blocks.
inline int balanced_grid(int64_t units, int64_t gmax) {
    // Measured: keeping every CU slot occupied (gmax blocks) beats equalising per-block work --
    // the streaming rate is set per CU (~24 GB/s each), so an idle slot costs more than a block
    // that runs one iteration longer.
    return (int)std::max<int64_t>(1, std::min<int64_t>(gmax, units));
}

template <typename F> int32_t dispatch(cdh_handle h, F&& f) {
    return h->dtype == CDH_F64 ? f((double*)nullptr) : f((float*)nullptr);
}

int32_t upload_ctrl(cdh_handle h) {
    *h->h_ctrl = h->ctrl;
    HIPCHK(h, hipMemcpyAsync(h->d_ctrl, h->h_ctrl, sizeof(Ctrl), hipMemcpyHostToDevice, h->stream));
    return CDH_OK;
}

// p2p_dead counts: a shard that lost its exchange must never fall into the single-process (fused)
// finalize kernels on its local rows -- every path then reaches allreduce(), which refuses
inline bool sharded(const cdh_handle_s* h) {
    return h->comm != nullptr || h->p2p_on || h->p2p_dead || h->lost_exchange || h->host_fn != nullptr;
}

// r (the buffer, or the residual it stands for) is about to change, or X / y / W are: what was known about it is void
inline void touch_r(cdh_handle_s* h) { h->r_pristine = false; h->dots_valid = false; }

// A shard that has lost its exchange refuses to sweep at all -- also where a pass could be served from sums exchanged
// earlier (the gradient cache): the ranks of one problem must fail together, not one by one as they come to need an exchange.
int32_t exchange_alive(cdh_handle h) {
    if (h->p2p_dead) return fail(h, CDH_RCCL_ERROR, "the shard lost its exchange (p2p timed out earlier); rebuild the handle");
    if (h->lost_exchange) return fail(h, CDH_RCCL_ERROR, "the shard's host exchange was removed and nothing replaced it: its sums would cover local rows only");
    return CDH_OK;
}

int32_t p2p_check(cdh_handle h) {
    // after a timeout the ranks no longer agree on what has been exchanged: the handle refuses every
    // later exchange (falling back to RCCL here could pair mismatched all-reduces and hang)
    if (h->p2p_on && *(volatile int*)h->p2p_timeout) {
        h->p2p_on = false;
        h->p2p_dead = true;
        h->err = "p2p exchange timed out waiting for a peer (rank died, or ranks ran different sweeps)";
        return CDH_RCCL_ERROR;
    }
    return CDH_OK;
}

static_assert(GramRec<4>::N <= cdk::kP2PMaxCount, "the widest block record must fit one inbox slot");
// Epochs count direct exchanges; 0 means "never written".  Consecutive epochs must alternate the
// inbox slot (parity), also across the 32-bit wrap.
constexpr unsigned kEpochWrap = 0xfffffff0u;
// Chunks start below this: every rank passes a chunk boundary with the same epoch count whether it replays a
// graph or launches node by node, so wrapping THERE keeps ranks on different paths in step (a chunk issues far
// fewer than 2^28 exchanges, so the hard wrap above is never reached inside one)
constexpr unsigned kEpochSoftWrap = 0xf0000000u;
inline unsigned epoch_after(unsigned e) { return e >= kEpochWrap ? ((e & 1u) ? 2u : 3u) : e + 1u; }
// the arguments of the next direct exchange.  While a graph is being recorded the epoch is
// (device-resident base) + (position of the exchange in the graph), so one graph serves every replay.
cdk::P2PCall next_p2p_call(cdh_handle h) {
    if (h->capturing) {
        h->cap_exchanges += 1;
        return cdk::P2PCall{h->p2p_peers, h->rank, h->p2p_ranks, h->cap_exchanges, h->p2p_spin_limit, h->p2p_timeout, h->d_p2p_base};
    }
    h->p2p_epoch = epoch_after(h->p2p_epoch);
    return cdk::P2PCall{h->p2p_peers, h->rank, h->p2p_ranks, h->p2p_epoch, h->p2p_spin_limit, h->p2p_timeout, nullptr};
}

// The one exchange seam of the row-sharded path: sum `count` doubles at dbuf (device) over all ranks,
// in stream order.  Behind it: the direct exchange (short records), RCCL, or a caller-supplied host
// transport.  Not sharded: nothing to do.
int32_t allreduce(cdh_handle h, double* dbuf, size_t count) {
    if (h->host_fn) {
        if (h->capturing) return fail(h, CDH_BAD_ARG, "the host-staged exchange cannot be recorded in a graph");
        for (size_t o = 0; o < count; o += h->h_xchg_doubles) {   // long records go through the staging buffer in pieces
            const size_t cnt = std::min(h->h_xchg_doubles, count - o);
            HIPCHK(h, hipMemcpyAsync(h->h_xchg, dbuf + o, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            const int32_t rc = h->host_fn(h->host_user, h->h_xchg, (int64_t)cnt);
            if (rc != 0) return fail(h, CDH_RCCL_ERROR, "the host exchange callback reported a failure");
            HIPCHK(h, hipMemcpyAsync(dbuf + o, h->h_xchg, sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));   // the staging buffer is reused by the next exchange
        }
        h->n_host_calls += 1;
        return CDH_OK;
    }
    if (h->p2p_on && (count <= (size_t)cdk::kP2PMaxCount || !h->comm)) {
        CHK(p2p_check(h));
        for (size_t o = 0; o < count; o += cdk::kP2PMaxCount) {
            const int c = (int)std::min<size_t>(cdk::kP2PMaxCount, count - o);
            hipLaunchKernelGGL(cdk::k_p2p_allreduce, dim3((c + 255) / 256), dim3(256), 0, h->stream, dbuf + o, c,
                               next_p2p_call(h));
            if (!h->capturing) h->n_p2p_calls += 1;
        }
        HIPCHK(h, hipGetLastError());
        return CDH_OK;
    }
    CHK(exchange_alive(h));
    if (!h->comm) return CDH_OK;
    int rc = g_rccl.AllReduce(dbuf, dbuf, count, kNcclDouble, kNcclSum, h->comm, h->stream);
    if (rc != 0) {
        h->err = std::string("ncclAllReduce failed: ") +
                 (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return CDH_RCCL_ERROR;
    }
    if (h->capturing) h->cap_rccl += 1; else h->n_rccl_calls += 1;
    return CDH_OK;
}

// ---- r catches up with the covariance-form visits: r -= sum_k pending_k X_k, 64 columns per launch ----------
// Called by everything that reads r -- the streaming kernels, the dots, the moments, the Gram entry point, the
// download (cdh_get_residual) -- and before X itself changes; nothing else needs r, so a warm-started path
// whose solves all run from the cache pays for one catch-up when its caller finally asks for the residual
// (moves of the same coordinate merge in the meantime).
int32_t rebuild_residual_from(cdh_handle h, const cdh::SupportList& x, bool upload_beta);
int32_t sync_r(cdh_handle h) {
    if (h->r_lazy) {                  // r = y - X * x_lazy, now that somebody wants it (beta on the device mirrors h->x: left alone)
        h->r_lazy = false;
        return rebuild_residual_from(h, h->x_lazy, false);
    }
    if (h->r_pending_list.empty()) return CDH_OK;
    h->r_pristine = false;            // (the residual the handle stands for does not change: dots taken of it stay good)
    std::vector<int64_t>& L = h->r_pending_list;
    for (size_t o = 0; o < L.size(); o += 64) {
        const int cnt = (int)std::min<size_t>(64, L.size() - o);
        for (int i = 0; i < cnt; ++i) { h->h_idx[i] = L[o + (size_t)i]; h->h_hs[i] = h->r_pending[(size_t)L[o + (size_t)i]]; }
        HIPCHK(h, hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)cnt, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_hs, h->h_hs, sizeof(double) * (size_t)cnt, hipMemcpyHostToDevice, h->stream));
        CHK(dispatch(h, [&](auto* t) {
            using T = std::remove_pointer_t<decltype(t)>;
            hipLaunchKernelGGL(k_multi_axpy<T>, dim3(h->step_grid), dim3(kBlock), 0, h->stream, (const T*)h->X, h->ld,
                               h->nvec, (T*)h->r, h->d_idx, h->d_hs, 0, cnt);
            return CDH_OK;
        }));
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(h->stream));   // the pinned staging arrays are reused
        h->r_roundings += 1;
    }
    for (int64_t k : L) { h->r_pending[(size_t)k] = 0.0; h->r_in_pending[(size_t)k] = 0; }
    L.clear();
    h->gc.n_reconcile += 1;
    return CDH_OK;
}
// r is about to be overwritten from scratch (initialize!, a new y): nothing to catch up with
inline void drop_r_pending(cdh_handle h) {
    for (int64_t k : h->r_pending_list) { h->r_pending[(size_t)k] = 0.0; h->r_in_pending[(size_t)k] = 0; }
    h->r_pending_list.clear();
}

// ---- column dots over columns [j0, j0+nc): d_colout[2*j + {0,1}] = (x.r (w), x.x (w)) ----
int32_t col_dots(cdh_handle h, int64_t j0, int64_t nc, const void* rvec, bool use_w,
                 const int64_t* d_cols = nullptr) {
    if (rvec == h->r) CHK(sync_r(h));
    // batches of at most 4096 columns keep the partial buffer small
    for (int64_t b0 = 0; b0 < nc; b0 += 4096) {
        const int64_t bc = std::min<int64_t>(4096, nc - b0);
        const int64_t groups = (bc + kColGroup - 1) / kColGroup;
        // enough row chunks to fill the chip (~8 blocks per CU) but no more than kColChunks
        const int64_t want_chunks = std::max<int64_t>(1, ((int64_t)h->cus * 8 + groups - 1) / groups);
        const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)kColChunks, want_chunks,
                                                                          (h->nvec + kBlock - 1) / kBlock}));
        if ((size_t)groups * chunks * 2 * kColGroup > h->partials_doubles) return fail(h, CDH_BAD_ARG, "partials too small");
        dim3 grid(chunks, (unsigned)groups);
        CHK(dispatch(h, [&](auto* t) {
            using T = std::remove_pointer_t<decltype(t)>;
            hipLaunchKernelGGL(k_col_dots<T>, grid, dim3(kBlock), 0, h->stream, (const T*)h->X, h->ld,
                               h->nvec, use_w ? (const T*)h->w : (const T*)nullptr, (const T*)rvec,
                               j0 + b0, (int)bc, d_cols, h->d_partials);
            return CDH_OK;
        }));
        hipLaunchKernelGGL(k_col_dots_reduce, dim3((unsigned)bc), dim3(64), 0, h->stream,
                           h->d_partials, chunks, h->d_colout + 2 * b0);
        HIPCHK(h, hipGetLastError());
    }
    CHK(allreduce(h, h->d_colout, (size_t)(2 * nc)));
    return CDH_OK;
}

int32_t resid_moments_dev(cdh_handle h, const void* vec = nullptr, double shift = 0.0) {  // -> d_red[0..2] = sum v, sum v^2, sum w v^2 (v = r unless given; minus shift)
    if (!vec) { CHK(sync_r(h)); vec = h->r; }
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (h->nvec + kBlock - 1) / kBlock));
    CHK(dispatch(h, [&](auto* t) {
        using T = std::remove_pointer_t<decltype(t)>;
        hipLaunchKernelGGL(k_resid_moments<T>, dim3(grid), dim3(kBlock), 0, h->stream, h->nvec,
                           (const T*)vec, h->has_w ? (const T*)h->w : (const T*)nullptr, h->d_partials, shift, h->n);
        return CDH_OK;
    }));
    hipLaunchKernelGGL(k_sum_records, dim3(1), dim3(kBlock), 0, h->stream, h->d_partials, grid, h->d_red);
    HIPCHK(h, hipGetLastError());
    CHK(allreduce(h, h->d_red, 4));
    return CDH_OK;
}

// ---- gradient cache: bookkeeping (the passes that use it are below run_chunk) ------------------------
constexpr int kGcEngage = 3;        // mode 1: this many screened full passes run the plain way first
constexpr int kGcBusy = 64;         // more inactive coordinates than this about to move: a plain pass is cheaper
constexpr int kGcMaxFetch = 64;     // more uncached movers than this: re-reference instead of fetching their columns
constexpr int kGcMaxSupport = 512;  // supports beyond this are not worth Gram columns
constexpr size_t kGcMaxBytes = (size_t)1 << 30;
constexpr int64_t kGcCovWindow = 2048;      // positions one covariance-form chunk of a full pass may span
constexpr int64_t kGcCovRefresh = 200000;   // covariance-form visits after which g is re-read from X
constexpr int64_t kGcRowsPerNnz = 400;   // rows the problem must have per non-zero for the HOST-side fold and re-check to pay ...
constexpr int64_t kGcRowsPerNnzDev = 32; // ... and when the passes run on the device (round 3: g, the fold and the re-check never leave it)
inline int64_t gc_rows_per_nnz(const cdh_handle_s* h) {
    const GradCache& c = h->gc;
    if (const char* e = getenv("CDH_GC_ROWS_PER_NNZ")) return std::max<int64_t>(1, atoll(e));   // experiments
    return (c.cov && (c.d_scan || c.g.empty())) ? kGcRowsPerNnzDev : kGcRowsPerNnz;   // (before the first sizing: assume the device path)
}

// When the support has outgrown what the cache pays for.  Long columns: a support column costs 1/32 of a pass over X to fetch
// and p doubles per move to apply, against the read of X a cached full pass saves -- the cache stands aside once n < 32 nnz
// (400 nnz while the fold and the re-check ran on the host), and beyond kGcMaxSupport columns.  SHORT columns (a streamed
// visit is launch-bound there: ~0.35 us as its share of a block of 64 whatever it reads, against ~0.13 us in covariance form,
// and fetching a batch of columns is a pass over a small X) keep the cache whatever n / nnz is, up to kGcMaxSupportShort
// columns: benchmark/cd_bench.jl's n = 3000, p = 5000 path down to 0.03 lambda_max (774 non-zeros) 0.204 -> 0.118 s with the
// rule off below 512 columns, before the cap was raised.
constexpr int kGcMaxSupportShort = 4096;
constexpr size_t kGcShortColumnBytes = (size_t)1 << 20;
inline bool gc_short_columns(const cdh_handle_s* h) {
    return h->gc.cov && (h->gc.d_scan || h->gc.g.empty()) && (size_t)h->n * h->esz <= kGcShortColumnBytes;
}
inline bool gc_support_outgrown(const cdh_handle_s* h) {        // the rows-per-non-zero rule (modes 1 and 2)
    if (h->gc.mode == 3 || gc_short_columns(h)) return false;
    return h->x.nnz() * gc_rows_per_nnz(h) > h->n_total;
}
inline int64_t gc_max_support(const cdh_handle_s* h) { return gc_short_columns(h) ? kGcMaxSupportShort : kGcMaxSupport; }

// every loss and both storage types (round 3); a weighted loss only once its weights are there (they are zero until
// cdh_set_obs_weights: every a_k would be zero and nothing could be settled)
inline bool gc_applicable(const cdh_handle_s* h) {
    return h->gc.mode != 0 && (h->loss != CDH_WLS || h->has_w);
}
// g no longer describes r (y or the loss changed); with `columns` the Gram columns are gone too (X changed)
void gc_invalidate(cdh_handle h, bool columns) {
    GradCache& c = h->gc;
    c.valid = false; c.beta_ok = false; c.q_valid = false;
    c.g_host_ok = true; c.g_dev_ok = false;      // whatever g held is void; the next reference pass fills the host copy
    for (int64_t j : c.moved) { c.dbeta[(size_t)j] = 0.0; c.in_moved[(size_t)j] = 0; }
    c.moved.clear();
    if (columns) {
        c.a_dev_ok = false;
        c.G.clear(); c.G.shrink_to_fit();
        std::fill(c.slot.begin(), c.slot.end(), -1);
        c.slot_dev_ok = false;           // d_slot still names the old columns until the next upload (cov_solve.hpp reads it)
        c.colmax_slots = 0;
        c.cs_table_reset = true;         // ... and so does the device loop's Gram table
        c.dev_slots = 0;                 // the device store is refilled from slot 0 (its memory is kept)
        c.full_seen = 0;
    }
}
// r was just set to y - X * (the handle's iterate) by a kernel: beta_ref follows; what changed against the
// previous reference becomes pending moves (a warm start from another x is a move like any other)
void gc_after_rebuild(cdh_handle h, const cdh::SupportList& x) {
    GradCache& c = h->gc;
    if (c.beta_ref.empty()) return;   // cache never sized (not applicable so far)
    if (c.valid && !c.beta_ok) gc_invalidate(h, false);
    std::vector<double> nb((size_t)h->p, 0.0);
    for (int64_t s_ = 0; s_ < x.nnz(); ++s_) nb[(size_t)x.coord(s_)] = x.slot_value(s_);
    if (c.valid) {
        for (int64_t k = 0; k < h->p; ++k) {
            const double d = nb[(size_t)k] - c.beta_ref[(size_t)k];
            if (d != 0.0) {
                c.dbeta[(size_t)k] += d;
                if (!c.in_moved[(size_t)k]) { c.in_moved[(size_t)k] = 1; c.moved.push_back(k); }
            }
        }
    }
    c.beta_ref.swap(nb);
    c.beta_ok = true;
    c.q_valid = false;
}
// the streamed visits of a chunk have updated r: g learns of them at the next fold
void gc_note_moves(cdh_handle h, const int64_t* idx0, int m) {
    GradCache& c = h->gc;
    if (!c.valid && !c.beta_ok) return;
    for (int i = 0; i < m; ++i) {
        const double hv = h->h_hs[i];
        if (hv == 0.0) continue;
        if (hv != hv) { gc_invalidate(h, false); return; }   // a NaN step: nothing is known about r any more
        const int64_t k = idx0[i];
        if (c.beta_ok) c.beta_ref[(size_t)k] += hv;
        c.q_valid = false;            // a streamed visit has changed r
        if (c.valid) {
            c.dbeta[(size_t)k] += hv;
            if (!c.in_moved[(size_t)k]) { c.in_moved[(size_t)k] = 1; c.moved.push_back(k); }
        }
    }
}

// ---- initialize!: upload support, r = y - X beta ------------------------------------
inline bool same_iterate(const cdh::SupportList& a, const cdh::SupportList& b) {
    if (a.size() != b.size() || a.nnz() != b.nnz()) return false;
    for (int64_t s = 0; s < a.nnz(); ++s) if (a.coord(s) != b.coord(s) || !(a.slot_value(s) == b.slot_value(s))) return false;
    return true;
}
int32_t rebuild_residual_from(cdh_handle h, const cdh::SupportList& x, bool upload_beta) {
    if (h->r_pristine && !h->r_lazy && h->r_pending_list.empty() && same_iterate(x, h->x_pristine)) {
        // the buffer already holds exactly what this rebuild would write
        if (upload_beta) {
            if (x.nnz() == 0) HIPCHK(h, hipMemsetAsync(h->beta, 0, sizeof(double) * h->p, h->stream));
            else {
                std::vector<double> dense((size_t)h->p, 0.0);
                for (int64_t s = 0; s < x.nnz(); ++s) dense[(size_t)x.coord(s)] = x.slot_value(s);
                HIPCHK(h, hipMemcpyAsync(h->beta, dense.data(), sizeof(double) * h->p, hipMemcpyHostToDevice, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
        }
        h->r_roundings = 1;
        h->r_consistent = true;
        h->n_rebuild_skipped += 1;
        gc_after_rebuild(h, x);
        return CDH_OK;
    }
    touch_r(h);
    drop_r_pending(h);
    h->r_lazy = false;
    h->r_roundings = 1;               // r = y - X beta, summed in fp64 and rounded once
    const int64_t nnz = x.nnz();
    if (nnz == 0) {   // the cold start: beta = 0, r = y; nothing host-side is in flight, so no wait either
        if (upload_beta) HIPCHK(h, hipMemsetAsync(h->beta, 0, sizeof(double) * h->p, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->r, h->y, (size_t)h->ld * h->esz, hipMemcpyDeviceToDevice, h->stream));
        h->r_consistent = true;
        h->r_pristine = true; h->x_pristine = x;
        gc_after_rebuild(h, x);
        return CDH_OK;
    }
    if (upload_beta) {
        std::vector<double> dense((size_t)h->p, 0.0);
        for (int64_t s = 0; s < nnz; ++s) dense[(size_t)x.coord(s)] = x.slot_value(s);
        HIPCHK(h, hipMemcpyAsync(h->beta, dense.data(), sizeof(double) * h->p, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));   // `dense` goes out of scope
    }
    std::vector<int64_t> si((size_t)nnz);
    std::vector<double> sv((size_t)nnz);
    for (int64_t s = 0; s < nnz; ++s) { si[(size_t)s] = x.coord(s); sv[(size_t)s] = x.slot_value(s); }
    HIPCHK(h, hipMemcpyAsync(h->d_sup_idx, si.data(), sizeof(int64_t) * nnz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_sup_val, sv.data(), sizeof(double) * nnz, hipMemcpyHostToDevice, h->stream));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (h->nvec + kBlock - 1) / kBlock));
    CHK(dispatch(h, [&](auto* t) {
        using T = std::remove_pointer_t<decltype(t)>;
        hipLaunchKernelGGL(k_init_resid<T>, dim3(grid), dim3(kBlock), 0, h->stream, (const T*)h->X, h->ld,
                           h->nvec, (const T*)h->y, (T*)h->r, h->d_sup_idx, h->d_sup_val, (int)nnz);
        return CDH_OK;
    }));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));  // host vectors above go out of scope
    h->r_consistent = true;
    h->r_pristine = true; h->x_pristine = x;
    gc_after_rebuild(h, x);
    return CDH_OK;
}
int32_t rebuild_residual(cdh_handle h) { return rebuild_residual_from(h, h->x, true); }

// does any coordinate repeat in the chunk?  (scheduler-made lists never do; caller-made ones may)
void note_duplicates(cdh_handle h, const int64_t* idx0, int m) {
    if ((int64_t)h->stamp.size() != h->p) h->stamp.assign((size_t)h->p, 0);
    h->chunk_dup = false;
    for (int i = 0; i < m; ++i) {
        if (h->stamp[(size_t)idx0[i]]) { h->chunk_dup = true; break; }
        h->stamp[(size_t)idx0[i]] = 1;
    }
    for (int i = 0; i < m; ++i) h->stamp[(size_t)idx0[i]] = 0;
}

// The visits of a chunk have been enqueued (streamed, or in covariance form): bring their results back,
// replay them on the host-side SparseIterate, and tell the gradient cache what moved.
int32_t finish_chunk(cdh_handle h, const int64_t* idx0, int m, double* maxH) {
    HIPCHK(h, hipMemcpyAsync(h->h_hs, h->d_hs, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_newval, h->d_newval, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_touched, h->d_touched, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_ctrl, h->d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(p2p_check(h));
    if (h->h_ctrl->domain_error) h->domain_error = true;
    const double mh = h->h_ctrl->maxH;
    if (mh > *maxH) *maxH = mh;   // a NaN h never raises maxH: `abs(h) > maxH` (coordinate_descent.jl:104)
    // replay the SparseIterate writes of the visits in order (x[k] += b/a ; cdprox!)
    for (int i = 0; i < m; ++i) {
        const int64_t k = idx0[i];
        if (h->h_touched[i] && h->x.get(k) == 0.0) h->x.set(k, 1.0);  // pre-prox non-zero: slot appended
        h->x.set(k, h->h_newval[i]);
    }
    gc_note_moves(h, idx0, m);
    return CDH_OK;
}

// ---- one chunk of a pass: visits idx0[0..m) -------------------------------------------
template <typename T, int B> int32_t launch_block_chunk(cdh_handle h, int m) {
    constexpr int NREC = BlockRec<B>::N;
    const int G = h->block_grid;
    int nprev = 0;
    for (int pos0 = 0; pos0 < m; pos0 += B) {
        const int nb = std::min(B, m - pos0);
        if (h->nt)
            hipLaunchKernelGGL((k_blockstep<T, B, true>), dim3(G), dim3(kBlock), 0, h->stream, (const T*)h->X,
                               h->ld, h->nvec, (T*)h->r, h->d_idx, h->d_hs, pos0, nb, nprev, h->d_partials);
        else
            hipLaunchKernelGGL((k_blockstep<T, B, false>), dim3(G), dim3(kBlock), 0, h->stream, (const T*)h->X,
                               h->ld, h->nvec, (T*)h->r, h->d_idx, h->d_hs, pos0, nb, nprev, h->d_partials);
        if (!sharded(h)) {
            hipLaunchKernelGGL((k_block_finalize<B, true>), dim3(1), dim3(1024), 0, h->stream,
                               h->d_partials, G, nb, h->d_ctrl, h->beta, h->omega, h->d_idx, h->d_hs,
                               h->d_newval, h->d_touched, pos0, h->d_red);
        } else {
            hipLaunchKernelGGL((k_block_finalize<B, false>), dim3(1), dim3(1024), 0, h->stream,
                               h->d_partials, G, nb, h->d_ctrl, h->beta, h->omega, h->d_idx, h->d_hs,
                               h->d_newval, h->d_touched, pos0, h->d_red);
            CHK(allreduce(h, h->d_red, NREC));
            hipLaunchKernelGGL((k_block_scalar<B>), dim3(1), dim3(64), 0, h->stream, h->d_red, nb,
                               h->d_ctrl, h->beta, h->omega, h->d_idx, h->d_hs, h->d_newval,
                               h->d_touched, pos0);
        }
        nprev = nb;
    }
    const int last0 = ((m - 1) / B) * B;
    hipLaunchKernelGGL((k_block_axpy<T, B>), dim3(G), dim3(kBlock), 0, h->stream, (const T*)h->X, h->ld,
                       h->nvec, (T*)h->r, h->d_idx, h->d_hs, last0, m - last0);
    return CDH_OK;
}

// blocks of k_gramstep per CU: as many as stay resident for the variant launched (register
// footprint: B = 64 holds 2 waves per SIMD, the narrower ones 3; the LDS-transposed variants are
// bounded by their 68-78 KB operand tiles to 2 blocks -- B = 32 launched with 3 per CU ran its
// third of the grid as a second, half-empty round: 933 -> 861 us per block at 1e7 rows), never
// more than the partial buffer was sized for
inline int gram_blocks_per_cu(cdh_handle h, int NG, bool lt) {
    if (lt) return h->lt_per_cu > 0 ? h->lt_per_cu : 2;
    return NG == 4 ? std::min(h->gram32_per_cu, 2) : NG == 2 ? h->gram32_per_cu : h->gram_per_cu;
}
inline int NGgrid(cdh_handle h, int NG, bool lt = false) {
    return balanced_grid(h->gram_units, (int64_t)h->cus * gram_blocks_per_cu(h, NG, lt));
}

// wide blocks (B = 16 / 32 / 64): MFMA-accumulated Gram kernel + two-stage reduction
template <typename T, int NG> int32_t launch_gram_chunk(cdh_handle h, int m) {
    using R = GramRec<NG>;
    constexpr int B = R::B;
    const T* wts = h->has_w ? (const T*)h->w : (const T*)nullptr;
    // LDS-transposed (fully coalesced) operand loads: measured -5 % at >= 5e6 rows for every
    // width; with short chunks also -10 % for B = 32 at 6.25e5 .. 1.25e6 rows and neutral for
    // B = 64.  B = 16 takes it on long columns only.  fp32 B = 64 would spill: fragment path.
    const bool use_lt = (h->lt == 1 || (h->lt == 2 && (NG >= 2 || h->n >= 2000000))) && !(NG == 4 && sizeof(T) == 4);
    int G = NGgrid(h, NG, use_lt);
    // short columns (a launch is fewer than kShortRounds rounds of 64-vector chunks): chunks of
    // one sub-chunk, so a partly filled last round costs a quarter (B = 64) or half (B = 32) as much
    const int64_t rounds64 = ((h->nvec + 63) / 64) / ((int64_t)G * kGramWaves);
    const bool short_chunks = use_lt && (h->ks == 1 || (h->ks == 0 && rounds64 < kShortRounds));
    // ... and as many blocks as there are chunks of THAT length to hand out, four per block (round 3: the grid was sized for
    // 64-vector chunks, which at 24 KB columns -- benchmark/cd_bench.jl's n = 3000 -- left 6 blocks to walk 94 short chunks,
    // four apiece and one after the other: 31 us per launch whatever it read)
    {
        const int64_t cvn = use_lt ? (int64_t)(64 / NG) * (short_chunks ? 1 : NG) : 64;
        const int64_t nchunks = (h->nvec + cvn - 1) / cvn;
        G = balanced_grid((nchunks + kGramWaves - 1) / kGramWaves, (int64_t)h->cus * gram_blocks_per_cu(h, NG, use_lt));
    }
    if ((size_t)G * R::N > h->partials_doubles) return fail(h, CDH_BAD_ARG, "partial buffer too small for this grid");
    int nprev = 0;
    for (int pos0 = 0; pos0 < m; pos0 += B) {
        const int nb = std::min(B, m - pos0);
        auto go = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(G), dim3(64 * kGramWaves), 0, h->stream, (const T*)h->X, h->ld, h->nvec,
                               wts, (T*)h->r, h->d_idx, h->d_hs, pos0, nb, nprev, h->d_partials);
        };
        if (use_lt) {
            if constexpr (NG == 1) {
                go(k_gramstep<T, 1, true, true>);
            } else if constexpr (NG == 4 && sizeof(T) == 4) {
                // (never reached: use_lt is false for fp32 B = 64 -- its LDS-transposed variants spilled 160 - 280 bytes per lane
                // and are not instantiated)
                go(k_gramstep<T, NG, true>);
            } else {
                if (short_chunks) go(k_gramstep<T, NG, true, true, 1>); else go(k_gramstep<T, NG, true, true>);
            }
        } else if (h->nt) {
            go(k_gramstep<T, NG, true>);
        } else {
            go(k_gramstep<T, NG, false>);
        }
        hipLaunchKernelGGL(k_gram_reduce, dim3((R::N + kReduceVals - 1) / kReduceVals), dim3(64 * kReduceWaves), 0, h->stream, h->d_partials, G, R::N, h->d_red);
        CHK(allreduce(h, h->d_red, R::N));
        hipLaunchKernelGGL((k_gram_scalar<NG>), dim3(1), dim3(64), 0, h->stream, h->d_red, nb, h->chunk_dup ? 1 : 0, h->d_ctrl,
                           h->beta, h->omega, h->d_idx, h->d_hs, h->d_newval, h->d_touched, pos0);
        nprev = nb;
    }
    const int last0 = ((m - 1) / B) * B;
    hipLaunchKernelGGL(k_multi_axpy<T>, dim3(h->step_grid), dim3(kBlock), 0, h->stream, (const T*)h->X, h->ld,
                       h->nvec, (T*)h->r, h->d_idx, h->d_hs, last0, m - last0);
    return CDH_OK;
}

template <typename T> int32_t launch_coord_chunk(cdh_handle h, int m) {
    const int G = h->step_grid;
    for (int pos = 0; pos < m; ++pos) {
        if (h->has_w)
            hipLaunchKernelGGL((k_step<T, true, true>), dim3(G), dim3(kBlock), 0, h->stream, (const T*)h->X,
                               h->ld, h->nvec, (const T*)h->w, (T*)h->r, h->d_idx, h->d_hs, pos, h->d_partials);
        else if (h->nt)
            hipLaunchKernelGGL((k_step<T, false, true>), dim3(G), dim3(kBlock), 0, h->stream, (const T*)h->X,
                               h->ld, h->nvec, (const T*)nullptr, (T*)h->r, h->d_idx, h->d_hs, pos, h->d_partials);
        else
            hipLaunchKernelGGL((k_step<T, false, false>), dim3(G), dim3(kBlock), 0, h->stream, (const T*)h->X,
                               h->ld, h->nvec, (const T*)nullptr, (T*)h->r, h->d_idx, h->d_hs, pos, h->d_partials);
        if (!sharded(h)) {
            hipLaunchKernelGGL(k_finalize<true>, dim3(1), dim3(kBlock), 0, h->stream, h->d_partials, G,
                               h->d_ctrl, h->beta, h->omega, h->d_idx, h->d_hs, h->d_newval, h->d_touched,
                               pos, h->d_red);
        } else {
            hipLaunchKernelGGL(k_finalize<false>, dim3(1), dim3(kBlock), 0, h->stream, h->d_partials, G,
                               h->d_ctrl, h->beta, h->omega, h->d_idx, h->d_hs, h->d_newval, h->d_touched,
                               pos, h->d_red);
            CHK(allreduce(h, h->d_red, 4));
            hipLaunchKernelGGL(k_scalar_update, dim3(1), dim3(64), 0, h->stream, h->d_red, h->d_ctrl,
                               h->beta, h->omega, h->d_idx, h->d_hs, h->d_newval, h->d_touched, pos);
        }
    }
    hipLaunchKernelGGL(k_axpy<T>, dim3(G), dim3(kBlock), 0, h->stream, (const T*)h->X, h->ld, h->nvec,
                       (T*)h->r, h->d_idx, h->d_hs, m);
    return CDH_OK;
}

int32_t all_ranks_agree(cdh_handle h, bool mine, bool* all);   // grad_cache.hpp
int32_t agree_default_width(cdh_handle h) {
    if (h->width_agreed || !h->width_default || !sharded(h) || h->nranks <= 1) return CDH_OK;
    bool all64 = false;
    CHK(all_ranks_agree(h, h->blockB == 64, &all64));
    if (!all64) h->blockB = 32;      // some shard is on the long side of the cut: every rank takes the long-column width
    h->width_agreed = true;
    return CDH_OK;
}

int32_t run_chunk(cdh_handle h, const int64_t* idx0, int m, double* maxH) {
    CHK(agree_default_width(h));
    CHK(sync_r(h));   // the streaming kernels read and write r
    touch_r(h);
    if (h->p2p_epoch >= kEpochSoftWrap) h->p2p_epoch = (h->p2p_epoch & 1u) ? 1u : 2u;   // slot parity keeps alternating
    std::memcpy(h->h_idx, idx0, sizeof(int64_t) * (size_t)m);
    note_duplicates(h, idx0, m);
    HIPCHK(h, hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    h->ctrl.maxH = 0.0;
    h->ctrl.domain_error = 0;
    CHK(upload_ctrl(h));
    if (h->prof) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    // observation weights: the wide-block kernel carries them; the vector-ALU blocks (B <= 8) do not
    const bool blocked = (h->mode == CDH_SWEEP_BLOCK) && (!h->has_w || h->blockB >= 16);
    auto enqueue = [&]() {
        return dispatch(h, [&](auto* t) {
            using T = std::remove_pointer_t<decltype(t)>;
            if (!blocked) return launch_coord_chunk<T>(h, m);
            if (h->blockB == 64) return launch_gram_chunk<T, 4>(h, m);
            if (h->blockB == 32) return launch_gram_chunk<T, 2>(h, m);
            if (h->blockB == 16) return launch_gram_chunk<T, 1>(h, m);
            if (h->blockB == 8) return launch_block_chunk<T, 8>(h, m);
            if (h->blockB == 4) return launch_block_chunk<T, 4>(h, m);
            return launch_block_chunk<T, 2>(h, m);
        });
    };
    // hipGraph replay: the launch sequence of an m-visit chunk depends only on (m, mode, B, exchange) --
    // every kernel takes its visit position as a literal and reads idx / hs / lambda from device
    // memory -- so one captured graph serves every later pass of that length.  Row shards too: the
    // direct exchange reads its epoch base from device memory (set before each replay), RCCL
    // all-reduces are recorded by RCCL itself as graph nodes.  Only the host-staged exchange cannot
    // be recorded.  A failed capture switches the handle back to node-by-node launches for good.
    bool launched = false;
    // (a graph of one or two launches saves nothing and costs a capture: short chunks go node by node)
    if (h->use_graph && !h->graph_broken && !h->host_fn && !h->p2p_dead && m >= (blocked ? 2 * h->blockB : 8)) {
        const uint64_t key = ((uint64_t)m << 20) | ((uint64_t)(blocked ? h->blockB : 0) << 8) |
                             (h->comm ? 32u : 0u) | (h->p2p_on ? 16u : 0u) |
                             (h->chunk_dup ? 4u : 0u) | (h->has_w ? 2u : 0u) | (h->nt ? 1u : 0u);
        cdh_handle_s::GraphEntry* entry = nullptr;
        for (auto& e : h->graphs) if (e.key == key) entry = &e;
        if (!entry) {
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            h->capturing = true; h->cap_exchanges = 0; h->cap_rccl = 0;
            hipError_t e0 = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal);
            const int32_t erc = e0 == hipSuccess ? enqueue() : (int32_t)CDH_HIP_ERROR;
            hipError_t e1 = e0 == hipSuccess ? hipStreamEndCapture(h->stream, &graph) : e0;
            h->capturing = false;
            if (erc == CDH_OK && e1 == hipSuccess && graph) e1 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (erc != CDH_OK || e1 != hipSuccess || !exec) {
                (void)hipGetLastError();
                h->graph_broken = true;
            } else {
                if (h->graphs.size() >= 32) {  // bounded cache: drop the oldest
                    (void)hipGraphExecDestroy(h->graphs.front().exec);
                    h->graphs.erase(h->graphs.begin());
                }
                h->graphs.push_back({key, exec, h->cap_exchanges, h->cap_rccl});
                entry = &h->graphs.back();
            }
        }
        if (entry) {
            if (entry->exchanges) {
                // epochs base+1 .. base+exchanges; across the 32-bit wrap the slot parity keeps alternating
                if ((uint64_t)h->p2p_epoch + entry->exchanges >= (uint64_t)kEpochWrap) h->p2p_epoch = (h->p2p_epoch & 1u) ? 1u : 2u;
                hipLaunchKernelGGL(cdk::k_set_u32, dim3(1), dim3(1), 0, h->stream, h->d_p2p_base, h->p2p_epoch);
                h->p2p_epoch += entry->exchanges;
                h->n_p2p_calls += entry->exchanges;
            }
            h->n_rccl_calls += entry->rccl;
            HIPCHK(h, hipGraphLaunch(entry->exec, h->stream));
            launched = true;
        }
    }
    if (!launched) CHK(enqueue());
    h->r_roundings += blocked ? (m + h->blockB - 1) / h->blockB + 1 : m + 1;   // one rewrite of r per launch that applies updates
    HIPCHK(h, hipGetLastError());
    if (h->prof) HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    CHK(finish_chunk(h, idx0, m, maxH));
    if (h->prof) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->prof_ms += ms;
        const double vec = (double)h->n * (double)h->esz;
        if (!blocked) {
            h->prof_launches += m;
            for (int i = 0; i < m; ++i) {
                const bool ap = i > 0 && h->h_hs[i - 1] != 0.0;
                h->prof_bytes += vec * ((h->has_w ? 3.0 : 2.0) + (ap ? 2.0 : 0.0));
            }
            if (h->h_hs[m - 1] != 0.0) h->prof_bytes += 3.0 * vec;
        } else {
            const int B = h->blockB;
            for (int pos0 = 0; pos0 < m; pos0 += B) {
                const int nb = std::min(B, m - pos0);
                int nzp = 0;
                for (int i = std::max(0, pos0 - B); i < pos0; ++i) nzp += h->h_hs[i] != 0.0;
                h->prof_bytes += vec * (nb + 1 + nzp + (nzp ? 1 : 0));
                h->prof_launches += 1;
            }
            int nzl = 0;
            for (int i = ((m - 1) / B) * B; i < m; ++i) nzl += h->h_hs[i] != 0.0;
            if (nzl) h->prof_bytes += vec * (nzl + 2);
        }
    }
    return CDH_OK;
}

constexpr int kScreenMinPass = 16;   // screens / the cache: passes over fewer coordinates than that are a block or two anyway
#include "grad_cache.hpp"   // gc_size, gc_validate, gc_fetch, gc_fold, gc_full_pass
#include "small_solve.hpp"  // small_applicable, small_prepare, small_solve
#include "cov_solve.hpp"    // k_cov_solve, cov_solve: the pass loop of a cache-served solve on the device

// _cdPass! (coordinate_descent.jl:94-110)
// Screening of a FULL pass (exact, no reference counterpart).  A visit of a coordinate with
// beta_k == 0 leaves everything unchanged unless |X_k'r| exceeds its threshold (LS: lambda n w_k,
// cd_differentiable_function.jl:101-104 with x[k] == 0; SQRT: lambda w_k ||r||, :276), and as long
// as nothing moves r does not change -- so one pass over S columns that only takes their dots
// settles a whole run of such visits, and the Gram machinery is started at the first visit that
// can move.  Same iterates as visiting one by one; a full pass over a sparse iterate then reads X
// about once, at the plain streaming rate.  A 1e-9 relative margin sends borderline coordinates
// through the exact path.  After a hit the next `cool` visits are not screened (doubling on
// repeated hits), so a pass in which everything moves pays for a handful of screens only; a screen
// that settles all of its columns doubles the next one (64 -> 1024 columns: one host round trip per
// screen, so long quiet stretches run at the plain streaming rate of k_col_dots).
constexpr int kScreen = 64, kScreenMax = 1024;

int32_t screened_full_pass(cdh_handle h, const int64_t* idx0, int64_t m, double* maxH) {
    {   // from the gradient cache when it is engaged: no read of X for the settled visits at all
        bool handled = false;
        CHK(gc_full_pass(h, idx0, m, maxH, &handled));
        if (handled) return CDH_OK;
    }
    const int B = (h->mode == CDH_SWEEP_BLOCK) ? h->blockB : 1;
    const double lam = h->ctrl.lambda0, nt = (double)h->n_total;
    const std::vector<double>& om = h->h_omega;
    std::vector<double> cd((size_t)(2 * kScreenMax));
    const int64_t scr_max = std::min<int64_t>({(int64_t)kScreenMax, h->p, h->cap});   // d_colout holds 2p values
    int64_t pos = 0, cool = 0, cool_len = kScreen, scr = std::min<int64_t>(kScreen, scr_max);
    while (pos < m) {
        if (cool > 0) {   // unscreened stretch: whole Gram blocks
            const int mm = (int)std::min<int64_t>({(int64_t)h->cap, m - pos, std::max<int64_t>(cool, B)});
            CHK(run_chunk(h, idx0 + pos, mm, maxH));
            pos += mm; cool -= mm;
            continue;
        }
        const int S = (int)std::min<int64_t>(scr, m - pos);
        std::memcpy(h->h_idx, idx0 + pos, sizeof(int64_t) * (size_t)S);
        HIPCHK(h, hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)S, hipMemcpyHostToDevice, h->stream));
        CHK(col_dots(h, 0, S, h->r, h->has_w, h->d_idx));   // X_k'W r, X_k'W X_k with observation weights
        HIPCHK(h, hipMemcpyAsync(cd.data(), h->d_colout, sizeof(double) * 2 * S, hipMemcpyDeviceToHost, h->stream));
        double rnorm = 0.0;
        if (h->loss == CDH_SQRT) {
            CHK(resid_moments_dev(h));
            HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->loss == CDH_SQRT) rnorm = std::sqrt(h->h_red[1]);
        int hit = S;
        for (int i = 0; i < S; ++i) {
            const int64_t k = idx0[pos + i];
            const double b = cd[(size_t)(2 * i)];
            const double w = h->has_omega ? om[(size_t)k] : 1.0;
            const double thr = (h->loss == CDH_SQRT ? lam * w * rnorm : lam * nt * w) * (1.0 - 1e-9);
            // a zero column (a == 0) goes to the exact path too: the reference turns it into NaN
            if (h->x.get(k) != 0.0 || !(std::fabs(b) <= thr) || !(cd[(size_t)(2 * i + 1)] > 0.0)) { hit = i; break; }
        }
        // visits [pos, pos + hit) are settled: h = 0; replay what the reference's SparseIterate
        // would have seen (LS: x[k] += b/a stores a slot when b != 0, cdprox! zeroes it)
        for (int i = 0; i < hit; ++i) {
            const int64_t k = idx0[pos + i];
            if (h->loss != CDH_SQRT && cd[(size_t)(2 * i)] != 0.0) h->x.set(k, 1.0);
            h->x.set(k, 0.0);
        }
        pos += hit;
        if (hit < S) {
            const int mm = (int)std::min<int64_t>(std::max(B, 1), m - pos);
            CHK(run_chunk(h, idx0 + pos, mm, maxH));
            pos += mm;
            cool = cool_len; cool_len = std::min<int64_t>(cool_len * 2, m);
            scr = std::min<int64_t>(kScreen, scr_max);
        } else {
            cool_len = kScreen;
            scr = std::min<int64_t>(scr * 2, scr_max);
        }
    }
    return CDH_OK;
}

// _cdPass! (coordinate_descent.jl:94-110)
int32_t run_pass(cdh_handle h, const int64_t* idx0, int64_t m, double* maxH, bool screen = false) {
    *maxH = 0.0;
    if (screen && h->screening && (h->loss != CDH_WLS || h->has_w) && m >= kScreenMinPass && h->x.nnz() * 4 <= h->p) {
        CHK(screened_full_pass(h, idx0, m, maxH));
    } else {
        for (int64_t off = 0; off < m; off += h->cap) {
            const int mm = (int)std::min<int64_t>(h->cap, m - off);
            // an active pass while the gradient cache holds the support's Gram columns: no read of X at all
            if (gc_ready_for_cov(h, idx0 + off, mm)) CHK(cov_chunk(h, idx0 + off, mm, maxH));
            else CHK(run_chunk(h, idx0 + off, mm, maxH));
        }
    }
    h->x.dropzeros();
    return CDH_OK;
}

// _coordinateDescent! (coordinate_descent.jl:65-92)
int32_t solve(cdh_handle h, const cdh_options* o, cdh::VisitScheduler& sched, cdh_stats* st) {
    bool prev_converged = false, converged = true;
    std::vector<int64_t> visit;
    st->converged = 0;
    for (int64_t iter = 0; iter < o->maxIter;) {
        // while the gradient cache can serve the passes, the loop itself runs on the device (cov_solve.hpp): one host round
        // trip per solve; what comes back is where the state machine stands
        {
            int outcome = kCsNotNow;
            CHK(cov_solve(h, o, sched, st, &prev_converged, &converged, &iter, &outcome));
            if (outcome == kCsFinished) break;
            if (outcome == kCsAgain) continue;
        }
        const bool full = converged;
        sched.next_pass(h->x, full, visit);
        double maxH = 0.0;
        if (!visit.empty()) CHK(run_pass(h, visit.data(), (int64_t)visit.size(), &maxH, full));
        else h->x.dropzeros();
        h->gc.prep_state = 0;
        st->passes += 1; st->visits += (int64_t)visit.size(); st->maxH = maxH;
        if (full) st->full_passes += 1;
        prev_converged = converged;
        converged = maxH < o->optTol;
        ++iter;
        if (prev_converged && converged) { st->converged = 1; break; }
    }
    st->domain_error = h->domain_error ? 1 : 0;
    return CDH_OK;
}

int32_t lambda_max(cdh_handle h, double* out, std::vector<double>* dots = nullptr /* the (X_k'r, a_k) pairs, for a caller that can use them */) {
    CHK(col_dots(h, 0, h->p, h->r, h->loss == CDH_WLS));
    std::vector<double> cd_own;
    std::vector<double>& cd = dots ? *dots : cd_own;
    cd.resize((size_t)(2 * h->p));
    HIPCHK(h, hipMemcpyAsync(cd.data(), h->d_colout, sizeof(double) * 2 * h->p, hipMemcpyDeviceToHost, h->stream));
    double denom = (double)h->n_total;
    if (h->loss == CDH_SQRT) {
        CHK(resid_moments_dev(h));
        HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dots_stash = cd; h->dots_valid = true; h->dots_w = h->loss == CDH_WLS;   // (col_dots has brought r up to date first)
    if (h->loss == CDH_SQRT) denom = std::sqrt(h->h_red[1]);
    const std::vector<double>& om = h->h_omega;
    double lmax = 0.0;
    for (int64_t k = 0; k < h->p; ++k) {
        double t = std::fabs(-cd[(size_t)(2 * k)] / denom);
        if (h->has_omega) t /= om[(size_t)k];
        if (t > lmax) lmax = t;
    }
    *out = lmax;
    return CDH_OK;
}

void free_all(cdh_handle h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (void* m : h->p2p_mapped) (void)hipIpcCloseMemHandle(m);
    if (h->p2p_inbox) (void)hipFree(h->p2p_inbox);
    if (h->d_p2p_base) (void)hipFree(h->d_p2p_base);
    if (h->gc.d_cross) (void)hipFree(h->gc.d_cross);
    if (h->gc.d_cols) (void)hipFree(h->gc.d_cols);
    if (h->gc.d_cross_part) (void)hipFree(h->gc.d_cross_part);
    if (h->gc.d_g) (void)hipFree(h->gc.d_g);
    if (h->gc.d_G) (void)hipFree(h->gc.d_G);
    for (double* q : h->gc.d_G_retired) (void)hipFree(q);
    if (h->gc.d_slot) (void)hipFree(h->gc.d_slot);
    if (h->gc.h_g_pin) (void)hipHostFree(h->gc.h_g_pin);
    if (h->small.d_G) (void)hipFree(h->small.d_G);
    if (h->small.d_iota) (void)hipFree(h->small.d_iota);
    if (h->small.d_io) (void)hipFree(h->small.d_io);
    if (h->small.d_ca) (void)hipFree(h->small.d_ca);
    if (h->small.h_io) (void)hipHostFree(h->small.h_io);
    {
        GradCache& c = h->gc;
        void* dv[] = {c.d_a, c.d_g_snap, c.d_beta_snap, c.d_qs, c.d_pass_idx, c.d_pos_of, c.d_scanbuf, c.d_setflag, c.d_pack};
        for (void* q : dv) if (q) (void)hipFree(q);
        if (c.h_scanbuf) (void)hipHostFree(c.h_scanbuf);
        if (c.h_pack) (void)hipHostFree(c.h_pack);
        if (c.cs_dev) (void)hipFree(c.cs_dev);
        if (c.cs_pin) (void)hipHostFree(c.cs_pin);
    }
    if (h->h_xchg) (void)hipHostFree(h->h_xchg);
    if (h->p2p_timeout) (void)hipHostFree(h->p2p_timeout);
    for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.exec);
    void* dev[] = {h->X, h->y, h->r, h->w, h->beta, h->omega, h->d_ctrl, h->d_idx, h->d_hs, h->d_newval,
                   h->d_touched, h->d_partials, h->d_red, h->d_colout, h->d_sup_idx, h->d_sup_val};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* pin[] = {h->h_idx, h->h_hs, h->h_newval, h->h_red, h->h_touched, h->h_ctrl};
    for (void* p : pin) if (p) (void)hipHostFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

}  // namespace

// ====================================================================================
extern "C" {

int32_t cdh_device_count(int32_t* out) {
    NEED_P(nullptr, out);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_create_error = hipGetErrorString(e); *out = 0; return CDH_HIP_ERROR; }
    *out = n;
    return CDH_OK;
}

const char* cdh_last_error(cdh_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int32_t cdh_create(cdh_handle* out, int32_t dtype, int32_t loss, int64_t n_local, int64_t n_total,
                   int64_t row_offset, int64_t p, int32_t device) {
    if (!out) return fail(nullptr, CDH_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (dtype != CDH_F64 && dtype != CDH_F32) return fail(nullptr, CDH_BAD_ARG, "dtype must be CDH_F64 or CDH_F32");
    if (loss != CDH_LS && loss != CDH_SQRT && loss != CDH_WLS) return fail(nullptr, CDH_BAD_ARG, "unknown loss");
    if (n_local <= 0 || p <= 0 || n_total < n_local || row_offset < 0 || row_offset + n_local > n_total)
        return fail(nullptr, CDH_DIM_MISMATCH, "need 0 < n_local <= n_total, p > 0, shard inside [0, n_total)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CDH_HIP_ERROR, "no HIP device: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, CDH_BAD_ARG, "device index out of range");
    cdh_handle h = new cdh_handle_s();
    h->dtype = dtype; h->loss = loss; h->device = device;
    h->n = n_local; h->n_total = n_total; h->row0 = row_offset; h->p = p;
    h->esz = dtype == CDH_F64 ? 8 : 4;
    const int NV = dtype == CDH_F64 ? 2 : 4;
    h->ld = round_up(n_local, 32);            // every column starts 128/256-B aligned
    h->nvec = round_up(n_local, NV) / NV;     // pad rows [n, ld) are zero in X, y, r, w
    h->cap = std::max<int64_t>(p, 4096);
    // The default sweep width: B = 32 streams fastest per visit on long columns; on SHORT ones (the grid is not even full:
    // fewer than 512 blocks x 4 waves x 64 vectors) a block of visits costs three launches whatever it reads, and B = 64
    // halves them (benchmark/cd_bench.jl's dense solve at n = 3000: 0.40 s at B = 32, 0.26 s at B = 64).  fp32 storage has
    // no LDS-transposed B = 64 kernel and keeps 32.
    if (dtype == CDH_F64 && n_local < (int64_t)512 * kGramWaves * 64 * 2) h->blockB = 64;
    h->x.resize(p);
    h->r_pending.assign((size_t)p, 0.0);
    h->r_in_pending.assign((size_t)p, 0);
    int32_t rc = [&]() -> int32_t {
        HIPCHK(h, hipSetDevice(device));
        HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreate(&h->ev0));
        HIPCHK(h, hipEventCreate(&h->ev1));
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, device));
        const int cus = std::max(1, prop.multiProcessorCount);
        // tuning knobs (experiments only; defaults are the measured best)
        auto env_int = [](const char* nm, int dflt) { const char* v = getenv(nm); return v ? atoi(v) : dflt; };
        h->nt = env_int("CDH_NT", 1) != 0;
        h->lt = env_int("CDH_LT", 2);
        h->ks = env_int("CDH_KS", 0);
        h->gc.mode = std::max(0, std::min(3, env_int("CDH_GRADIENT_CACHE", 1)));
        h->gc.cov = env_int("CDH_GC_COV", 1) != 0;
        h->gc.cs_enabled = env_int("CDH_COV_SOLVE", 1) != 0;
        h->gc.cs_helpers = std::max(0, std::min(kCsCrewMax, env_int("CDH_CS_CREW", 31)));
        h->gc.cs_ucap_limit = std::max(0, env_int("CDH_CS_UCAP", 0));
        h->small.enabled = env_int("CDH_SMALL_PATH", 1) != 0;
        if (const char* e = getenv("CDH_SMALL_MAX_BYTES")) h->small.max_bytes = std::atoll(e);
        h->small.zero_copy = env_int("CDH_SMALL_ZEROCOPY", 1) != 0;
        if (const char* e = getenv("CDH_SMALL_ALWAYS_BYTES")) h->small.always_bytes = std::atoll(e);
        h->gc.inject_rollback = std::max(0, env_int("CDH_GC_INJECT_ROLLBACK", 0));
        h->gc.refresh_after = std::max(1, env_int("CDH_GC_REFRESH", (int)kGcCovRefresh));
        const int step_per_cu = std::max(1, env_int("CDH_STEP_GRID_PER_CU", 8));
        const int block_per_cu = std::max(1, std::min(kBlockGridPerCU, env_int("CDH_BLOCK_GRID_PER_CU", 3)));
        const int64_t want = (h->nvec + (int64_t)kBlock * kUnroll - 1) / ((int64_t)kBlock * kUnroll);
        h->step_grid = balanced_grid(want, std::min<int64_t>((int64_t)kMaxStepGrid, (int64_t)cus * step_per_cu));
        const int64_t wantb = (h->nvec + kBlock - 1) / kBlock;
        h->block_grid = balanced_grid(wantb, (int64_t)cus * block_per_cu);
        h->cus = cus;
        h->gram32_per_cu = std::max(1, env_int("CDH_GRAM32_GRID_PER_CU", 3));
        const int64_t wantg = (h->nvec + 64 * kGramWaves - 1) / (64 * kGramWaves);
        h->gram_per_cu = std::max(1, std::min(4, env_int("CDH_GRAM_GRID_PER_CU", 2)));
        h->lt_per_cu = std::max(0, std::min(4, env_int("CDH_LT_PER_CU", 0)));
        h->gram_units = wantg;
        const size_t colbytes = (size_t)h->ld * h->esz;
        HIPCHK(h, hipMalloc(&h->X, colbytes * (size_t)p));
        HIPCHK(h, hipMalloc(&h->y, colbytes));
        HIPCHK(h, hipMalloc(&h->r, colbytes));
        HIPCHK(h, hipMemsetAsync(h->y, 0, colbytes, h->stream));
        HIPCHK(h, hipMemsetAsync(h->r, 0, colbytes, h->stream));
        if (loss == CDH_WLS) {
            HIPCHK(h, hipMalloc(&h->w, colbytes));
            HIPCHK(h, hipMemsetAsync(h->w, 0, colbytes, h->stream));
        }
        HIPCHK(h, hipMalloc(&h->beta, sizeof(double) * p));
        HIPCHK(h, hipMemsetAsync(h->beta, 0, sizeof(double) * p, h->stream));
        HIPCHK(h, hipMalloc(&h->omega, sizeof(double) * p));
        HIPCHK(h, hipMalloc(&h->d_ctrl, sizeof(Ctrl)));
        HIPCHK(h, hipMalloc(&h->d_idx, sizeof(int64_t) * h->cap));
        HIPCHK(h, hipMalloc(&h->d_hs, sizeof(double) * h->cap));
        HIPCHK(h, hipMalloc(&h->d_newval, sizeof(double) * h->cap));
        HIPCHK(h, hipMalloc(&h->d_touched, sizeof(int32_t) * h->cap));
        h->partials_doubles = std::max<size_t>({(size_t)kMaxStepGrid * kNSum,
                                                (size_t)cus * kBlockGridPerCU * BlockRec<kMaxBlockB>::N,
                                                (size_t)4096 * kColChunks * 2,
                                                (size_t)cus * 2 * GramRec<4>::N,
                                                (size_t)cus * std::max(h->gram32_per_cu, 4) * GramRec<2>::N,
                                                (size_t)cus * 4 * GramRec<1>::N});
        HIPCHK(h, hipMalloc(&h->d_partials, sizeof(double) * h->partials_doubles));
        HIPCHK(h, hipMalloc(&h->d_red, sizeof(double) * 4096));
        HIPCHK(h, hipMalloc(&h->d_colout, sizeof(double) * 2 * p));
        HIPCHK(h, hipMalloc(&h->d_sup_idx, sizeof(int64_t) * p));
        HIPCHK(h, hipMalloc(&h->d_sup_val, sizeof(double) * p));
        HIPCHK(h, hipHostMalloc(&h->h_idx, sizeof(int64_t) * h->cap));
        HIPCHK(h, hipHostMalloc(&h->h_hs, sizeof(double) * h->cap));
        HIPCHK(h, hipHostMalloc(&h->h_newval, sizeof(double) * h->cap));
        HIPCHK(h, hipHostMalloc(&h->h_touched, sizeof(int32_t) * h->cap));
        HIPCHK(h, hipHostMalloc(&h->h_red, sizeof(double) * 64));
        HIPCHK(h, hipHostMalloc(&h->h_ctrl, sizeof(Ctrl)));
        // zero the pad rows of X once (uploads / the generator only write rows < n)
        if (h->ld > h->n) HIPCHK(h, hipMemsetAsync(h->X, 0, colbytes * (size_t)p, h->stream));
        h->ctrl.lambda0 = 0.0; h->ctrl.n_total = (double)n_total; h->ctrl.maxH = 0.0;
        h->ctrl.loss = loss; h->ctrl.has_omega = 0; h->ctrl.domain_error = 0;
        CHK(upload_ctrl(h));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return CDH_OK;
    }();
    if (rc != CDH_OK) { g_create_error = h->err; free_all(h); return rc; }
    *out = h;
    return CDH_OK;
}

int32_t cdh_destroy(cdh_handle h) {
    if (!h) return CDH_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    free_all(h);
    return CDH_OK;
}

int32_t cdh_synchronize(cdh_handle h) {
    NEED_H(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

int32_t cdh_set_X_cols(cdh_handle h, int64_t j0, int64_t ncols, const void* host, int64_t ld) {
    NEED_H(h);
    if (ncols > 0) NEED_P(h, host);
    if (!h->r_pending_list.empty() || h->r_lazy) {   // r still owes updates (or its rebuild) in terms of the OLD columns
        HIPCHK(h, hipSetDevice(h->device));
        CHK(sync_r(h));
    }
    h->r_consistent = false;
    touch_r(h);
    gc_invalidate(h, true);
    h->small.G_valid = false; h->small.c_valid = false; h->small.rent_paid = 0.0;
    if (j0 < 0 || ncols < 0 || j0 + ncols > h->p || ld < h->n) return fail(h, CDH_DIM_MISMATCH, "column block outside X");
    if (ncols == 0) return CDH_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync((char*)h->X + (size_t)j0 * h->ld * h->esz, (size_t)h->ld * h->esz, host,
                               (size_t)ld * h->esz, (size_t)h->n * h->esz, (size_t)ncols,
                               hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

int32_t cdh_get_X_cols(cdh_handle h, int64_t j0, int64_t ncols, void* host, int64_t ld) {
    NEED_H(h);
    if (ncols > 0) NEED_P(h, host);
    if (j0 < 0 || ncols < 0 || j0 + ncols > h->p || ld < h->n) return fail(h, CDH_DIM_MISMATCH, "column block outside X");
    if (ncols == 0) return CDH_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy2DAsync(host, (size_t)ld * h->esz, (char*)h->X + (size_t)j0 * h->ld * h->esz,
                               (size_t)h->ld * h->esz, (size_t)h->n * h->esz, (size_t)ncols,
                               hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

int32_t cdh_set_y(cdh_handle h, const void* host_y) {
    NEED_H(h);
    NEED_P(h, host_y);
    h->r_consistent = false;
    touch_r(h);
    gc_invalidate(h, false);
    h->gc.yy_ok = false;
    h->small.c_valid = false;
    drop_r_pending(h);                 // r = copy(y) below
    h->r_lazy = false;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->y, host_y, (size_t)h->n * h->esz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->r, h->y, (size_t)h->n * h->esz, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->y_set = true;
    return CDH_OK;
}

int32_t cdh_get_y(cdh_handle h, void* host_y) {
    NEED_H(h);
    NEED_P(h, host_y);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(host_y, h->y, (size_t)h->n * h->esz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

static int32_t ensure_weights_buffer(cdh_handle h) {
    if (h->w) return CDH_OK;
    const size_t colbytes = (size_t)h->ld * h->esz;
    HIPCHK(h, hipMalloc(&h->w, colbytes));
    HIPCHK(h, hipMemsetAsync(h->w, 0, colbytes, h->stream));
    return CDH_OK;
}

int32_t cdh_set_obs_weights(cdh_handle h, const void* host_w) {
    NEED_H(h);
    NEED_P(h, host_w);
    h->r_consistent = false;
    touch_r(h);
    gc_invalidate(h, true);
    h->small.G_valid = false; h->small.c_valid = false; h->small.rent_paid = 0.0;
    if (h->loss != CDH_WLS) return fail(h, CDH_BAD_ARG, "observation weights need the CDH_WLS loss");
    HIPCHK(h, hipSetDevice(h->device));
    CHK(ensure_weights_buffer(h));
    HIPCHK(h, hipMemcpyAsync(h->w, host_w, (size_t)h->n * h->esz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->has_w = true;
    return CDH_OK;
}

int32_t cdh_set_loss(cdh_handle h, int32_t loss) {
    NEED_H(h);
    if (loss != CDH_LS && loss != CDH_SQRT && loss != CDH_WLS) return fail(h, CDH_BAD_ARG, "unknown loss");
    if (loss == h->loss) return CDH_OK;
    HIPCHK(h, hipSetDevice(h->device));
    if (loss == CDH_WLS) CHK(ensure_weights_buffer(h));
    h->loss = loss;
    h->ctrl.loss = loss;          // goes to the device with the next chunk's control block
    if (h->has_w) { h->small.G_valid = false; h->small.c_valid = false; h->small.rent_paid = 0.0; gc_invalidate(h, true); }   // X'WX is not X'X
    h->has_w = false;             // a weighted loss gets its weights from cdh_set_obs_weights
    h->r_consistent = false;
    touch_r(h);
    gc_invalidate(h, false);
    h->domain_error = false;
    return CDH_OK;
}

static int32_t cdh_generate_impl(cdh_handle h, uint64_t seed, int64_t s, double noise, double* out_beta_star) {
    if (s < 0 || s > h->p) return fail(h, CDH_BAD_ARG, "need 0 <= s <= p");
    touch_r(h);
    gc_invalidate(h, true);
    h->gc.yy_ok = false;
    h->small.G_valid = false; h->small.c_valid = false; h->small.rent_paid = 0.0;
    drop_r_pending(h);
    h->r_lazy = false;
    HIPCHK(h, hipSetDevice(h->device));
    // planted coefficients: beta*_j = z_j (1 + u_j) (benchmark/cd_bench.jl:14), stream 2
    std::vector<double> bstar((size_t)std::max<int64_t>(s, 1), 0.0);
    for (int64_t j = 0; j < s; ++j) {
        double u1, u2;
        const double z = Philox::normal(seed, (uint64_t)(2 * j), 0u, 2u);
        Philox::uniforms(seed, (uint64_t)j, 1u, 2u, u1, u2);
        bstar[(size_t)j] = z * (1.0 + u1);
    }
    if (out_beta_star) std::memcpy(out_beta_star, bstar.data(), sizeof(double) * (size_t)s);
    HIPCHK(h, hipMemcpyAsync(h->d_sup_val, bstar.data(), sizeof(double) * bstar.size(), hipMemcpyHostToDevice, h->stream));
    const int64_t pairs = (h->n + 3) / 2;
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (pairs + kBlock - 1) / kBlock));
    for (int64_t j0 = 0; j0 < h->p; j0 += 32768) {
        const int64_t nc = std::min<int64_t>(32768, h->p - j0);
        CHK(dispatch(h, [&](auto* t) {
            using T = std::remove_pointer_t<decltype(t)>;
            hipLaunchKernelGGL(k_gen_X<T>, dim3(gx, (unsigned)nc), dim3(kBlock), 0, h->stream, (T*)h->X,
                               h->ld, h->n, h->row0, j0, seed);
            return CDH_OK;
        }));
    }
    const int gy = (int)std::max<int64_t>(1, std::min<int64_t>(4096, (h->n + kBlock - 1) / kBlock));
    CHK(dispatch(h, [&](auto* t) {
        using T = std::remove_pointer_t<decltype(t)>;
        hipLaunchKernelGGL(k_gen_y<T>, dim3(gy), dim3(kBlock), 0, h->stream, (const T*)h->X, h->ld, h->n,
                           h->row0, s, h->d_sup_val, noise, seed, (T*)h->y, (T*)h->r);
        return CDH_OK;
    }));
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->y_set = true;
    h->x.clear();
    HIPCHK(h, hipMemsetAsync(h->beta, 0, sizeof(double) * h->p, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

static int32_t cdh_set_penalty_impl(cdh_handle h, double lambda0, const double* omega, int64_t n_omega) {
    HIPCHK(h, hipSetDevice(h->device));
    bool uploaded = false;
    if (omega) {
        if (n_omega != h->p) return fail(h, CDH_DIM_MISMATCH, "length(g.lambda) != numCoordinates(f)");
        // a path hands over the same weights at every lambda (lasso.jl:251: ProxL1(lambda, stdX)): what the device holds is kept
        if (!(h->omega_dev_ok && (int64_t)h->h_omega.size() == h->p && std::memcmp(h->h_omega.data(), omega, sizeof(double) * (size_t)h->p) == 0)) {
            HIPCHK(h, hipMemcpyAsync(h->omega, omega, sizeof(double) * h->p, hipMemcpyHostToDevice, h->stream));
            h->h_omega.assign(omega, omega + h->p);
            h->omega_dev_ok = true;
            uploaded = true;
        }
        h->has_omega = true;
    } else {
        h->has_omega = false;
    }
    h->ctrl.lambda0 = lambda0;
    h->ctrl.has_omega = h->has_omega ? 1 : 0;
    // the control block goes to the device at the start of every chunk (run_chunk); only the caller's
    // weight buffer, borrowed for this call, has to be consumed before returning
    if (uploaded) HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

int32_t cdh_num_coordinates(cdh_handle h, int64_t* out) {
    NEED_H(h);
    NEED_P(h, out);
    *out = h->p;
    return CDH_OK;
}

static int32_t cdh_set_iterate_impl(cdh_handle h, int64_t x_length, int64_t nnz, const int64_t* idx1, const double* val) {
    if (x_length != h->p) return fail(h, CDH_DIM_MISMATCH, "numCoordinates(x) != numCoordinates(f)");
    if (nnz < 0 || nnz > h->p) return fail(h, CDH_BAD_ARG, "nnz out of range");
    if (nnz > 0) { NEED_P(h, idx1); NEED_P(h, val); }
    for (int64_t i = 0; i < nnz; ++i)
        if (idx1[i] < 1 || idx1[i] > h->p) return fail(h, CDH_BAD_ARG, "support index out of range");
    HIPCHK(h, hipSetDevice(h->device));
    h->r_consistent = false;
    h->x.clear();
    for (int64_t i = 0; i < nnz; ++i) {
        // a stored zero keeps its slot in the reference's SparseIterate; mirror that
        if (val[i] == 0.0) { h->x.set(idx1[i] - 1, 1.0); h->x.set(idx1[i] - 1, 0.0); }
        else h->x.set(idx1[i] - 1, val[i]);
    }
    if (h->x.nnz() == 0) {   // beta = 0: nothing to upload, nothing to wait for
        HIPCHK(h, hipMemsetAsync(h->beta, 0, sizeof(double) * h->p, h->stream));
        return CDH_OK;
    }
    std::vector<double> dense((size_t)h->p, 0.0);
    for (int64_t s = 0; s < h->x.nnz(); ++s) dense[(size_t)h->x.coord(s)] = h->x.slot_value(s);
    HIPCHK(h, hipMemcpyAsync(h->beta, dense.data(), sizeof(double) * h->p, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

static int32_t cdh_initialize_impl(cdh_handle h, int64_t x_length, int64_t nnz, const int64_t* idx1, const double* val) {
    CHK(cdh_set_iterate_impl(h, x_length, nnz, idx1, val));
    return rebuild_residual(h);
}

static int32_t cdh_gradient_impl(cdh_handle h, int64_t k1, double* out) {
    NEED_P(h, out);
    if (k1 < 1 || k1 > h->p) return fail(h, CDH_BAD_ARG, "coordinate out of range");
    HIPCHK(h, hipSetDevice(h->device));
    CHK(col_dots(h, k1 - 1, 1, h->r, h->loss == CDH_WLS));
    HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_colout, sizeof(double) * 2, hipMemcpyDeviceToHost, h->stream));
    double denom = (double)h->n_total;
    double xr;
    if (h->loss == CDH_SQRT) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        xr = h->h_red[0];
        CHK(resid_moments_dev(h));
        HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        denom = std::sqrt(h->h_red[1]);
    } else {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        xr = h->h_red[0];
    }
    *out = -xr / denom;
    return CDH_OK;
}

static int32_t cdh_descend_impl(cdh_handle h, int64_t k1, double* out_h) {
    if (k1 < 1 || k1 > h->p) return fail(h, CDH_BAD_ARG, "coordinate out of range");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t k0 = k1 - 1;
    double maxH = 0.0;
    const int save_mode = h->mode;
    h->mode = CDH_SWEEP_COORD;
    int32_t rc = run_chunk(h, &k0, 1, &maxH);  // no dropzeros!: that is _cdPass!'s job
    h->mode = save_mode;
    if (rc != CDH_OK) return rc;
    if (out_h) *out_h = h->h_hs[0];
    return CDH_OK;
}

static int32_t cdh_lambda_max_impl(cdh_handle h, double* out) {
    NEED_P(h, out);
    HIPCHK(h, hipSetDevice(h->device));
    return lambda_max(h, out);
}

static int32_t cdh_pass_impl(cdh_handle h, int64_t m, const int64_t* idx1, double* out_maxH) {
    if (m < 0) return fail(h, CDH_BAD_ARG, "m < 0");
    CHK(exchange_alive(h));
    if (m > 0) NEED_P(h, idx1);
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int64_t> idx0((size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        if (idx1[i] < 1 || idx1[i] > h->p) return fail(h, CDH_BAD_ARG, "coordinate out of range");
        idx0[(size_t)i] = idx1[i] - 1;
    }
    double maxH = 0.0;
    h->gc.prep_state = 0;
    if (m > 0) CHK(run_pass(h, idx0.data(), m, &maxH, h->screening >= 2));
    else h->x.dropzeros();
    if (out_maxH) *out_maxH = maxH;
    return CDH_OK;
}

static int32_t cdh_solve_impl(cdh_handle h, const cdh_options* opt, cdh_stats* out) {
    NEED_P(h, opt);
    CHK(exchange_alive(h));
    HIPCHK(h, hipSetDevice(h->device));
    cdh_stats st{};
    if (small_applicable(h, opt)) {
        CHK(small_prepare(h));
        if (h->small.enabled) {     // (the preparation may find no room and switch the path off)
            uint64_t rng = opt->seed;
            const double lam = h->ctrl.lambda0;
            int32_t rcs = small_solve(h, opt, &lam, 1, &rng, &st, false);
            if (rcs != kSmallPrecisionLost) {
                if (out) *out = st;
                return rcs;
            }
            st = cdh_stats{};            // (r'r out of digits: the streamed kernels below take the call)
        }
    }
    cdh::VisitScheduler sched(h->p, opt->randomize != 0, opt->seed);
    const SmallRent rent(h, opt);
    int32_t rc = solve(h, opt, sched, &st);
    rent.pay(st);
    if (out) *out = st;
    return rc;
}

static int32_t cdh_coordinate_descent_impl(cdh_handle h, const cdh_options* opt, cdh_stats* out) {
    NEED_P(h, opt);
    CHK(exchange_alive(h));
    HIPCHK(h, hipSetDevice(h->device));
    cdh_stats st{};
    cdh::VisitScheduler sched(h->p, opt->randomize != 0, opt->seed);
    h->domain_error = false;
    int32_t rc = CDH_OK;
    bool small = small_applicable(h, opt, !opt->warmStart);   // a cold start is numSteps + 1 solves: it buys the Gram form outright
    if (small) { CHK(small_prepare(h)); small = h->small.enabled; }
    const SmallRent rent(h, opt);
    uint64_t rng = opt->seed;       // the one-launch solve carries the scheduler's generator state itself
    // ---- in one launch (small_solve.hpp) where the Gram form applies -- unless its r'r runs out of digits (kSmallPrecisionLost:
    // nothing of the launch is used), in which case the streamed kernels below take the call ----
    if (small && opt->warmStart) {
        // initialize!(f, x) (:21) makes r = y - X x by definition: the one-launch solve derives its gradient from that
        // identity (g = X'y - G x) and leaves the residual to be formed when somebody reads it
        const double lam = h->ctrl.lambda0;
        rc = small_solve(h, opt, &lam, 1, &rng, &st, true);
    } else if (small) {
        const double target = h->ctrl.lambda0;
        h->x.clear();                                   // fill!(x, 0)           (:25)
        HIPCHK(h, hipMemsetAsync(h->beta, 0, sizeof(double) * h->p, h->stream));
        // _findLambdaMax (:29) at r = y: max_k |X_k'y| / n / omega_k (sqrt-lasso: / ||y||) -- from the cached X'y
        double lmax = 0.0;
        const double denom = h->loss == CDH_SQRT ? std::sqrt(h->small.yy) : (double)h->n_total;
        for (int64_t k = 0; k < h->p; ++k) {
            double t = std::fabs(-h->small.h_c[(size_t)(2 * k)] / denom);
            if (h->has_omega) t /= h->h_omega[(size_t)k];
            if (t > lmax) lmax = t;
        }
        st.lambda_max = lmax;
        const double l1 = std::log(lmax), l2 = std::log(target);
        const double step = (l2 - l1) / (double)opt->numSteps;
        if (step == 0.0 || step != step)
            return fail(h, CDH_BAD_ARG, "cold start: the range log(lambda_max):step:log(lambda0) has a zero step");
        std::vector<double> grid((size_t)opt->numSteps + 1);   // the numSteps + 1 solves of (:32-36) inside one launch
        for (int64_t j = 0; j <= opt->numSteps; ++j) grid[(size_t)j] = std::exp((j == opt->numSteps) ? l2 : l1 + (double)j * step);
        rc = small_solve(h, opt, grid.data(), (int)grid.size(), &rng, &st, true);
    }
    if (small && rc == kSmallPrecisionLost) { small = false; st = cdh_stats{}; rc = CDH_OK; }
    if (small) {
        // done above
    } else if (opt->warmStart) {
        // initialize!(f, x) (:21).  Optional shortcut for warm-started paths (LassoPath): the
        // carried residual already equals y - X beta, so the rebuild only re-rounds it.
        if (!(h->reuse_residual && h->r_consistent)) CHK(rebuild_residual(h));
        rc = solve(h, opt, sched, &st);
    } else {
        const double target = h->ctrl.lambda0;          // g itself is never mutated by the reference:
        rc = [&]() -> int32_t {                         // whatever happens below, lambda0 is put back
            h->x.clear();                               // fill!(x, 0)           (:25)
            CHK(rebuild_residual(h));                   // initialize!(f, x)     (:26)
            // 51 solves on the same X follow: the gradient cache (mode 1) need not wait for evidence
            h->gc.full_seen = std::max<int64_t>(h->gc.full_seen, 1000);
            double lmax = 0.0;
            std::vector<double> dots;
            CHK(lambda_max(h, &lmax, &dots));           // _findLambdaMax        (:29)
            CHK(gc_adopt_dots(h, dots));                // ... whose pass over X is the cache's reference pass as well
            st.lambda_max = lmax;
            const double l1 = std::log(lmax), l2 = std::log(target);
            const double step = (l2 - l1) / (double)opt->numSteps;
            if (step == 0.0 || step != step)
                return fail(h, CDH_BAD_ARG, "cold start: the range log(lambda_max):step:log(lambda0) has a zero step");
            for (int64_t j = 0; j <= opt->numSteps; ++j) {  // (:32-36)
                const double l = (j == opt->numSteps) ? l2 : l1 + (double)j * step;
                h->ctrl.lambda0 = std::exp(l);
                CHK(solve(h, opt, sched, &st));
            }
            return CDH_OK;
        }();
        h->ctrl.lambda0 = target;
        if (rc == CDH_OK) {
            CHK(upload_ctrl(h));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
    }
    if (!small) rent.pay(st);
    if (out) *out = st;
    return rc;
}

int32_t cdh_get_beta(cdh_handle h, double* out_p) {
    NEED_H(h);
    NEED_P(h, out_p);
    std::memset(out_p, 0, sizeof(double) * (size_t)h->p);
    for (int64_t s = 0; s < h->x.nnz(); ++s) out_p[h->x.coord(s)] = h->x.slot_value(s);
    return CDH_OK;
}

int32_t cdh_get_support(cdh_handle h, int64_t* out_idx1, int64_t* out_nnz) {
    NEED_H(h);
    NEED_P(h, out_idx1);
    NEED_P(h, out_nnz);
    for (int64_t s = 0; s < h->x.nnz(); ++s) out_idx1[s] = h->x.coord(s) + 1;
    *out_nnz = h->x.nnz();
    return CDH_OK;
}

int32_t cdh_get_residual(cdh_handle h, void* out_n_local) {
    NEED_H(h);
    NEED_P(h, out_n_local);
    HIPCHK(h, hipSetDevice(h->device));
    CHK(sync_r(h));
    HIPCHK(h, hipMemcpyAsync(out_n_local, h->r, (size_t)h->n * h->esz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}

static int32_t cdh_col_rms_impl(cdh_handle h, double* out_p) {
    NEED_P(h, out_p);
    HIPCHK(h, hipSetDevice(h->device));
    CHK(col_dots(h, 0, h->p, h->r, false));
    std::vector<double> cd((size_t)(2 * h->p));
    HIPCHK(h, hipMemcpyAsync(cd.data(), h->d_colout, sizeof(double) * 2 * h->p, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t j = 0; j < h->p; ++j) out_p[j] = std::sqrt(cd[(size_t)(2 * j + 1)] / (double)h->n_total);
    return CDH_OK;
}

static int32_t cdh_xt_r_impl(cdh_handle h, double* out_p) {
    NEED_P(h, out_p);
    HIPCHK(h, hipSetDevice(h->device));
    CHK(col_dots(h, 0, h->p, h->r, false));
    std::vector<double> cd((size_t)(2 * h->p));
    HIPCHK(h, hipMemcpyAsync(cd.data(), h->d_colout, sizeof(double) * 2 * h->p, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int64_t j = 0; j < h->p; ++j) out_p[j] = cd[(size_t)(2 * j)];
    h->dots_stash.swap(cd); h->dots_valid = true; h->dots_w = false;
    return CDH_OK;
}

// one launch of the wide-block Gram kernel over up to 64 columns (0-based, in h->h_idx[0 .. m)): rec <- (G, c = X_S'r, q = r'r)
static int32_t gram_launch(cdh_handle h, int m, std::vector<double>& rec) {
    using R = GramRec<4>;
    HIPCHK(h, hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    // one k_gramstep launch with no pending update: r is only read
    const int G = NGgrid(h, 4);
    if ((size_t)G * R::N > h->partials_doubles) return fail(h, CDH_BAD_ARG, "partial buffer too small for this grid");
    CHK(dispatch(h, [&](auto* t) {
        using T = std::remove_pointer_t<decltype(t)>;
        hipLaunchKernelGGL((k_gramstep<T, 4, true>), dim3(G), dim3(64 * kGramWaves), 0, h->stream,
                           (const T*)h->X, h->ld, h->nvec, (const T*)nullptr, (T*)h->r, h->d_idx, h->d_hs, 0, m, 0,
                           h->d_partials);
        return CDH_OK;
    }));
    hipLaunchKernelGGL(k_gram_reduce, dim3((R::N + kReduceVals - 1) / kReduceVals), dim3(64 * kReduceWaves), 0, h->stream, h->d_partials, G, R::N, h->d_red);
    HIPCHK(h, hipGetLastError());
    CHK(allreduce(h, h->d_red, R::N));
    rec.resize((size_t)R::N);
    HIPCHK(h, hipMemcpyAsync(rec.data(), h->d_red, sizeof(double) * R::N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return CDH_OK;
}
constexpr int64_t kGramMaxCols = 4096;
static int32_t cdh_gram_impl(cdh_handle h, int64_t m, const int64_t* idx1, double* out_G, double* out_c, double* out_q) {
    if (m < 1 || m > kGramMaxCols) return fail(h, CDH_BAD_ARG, "need 1 <= m <= 4096 columns");
    NEED_P(h, idx1);
    NEED_P(h, out_G);
    for (int64_t i = 0; i < m; ++i)
        if (idx1[i] < 1 || idx1[i] > h->p) return fail(h, CDH_BAD_ARG, "coordinate out of range");
    HIPCHK(h, hipSetDevice(h->device));
    CHK(sync_r(h));
    using R = GramRec<4>;
    std::vector<double> rec;
    if (m <= 64) {
        for (int64_t i = 0; i < m; ++i) h->h_idx[i] = idx1[i] - 1;
        CHK(gram_launch(h, (int)m, rec));
        for (int64_t i = 0; i < m; ++i) {
            for (int64_t j = 0; j < m; ++j) {
                const int s = (int)std::min(i, j), l = (int)std::max(i, j);
                out_G[i * m + j] = rec[(size_t)R::g(s, l)];
            }
            if (out_c) out_c[i] = rec[(size_t)(R::OFF_C + i)];
        }
        if (out_q) *out_q = rec[(size_t)R::OFF_Q];
        return CDH_OK;
    }
    // More than one launch's worth (round 4; _findInitResiduals! takes any s, src/utils.jl:65-77): groups of 32 columns, one
    // launch per PAIR of groups -- the 64 columns of a pair give the pair's two diagonal blocks and the block between them.
    const int64_t ng = (m + 31) / 32;
    for (int64_t a = 0; a < ng; ++a)
        for (int64_t b = a + 1; b < ng; ++b) {
            const int64_t a0 = 32 * a, a1 = std::min<int64_t>(a0 + 32, m), b0 = 32 * b, b1 = std::min<int64_t>(b0 + 32, m);
            const int na = (int)(a1 - a0), nb = (int)(b1 - b0);
            for (int i = 0; i < na; ++i) h->h_idx[i] = idx1[a0 + i] - 1;
            for (int i = 0; i < nb; ++i) h->h_idx[na + i] = idx1[b0 + i] - 1;
            CHK(gram_launch(h, na + nb, rec));
            // column -> position inside the launch; every block and dot the launch holds is written (a diagonal block comes out
            // of every pair its group is in, each time from the same sums over the same rows)
            auto col_of = [&](int pos) { return pos < na ? a0 + pos : b0 + (pos - na); };
            for (int pi = 0; pi < na + nb; ++pi) {
                const int64_t i = col_of(pi);
                for (int pj = 0; pj < na + nb; ++pj)
                    out_G[i * m + col_of(pj)] = rec[(size_t)R::g(std::min(pi, pj), std::max(pi, pj))];
                if (out_c) out_c[i] = rec[(size_t)(R::OFF_C + pi)];
            }
            if (out_q) *out_q = rec[(size_t)R::OFF_Q];
        }
    return CDH_OK;
}

// X_S'r (X_S'Wr for the weighted loss) for a list of columns: the refinement step of the screening init reads the normal
// equations' residual off it (src/utils.jl:65-77 solves Xs \ y by QR; here: the Gram block plus refinement)
static int32_t cdh_xt_r_cols_impl(cdh_handle h, int64_t m, const int64_t* idx1, double* out_m) {
    if (m < 1 || m > h->cap) return fail(h, CDH_BAD_ARG, "need 1 <= m <= max(p, 4096) columns");
    NEED_P(h, idx1);
    NEED_P(h, out_m);
    for (int64_t i = 0; i < m; ++i)
        if (idx1[i] < 1 || idx1[i] > h->p) return fail(h, CDH_BAD_ARG, "coordinate out of range");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<double> cd(2 * (size_t)std::min<int64_t>(m, h->p));
    for (int64_t o = 0; o < m; o += h->p) {          // d_colout holds 2p values
        const int64_t mm = std::min<int64_t>(h->p, m - o);
        for (int64_t i = 0; i < mm; ++i) h->h_idx[i] = idx1[o + i] - 1;
        HIPCHK(h, hipMemcpyAsync(h->d_idx, h->h_idx, sizeof(int64_t) * (size_t)mm, hipMemcpyHostToDevice, h->stream));
        CHK(col_dots(h, 0, mm, h->r, h->loss == CDH_WLS, h->d_idx));
        HIPCHK(h, hipMemcpyAsync(cd.data(), h->d_colout, sizeof(double) * 2 * (size_t)mm, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int64_t i = 0; i < mm; ++i) out_m[o + i] = cd[(size_t)(2 * i)];
    }
    return CDH_OK;
}

// std(f.r) as Statistics.std computes it (two passes: the mean, then the centred sum of squares; Bessel-corrected) --
// lasso.jl:37,52,81,97,143 -- without the cancellation of the one-pass form when the mean is large against the spread
static int32_t cdh_resid_std_impl(cdh_handle h, double* out_std, double* out_mean) {
    NEED_P(h, out_std);
    HIPCHK(h, hipSetDevice(h->device));
    CHK(resid_moments_dev(h));
    HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const double n = (double)h->n_total, mean = h->h_red[0] / n;
    double ss = h->h_red[1], s1 = h->h_red[0];
    if (mean != 0.0 && mean == mean) {
        CHK(resid_moments_dev(h, nullptr, mean));
        HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        s1 = h->h_red[0]; ss = h->h_red[1];         // of the centred values: s1 is rounding-sized
    } else {
        s1 = 0.0;
    }
    *out_std = std::sqrt(std::max(ss - s1 * s1 / n, 0.0) / (n - 1.0));
    if (out_mean) *out_mean = mean;
    return CDH_OK;
}

int32_t cdh_set_reuse_residual(cdh_handle h, int32_t on) {
    NEED_H(h);
    h->reuse_residual = on != 0;
    return CDH_OK;
}

int32_t cdh_resid_moments(cdh_handle h, double* out_sum, double* out_sumsq) {
    NEED_H(h);
    HIPCHK(h, hipSetDevice(h->device));
    CHK(resid_moments_dev(h));
    HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (out_sum) *out_sum = h->h_red[0];
    if (out_sumsq) *out_sumsq = h->h_red[1];
    return CDH_OK;
}

static int32_t cdh_objective_impl(cdh_handle h, double* out) {
    NEED_P(h, out);
    HIPCHK(h, hipSetDevice(h->device));
    CHK(resid_moments_dev(h));
    HIPCHK(h, hipMemcpyAsync(h->h_red, h->d_red, sizeof(double) * 4, hipMemcpyDeviceToHost, h->stream));
    const std::vector<double>& om = h->h_omega;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double pen = 0.0;
    for (int64_t s = 0; s < h->x.nnz(); ++s)
        pen += std::fabs(h->x.slot_value(s)) * (h->has_omega ? om[(size_t)h->x.coord(s)] : 1.0);
    pen *= h->ctrl.lambda0;
    const double ss = h->loss == CDH_WLS ? h->h_red[2] : h->h_red[1];
    *out = (h->loss == CDH_SQRT ? std::sqrt(ss) : ss / (2.0 * (double)h->n_total)) + pen;
    return CDH_OK;
}

int32_t cdh_set_sweep_mode(cdh_handle h, int32_t mode, int32_t block) {
    NEED_H(h);
    if (mode != CDH_SWEEP_COORD && mode != CDH_SWEEP_BLOCK) return fail(h, CDH_BAD_ARG, "unknown sweep mode");
    if (mode == CDH_SWEEP_BLOCK && block != 2 && block != 4 && block != 8 && block != 16 && block != 32 && block != 64)
        return fail(h, CDH_BAD_ARG, "block size must be 2, 4, 8, 16, 32 or 64");
    h->mode = mode;
    if (mode == CDH_SWEEP_BLOCK) { h->blockB = block; h->width_default = false; }   // the caller's choice: the caller keeps the ranks alike
    return CDH_OK;
}

int32_t cdh_set_screening(cdh_handle h, int32_t on) {
    NEED_H(h);
    if (on < 0 || on > 2) return fail(h, CDH_BAD_ARG, "screening: 0 = never, 1 = solves, 2 = cdh_pass as well");
    h->screening = on;
    return CDH_OK;
}

int32_t cdh_set_gradient_cache(cdh_handle h, int32_t mode) {
    NEED_H(h);
    if (mode < 0 || mode > 3) return fail(h, CDH_BAD_ARG, "gradient cache: 0 = off, 1 = rent-or-buy, 2 = from the first full pass, 3 = 2 without the size guard");
    if (mode == 0) gc_invalidate(h, true);
    h->gc.mode = mode;
    h->gc.cooldown = 0; h->gc.backoff = 1;
    return CDH_OK;
}

int32_t cdh_set_onchip_solve(cdh_handle h, int32_t on) {
    NEED_H(h);
    h->small.enabled = on != 0;
    return CDH_OK;
}

int32_t cdh_onchip_stats(cdh_handle h, int64_t* out2) {
    NEED_H(h);
    NEED_P(h, out2);
    out2[0] = h->small.n_solves; out2[1] = h->small.n_gram;
    return CDH_OK;
}

int32_t cdh_onchip_last(cdh_handle h, int64_t* out3) {
    NEED_H(h);
    NEED_P(h, out3);
    out3[0] = out3[1] = out3[2] = 0;
    if (h->small.h_ctl) { out3[0] = h->small.h_ctl->steps; out3[1] = (int64_t)h->small.h_ctl->cycles; out3[2] = (int64_t)h->small.h_ctl->ticks; }
    return CDH_OK;
}

int32_t cdh_get_gradient_cache(cdh_handle h, int32_t* out_mode) {
    NEED_H(h);
    NEED_P(h, out_mode);
    *out_mode = h->gc.mode;
    return CDH_OK;
}

static int32_t cdh_cache_drift_impl(cdh_handle h, int32_t rereference_now, double* out3) {
    NEED_P(h, out3);
    GradCache& c = h->gc;
    if (rereference_now && c.valid && gc_applicable(h)) {
        HIPCHK(h, hipSetDevice(h->device));
        CHK(gc_rereference(h));
    }
    out3[0] = c.drift_last; out3[1] = c.drift_max; out3[2] = (double)c.n_drift;
    return CDH_OK;
}

int32_t cdh_cache_stats(cdh_handle h, int64_t* out10) {
    NEED_H(h);
    NEED_P(h, out10);
    const GradCache& c = h->gc;
    out10[0] = c.n_passes; out10[1] = c.n_certified; out10[2] = c.n_exact;
    out10[3] = c.n_validate; out10[4] = c.n_batches; out10[5] = c.n_columns;
    out10[6] = c.n_cov; out10[7] = c.n_reconcile; out10[8] = c.n_rollbacks;
    out10[9] = c.n_dev_passes;
    return CDH_OK;
}

int32_t cdh_cache_gram_column(cdh_handle h, int64_t k1, double* out_p, double* out_eps) {
    NEED_H(h);
    NEED_P(h, out_p);
    if (k1 < 1 || k1 > h->p) return fail(h, CDH_BAD_ARG, "coordinate out of range");
    GradCache& c = h->gc;
    if (c.slot.empty() || c.slot[(size_t)(k1 - 1)] < 0 || (size_t)c.slot[(size_t)(k1 - 1)] >= c.G.size())
        return fail(h, CDH_BAD_ARG, "the gradient cache holds no Gram column for this coordinate");
    HIPCHK(h, hipSetDevice(h->device));
    CHK(gc_host_column(h, c.slot[(size_t)(k1 - 1)]));
    const std::vector<double>& col = c.G[(size_t)c.slot[(size_t)(k1 - 1)]];
    std::memcpy(out_p, col.data(), sizeof(double) * (size_t)h->p);
    if (out_eps) *out_eps = h->dtype == CDH_F32 ? 5.9604644775390625e-8 * kCrossF32EpsFactor / std::sqrt((double)h->n_total) : 0.0;
    return CDH_OK;
}

int32_t cdh_set_device_loop(cdh_handle h, int32_t on) {
    NEED_H(h);
    h->gc.cs_enabled = on != 0;
    if (on == 2) h->gc.cs_helpers = 0;                 // the loop without its helper workgroups
    else if (on > 2) h->gc.cs_helpers = std::min(kCsCrewMax, on);
    return CDH_OK;
}

int32_t cdh_device_loop_stats(cdh_handle h, int64_t* out12) {
    NEED_H(h);
    NEED_P(h, out12);
    out12[0] = h->gc.n_cs_launches; out12[1] = h->gc.n_cs_passes; out12[2] = h->gc.n_cs_folds; out12[3] = h->gc.n_cs_exact;
    for (int i = 0; i < 8; ++i) out12[4 + i] = h->gc.cs_ticks[i];
    if (getenv("CDH_COV_SOLVE_CLOCK")) fprintf(stderr, "k_cov_solve: %lld cycles in %lld ticks of 10 ns: %.3f GHz\n", (long long)h->gc.cs_cycles, (long long)h->gc.cs_ticks_total, h->gc.cs_ticks_total ? (double)h->gc.cs_cycles / (double)h->gc.cs_ticks_total * 0.1 : 0.0);
    return CDH_OK;
}

int32_t cdh_device_loop_table(cdh_handle h, int64_t* out6 /* eight values by now */) {
    NEED_H(h);
    NEED_P(h, out6);
    out6[0] = h->gc.n_cs_table_passes; out6[1] = h->gc.n_cs_table_rows; out6[2] = h->gc.cs_table_reset ? 0 : h->gc.cs_ncid; out6[3] = kCsTableCap;
    out6[4] = h->gc.n_forced_rounds; out6[5] = h->gc.n_cs_forced_rounds; out6[6] = h->gc.n_cs_crew_passes; out6[7] = h->gc.n_cs_crew_jobs;
    return CDH_OK;
}

int32_t cdh_set_use_graph(cdh_handle h, int32_t on) {
    NEED_H(h);
    h->use_graph = on != 0;
    if (on) h->graph_broken = false;   // asking again retries a capture that failed earlier
    return CDH_OK;
}

int32_t cdh_comm_unique_id(void* out_128_bytes) {
    NEED_P(nullptr, out_128_bytes);
    std::string err;
    if (!load_rccl(err)) { g_create_error = err; return CDH_RCCL_ERROR; }
    UniqueId id;
    std::memset(&id, 0, sizeof id);
    if (g_rccl.GetUniqueId(&id) != 0) { g_create_error = "ncclGetUniqueId failed"; return CDH_RCCL_ERROR; }
    std::memcpy(out_128_bytes, &id, sizeof id);
    return CDH_OK;
}

int32_t cdh_comm_init(cdh_handle h, const void* id_128_bytes, int32_t rank, int32_t nranks) {
    NEED_H(h);
    NEED_P(h, id_128_bytes);
    if (h->host_fn) return fail(h, CDH_BAD_ARG, "the handle already exchanges through a host transport");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, CDH_BAD_ARG, "bad rank / nranks");
    // a 1-rank communicator is only built when asked for (exercises the RCCL path on one GPU)
    if (nranks == 1 && !getenv("CDH_FORCE_RCCL")) { h->rank = 0; h->nranks = 1; return CDH_OK; }
    std::string err;
    if (!load_rccl(err)) { h->err = err; return CDH_RCCL_ERROR; }
    HIPCHK(h, hipSetDevice(h->device));
    UniqueId id;
    std::memcpy(&id, id_128_bytes, sizeof id);
    int rc = ((comm_init_rank_fn)g_rccl.CommInitRank)(&h->comm, nranks, id, rank);
    if (rc != 0) {
        h->err = std::string("ncclCommInitRank failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return CDH_RCCL_ERROR;
    }
    h->rank = rank; h->nranks = nranks;
    h->lost_exchange = false;          // an exchange is installed again (after a drop: cdh_comm_drop)
    return CDH_OK;
}

int32_t cdh_comm_drop(cdh_handle h) {
    NEED_H(h);
    if (!h->comm) return CDH_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto& e : h->graphs) (void)hipGraphExecDestroy(e.exec);   // captured passes hold the communicator's all-reduces
    h->graphs.clear();
    if (g_rccl.CommAbort) g_rccl.CommAbort(h->comm); else if (g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    h->comm = nullptr;
    if (h->nranks > 1 && !h->p2p_on && !h->host_fn) h->lost_exchange = true;
    return CDH_OK;
}

// ---- optional direct exchange (p2p_exchange.hpp) -------------------------------------------------
int32_t cdh_p2p_local_handle(cdh_handle h, void* out_64_bytes) {
    NEED_H(h);
    NEED_P(h, out_64_bytes);
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size is part of the ABI");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->p2p_inbox) {
        // polled by this GPU while peers write it: must not be served from a stale L2 line
        void* q = nullptr;
        if (hipExtMallocWithFlags(&q, kP2PInboxBytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(h, hipExtMallocWithFlags(&q, kP2PInboxBytes, hipDeviceMallocFinegrained));
        }
        h->p2p_inbox = (unsigned long long*)q;
        HIPCHK(h, hipMemset(h->p2p_inbox, 0, kP2PInboxBytes));  // tag 0 = never written; epochs start at 1
        HIPCHK(h, hipMalloc((void**)&h->d_p2p_base, sizeof(unsigned)));
        HIPCHK(h, hipMemset(h->d_p2p_base, 0, sizeof(unsigned)));
        HIPCHK(h, hipHostMalloc((void**)&h->p2p_timeout, sizeof(int), hipHostMallocDefault));
        *h->p2p_timeout = 0;
        HIPCHK(h, hipDeviceSynchronize());
    }
    hipIpcMemHandle_t ipc;
    HIPCHK(h, hipIpcGetMemHandle(&ipc, h->p2p_inbox));
    std::memcpy(out_64_bytes, &ipc, sizeof ipc);
    return CDH_OK;
}

int32_t cdh_p2p_connect(cdh_handle h, const void* handles_64_bytes_each, int32_t rank, int32_t nranks) {
    NEED_H(h);
    NEED_P(h, handles_64_bytes_each);
    if (h->host_fn) return fail(h, CDH_BAD_ARG, "the handle already exchanges through a host transport");
    if (nranks < 1 || nranks > kP2PMaxRanks || rank < 0 || rank >= nranks)
        return fail(h, CDH_BAD_ARG, "p2p exchange: bad rank / nranks (at most 8 ranks)");
    if (!h->p2p_inbox) return fail(h, CDH_BAD_ARG, "cdh_p2p_local_handle must be called first");
    if (h->comm && (rank != h->rank || nranks != h->nranks))
        return fail(h, CDH_BAD_ARG, "p2p exchange: rank / nranks differ from the RCCL communicator's");
    if (h->p2p_ranks) return fail(h, CDH_BAD_ARG, "p2p exchange is already connected");
    HIPCHK(h, hipSetDevice(h->device));
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) { h->p2p_peers.inbox[q] = h->p2p_inbox; continue; }
        hipIpcMemHandle_t ipc;
        std::memcpy(&ipc, (const char*)handles_64_bytes_each + 64 * (size_t)q, sizeof ipc);
        void* mapped = nullptr;
        HIPCHK(h, hipIpcOpenMemHandle(&mapped, ipc, hipIpcMemLazyEnablePeerAccess));
        h->p2p_mapped.push_back(mapped);
        h->p2p_peers.inbox[q] = (unsigned long long*)mapped;
    }
    h->rank = rank; h->nranks = nranks; h->p2p_ranks = nranks;
    if (const char* e = getenv("CDH_P2P_SPIN_LIMIT")) h->p2p_spin_limit = (unsigned)std::max<long long>(1, atoll(e));
    return CDH_OK;
}

int32_t cdh_p2p_enable(cdh_handle h, int32_t on) {
    NEED_H(h);
    if (on && !h->p2p_ranks) return fail(h, CDH_BAD_ARG, "p2p exchange is not connected");
    if (on && *(volatile int*)h->p2p_timeout)
        return fail(h, CDH_RCCL_ERROR, "p2p exchange timed out earlier on this handle; it stays off");
    h->p2p_on = on != 0;
    if (h->p2p_on) h->lost_exchange = false;   // the direct exchange serves the shard again
    return CDH_OK;
}

int32_t cdh_set_host_exchange(cdh_handle h, cdh_host_allreduce_fn fn, void* user, int32_t rank, int32_t nranks) {
    NEED_H(h);
    if (!fn) {
        // a shard of a multi-rank problem must not quietly fall into the single-process kernels on its local rows
        if (h->host_fn && h->nranks > 1 && !h->comm && !h->p2p_ranks) h->lost_exchange = true;
        h->host_fn = nullptr; h->host_user = nullptr;
        return CDH_OK;
    }
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, CDH_BAD_ARG, "bad rank / nranks");
    if (h->comm || h->p2p_ranks) return fail(h, CDH_BAD_ARG, "the handle already has an exchange (RCCL / direct)");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->h_xchg) {   // the longest record is the 2p column dots of _findLambdaMax / _stdX!
        h->h_xchg_doubles = (size_t)std::max<int64_t>(4096, 2 * h->p);
        HIPCHK(h, hipHostMalloc((void**)&h->h_xchg, sizeof(double) * h->h_xchg_doubles));
    }
    h->host_fn = fn; h->host_user = user; h->rank = rank; h->nranks = nranks;
    h->lost_exchange = false;
    return CDH_OK;
}

int32_t cdh_exchange_stats(cdh_handle h, int64_t* out_rccl_calls, int64_t* out_p2p_calls, int64_t* out_host_calls,
                           int32_t* out_nranks) {
    NEED_H(h);
    if (out_rccl_calls) *out_rccl_calls = h->n_rccl_calls;
    if (out_p2p_calls) *out_p2p_calls = h->n_p2p_calls;
    if (out_host_calls) *out_host_calls = h->n_host_calls;
    if (out_nranks) {
        int n = 1;
        if (h->comm) {   // what the communicator itself says, not what we asked for
            n = h->nranks;
            if (g_rccl.CommCount && g_rccl.CommCount(h->comm, &n) != 0) n = -1;
        } else if (h->p2p_on) {
            n = h->p2p_ranks;
        } else if (h->host_fn) {
            n = h->nranks;
        }
        *out_nranks = n;
    }
    return CDH_OK;
}

int32_t cdh_exchange_probe(cdh_handle h, double* inout, int64_t count) {
    NEED_H(h);
    if (count < 0 || count > 4096 || (count > 0 && !inout)) return fail(h, CDH_BAD_ARG, "probe: 0 <= count <= 4096");
    if (!h->d_red) return fail(h, CDH_BAD_ARG, "probe: handle has no data yet");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d_red, inout, sizeof(double) * count, hipMemcpyHostToDevice, h->stream));
    CHK(allreduce(h, h->d_red, (size_t)count));
    HIPCHK(h, hipMemcpyAsync(inout, h->d_red, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return p2p_check(h);
}

int32_t cdh_exchange_latency(cdh_handle h, int64_t count, int32_t iters, double* out_us) {
    NEED_H(h);
    if (count < 1 || count > 4096 || iters < 1 || iters > 100000 || !out_us) return fail(h, CDH_BAD_ARG, "latency probe: 1 <= count <= 4096, 1 <= iters <= 100000");
    if (!h->d_red) return fail(h, CDH_BAD_ARG, "probe: handle has no data yet");
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->ev0) { HIPCHK(h, hipEventCreate(&h->ev0)); HIPCHK(h, hipEventCreate(&h->ev1)); }
    HIPCHK(h, hipMemsetAsync(h->d_red, 0, sizeof(double) * (size_t)count, h->stream));
    for (int i = 0; i < 3; ++i) CHK(allreduce(h, h->d_red, (size_t)count));   // warm the path
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < iters; ++i) CHK(allreduce(h, h->d_red, (size_t)count));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(p2p_check(h));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *out_us = (double)ms * 1e3 / iters;
    return CDH_OK;
}

int32_t cdh_profile_begin(cdh_handle h) {
    NEED_H(h);
    h->prof = true; h->prof_ms = 0.0; h->prof_bytes = 0.0; h->prof_launches = 0;
    return CDH_OK;
}

int32_t cdh_profile_end(cdh_handle h, double* out_ms, int64_t* out_launches, double* out_algorithmic_bytes) {
    NEED_H(h);
    h->prof = false;
    if (out_ms) *out_ms = h->prof_ms;
    if (out_launches) *out_launches = h->prof_launches;
    if (out_algorithmic_bytes) *out_algorithmic_bytes = h->prof_bytes;
    return CDH_OK;
}

// ---- exception firewall: nothing may unwind across the C ABI ----------------------------------
#define CDH_CATCH(h)                                                              \
    catch (const std::bad_alloc&) { return fail((h), CDH_OOM, "host allocation failed"); } \
    catch (const std::exception& e) { return fail((h), CDH_BAD_ARG, e.what()); }  \
    catch (...) { return fail((h), CDH_BAD_ARG, "unknown C++ exception"); }

int32_t cdh_pass(cdh_handle h, int64_t m, const int64_t* idx1, double* out_maxH) {
    NEED_H(h);
    try { return cdh_pass_impl(h, m, idx1, out_maxH); }
    CDH_CATCH(h)
}

int32_t cdh_solve(cdh_handle h, const cdh_options* opt, cdh_stats* out) {
    NEED_H(h);
    try { return cdh_solve_impl(h, opt, out); }
    CDH_CATCH(h)
}

int32_t cdh_coordinate_descent(cdh_handle h, const cdh_options* opt, cdh_stats* out) {
    NEED_H(h);
    try { return cdh_coordinate_descent_impl(h, opt, out); }
    CDH_CATCH(h)
}

int32_t cdh_initialize(cdh_handle h, int64_t x_length, int64_t nnz, const int64_t* idx1, const double* val) {
    NEED_H(h);
    try { return cdh_initialize_impl(h, x_length, nnz, idx1, val); }
    CDH_CATCH(h)
}

int32_t cdh_set_iterate(cdh_handle h, int64_t x_length, int64_t nnz, const int64_t* idx1, const double* val) {
    NEED_H(h);
    try { return cdh_set_iterate_impl(h, x_length, nnz, idx1, val); }
    CDH_CATCH(h)
}

int32_t cdh_lambda_max(cdh_handle h, double* out) {
    NEED_H(h);
    try { return cdh_lambda_max_impl(h, out); }
    CDH_CATCH(h)
}

int32_t cdh_col_rms(cdh_handle h, double* out_p) {
    NEED_H(h);
    try { return cdh_col_rms_impl(h, out_p); }
    CDH_CATCH(h)
}

int32_t cdh_xt_r(cdh_handle h, double* out_p) {
    NEED_H(h);
    try { return cdh_xt_r_impl(h, out_p); }
    CDH_CATCH(h)
}

int32_t cdh_gram(cdh_handle h, int64_t m, const int64_t* idx1, double* out_G, double* out_c, double* out_q) {
    NEED_H(h);
    try { return cdh_gram_impl(h, m, idx1, out_G, out_c, out_q); }
    CDH_CATCH(h)
}

int32_t cdh_xt_r_cols(cdh_handle h, int64_t m, const int64_t* idx1, double* out_m) {
    NEED_H(h);
    try { return cdh_xt_r_cols_impl(h, m, idx1, out_m); }
    CDH_CATCH(h)
}

int32_t cdh_resid_std(cdh_handle h, double* out_std, double* out_mean) {
    NEED_H(h);
    try { return cdh_resid_std_impl(h, out_std, out_mean); }
    CDH_CATCH(h)
}

int32_t cdh_generate(cdh_handle h, uint64_t seed, int64_t s, double noise, double* out_beta_star) {
    NEED_H(h);
    try { return cdh_generate_impl(h, seed, s, noise, out_beta_star); }
    CDH_CATCH(h)
}

int32_t cdh_objective(cdh_handle h, double* out) {
    NEED_H(h);
    try { return cdh_objective_impl(h, out); }
    CDH_CATCH(h)
}

int32_t cdh_gradient(cdh_handle h, int64_t k1, double* out) {
    NEED_H(h);
    try { return cdh_gradient_impl(h, k1, out); }
    CDH_CATCH(h)
}

int32_t cdh_descend(cdh_handle h, int64_t k1, double* out_h) {
    NEED_H(h);
    try { return cdh_descend_impl(h, k1, out_h); }
    CDH_CATCH(h)
}

int32_t cdh_cache_drift(cdh_handle h, int32_t rereference_now, double* out3) {
    NEED_H(h);
    try { return cdh_cache_drift_impl(h, rereference_now, out3); }
    CDH_CATCH(h)
}

int32_t cdh_set_penalty(cdh_handle h, double lambda0, const double* omega, int64_t n_omega) {
    NEED_H(h);
    try { return cdh_set_penalty_impl(h, lambda0, omega, n_omega); }
    CDH_CATCH(h)
}

}  // extern "C"
