// cov_solve_types.hpp -- what the device-resident pass loop (cov_solve.hpp) exchanges with the host: declared ahead of
// the handle, which keeps these blocks.
#pragma once
#include <stdint.h>

enum { kCsConverged = 0, kCsMaxIter = 1, kCsNeedColumns = 2, kCsRollback = 3, kCsBusy = 4, kCsRefresh = 5, kCsOutgrown = 6, kCsNeedQ = 7, kCsNeedFold = 8, kCsHostFull = 9, kCsCrewLost = 10, kCsNeedBig = 11 };

struct CovSolveCtl {
    // in
    double lambda0, n_total, optTol, cert_abs;
    double q_floor;              // sqrt-lasso: r'r below this has run out of digits (the host sums it from r): kCsNeedQ
    int64_t max_passes;          // passes this launch may run (maxIter - those already done)
    int64_t cov_budget;          // covariance-form visits this launch may make before g is due to be re-read from X
    int32_t loss, has_omega, randomize, nnz_limit /* support size beyond which the cache stands aside */;
    int32_t busy_limit, inject_every;
    int32_t fold_limit;          // pending moves beyond which a fold is the host's (p x moves gathers: one CU against the chip): kCsNeedFold
    int32_t tcap;                // rows of the tracked coordinates' Gram TABLE in device memory (0: none)
    int32_t full_cap;            // supports beyond this have their FULL passes run by the host (p x moves work per pass: the chip's, not one CU's)
    int32_t ucap_limit;          // > 0: visit lists longer than this leave the LDS block even where it would hold them (tests: small problems then exercise the table and the helpers)
    // in / out
    uint64_t rng;
    double q;                    // r'r (sqrt-lasso)
    int32_t nnz, prev_conv, conv, inject_count;
    int32_t ncid, tepoch;        // the table: coordinates it holds (kept from launch to launch), the epoch its carried gradients belong to
    // out
    int32_t status, n_list /* kCsNeedColumns / kCsBusy: coordinates that want a Gram column (out_list) */, n_moved, domain_error;
    int64_t passes, full_passes, visits, cov_visits, cov_visits_full, settled, folds, exact_rechecks, table_passes, table_rows, forced_rounds, crew_passes, crew_jobs;
    double maxH;
    int64_t cycles, ticks_total; // shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) the launch ran for
    int64_t ticks[8];            // 100 MHz ticks the kernel spent per phase (list, scan, exact gradients, visits, re-check, accept, bookkeeping, dropzeros! + the rest)
};

// The crew: helper workgroups of the same launch that keep g = X'r current for ALL p coordinates while workgroup 0 visits
// (cov_solve.hpp, "crew passes").  Workgroup 0 posts one job per block of visits; the helpers take them in order.
constexpr int kCsCrewMax = 64;
constexpr int kCsCrewRing = 16;        // jobs in flight: a full pass over 1024 visits posts without ever waiting
enum { kCrewUpdate = 1, kCrewSnapshot = 2, kCrewRestore = 3, kCrewFold = 4 };
struct CsCrewJob {
    int32_t kind, nmove, j0, nb, chk, last_block, cnt, pad;
    uint32_t pad1, pad2;
    double q_start;
    double h[64], q[64];
    int64_t off[64];
    int32_t pos[64];
};
struct CsCrew {
    uint32_t posted, exit_flag, bad, lost;
    uint32_t done[kCsCrewMax];
    CsCrewJob ring[kCsCrewRing];
};

struct CovSolveBufs {
    int64_t p;
    double* g; const double* Gcols; const int32_t* slot; const double* a; const double* colmax; const double* omega;
    double* beta;
    double *gx, *bfold, *bsnap, *hs, *newval, *qs, *tv, *pendv, *ubeta, *uom, *ugx;
    int64_t *uk, *poff, *voff, *uprev, *iota;
    int32_t *touched, *s2i, *i2s, *list, *vb, *moved, *holes, *fills, *gxp, *upos, *aidx, *occ;
    uint8_t *setflag, *inmoved, *forced;
    // the Gram table of large supports (cov_solve.hpp, "table mode"): Gc[c][d] = X_k(c)' X_k(d) by table id, tcap x tcap, symmetric
    double *Gc, *gxc;
    int64_t* cidk;
    int32_t *cidof, *ucid, *gxe, *newc;
    // crew passes
    CsCrew* crew;
    double* g_snap;
    const int32_t* in_sup;                   // the support in slot order (pinned host memory, read once)
    int32_t *out_sup_idx, *out_moved_idx, *out_list;   // pinned host memory, written once at the end
    double *out_sup_val, *out_moved_val;
};

