// p2p_exchange.hpp -- one-shot all-reduce of a short fp64 record over the xGMI full mesh
// (SURVEY.md section 5): every GPU stores its record into a slot of every peer's inbox, polls its
// own inbox until all sources have arrived, and sums the sources in rank order -- one hop of
// latency, all links in parallel, and bit-identical sums on every rank.  OPT-IN (CDH exchange
// "p2p"): the default exchange is RCCL.  Validated in this pipeline only with two processes on ONE
// GPU (IPC plumbing, epochs, double buffering); cross-GPU coherence rests on the protocol below.
//
// Protocol (the LL idea: data and flag travel in ONE naturally aligned 8-byte store, so no fence
// and no separate flag can be overtaken): a double is sent as two words {epoch:32 | lo:32} and
// {epoch:32 | hi:32}, each by one system-scope relaxed atomic store; the reader polls each word with
// system-scope relaxed atomic loads until its tag equals the epoch.  Inboxes are allocated uncached
// (hipDeviceMallocUncached) so a poll is never served from a stale L2 line.  Two slots alternate by
// epoch parity: a rank can start exchange e+1 while a slow peer still reads e, and nobody can reach
// e+2 (same slot as e) before every rank has finished reading e, because e+1 needs every rank's
// contribution, which a rank sends only after it has finished e.  Every value is handled by one
// thread end to end, so the kernel needs no inter-block synchronisation; spins are bounded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cdk {

constexpr int kP2PMaxRanks = 8;
constexpr int kP2PMaxCount = 2688;                 // >= the widest block record (2625 doubles)
constexpr int kP2PWords = 2 * kP2PMaxCount;        // 8-byte words per (slot, source)
constexpr size_t kP2PInboxBytes = (size_t)2 * kP2PMaxRanks * kP2PWords * sizeof(unsigned long long);
constexpr unsigned kP2PSpinLimit = 2u * 1000u * 1000u;  // default bound: a few seconds (ranks run in lockstep; env CDH_P2P_SPIN_LIMIT)

struct P2PPeers { unsigned long long* inbox[kP2PMaxRanks]; };

// One value through the exchange: store it (epoch-tagged halves) into slot [rank] of every inbox, wait for
// every source in the own inbox, return the rank-ordered sum.  Called by one thread per value.
__device__ __forceinline__ double p2p_exchange_value(double mine, int v, const P2PPeers& peers, int rank,
                                                     int nranks, unsigned epoch, unsigned limit,
                                                     int* timeout_flag) {
    const size_t slot_off = (size_t)(epoch & 1u) * kP2PMaxRanks * kP2PWords;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(mine);
    const unsigned long long tag = (unsigned long long)epoch << 32;
    const unsigned long long w0 = tag | (bits & 0xffffffffull), w1 = tag | (bits >> 32);
    for (int q = 0; q < nranks; ++q) {   // my words into slot [rank] of every peer's inbox
        if (q == rank) continue;         // my own contribution never goes through memory
        unsigned long long* dst = peers.inbox[q] + slot_off + (size_t)rank * kP2PWords + 2 * (size_t)v;
        __hip_atomic_store(dst, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // poll every source at once (one round trip when the data is there: a source-by-source wait
    // would chain nranks uncached-load latencies), re-reading only the slots still behind
    const unsigned long long* src0 = peers.inbox[rank] + slot_off + 2 * (size_t)v;
    unsigned long long a[kP2PMaxRanks] = {}, b[kP2PMaxRanks] = {};
    unsigned pending = ((1u << nranks) - 1u) & ~(1u << rank), spins = 0;
    while (pending) {
#pragma unroll
        for (int s = 0; s < kP2PMaxRanks; ++s)
            if ((pending >> s) & 1u) {
                a[s] = __hip_atomic_load(src0 + (size_t)s * kP2PWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                b[s] = __hip_atomic_load(src0 + (size_t)s * kP2PWords + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
        for (int s = 0; s < kP2PMaxRanks; ++s)
            if (((pending >> s) & 1u) && (unsigned)(a[s] >> 32) == epoch && (unsigned)(b[s] >> 32) == epoch)
                pending &= ~(1u << s);
        if (pending) {
            if (++spins > limit) { *(volatile int*)timeout_flag = 1; break; }   // never hang the GPU
            __builtin_amdgcn_s_sleep(8);
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int s = 0; s < kP2PMaxRanks; ++s)   // rank order: the same sum on every rank
        if (s < nranks) sum += (s == rank) ? mine : __longlong_as_double((long long)((a[s] & 0xffffffffull) | (b[s] << 32)));
    return sum;
}

// what a kernel needs to take part in exchange number `epoch`
struct P2PCall {
    P2PPeers peers;
    int rank, nranks;
    unsigned epoch, spin_limit;
    int* timeout_flag;
    // replayed from a hipGraph: the exchange's epoch is *epoch_base + epoch (its position in the graph);
    // the host sets the base before every replay.  nullptr: `epoch` is the epoch itself.
    const unsigned* epoch_base;
};
// an exchange that already timed out poisons the ones queued behind it: they do not wait again
__device__ __forceinline__ unsigned p2p_spin_budget(const P2PCall& c) {
    return *(volatile int*)c.timeout_flag ? 0u : c.spin_limit;
}

__global__ void k_set_u32(unsigned* dst, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v; }

__global__ __launch_bounds__(256) void k_p2p_allreduce(double* __restrict__ buf, int count, P2PCall c) {
    const unsigned limit = p2p_spin_budget(c);
    const unsigned epoch = c.epoch_base ? *c.epoch_base + c.epoch : c.epoch;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < count; v += gridDim.x * blockDim.x)
        buf[v] = p2p_exchange_value(buf[v], v, c.peers, c.rank, c.nranks, epoch, limit, c.timeout_flag);
}

}  // namespace cdk
