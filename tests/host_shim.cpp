// Test-only C shim over the product's host-side bookkeeping (csrc/sparse_iterate.hpp) so the CPU
// suite can drive it without a GPU and compare it with the oracle.
#include <cstdint>
#include <vector>
#include "../coordinatedescent.jl_amd/csrc/sparse_iterate.hpp"

extern "C" {
void* sl_new(int64_t p) { return new cdh::SupportList(p); }
void sl_free(void* s) { delete (cdh::SupportList*)s; }
void sl_set(void* s, int64_t k0, double v) { ((cdh::SupportList*)s)->set(k0, v); }
double sl_get(void* s, int64_t k0) { return ((cdh::SupportList*)s)->get(k0); }
void sl_dropzeros(void* s) { ((cdh::SupportList*)s)->dropzeros(); }
void sl_clear(void* s) { ((cdh::SupportList*)s)->clear(); }
int64_t sl_nnz(void* s) { return ((cdh::SupportList*)s)->nnz(); }
void sl_support(void* s, int64_t* out) {
    auto* x = (cdh::SupportList*)s;
    for (int64_t i = 0; i < x->nnz(); ++i) out[i] = x->coord(i);
}
void* vs_new(int64_t p, int randomize, uint64_t seed) { return new cdh::VisitScheduler(p, randomize != 0, seed); }
void vs_free(void* v) { delete (cdh::VisitScheduler*)v; }
int64_t vs_next(void* v, void* s, int full, int64_t* out) {
    std::vector<int64_t> visit;
    ((cdh::VisitScheduler*)v)->next_pass(*(cdh::SupportList*)s, full != 0, visit);
    for (size_t i = 0; i < visit.size(); ++i) out[i] = visit[i];
    return (int64_t)visit.size();
}
}
