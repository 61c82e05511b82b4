// sparse_iterate.hpp -- host-side bookkeeping of the iterate's support and the
// coordinate schedulers.  The device holds the dense beta; this mirrors what the
// reference keeps in ProximalBase's SparseIterate (insertion-ordered support,
// zeros kept until dropzeros!) and in src/atom_iterator.jl,
// because the ORDER of the support is the visit order of an active pass
// (atom_iterator.jl:18-26).  Coordinates are 0-based here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace cdh {

class SupportList {
public:
    explicit SupportList(int64_t p = 0) { resize(p); }
    void resize(int64_t p) {
        p_ = p; nnz_ = 0;
        val_.assign((std::size_t)p, 0.0); slot2ind_.assign((std::size_t)p, 0); ind2slot_.assign((std::size_t)p, 0);
    }
    int64_t size() const { return p_; }
    int64_t nnz() const { return nnz_; }
    int64_t coord(int64_t slot) const { return slot2ind_[(std::size_t)slot]; }
    double slot_value(int64_t slot) const { return val_[(std::size_t)slot]; }
    double get(int64_t k) const {
        const int64_t s = ind2slot_[(std::size_t)k];
        return s ? val_[(std::size_t)(s - 1)] : 0.0;
    }
    // setindex!: non-zero to an unstored coordinate appends; zero to a stored one keeps the slot
    void set(int64_t k, double v) {
        const int64_t s = ind2slot_[(std::size_t)k];
        if (s) { val_[(std::size_t)(s - 1)] = v; return; }
        if (v != 0.0) {
            val_[(std::size_t)nnz_] = v; slot2ind_[(std::size_t)nnz_] = k; ++nnz_;
            ind2slot_[(std::size_t)k] = nnz_;
        }
    }
    void clear() {  // fill!(x, 0)
        for (int64_t i = 0; i < nnz_; ++i) ind2slot_[(std::size_t)slot2ind_[(std::size_t)i]] = 0;
        nnz_ = 0;
    }
    // dropzeros! -- resulting order is not pinned by any reference test; swap-with-last
    void dropzeros() {
        int64_t i = 0;
        while (i < nnz_) {
            if (val_[(std::size_t)i] == 0.0) {
                ind2slot_[(std::size_t)slot2ind_[(std::size_t)i]] = 0;
                const int64_t last = nnz_ - 1;
                if (i != last) {
                    val_[(std::size_t)i] = val_[(std::size_t)last];
                    slot2ind_[(std::size_t)i] = slot2ind_[(std::size_t)last];
                    ind2slot_[(std::size_t)slot2ind_[(std::size_t)i]] = i + 1;
                }
                --nnz_;
            } else {
                ++i;
            }
        }
    }

private:
    int64_t p_ = 0, nnz_ = 0;
    std::vector<double> val_;
    std::vector<int64_t> slot2ind_, ind2slot_;
};

// OrderedIterator / RandomIterator (atom_iterator.jl:9-75).  The reference shuffles
// with Julia's global RNG; the documented substitute is splitmix64 with
// j = i + next() mod (L - i)   (0-based i, L = pass length).
class VisitScheduler {
public:
    VisitScheduler(int64_t p, bool randomize, uint64_t seed)
        : p_(p), randomize_(randomize), state_(seed), order_((std::size_t)p) {
        for (int64_t i = 0; i < p; ++i) order_[(std::size_t)i] = i;
    }
    // reset!(it, fullPass) followed by collect(it): the 0-based visit list of the pass
    void next_pass(const SupportList& x, bool fullPass, std::vector<int64_t>& out) {
        const int64_t L = fullPass ? p_ : x.nnz();
        out.resize((std::size_t)L);
        if (randomize_) {
            for (int64_t i = 0; i < L; ++i) order_[(std::size_t)i] = i;
            for (int64_t i = 0; i + 1 < L; ++i) {
                const int64_t j = i + (int64_t)(next() % (uint64_t)(L - i));
                std::swap(order_[(std::size_t)i], order_[(std::size_t)j]);
            }
        }
        for (int64_t i = 0; i < L; ++i) {
            const int64_t o = randomize_ ? order_[(std::size_t)i] : i;
            out[(std::size_t)i] = fullPass ? o : x.coord(o);
        }
    }

    // the generator's state: the device-resident pass loops carry it themselves (small_solve.hpp, cov_solve.hpp)
    uint64_t state() const { return state_; }
    void set_state(uint64_t s) { state_ = s; }

private:
    uint64_t next() {
        uint64_t z = (state_ += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    int64_t p_;
    bool randomize_;
    uint64_t state_;
    std::vector<int64_t> order_;
};

}  // namespace cdh
