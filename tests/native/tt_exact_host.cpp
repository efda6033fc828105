// tt_exact_host.cpp -- TEST-ONLY host build of csrc/tt_exact.h: the block walk tsvq_wide6_kernels.hip (tt_walk) does for the reference's double accumulators --
// blocks that pass tt::block_is_exact taken in one step with a tree-summed block total, the others added member by member -- against the plain sequential sum
//     s <- s + (double)a[i]
// it must reproduce bit for bit. Compiled by tests/helpers.py with g++ -O2 -ffp-contract=off.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../basis_universal_amd/csrc/tt_exact.h"

using namespace bu::tt;

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

extern "C" {

double tt_sequential(const float* a, uint64_t n) {
    volatile double s = 0.0;
    for (uint64_t i = 0; i < n; i++) s = s + (double)a[i];
    return s;
}

// stats[0] = blocks taken in one step, stats[1] = blocks added member by member, stats[2] = blocks that passed the test although two different summation orders of
// the block (a pairwise tree, and last-to-first) disagree or differ from the in-order result -- must stay 0: the test promises exactness in ANY order
double tt_blocked(const float* a, uint64_t n, uint32_t block, uint64_t* stats) {
    double s = 0.0;
    int L = L_FREE;
    stats[0] = stats[1] = stats[2] = 0;
    std::vector<double> t(block);
    for (uint64_t i0 = 0; i0 < n; i0 += block) {
        const uint64_t m = i0 + block < n ? block : n - i0;
        uint32_t summary = E_NONE;
        for (uint64_t j = 0; j < m; j++) { const uint32_t e = addend_exp(f2u(a[i0 + j])); if (e < summary) summary = e; }
        // the block total as a pairwise tree (what a wave reduction computes)
        for (uint64_t j = 0; j < m; j++) t[j] = (double)a[i0 + j];
        for (uint64_t w = m; w > 1; w = (w + 1) / 2) {
            const uint64_t h = (w + 1) / 2;
            for (uint64_t j = 0; j + h < w; j++) { volatile double x = t[j] + t[j + h]; t[j] = x; }
        }
        const double bs = t[0];
        double s_end;
        if (block_is_exact(s, bs, L, summary, &s_end)) {
            volatile double fwd = s, rev = 0.0;
            for (uint64_t j = 0; j < m; j++) fwd = fwd + (double)a[i0 + j];
            for (uint64_t j = m; j-- > 0;) rev = rev + (double)a[i0 + j];
            volatile double rev_total = s + rev;
            if (fwd != s_end || rev_total != s_end) stats[2]++;
            s = s_end;
            const int Lb = block_low(summary);
            if (Lb < L) L = Lb;
            stats[0]++;
        } else {
            volatile double f = s;
            for (uint64_t j = 0; j < m; j++) f = f + (double)a[i0 + j];
            s = f;
            L = low_bit(s);
            stats[1]++;
        }
    }
    return s;
}

int tt_low_bit(double x) { return low_bit(x); }

}
