// fsum_host.cpp -- TEST-ONLY host build of csrc/fsum_scan.h (the order-preserving float sum the wide TSVQ kernels use):
// the blocked algorithm (per-block stretches for predicted binades, serial walk with plain adds as the fallback) against the
// plain sequential float sum it must reproduce bit for bit. Compiled by tests/helpers.py with g++ -O2 -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../basis_universal_amd/csrc/fsum_scan.h"

using namespace bu::fsum;

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static int g_fast = 0;

extern "C" {

void fsum_set_fast(int on) { g_fast = on; }

// s <- RN(s + a[i]) for i = 0..n-1, from `start`
float fsum_sequential(const float* a, uint64_t n, float start) {
    volatile float s = start;
    for (uint64_t i = 0; i < n; i++) s = s + a[i];
    return s;
}

// The blocked form. block: addends per block; stats[0] = blocks applied as a stretch, stats[1] = blocks walked with plain adds,
// stats[2] = blocks whose predicted binades did not contain the state's.
float fsum_blocked(const float* a, uint64_t n, float start, uint32_t block, uint64_t* stats) {
    const uint64_t nb = (n + block - 1) / block;
    struct summ { int E; bool neg; stretch s[2]; };
    std::vector<summ> sm(nb);
    // "stage 0/1": exact-ish prefix in double predicts the binade at every block start; two candidate exponents per block
    double P = (double)start;
    for (uint64_t b = 0; b < nb; b++) {
        const uint64_t i0 = b * block, i1 = i0 + block < n ? i0 + block : n;
        const double lo = std::fabs(P) * (1.0 - (double)(i0 + 1) * 5.9604644775390625e-08); // |s| >= |P| (1 - i 2^-24) for monotone chains
        const float lof = (float)(lo > 0 ? lo : 0);
        summ& m = sm[b];
        m.neg = P < 0;
        m.E = state_exp(f2u(lof));
        if (m.E > 0 && u2f(f2u(lof)) > lo) m.E = state_exp(f2u(std::nextafterf(lof, 0.0f))); // rounded up across a power of two
        for (int c = 0; c < 2; c++) {
            stretch s = identity();
            if (g_fast && i1 - i0 <= 512) {   // the kernels' form: no saturation, a `bad` flag instead
                bool bad = false;
                for (uint64_t i = i0; i < i1; i++) {
                    if ((f2u(a[i]) << 1) == 0) continue;
                    push_fast(s, decode_fast(split(f2u(a[i]), m.neg), m.E + c, bad));
                }
                if (bad || m.E + c < 1 || m.E + c > 253) poison(s);
            } else
                for (uint64_t i = i0; i < i1; i++) push(s, decode(f2u(a[i]), m.E + c, m.neg));
            m.s[c] = s;
        }
        for (uint64_t i = i0; i < i1; i++) P += (double)a[i];
    }
    // "stage 2": the serial walk
    uint32_t s = f2u(start);
    for (uint64_t b = 0; b < nb; b++) {
        const summ& m = sm[b];
        const int c = state_exp(s) - m.E;
        const bool neg = (s >> 31) != 0;
        if (state_ok(s) && neg == m.neg && (c == 0 || c == 1)) {
            // g_fast == 2: the walk's rule for monotone chains (addends >= 0 on a positive sum): only the result offsets are looked at
            const bool ok = g_fast == 2 ? (state_k(s) + m.s[c].d[state_k(s) & 1] < K_HI) : applies(m.s[c], state_k(s));
            if (ok) { s = apply(m.s[c], s); if (stats) stats[0]++; continue; }
        } else if (stats) stats[2]++;
        const uint64_t i0 = b * block, i1 = i0 + block < n ? i0 + block : n;
        volatile float f = u2f(s);
        for (uint64_t i = i0; i < i1; i++) f = f + a[i];
        s = f2u(f);
        if (stats) stats[1]++;
    }
    return u2f(s);
}

// Composition check: the stretch of a whole range built by composing per-piece stretches equals the stretch pushed in one go.
int fsum_compose_check(const float* a, uint64_t n, int E, int neg, uint32_t piece) {
    stretch whole = identity(), acc = identity();
    for (uint64_t i = 0; i < n; i++) push(whole, decode(f2u(a[i]), E, neg != 0));
    for (uint64_t i0 = 0; i0 < n; i0 += piece) {
        stretch s = identity();
        for (uint64_t i = i0; i < n && i < i0 + piece; i++) push(s, decode(f2u(a[i]), E, neg != 0));
        acc = compose(acc, s);
    }
    // saturated values need not agree digit for digit; what must agree is applicability and, where applicable, the result
    for (int p = 0; p < 2; p++) {
        const bool wa = whole.lo[p] > -(1 << 25) && whole.hi[p] < (1 << 25), ca = acc.lo[p] > -(1 << 25) && acc.hi[p] < (1 << 25);
        if (wa != ca) return 0;
        if (wa && (whole.d[p] != acc.d[p] || whole.lo[p] != acc.lo[p] || whole.hi[p] != acc.hi[p])) return 0;
    }
    return 1;
}

} // extern "C"
