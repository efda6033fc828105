// fsum_scan64.h -- fsum_scan.h for a binary64 running sum that adds binary32 values: s <- RN64(s + (double)a_i).
//
// That is the shape of the reference's double accumulators in tree_vector_quant<>::split_node (ttsum, l_weight / r_weight:
// encoder/basisu_enc.h:1873-1881, 1996-2006): the selector side replaces them by integer reductions (tsvq_common.h, exact_acc), which is not
// possible for the 6-float endpoint vectors -- their addends w * |v|^2 are not integers. This header is the piece a many-workgroup endpoint
// split needs next to fsum_scan.h: the same parity maps on a 53-bit significand, in 64-bit integers.
//
// s = k * u, u = 2^(E - 1075) the ulp of s's binade (E its 11-bit exponent field), 2^52 <= k < 2^53. An addend is a = +-A * 2^(e - 150) with a
// 24-bit A, so a / u = +-A * 2^(e - E + 925):
//   e - E + 925 >= 0 : a is a whole number of ulps -- the add is EXACT while the sum stays in the binade (no rounding class at all);
//   otherwise        : q = floor(a / u), and the fraction decides the bump exactly as in fsum_scan.h (tie -> parity of k + q).
// Composition, validity (least floor offset >= 2^52, greatest result offset < 2^53) and the caller's contract (anything else is added with real
// adds, in order) are those of fsum_scan.h.
//
// NOT used by any kernel yet: compiled and held to the sequential double sum by tests/native/fsum64_host.cpp / tests/test_fsum64_host.py only.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FS64_FN __host__ __device__ __forceinline__
#else
#define FS64_FN inline
#endif

namespace bu {
namespace fsum64 {

constexpr int64_t K_LO = (int64_t)1 << 52;   // significand range of a normal double
constexpr int64_t K_HI = (int64_t)1 << 53;
constexpr int64_t Q_SAT = (int64_t)1 << 58;  // |q| of one addend saturates here (far outside any valid window)
constexpr int64_t D_SAT = (int64_t)1 << 60;  // running offsets saturate here

struct addend { int64_t q; uint32_t c; };    // c: 0 = fraction below half (or none), 1 = above half, 2 = tie

// the float a (given by its bits) measured in ulps of a double state with exponent field E and the given sign
FS64_FN addend decode(uint32_t a_bits, int E, bool state_negative) {
    const uint32_t ea = (a_bits >> 23) & 0xffu, ma = a_bits & 0x7fffffu;
    const bool neg = ((a_bits >> 31) != 0) != state_negative;
    addend r;
    if (ea == 0xffu) { r.q = Q_SAT; r.c = 0; return r; }              // inf / nan: never valid
    const int64_t A = ea ? (int64_t)(ma | 0x800000u) : (int64_t)ma;   // a = +-A * 2^(e - 150)
    const int e = ea ? (int)ea : 1;
    if (A == 0) { r.q = 0; r.c = 0; return r; }
    const int sh = e - E + 925;                                       // a / u = +-A * 2^sh
    if (sh >= 0) {
        const int64_t q = sh >= 34 ? Q_SAT : (A << sh);               // A < 2^24: below 2^58 up to sh = 33
        r.q = neg ? -q : q; r.c = 0;
        return r;
    }
    const int d = -sh;
    if (d >= 25) {                                                    // |a| < u / 2
        r.q = neg ? -1 : 0; r.c = neg ? 1u : 0u;
        return r;
    }
    const int64_t SA = neg ? -A : A;
    const int64_t rem = SA & (((int64_t)1 << d) - 1), half = (int64_t)1 << (d - 1);
    r.q = SA >> d;                                                    // arithmetic shift = floor
    r.c = rem > half ? 1u : (rem == half ? 2u : 0u);
    return r;
}

FS64_FN int64_t sat(int64_t t) { return t < -D_SAT ? -D_SAT : (t > D_SAT ? D_SAT : t); }
FS64_FN int64_t sat_add(int64_t a, int64_t b) { return sat(a + b); }

// d[p] = offset of the result for a state of parity p; lo[p] = least floor offset over the steps; hi[p] = greatest result offset.
struct stretch { int64_t d[2], lo[2], hi[2]; };
FS64_FN stretch identity() { stretch s; s.d[0] = s.d[1] = 0; s.lo[0] = s.lo[1] = D_SAT; s.hi[0] = s.hi[1] = -D_SAT; return s; }
FS64_FN void push(stretch& s, addend a) {
    for (int p = 0; p < 2; p++) {
        const int64_t fl = sat(s.d[p] + a.q);                          // floor offset of the exact sum
        const uint32_t odd = (uint32_t)(((uint64_t)p + (uint64_t)fl) & 1u);
        const int64_t t = fl + (int64_t)((a.c == 1u) | ((a.c == 2u) & odd));
        s.d[p] = t;
        s.lo[p] = fl < s.lo[p] ? fl : s.lo[p];
        s.hi[p] = t > s.hi[p] ? t : s.hi[p];
    }
}
// f first, then g
FS64_FN stretch compose(const stretch& f, const stretch& g) {
    stretch h;
    for (int p = 0; p < 2; p++) {
        const bool pg = (((uint64_t)p + (uint64_t)f.d[p]) & 1u) != 0;
        h.d[p] = sat_add(f.d[p], pg ? g.d[1] : g.d[0]);
        const int64_t l = sat_add(f.d[p], pg ? g.lo[1] : g.lo[0]), u = sat_add(f.d[p], pg ? g.hi[1] : g.hi[0]);
        h.lo[p] = f.lo[p] < l ? f.lo[p] : l;
        h.hi[p] = f.hi[p] > u ? f.hi[p] : u;
    }
    return h;
}

// state helpers on double bits
FS64_FN bool state_ok(uint64_t s_bits) { const uint32_t e = (uint32_t)(s_bits >> 52) & 0x7ffu; return e >= 1u && e <= 2045u; } // normal, room above
FS64_FN int state_exp(uint64_t s_bits) { return (int)((s_bits >> 52) & 0x7ffu); }
FS64_FN int64_t state_k(uint64_t s_bits) { return (int64_t)((s_bits & 0xfffffffffffffull) | 0x10000000000000ull); }
FS64_FN bool applies(const stretch& s, int64_t k) {
    const bool odd = (k & 1) != 0;
    return k + (odd ? s.lo[1] : s.lo[0]) >= K_LO && k + (odd ? s.hi[1] : s.hi[0]) < K_HI;
}
FS64_FN uint64_t apply(const stretch& s, uint64_t s_bits) {
    const int64_t k = state_k(s_bits);
    const int64_t K = k + ((k & 1) ? s.d[1] : s.d[0]);
    return (s_bits & 0xfff0000000000000ull) | ((uint64_t)K & 0xfffffffffffffull);
}

} // namespace fsum64
} // namespace bu
