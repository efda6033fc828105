// fsum_scan.h -- an ORDER-PRESERVING float sum that does not have to run in order.
//
// The reference's TSVQ (encoder/basisu_enc.h:1708-2077) accumulates centroids and covariances as running IEEE binary32 sums over a
// node's members in list order: s <- RN(s + a_i). RN is not associative, so the obvious parallel sum gives other bits. But the
// recurrence has very little state. Write the running sum as s = k * u with u = 2^(E-150) the ulp of s's binade (E its biased
// exponent field) and k its 24-bit significand, 2^23 <= k < 2^24. As long as the exact value s + a stays inside that binade,
//
//      RN(s + a) = (k + q + r) * u,   a / u = q + f,  q = floor(a / u),  f in [0, 1),
//      r = 0 if f < 1/2,  1 if f > 1/2,  and for a tie (f == 1/2) the parity of (k + q)            [round half to even]
//
// i.e. every addend is a map on k that only looks at k's PARITY: k -> k + d[k & 1]. Such maps compose associatively,
// (f then g)[p] = f[p] + g[(p + f[p]) & 1], so any stretch of addends over which the sum stays inside one binade collapses to two
// integers, computed in any order / in parallel, and applying the stretch to a state costs one add. Binade changes (a few dozen
// per monotone chain) are the only places where single addends have to be added one by one with a real float add.
//
// Validity: a stretch may be applied to a state k only when, at every step, the exact pre-rounding value (k' + q + f) * u is at
// least 2^23 u (tracked as the least "floor offset" k' + q over the stretch) and the rounded result stays below 2^24 u (the
// greatest result offset); then the rounding grid was u at every step.
// Anything else -- zero / denormal / negative-crossing states, infinities, addends far above the state -- is "invalid" and is
// handled by the caller with plain sequential adds, which are always right.
//
// Shared by the device kernels (tsvq_wide_kernels.hip) and, compiled with g++, by the CPU tests (tests/native/fsum_host.cpp):
// test-only host build, the product runs this code on the GPU only.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FS_FN __host__ __device__ __forceinline__
#else
#define FS_FN inline
#endif

namespace bu {
namespace fsum {

constexpr int32_t K_LO = 1 << 23;        // significand range of a normal float
constexpr int32_t K_HI = 1 << 24;
constexpr int32_t Q_SAT = 1 << 26;       // |q| of one addend saturates here (far outside any valid window)
constexpr int32_t D_SAT = 1 << 28;       // running offsets saturate here

struct addend { int32_t q; uint32_t c; }; // c: 0 = fraction below half, 1 = above half, 2 = tie

// a (given by its bits) measured in ulps of a state with biased exponent E and the given sign
FS_FN addend decode(uint32_t a_bits, int E, bool state_negative) {
    const uint32_t ea = (a_bits >> 23) & 0xffu, ma = a_bits & 0x7fffffu;
    const bool neg = ((a_bits >> 31) != 0) != state_negative;
    addend r;
    if (ea == 0xffu) { r.q = Q_SAT; r.c = 0; return r; }              // inf / nan: never valid
    const int32_t A = ea ? (int32_t)(ma | 0x800000u) : (int32_t)ma;   // a = +-A * 2^(e - 150)
    const int e = ea ? (int)ea : 1;
    if (A == 0) { r.q = 0; r.c = 0; return r; }
    const int d = E - e;
    if (d <= 0) {                                                     // a is a multiple of u
        const int sh = -d;
        const int32_t q = (sh >= 3) ? Q_SAT : ((A << sh) >= Q_SAT ? Q_SAT : (A << sh));
        r.q = neg ? -q : q; r.c = 0;
        return r;
    }
    if (d >= 25) {                                                    // |a| < u / 2
        r.q = neg ? -1 : 0; r.c = neg ? 1u : 0u;
        return r;
    }
    const int32_t SA = neg ? -A : A;
    const int32_t rem = SA & ((1 << d) - 1), half = 1 << (d - 1);
    r.q = SA >> d;                                                    // arithmetic shift = floor
    r.c = rem > half ? 1u : (rem == half ? 2u : 0u);
    return r;
}

FS_FN int32_t sat(int32_t t) { return t < -D_SAT ? -D_SAT : (t > D_SAT ? D_SAT : t); }
FS_FN int32_t sat_add(int32_t a, int32_t b) { return sat(a + b); }

// The composed map of a stretch for a state of parity p: d[p] = offset of the result; lo[p] = least floor offset (before the
// rounding bump) over the steps; hi[p] = greatest result offset over the steps. The empty stretch has lo = +SAT, hi = -SAT.
struct stretch {
    int32_t d[2], lo[2], hi[2];
};
FS_FN stretch identity() { stretch s; s.d[0] = s.d[1] = 0; s.lo[0] = s.lo[1] = D_SAT; s.hi[0] = s.hi[1] = -D_SAT; return s; }
FS_FN void push(stretch& s, addend a) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int32_t fl = sat(s.d[p] + a.q);                           // floor offset of the exact sum
        const uint32_t odd = ((uint32_t)p + (uint32_t)fl) & 1u;
        const int32_t t = fl + (int32_t)((a.c == 1u) | ((a.c == 2u) & odd));
        s.d[p] = t;
        s.lo[p] = fl < s.lo[p] ? fl : s.lo[p];
        s.hi[p] = t > s.hi[p] ? t : s.hi[p];
    }
}
// f first, then g
FS_FN stretch compose(const stretch& f, const stretch& g) {
    stretch h;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const bool pg = (((uint32_t)p + (uint32_t)f.d[p]) & 1u) != 0;   // g's half for the parity f leaves (selects: no dynamic indexing of registers)
        h.d[p] = sat_add(f.d[p], pg ? g.d[1] : g.d[0]);
        const int32_t l = sat_add(f.d[p], pg ? g.lo[1] : g.lo[0]), u = sat_add(f.d[p], pg ? g.hi[1] : g.hi[0]);
        h.lo[p] = f.lo[p] < l ? f.lo[p] : l;
        h.hi[p] = f.hi[p] > u ? f.hi[p] : u;
    }
    return h;
}

// ---- the same two operations in the form the per-block kernels use: branch-light, no saturation. An addend whose |q| reaches 2^22 (it is within
// a factor of four of the state, or above it) cannot be part of a valid 256-addend stretch anyway (offsets must stay within +-2^24); it raises
// `bad` instead, and with |q| < 2^22 the int32 offsets of up to 512 addends cannot wrap. The caller turns a bad stretch into one that never applies.
struct parts { int32_t SA; int32_t e; };   // a = SA * 2^(e - 150), SA sign-adjusted for the state's sign; e == 255: inf / nan
FS_FN parts split(uint32_t a_bits, bool state_negative) {
    const uint32_t ea = (a_bits >> 23) & 0xffu, ma = a_bits & 0x7fffffu;
    const bool neg = ((a_bits >> 31) != 0) != state_negative;
    const int32_t A = ea ? (int32_t)(ma | 0x800000u) : (int32_t)ma;
    parts p; p.SA = neg ? -A : A; p.e = ea ? (int32_t)ea : 1;
    if (ea == 0xffu) p.e = 255;
    return p;
}
FS_FN addend decode_fast(parts p, int E, bool& bad) {
    int d = E - p.e;
    bad |= (d < 2) & (p.SA != 0);            // |a| >= u * 2^22 (d <= 1 with a 24-bit significand), or inf / nan (e = 255)
    d = d < 1 ? 1 : (d > 25 ? 25 : d);       // beyond 25 the shift result and the rounding class no longer change
    const int32_t rem = p.SA & ((1 << d) - 1), half = 1 << (d - 1);
    addend r;
    r.q = p.SA >> d;
    r.c = rem > half ? 1u : (rem == half ? 2u : 0u);
    return r;
}
FS_FN void push_fast(stretch& s, addend a) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int32_t fl = s.d[p] + a.q;
        const uint32_t odd = ((uint32_t)p + (uint32_t)fl) & 1u;
        const int32_t t = fl + (int32_t)((a.c == 1u) | ((a.c == 2u) & odd));
        s.d[p] = t;
        s.lo[p] = fl < s.lo[p] ? fl : s.lo[p];
        s.hi[p] = t > s.hi[p] ? t : s.hi[p];
    }
}
FS_FN void poison(stretch& s) { s.d[0] = s.d[1] = D_SAT; s.lo[0] = s.lo[1] = -D_SAT; s.hi[0] = s.hi[1] = D_SAT; }   // never applies, whichever fields a consumer looks at

// state helpers on float bits
FS_FN bool state_ok(uint32_t s_bits) { const uint32_t e = (s_bits >> 23) & 0xffu; return e >= 1u && e <= 253u; } // normal, room above
FS_FN int state_exp(uint32_t s_bits) { return (int)((s_bits >> 23) & 0xffu); }
FS_FN int32_t state_k(uint32_t s_bits) { return (int32_t)((s_bits & 0x7fffffu) | 0x800000u); }
// may the stretch be applied to this state? (every step stays inside the binade)
FS_FN bool applies(const stretch& s, int32_t k) {
    const bool odd = (k & 1) != 0;   // selects: a dynamic index would send a register-resident stretch through scratch
    return k + (odd ? s.lo[1] : s.lo[0]) >= K_LO && k + (odd ? s.hi[1] : s.hi[0]) < K_HI;
}
FS_FN uint32_t apply(const stretch& s, uint32_t s_bits) {
    const int32_t k = state_k(s_bits);
    const int32_t K = k + ((k & 1) ? s.d[1] : s.d[0]);
    return (s_bits & 0xff800000u) | ((uint32_t)K & 0x7fffffu);
}

} // namespace fsum
} // namespace bu
