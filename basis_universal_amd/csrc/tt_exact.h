// tt_exact.h -- when is a run of `s <- RN64(s + (double)a_i)` over a block of float addends a_i >= 0 EXACT, i.e. independent of the order and equal to s + (block sum)?
//
// That is the shape of the reference's double accumulators in tree_vector_quant<>::split_node (l_ttsum / r_ttsum, encoder/basisu_enc.h:1996-2006) for the 6-float
// endpoint vectors, whose addends w * |v|^2 are not integers (the selector side replaces them by integer reductions: tsvq_common.h, exact_acc). A double has 29 bits more
// than the float it adds, so an add rounds only when bits fall off the low end of the sum. Let L be (a lower bound of) the exponent of the lowest bit that can be set in s,
// Lb the same for the block's addends (exponent of the smallest non-zero addend - 23), and E the exponent of an upper bound of the sum after the block. Every partial sum of
// the block, in any order, is a multiple of 2^min(L, Lb) and at most that upper bound: all of them fit into 53 bits -- every add is exact -- iff min(L, Lb) >= E - 52.
// A block that fails the test is added member by member with real double adds (the caller's job), after which L is the lowest set bit of the resulting double.
// The test only ever errs on the safe side. Shared by tsvq_wide6_kernels.hip (tt_walk) and its test-only host build (tests/native/tt_exact_host.cpp, tests/test_tt_exact_host.py).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define TT_FN __host__ __device__ __forceinline__
#else
#define TT_FN inline
#endif

namespace bu {
namespace tt {

constexpr int L_FREE = 1 << 20;          // "no low bit to lose": an empty or zero sum
constexpr uint32_t E_NONE = 0xffffu;     // block summary: no non-zero addend
constexpr uint32_t E_UNSAFE = 0u;        // block summary: a denormal, negative or non-finite addend -- never exact by this test

TT_FN uint64_t bits_of(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

// what one addend contributes to its block's summary (combine with min): its exponent field, or E_NONE for +-0, or E_UNSAFE
TT_FN uint32_t addend_exp(uint32_t float_bits) {
    if ((float_bits << 1) == 0) return E_NONE;
    const uint32_t e = (float_bits >> 23) & 0xffu;
    return (e == 0xffu || e == 0u || (float_bits >> 31)) ? E_UNSAFE : e;
}
// exponent (power of two) of the lowest bit a block with that summary can set
TT_FN int block_low(uint32_t summary) { return summary == E_NONE ? L_FREE : (summary == E_UNSAFE ? -L_FREE : (int)summary - 150); }

// exponent of the lowest SET bit of a finite double > 0 (what L becomes after real adds); L_FREE for 0, -L_FREE for denormals
TT_FN int low_bit(double x) {
    if (x == 0.0) return L_FREE;
    const uint64_t b = bits_of(x);
    const uint32_t ef = (uint32_t)(b >> 52) & 0x7ffu;
    if (ef == 0u || ef == 0x7ffu) return -L_FREE;
    uint64_t sig = (b & 0xfffffffffffffull) | (1ull << 52);
    int z = 0;
    while (!(sig & 1ull)) { sig >>= 1; z++; }
    return (int)ef - 1023 - 52 + z;
}

// s: the exact running sum so far; bs: the block's sum as a tree of double adds computed it (it may have rounded when the test is about to fail: a few parts in 2^53,
// covered by the factor below); L: see above. true: every add of the block is exact, the sum after it is *s_end = s + bs, and L becomes min(L, block_low(summary)).
TT_FN bool block_is_exact(double s, double bs, int L, uint32_t summary, double* s_end) {
    *s_end = s + bs;
    if (summary == E_NONE) return true;
    const double u = *s_end * 1.0000000000009095;   // 1 + 2^-40: an upper bound of the true sum after the block
    const uint32_t ef = (uint32_t)(bits_of(u) >> 52) & 0x7ffu;
    if (ef == 0x7ffu || ef == 0u) return false;
    const int Lb = block_low(summary);
    return (L < Lb ? L : Lb) >= (int)ef - 1023 - 52;
}

} // namespace tt
} // namespace bu
