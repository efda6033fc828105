// tsvq_common.h -- device helpers shared by tsvq_kernels.hip (one workgroup per node, chained sums) and tsvq_wide_kernels.hip
// (many workgroups per node, order-preserving sums through fsum_scan.h). Internal to libbasisu_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bu {

template <int N> __device__ __forceinline__ float dot_seq(const float* a, const float* b) {
    float r = a[0] * b[0];
#pragma unroll
    for (int i = 1; i < N; i++) r += a[i] * b[i];
    return r;
}

// compute_pca_from_covar (enc.h:605-648) on one thread: 8 power iterations, double row sums, float early-out.
template <int N>
__device__ __noinline__ void principal_axis(const float (*cov)[16], float* out_axis) {
    float axis[N], prev[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float t = (float)(uint32_t)i * (1.0f / (float)(N - 1 > 1 ? N - 1 : 1));
        axis[i] = .75f + (1.25f - .75f) * t;
        prev[i] = axis[i];
    }
    for (int iter = 0; iter < 8; iter++) {
        float trial[N];
        double max_sum = 0;
        for (int i = 0; i < N; i++) {
            double sum = 0;
            for (int j = 0; j < N; j++) { const float p = cov[i][j] * axis[j]; sum += p; }
            trial[i] = (float)sum;
            const double a = fabs(sum);
            if (a > max_sum) max_sum = a;
        }
        if (max_sum != 0.0) {
            const float s = (float)(1.0 / max_sum);
            for (int i = 0; i < N; i++) trial[i] *= s;
        }
        float delta[N];
        for (int i = 0; i < N; i++) delta[i] = prev[i] - trial[i];
        for (int i = 0; i < N; i++) { prev[i] = axis[i]; axis[i] = trial[i]; }
        if (dot_seq<N>(delta, delta) < .0024f) break;
    }
    const float len = sqrtf(dot_seq<N>(axis, axis));
    if (len != 0.0f) {
        const float s = 1.0f / len;
        for (int i = 0; i < N; i++) axis[i] *= s;
    }
    for (int i = 0; i < N; i++) out_axis[i] = axis[i];
}

// The same on ONE WAVE (all 64 lanes must call it): lane i < N owns row i of every matrix-vector product -- the same float products and the
// same j = 0..N-1 double summation as the serial form -- the vector is kept replicated in every lane (gathered by shuffles), the order-dependent
// float dot products are evaluated redundantly by every lane in the serial order. ~10x fewer dependent instructions than one thread doing all rows.
template <int N>
__device__ __forceinline__ void principal_axis_wave(const float (*cov)[16], float* out_axis) {
    const int lane = threadIdx.x & 63;
    const int row = lane < N ? lane : 0;
    float crow[N];
#pragma unroll
    for (int j = 0; j < N; j++) crow[j] = cov[row][j];
    float axis[N], prev[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float t = (float)(uint32_t)i * (1.0f / (float)(N - 1 > 1 ? N - 1 : 1));
        axis[i] = .75f + (1.25f - .75f) * t;
        prev[i] = axis[i];
    }
    for (int iter = 0; iter < 8; iter++) {
        double sum = 0;
#pragma unroll
        for (int j = 0; j < N; j++) { const float p = crow[j] * axis[j]; sum += p; }
        // the largest |row sum|: a non-negative double orders like its bit pattern, so the maximum over lanes 0..N-1 is a DPP prefix maximum (rows of 16
        // lanes; N <= 16) read from lane 15 -- no trip through the LDS crossbar per step
        unsigned long long mb = lane < N ? (unsigned long long)__double_as_longlong(fabs(sum)) : 0ull;
#pragma unroll
        for (int step = 0; step < 4; step++) {
            uint32_t lo, hi;
            if (step == 0) { lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)mb, 0x111, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(mb >> 32), 0x111, 0xf, 0xf, false); }
            else if (step == 1) { lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)mb, 0x112, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(mb >> 32), 0x112, 0xf, 0xf, false); }
            else if (step == 2) { lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)mb, 0x114, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(mb >> 32), 0x114, 0xf, 0xf, false); }
            else { lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)mb, 0x118, 0xf, 0xf, false); hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(mb >> 32), 0x118, 0xf, 0xf, false); }
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            mb = o > mb ? o : mb;
        }
        const double max_sum = __longlong_as_double((long long)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mb >> 32), 15) << 32) |
                                                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mb, 15)));
        float mine = (float)sum;
        if (max_sum != 0.0) mine *= (float)(1.0 / max_sum);
        float trial[N], delta[N];
#pragma unroll
        for (int i = 0; i < N; i++) trial[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), i));
#pragma unroll
        for (int i = 0; i < N; i++) delta[i] = prev[i] - trial[i];
#pragma unroll
        for (int i = 0; i < N; i++) { prev[i] = axis[i]; axis[i] = trial[i]; }
        if (dot_seq<N>(delta, delta) < .0024f) break;
    }
    const float len = sqrtf(dot_seq<N>(axis, axis));
    if (len != 0.0f) {
        const float s = 1.0f / len;
#pragma unroll
        for (int i = 0; i < N; i++) axis[i] *= s;
    }
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < N; i++) out_axis[i] = axis[i];
}

// wave64 sums by DPP (rows of 16 lanes: row_shr 1/2/4/8, then row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3): an inclusive prefix whose LAST lane
// holds the wave's total; lanes without a source add 0. No LDS traffic, six dependent adds.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_src_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {   // valid in lane 63
    v += dpp_src_u32<0x111, 0xf>(v); v += dpp_src_u32<0x112, 0xf>(v); v += dpp_src_u32<0x114, 0xf>(v); v += dpp_src_u32<0x118, 0xf>(v);
    v += dpp_src_u32<0x142, 0xa>(v); v += dpp_src_u32<0x143, 0xc>(v);
    return v;
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint64_t dpp_src_u64(uint64_t v) {
    return ((uint64_t)dpp_src_u32<CTRL, ROW_MASK>((uint32_t)(v >> 32)) << 32) | dpp_src_u32<CTRL, ROW_MASK>((uint32_t)v);
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {   // valid in lane 63
    v += dpp_src_u64<0x111, 0xf>(v); v += dpp_src_u64<0x112, 0xf>(v); v += dpp_src_u64<0x114, 0xf>(v); v += dpp_src_u64<0x118, 0xf>(v);
    v += dpp_src_u64<0x142, 0xa>(v); v += dpp_src_u64<0x143, 0xc>(v);
    return v;
}

// The reference's double accumulators (ttsum, l_weight / r_weight) only ever add floats. When every addend is a non-negative
// INTEGER-valued float below 2^53 and the total stays below 2^53 (always the case for selector vectors with real weights), each
// double add is exact, so the running sum equals the integer sum and its order does not matter: the "exact" kernel variants
// replace those two chains by an integer reduction (low / high 32-bit halves summed separately so nothing overflows). When the
// condition fails the kernel reports it and the caller re-runs the node with the chained variant.
struct exact_acc {
    uint64_t lo = 0, hi = 0;
    __device__ __forceinline__ bool add(float t) { // returns false when t is outside the exact range
        if (!(t < 9007199254740992.0f)) return false;
        const uint64_t ti = (uint64_t)t;
        lo += ti & 0xffffffffull; hi += ti >> 32;
        return true;
    }
    // add(t) when `take`, nothing otherwise -- without a branch, so that two accumulators picked by a per-member flag stay in registers
    __device__ __forceinline__ bool add_if(float t, bool take) {
        const bool in_range = t < 9007199254740992.0f;
        const uint64_t ti = (in_range && take) ? (uint64_t)t : 0ull;
        lo += ti & 0xffffffffull; hi += ti >> 32;
        return in_range || !take;
    }
};
__device__ __forceinline__ bool exact_total(uint64_t lo, uint64_t hi, double* out) {
    const uint64_t h = hi + (lo >> 32);
    if (h >= (1ull << 21)) return false;
    *out = (double)((h << 32) | (lo & 0xffffffffull));
    return true;
}

// value k (0..15) of a packed selector vector: element 0 in the top two bits (the order the frontend's de-duplication keys use)
__device__ __forceinline__ uint32_t packed16_value(uint32_t key, int k) { return (key >> (30 - 2 * k)) & 3u; }

} // namespace bu
