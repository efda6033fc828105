// kmeans_kernels.hip -- the codebook builders' FAST mode: weighted k-means over the distinct training vectors with
// the assignment step on the matrix cores, in place of the order-dependent TSVQ (encoder/basisu_enc.h:1546-2354). EXPLICITLY NOT
// bit-identical to the reference -- it produces different (typically slightly better) codebooks -- and therefore off by default and gated
// by the reference's own tolerances on size and PSNR (basisu_tool.cpp:6786-6793; tests/test_gpu_fast_codebooks.py).
//
// It is deterministic all the same: the assignment is an argmin with index tie-breaks over MFMA results (the same instruction stream gives
// the same bits run to run), and the centroid update accumulates INTEGERS (vector components are small integers: selector values 0..3,
// endpoint colours 0..255; weights are integers) with 64-bit atomics, whose sum does not depend on the order of arrival.
//
// Assignment as a GEMM: argmin_c |u - c|^2 = argmin_c (|c|^2 - 2 c.u). One v_mfma_f32_32x32x16_f16 computes a 32-centroid x 32-vector tile
// of (-2 c).u over all 16 dimensions (endpoint vectors: 6 real + 10 zero); the accumulator is preloaded with |c|^2, so the instruction's
// output IS the comparison key. Centroids are fractional: -2c is split into two halves' worth of precision (hi + lo), two MFMAs per tile;
// vector components are exact in f16. Rows of the tile are centroids, columns are vectors, so every lane owns ONE vector (column lane & 31)
// and sees 16 centroids per tile in its accumulator registers: the running argmin never leaves the lane until the two half-waves merge.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include "kmeans_kernels.h"

namespace bu {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int KM_DIM = 16;

// distinct vectors -> [U][16] f16: selector vectors from the packed key (value 0 in the top two bits) ...
__global__ __launch_bounds__(256) void k_km_unpack_selectors(const uint32_t* __restrict__ keys, uint32_t n, _Float16* __restrict__ out) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const uint32_t key = keys[u];
    half8 a, b;
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)(float)((key >> (30 - 2 * k)) & 3u); b[k] = (_Float16)(float)((key >> (14 - 2 * k)) & 3u); }
    half8* o = reinterpret_cast<half8*>(out + (size_t)u * KM_DIM);
    o[0] = a; o[1] = b;
}
// ... endpoint vectors from the 48-bit key (low rgb in bits 47..24, high rgb in bits 23..0), padded with zeros; weight = 2 x group size
__global__ __launch_bounds__(256) void k_km_unpack_endpoints(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ goffs, uint32_t n, _Float16* __restrict__ out,
                                                             uint64_t* __restrict__ weights) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const uint64_t key = keys[u];
    half8 a, b;
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = (_Float16)(k < 6 ? (float)((key >> (40 - 8 * k)) & 255u) : 0.0f); b[k] = (_Float16)0.0f; }
    half8* o = reinterpret_cast<half8*>(out + (size_t)u * KM_DIM);
    o[0] = a; o[1] = b;
    weights[u] = 2ull * (goffs[u + 1] - goffs[u]);
}

// initial centroids: the distinct vectors at the k weight quantiles (j + 1/2) / k of the (sorted) list -- more centres where the weight is --
// made distinct by moving on to the next unused vector (cum = inclusive prefix sums of the weights)
__global__ __launch_bounds__(256) void k_km_seed_pick(const uint64_t* __restrict__ cum, uint32_t n, uint32_t k, uint32_t* __restrict__ pick) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k) return;
    const unsigned __int128 total = cum[n - 1];
    const uint64_t target = (uint64_t)((total * (2ull * c + 1)) / (2ull * k));
    uint32_t lo = 0, hi = n - 1;   // first u with cum[u] > target
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] > target) hi = mid; else lo = mid + 1; }
    pick[c] = lo;
}
// picks are ascending but may repeat: u_c = c + max_{j <= c}(pick_j - j) makes them strictly ascending, capped so that the last centres still
// find a vector (k <= n). One 1024-thread workgroup: a prefix maximum over chunks.
__global__ __launch_bounds__(1024) void k_km_seed_distinct(uint32_t* __restrict__ pick, uint32_t n, uint32_t k) {
    __shared__ int32_t s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (k + 1023) / 1024, c0 = min((uint32_t)tid * per, k), c1 = min(c0 + per, k);
    int32_t m = INT32_MIN;
    for (uint32_t c = c0; c < c1; c++) m = max(m, (int32_t)pick[c] - (int32_t)c);
    int32_t incl = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl = max(incl, t); }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int32_t before = INT32_MIN;
    for (int w = 0; w < wave; w++) before = max(before, s_wave[w]);
    const int32_t prev_lane = __shfl_up(incl, 1, 64);
    int32_t run = max(before, lane ? prev_lane : INT32_MIN);   // maximum over everything before this thread's chunk
    const int32_t cap = (int32_t)(n - k);
    for (uint32_t c = c0; c < c1; c++) {
        run = max(run, (int32_t)pick[c] - (int32_t)c);
        pick[c] = c + (uint32_t)min(run, cap);
    }
}
__global__ __launch_bounds__(256) void k_km_seed(const _Float16* __restrict__ vec, const uint32_t* __restrict__ pick, uint32_t k, float* __restrict__ cen) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k * KM_DIM) return;
    cen[i] = (float)vec[(size_t)pick[i / KM_DIM] * KM_DIM + (i % KM_DIM)];
}

// clusters that lost all their members are moved onto the vectors that are represented worst (largest weighted error first)
__global__ __launch_bounds__(1024) void k_km_list_empty(const uint64_t* __restrict__ sums, uint32_t k, uint32_t* __restrict__ empty, uint32_t* __restrict__ n_empty) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < k; c0 += 1024) {   // index order is kept: chunk by chunk, wave by wave, lane by lane
        const uint32_t c = c0 + (uint32_t)tid;
        const bool e = c < k && sums[(size_t)c * 17 + 16] == 0;
        const uint64_t mask = __ballot(e);
        if (lane == 0) s_wave[wave] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t at = s_base;
        for (int w = 0; w < wave; w++) at += s_wave[w];
        if (e) empty[at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = c;
        __syncthreads();
        if (tid == 0) { uint32_t t = 0; for (int w = 0; w < 16; w++) t += s_wave[w]; s_base += t; }
        __syncthreads();
    }
    if (tid == 0) *n_empty = s_base;
}
__global__ __launch_bounds__(256) void k_km_reseed(const _Float16* __restrict__ vec, const uint32_t* __restrict__ worst, const uint32_t* __restrict__ empty,
                                                   const uint32_t* __restrict__ n_empty, uint32_t n, float* __restrict__ cen, uint64_t* __restrict__ sums) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t m = min(*n_empty, n);
    if (i >= m * KM_DIM) return;
    const uint32_t j = i / KM_DIM, d = i % KM_DIM, c = empty[j];
    cen[(size_t)c * KM_DIM + d] = (float)vec[(size_t)worst[j] * KM_DIM + d];
    if (d == 0) sums[(size_t)c * 17 + 16] = 1;   // live again for the next assignment
}

// float centroids -> the GEMM operands: -2c as hi + lo halves, |c|^2 (rows past k and empty clusters get +inf: never the minimum)
__global__ __launch_bounds__(256) void k_km_prepare(const float* __restrict__ cen, const uint64_t* __restrict__ sums, uint32_t k, uint32_t k_pad, int have_sums,
                                                    _Float16* __restrict__ hi, _Float16* __restrict__ lo, float* __restrict__ cnorm) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k_pad) return;
    const bool live = c < k && (!have_sums || sums[(size_t)c * 17 + 16] != 0);
    float nrm = 0.0f;
#pragma unroll
    for (int d = 0; d < KM_DIM; d++) {
        const float a = c < k ? -2.0f * cen[(size_t)c * KM_DIM + d] : 0.0f;
        const _Float16 h = (_Float16)a;
        const _Float16 l = (_Float16)(a - (float)h);
        hi[(size_t)c * KM_DIM + d] = h; lo[(size_t)c * KM_DIM + d] = l;
        const float cc = -0.5f * ((float)h + (float)l);   // the centroid the GEMM really uses
        nrm += cc * cc;
    }
    cnorm[c] = live ? nrm : __builtin_inff();
}

// sums -> float centroids (empty clusters keep their place)
__global__ __launch_bounds__(256) void k_km_update(const uint64_t* __restrict__ sums, uint32_t k, float* __restrict__ cen) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k * KM_DIM) return;
    const uint32_t c = i / KM_DIM, d = i % KM_DIM;
    const uint64_t w = sums[(size_t)c * 17 + 16];
    if (w) cen[i] = (float)((double)sums[(size_t)c * 17 + d] / (double)w);
}

// One wave per 64 vectors (two 32-column tiles share every centroid tile); a 256-thread workgroup keeps |c|^2 of all centroids in LDS.
__global__ __launch_bounds__(256) void k_km_assign(const _Float16* __restrict__ vec, const uint64_t* __restrict__ weights, uint32_t n, const _Float16* __restrict__ hi,
                                                   const _Float16* __restrict__ lo, const float* __restrict__ cnorm, uint32_t k_pad, uint32_t* __restrict__ assign,
                                                   unsigned long long* __restrict__ sums, int dims, float* __restrict__ err_key, uint32_t* __restrict__ err_idx) {
    extern __shared__ float s_norm[];
    for (uint32_t i = threadIdx.x; i < k_pad; i += 256) s_norm[i] = cnorm[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = (blockIdx.x * 4 + (uint32_t)wave) * 64;
    if (base >= n) return;
    const int col = lane & 31, kb = lane >> 5;
    half8 b[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint32_t u = min(base + (uint32_t)(t * 32 + col), n - 1);
        b[t] = *reinterpret_cast<const half8*>(vec + (size_t)u * KM_DIM + kb * 8);
    }
    float best[2] = {__builtin_inff(), __builtin_inff()};
    uint32_t bi[2] = {0, 0};
    for (uint32_t c0 = 0; c0 < k_pad; c0 += 32) {
        const half8 ah = *reinterpret_cast<const half8*>(hi + (size_t)(c0 + col) * KM_DIM + kb * 8);
        const half8 al = *reinterpret_cast<const half8*>(lo + (size_t)(c0 + col) * KM_DIM + kb * 8);
        float16v init;
#pragma unroll
        for (int r = 0; r < 16; r++) init[r] = s_norm[c0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * kb)];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            float16v acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b[t], init, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[t], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) {   // rows ascend with r inside a lane: strict < keeps the lowest index among equals
                const bool better = acc[r] < best[t];
                best[t] = better ? acc[r] : best[t];
                bi[t] = better ? c0 + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * kb) : bi[t];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {   // the other half-wave saw the other 16 rows of every tile
        const float ob = __shfl_xor(best[t], 32, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi[t], 32, 64);
        if (ob < best[t] || (ob == best[t] && oi < bi[t])) { best[t] = ob; bi[t] = oi; }
    }
    // lanes 0..31 own tile 0's vectors, lanes 32..63 tile 1's
    const int t = kb;
    const uint32_t u = base + (uint32_t)(t * 32 + col);
    if (u >= n) return;
    const uint32_t c = t ? bi[1] : bi[0];
    assign[u] = c;
    const uint64_t w = weights[u];
    const _Float16* row = vec + (size_t)u * KM_DIM;
    float unorm = 0.0f;
    for (int d = 0; d < dims; d++) {
        const uint32_t v = (uint32_t)(float)row[d];
        unorm += (float)(v * v);
        if (v) atomicAdd(&sums[(size_t)c * 17 + d], (unsigned long long)(w * v));
    }
    atomicAdd(&sums[(size_t)c * 17 + 16], (unsigned long long)w);
    if (err_key) { err_key[u] = fmaxf((t ? best[1] : best[0]) + unorm, 0.0f) * (float)w; err_idx[u] = u; }
}

} // namespace

size_t kmeans_workspace_bytes(uint32_t n, uint32_t k) {
    const size_t k_pad = ((size_t)k + 31) / 32 * 32;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t cub = 0, cub2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub, (const float*)nullptr, (float*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, cub2, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n);
    return up((size_t)n * KM_DIM * 2) + up((size_t)n * 8) + 2 * up(k_pad * KM_DIM * 2) + up(k_pad * 4) + up((size_t)k * KM_DIM * 4) + up((size_t)k * 17 * 8) +
           4 * up((size_t)n * 4) + up((size_t)n * 8) + 2 * up((size_t)k * 4 + 4) + up(cub > cub2 ? cub : cub2);
}

kmeans_buffers kmeans_carve(void* ws, uint32_t n, uint32_t k) {
    const size_t k_pad = ((size_t)k + 31) / 32 * 32;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    char* p = static_cast<char*>(ws);
    kmeans_buffers b;
    b.k_pad = (uint32_t)k_pad;
    b.vec = p; p += up((size_t)n * KM_DIM * 2);
    b.weights = reinterpret_cast<uint64_t*>(p); p += up((size_t)n * 8);
    b.hi = p; p += up(k_pad * KM_DIM * 2);
    b.lo = p; p += up(k_pad * KM_DIM * 2);
    b.cnorm = reinterpret_cast<float*>(p); p += up(k_pad * 4);
    b.cen = reinterpret_cast<float*>(p); p += up((size_t)k * KM_DIM * 4);
    b.sums = reinterpret_cast<uint64_t*>(p); p += up((size_t)k * 17 * 8);
    b.err_key = reinterpret_cast<float*>(p); p += up((size_t)n * 4);
    b.err_idx = reinterpret_cast<uint32_t*>(p); p += up((size_t)n * 4);
    b.err_key_sorted = reinterpret_cast<float*>(p); p += up((size_t)n * 4);
    b.worst = reinterpret_cast<uint32_t*>(p); p += up((size_t)n * 4);
    b.cum = reinterpret_cast<uint64_t*>(p); p += up((size_t)n * 8);
    b.pick = reinterpret_cast<uint32_t*>(p); p += up((size_t)k * 4 + 4);
    b.empty = reinterpret_cast<uint32_t*>(p); p += up((size_t)k * 4 + 4);   // last word: the count
    b.cub = p;
    size_t cub = 0, cub2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub, (const float*)nullptr, (float*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, cub2, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n);
    b.cub_bytes = cub > cub2 ? cub : cub2;
    return b;
}

hipError_t launch_kmeans(hipStream_t st, int endpoints, const void* d_keys, const uint64_t* d_weights, const uint32_t* d_goffs, uint32_t n, uint32_t k, uint32_t iterations,
                         const kmeans_buffers& b, uint32_t* d_assign) {
    if (!n || !k) return hipErrorInvalidValue;
    _Float16* vec = static_cast<_Float16*>(b.vec);
    _Float16* hi = static_cast<_Float16*>(b.hi);
    _Float16* lo = static_cast<_Float16*>(b.lo);
    const dim3 gu((n + 255) / 256), blk(256);
    const uint64_t* weights = d_weights;
    if (endpoints) {
        hipLaunchKernelGGL(k_km_unpack_endpoints, gu, blk, 0, st, static_cast<const uint64_t*>(d_keys), d_goffs, n, vec, b.weights);
        weights = b.weights;
    } else hipLaunchKernelGGL(k_km_unpack_selectors, gu, blk, 0, st, static_cast<const uint32_t*>(d_keys), n, vec);
    const dim3 gk((k * KM_DIM + 255) / 256), gp((b.k_pad + 255) / 256);
    size_t bytes = b.cub_bytes;
    hipError_t e = hipcub::DeviceScan::InclusiveSum(b.cub, bytes, weights, b.cum, (int)n, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_km_seed_pick, dim3((k + 255) / 256), blk, 0, st, b.cum, n, k, b.pick);
    hipLaunchKernelGGL(k_km_seed_distinct, dim3(1), dim3(1024), 0, st, b.pick, n, k);
    hipLaunchKernelGGL(k_km_seed, gk, blk, 0, st, vec, b.pick, k, b.cen);
    const size_t lds = (size_t)b.k_pad * 4;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_km_assign), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    const dim3 ga((n + 255) / 256);
    const int dims = endpoints ? 6 : 16;
    for (uint32_t it = 0; it <= iterations; it++) {
        hipLaunchKernelGGL(k_km_prepare, gp, blk, 0, st, b.cen, b.sums, k, b.k_pad, it != 0, hi, lo, b.cnorm);
        if ((e = hipMemsetAsync(b.sums, 0, (size_t)k * 17 * 8, st)) != hipSuccess) return e;
        // the last round only assigns (its sums tell which clusters ended up non-empty)
        const bool more = it < iterations;
        hipLaunchKernelGGL(k_km_assign, ga, blk, lds, st, vec, weights, n, hi, lo, b.cnorm, b.k_pad, d_assign, reinterpret_cast<unsigned long long*>(b.sums), dims,
                           more ? b.err_key : nullptr, b.err_idx);
        if (more) {
            hipLaunchKernelGGL(k_km_update, gk, blk, 0, st, b.sums, k, b.cen);
            // empty clusters move onto the worst represented vectors (weighted error descending, index ascending among equals: the sort is stable)
            bytes = b.cub_bytes;
            if ((e = hipcub::DeviceRadixSort::SortPairsDescending(b.cub, bytes, b.err_key, b.err_key_sorted, b.err_idx, b.worst, (int)n, 0, 32, st)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_km_list_empty, dim3(1), dim3(1024), 0, st, b.sums, k, b.empty, b.empty + k);
            hipLaunchKernelGGL(k_km_reseed, gk, blk, 0, st, vec, b.worst, b.empty, b.empty + k, n, b.cen, b.sums);
        }
    }
    return hipGetLastError();
}

} // namespace bu
