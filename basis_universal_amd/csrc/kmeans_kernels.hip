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
#include <cstdlib>

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

// clusters that lost all their members are moved onto badly represented vectors: every workgroup of the assignment kernel reports the vector of its 512 with
// the largest weighted error, and the empty clusters take those in descending order (a sort of ~n / 512 keys instead of all n)
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
__global__ __launch_bounds__(256) void k_km_reseed(const _Float16* __restrict__ vec, const unsigned long long* __restrict__ worst_sorted, uint32_t n_worst,
                                                   const uint32_t* __restrict__ empty, const uint32_t* __restrict__ n_empty, float* __restrict__ cen, uint64_t* __restrict__ sums) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t m = min(*n_empty, n_worst);
    if (i >= m * KM_DIM) return;
    const uint32_t j = i / KM_DIM, d = i % KM_DIM, c = empty[j];
    const uint32_t u = 0xFFFFFFFFu - (uint32_t)worst_sorted[j];   // the j-th worst of the workgroups' worst represented vectors (largest weighted error first, lowest index among equals)
    cen[(size_t)c * KM_DIM + d] = (float)vec[(size_t)u * KM_DIM + d];
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

// The assignment GEMM. A 256-thread workgroup owns KM_WG_VECS = 512 vectors: each of its four waves keeps KM_T = 4 column tiles (128 vectors) as B operands in
// registers for the whole sweep, and the workgroup walks the centroid tiles (32 rows: 1 KiB of hi + 1 KiB of lo halves + 32 norms) through a double-buffered LDS
// stage, so a centroid tile is fetched from L2 once per 512 vectors and read from LDS once per 128 (8 column tiles per wave spill: measured) (round 2: once per 64 -- 1.9 GB of L2 traffic per assignment round, which bound the kernel).
// Per centroid tile and column tile: two MFMAs (hi, lo) on an accumulator preloaded with |c|^2, then ONLY the minimum of the 16 results a lane sees (v_min3_f32
// trees) and which tile it came from (a compare and two selects): 17 VALU instructions, ~68 cycles against 64 MFMA cycles. WHICH of the
// tile's 32 rows gave the minimum is narrowed to a group of four rows by two tag bits in the key, and the row inside the group is recovered once per vector after
// the sweep by evaluating those four rows directly (256 B of centroid data per vector) -- instead of an index update beside every compare.
constexpr int KM_T = 4;
constexpr uint32_t KM_WG_VECS = 4 * 32 * KM_T;
constexpr int KM_STAGE_TILES = 2;        // centroid tiles per LDS stage = per workgroup barrier (k_pad is a multiple of 32 * KM_STAGE_TILES)
constexpr uint32_t KM_SLOTS = 256;       // LDS accumulator slots of a workgroup (its 512 vectors rarely reach a fifth of that many distinct clusters)
constexpr uint32_t KM_EMPTY = 0xFFFFFFFFu;

struct km_stage { uint4 hi[64 * KM_STAGE_TILES]; uint4 lo[64 * KM_STAGE_TILES]; float norm[32 * KM_STAGE_TILES]; };   // rows x 16 halves = 64 x 16 B per tile; norms permuted (below)

__global__ __launch_bounds__(256, 2) void k_km_assign(const _Float16* __restrict__ vec, const uint64_t* __restrict__ weights, uint32_t n, const _Float16* __restrict__ hi,
                                                   const _Float16* __restrict__ lo, const float* __restrict__ cnorm, uint32_t k_pad, uint32_t* __restrict__ assign,
                                                   unsigned long long* __restrict__ sums, int dims, unsigned long long* __restrict__ wg_worst,
                                                   const uint32_t* __restrict__ packed_ok, uint32_t debug_skip) {
    __shared__ km_stage s_stage[2];
    __shared__ uint32_t s_key[KM_SLOTS];
    __shared__ unsigned long long s_acc[KM_SLOTS][17];
    __shared__ unsigned long long s_worst;   // the workgroup's worst represented vector: (weighted error bits, ~index), see k_km_reseed
    if (threadIdx.x == 0) s_worst = 0ull;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kb = lane >> 5;
    const uint32_t base = blockIdx.x * KM_WG_VECS + (uint32_t)wave * (32u * KM_T);
    s_key[tid] = KM_EMPTY;
    for (uint32_t i = (uint32_t)tid; i < KM_SLOTS * 17; i += 256) (&s_acc[0][0])[i] = 0ull;
    half8 b[KM_T];
#pragma unroll
    for (int t = 0; t < KM_T; t++) {
        const uint32_t u = min(base + (uint32_t)(t * 32 + col), n - 1);
        b[t] = *reinterpret_cast<const half8*>(vec + (size_t)u * KM_DIM + kb * 8);
    }
    float best[KM_T];
    uint32_t btile[KM_T];
#pragma unroll
    for (int t = 0; t < KM_T; t++) { best[t] = __builtin_inff(); btile[t] = 0; }

    // staging: threads 0..127 fetch the hi halves of a stage's tiles (16 B each), 128..255 the lo halves, threads 0..63 also the norms, permuted so that the
    // 16 rows a lane's accumulator registers hold (row = (r & 3) + 8 (r >> 2) + 4 kb) are 16 consecutive floats: norm[tile * 32 + kb * 16 + r]
    const uint32_t stages = k_pad / (32 * KM_STAGE_TILES);
    uint4 pre = make_uint4(0, 0, 0, 0);
    float pre_n = 0.0f;
    auto fetch = [&](uint32_t stage) {
        const size_t first_row = (size_t)stage * 32 * KM_STAGE_TILES;
        if (tid < 128) pre = reinterpret_cast<const uint4*>(hi + first_row * KM_DIM)[tid];
        else pre = reinterpret_cast<const uint4*>(lo + first_row * KM_DIM)[tid - 128];
        if (tid < 64) { const int tl = tid >> 5, i = tid & 31, r = i & 15, kk = i >> 4; pre_n = cnorm[first_row + (size_t)(tl * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk)]; }
    };
    auto stash = [&](int buf) {
        if (tid < 128) s_stage[buf].hi[tid] = pre;
        else s_stage[buf].lo[tid - 128] = pre;
        if (tid < 64) s_stage[buf].norm[tid] = pre_n;
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (uint32_t stage = 0; stage < ((debug_skip & 2u) ? 1u : stages); stage++) {   // bit 1 of BU_KM_DEBUG_SKIP: one stage only (timing experiments)
        const int buf = (int)(stage & 1u);
        if (stage + 1 < stages) fetch(stage + 1);
        const km_stage& S = s_stage[buf];
#pragma unroll
        for (int tl = 0; tl < KM_STAGE_TILES; tl++) {
            const uint32_t tile = stage * KM_STAGE_TILES + (uint32_t)tl;
            // A operand of lane (col, kb): row `col` of the tile, halves kb*8 .. kb*8+7 = the (col * 2 + kb)-th 16-byte piece
            const uint4 ahv = S.hi[tl * 64 + col * 2 + kb], alv = S.lo[tl * 64 + col * 2 + kb];
            half8 ah, al;
            __builtin_memcpy(&ah, &ahv, 16); __builtin_memcpy(&al, &alv, 16);
            float16v init;
            {
                const float4* nv = reinterpret_cast<const float4*>(&S.norm[tl * 32 + kb * 16]);
#pragma unroll
                for (int q = 0; q < 4; q++) { const float4 v = nv[q]; init[q * 4 + 0] = v.x; init[q * 4 + 1] = v.y; init[q * 4 + 2] = v.z; init[q * 4 + 3] = v.w; }
            }
            // the four hi products first, then the four lo products: an accumulator is touched again only four MFMAs (128 cycles) later, so the matrix pipe
            // never waits for its own result, and the reductions of column tile t run beside the MFMAs of the tiles after it
#pragma unroll
            for (int g = 0; g < KM_T; g += 4) {   // four column tiles at a time share the accumulator registers
                float16v acc[4];
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b[g + t], init, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b[g + t], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    // the minimum of each group of four consecutive rows (registers 4q .. 4q+3), tagged with q in the two lowest mantissa bits (a 2^-21
                    // relative perturbation of a comparison key), then the minimum of the four tagged values: 8 + 4 + 2 VALU instructions for 16 results
                    float key[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float m4 = __builtin_fminf(__builtin_fminf(__builtin_fminf(acc[t][4 * q], acc[t][4 * q + 1]), acc[t][4 * q + 2]), acc[t][4 * q + 3]);
                        key[q] = __uint_as_float((__float_as_uint(m4) & ~3u) | (uint32_t)q);
                    }
                    const float m = __builtin_fminf(__builtin_fminf(__builtin_fminf(key[0], key[1]), key[2]), key[3]);
                    const bool better = m < best[g + t];   // strict: the earlier tile keeps a tie
                    best[g + t] = better ? m : best[g + t];
                    btile[g + t] = better ? tile : btile[g + t];
                }
            }
        }
        if (stage + 1 < stages) stash(buf ^ 1);
        __syncthreads();
    }
    // the other half-wave saw the other 16 rows of every tile: the smaller key decides, the earlier tile a tie, then the lower half
    uint32_t bhalf[KM_T];
#pragma unroll
    for (int t = 0; t < KM_T; t++) {
        const float ob = __shfl_xor(best[t], 32, 64);
        const uint32_t ot = (uint32_t)__shfl_xor((int)btile[t], 32, 64);
        const bool other = ob < best[t] || (ob == best[t] && (ot < btile[t] || (ot == btile[t] && kb == 1)));
        bhalf[t] = other ? (uint32_t)(kb ^ 1) : (uint32_t)kb;
        if (other) { best[t] = ob; btile[t] = ot; }
    }
    // lanes with kb = 0 finish the first half of the column tiles, lanes with kb = 1 the second half. The key names the winning tile, half and group of four rows
    // (rows 8 q + 4 half .. + 3 of the tile); WHICH of the four it is comes from evaluating them directly, lowest row first on ties.
    // The centroid update's sums go through the workgroup's LDS accumulators first: one slot per distinct cluster (open addressing on the cluster id), LDS
    // atomics per vector, and ONE global atomic per slot and word at the end -- a tenth of the global atomics of "every vector adds its components".
    const bool packed = *packed_ok != 0;
#pragma unroll
    for (int h = 0; h < KM_T / 2; h++) {
        const int t = kb * (KM_T / 2) + h;
        uint32_t tile_w = 0, half_w = 0, q_w = 0;
#pragma unroll
        for (int tt = 0; tt < KM_T; tt++) {   // selects: the arrays stay in registers
            tile_w = (tt == t) ? btile[tt] : tile_w;
            half_w = (tt == t) ? bhalf[tt] : half_w;
            q_w = (tt == t) ? (__float_as_uint(best[tt]) & 3u) : q_w;
        }
        const uint32_t u = base + (uint32_t)(t * 32 + col);
        if (u < n) {
            const _Float16* row = vec + (size_t)u * KM_DIM;
            float uf[KM_DIM];
            {
                const half8 r0 = *reinterpret_cast<const half8*>(row), r1 = *reinterpret_cast<const half8*>(row + 8);
#pragma unroll
                for (int d = 0; d < 8; d++) { uf[d] = (float)r0[d]; uf[8 + d] = (float)r1[d]; }
            }
            float bd = __builtin_inff();
            const uint32_t first_row = tile_w * 32 + 8 * q_w + 4 * half_w;
            uint32_t bc = first_row;
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t c = first_row + r;
                const bool live = cnorm[c] < __builtin_inff();   // not a row past k, nor an empty cluster
                const half8 h0 = *reinterpret_cast<const half8*>(hi + (size_t)c * KM_DIM), h1 = *reinterpret_cast<const half8*>(hi + (size_t)c * KM_DIM + 8);
                const half8 l0 = *reinterpret_cast<const half8*>(lo + (size_t)c * KM_DIM), l1 = *reinterpret_cast<const half8*>(lo + (size_t)c * KM_DIM + 8);
                float dist = 0.0f;
#pragma unroll
                for (int d = 0; d < 8; d++) {
                    const float c0 = -0.5f * ((float)h0[d] + (float)l0[d]), c1 = -0.5f * ((float)h1[d] + (float)l1[d]);   // the centroid the GEMM uses
                    const float e0 = uf[d] - c0, e1 = uf[8 + d] - c1;
                    dist += e0 * e0; dist += e1 * e1;
                }
                if (live && dist < bd) { bd = dist; bc = c; }   // ascending rows, strict: the lowest index among equals
            }
            assign[u] = bc;
            const uint64_t w = weights[u];
            if (wg_worst) atomicMax(&s_worst, ((unsigned long long)__float_as_uint(bd * (float)w) << 32) | (unsigned long long)(0xFFFFFFFFu - u));   // non-negative floats order like their bits
            if (!(debug_skip & 1u)) {   // bit 0 of BU_KM_DEBUG_SKIP: no accumulation (timing experiments)
                // the workgroup's slot of cluster bc, or none when the table is full (then straight to memory)
                uint32_t slot = (bc * 2654435761u) >> 24;
                bool have = false;
                for (uint32_t tries = 0; tries < KM_SLOTS; tries++) {
                    const uint32_t old = atomicCAS(&s_key[slot], KM_EMPTY, bc);
                    if (old == KM_EMPTY || old == bc) { have = true; break; }
                    slot = (slot + 1) & (KM_SLOTS - 1);
                }
                unsigned long long* dst = have ? &s_acc[slot][0] : &sums[(size_t)bc * 17];
                if (packed) {   // two components per 64-bit accumulator: every 32-bit half stays below 2^32 (k_km_flags checked total weight x largest value)
#pragma unroll
                    for (int d = 0; d < KM_DIM; d += 2) {   // fixed trip count + predicate: a run-time bound would index uf[] dynamically (scratch)
                        const uint64_t v0 = (uint64_t)(uint32_t)uf[d], v1 = (uint64_t)(uint32_t)uf[d + 1];
                        if (d < dims && (v0 | v1)) atomicAdd(dst + d, (unsigned long long)((w * v0) | ((w * v1) << 32)));
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < KM_DIM; d++) {
                        const uint32_t v = (uint32_t)uf[d];
                        if (d < dims && v) atomicAdd(dst + d, (unsigned long long)(w * v));
                    }
                }
                atomicAdd(dst + 16, (unsigned long long)w);
            }
        }
    }
    __syncthreads();
    if (wg_worst && tid == 0) wg_worst[blockIdx.x] = s_worst;
    {
        const uint32_t c = s_key[tid];   // one slot per thread
        if (c != KM_EMPTY) {
#pragma unroll
            for (int d = 0; d < 17; d++) {
                const unsigned long long v = s_acc[tid][d];
                if (v) atomicAdd(&sums[(size_t)c * 17 + d], v);
            }
        }
    }
}

// may two component sums share one 64-bit accumulator? (total weight x the largest component value must stay below 2^32)
__global__ void k_km_flags(const uint64_t* __restrict__ cum, uint32_t n, uint32_t max_value, uint32_t* __restrict__ packed_ok) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *packed_ok = ((unsigned __int128)cum[n - 1] * max_value < ((unsigned __int128)1 << 32)) ? 1u : 0u;
}

// the packed accumulators back into one sum per component (what k_km_update and the callers read)
__global__ __launch_bounds__(256) void k_km_unpack_sums(uint64_t* __restrict__ sums, uint32_t k, const uint32_t* __restrict__ packed_ok) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k * 8 || !*packed_ok) return;
    const uint32_t c = i >> 3, d = (i & 7u) * 2;
    const uint64_t v = sums[(size_t)c * 17 + d];
    sums[(size_t)c * 17 + d] = v & 0xffffffffull;
    sums[(size_t)c * 17 + d + 1] = v >> 32;
}

} // namespace

// temporary storage of the two library calls launch_kmeans makes, sized from THOSE calls: the 64-bit key sort of one (error, index) key per assignment workgroup
// and the inclusive weight scan of the seeding
static size_t kmeans_cub_bytes(uint32_t n) {
    size_t sort_bytes = 0, scan_bytes = 0;
    const int groups = (int)((n + KM_WG_VECS - 1) / KM_WG_VECS);
    (void)hipcub::DeviceRadixSort::SortKeysDescending(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, groups > 0 ? groups : 1, 0, 64);
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)n);
    return sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
}

size_t kmeans_workspace_bytes(uint32_t n, uint32_t k) {
    const size_t k_pad = ((size_t)k + 63) / 64 * 64;   // whole LDS stages of the assignment kernel (two centroid tiles)
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t cub = kmeans_cub_bytes(n);
    return up((size_t)n * KM_DIM * 2) + up((size_t)n * 8) + 2 * up(k_pad * KM_DIM * 2) + up(k_pad * 4) + up((size_t)k * KM_DIM * 4) + up((size_t)k * 17 * 8) +
           4 * up((size_t)n * 4) + up((size_t)n * 8) + up((size_t)k * 4 + 4) + up((size_t)k * 4 + 8) + up(cub);
}

kmeans_buffers kmeans_carve(void* ws, uint32_t n, uint32_t k) {
    const size_t k_pad = ((size_t)k + 63) / 64 * 64;   // whole LDS stages of the assignment kernel (two centroid tiles)
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
    b.empty = reinterpret_cast<uint32_t*>(p); p += up((size_t)k * 4 + 8);   // behind the list: the count, then launch_kmeans' "packed sums" flag
    b.cub = p;
    b.cub_bytes = kmeans_cub_bytes(n);
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
    const dim3 ga((n + KM_WG_VECS - 1) / KM_WG_VECS);
    const int dims = endpoints ? 6 : 16;
    uint32_t* packed_ok = b.empty + k + 1;   // one more word behind the empty-cluster count
    unsigned long long* wg_worst = reinterpret_cast<unsigned long long*>(b.err_key);            // one key per workgroup of k_km_assign (the buffers hold n floats each)
    unsigned long long* wg_worst_sorted = reinterpret_cast<unsigned long long*>(b.err_key_sorted);
    static const uint32_t debug_skip = [] { const char* e = std::getenv("BU_KM_DEBUG_SKIP"); return e ? (uint32_t)std::atoi(e) : 0u; }();
    hipLaunchKernelGGL(k_km_flags, dim3(1), dim3(64), 0, st, b.cum, n, endpoints ? 255u : 3u, packed_ok);
    for (uint32_t it = 0; it <= iterations; it++) {
        hipLaunchKernelGGL(k_km_prepare, gp, blk, 0, st, b.cen, b.sums, k, b.k_pad, it != 0, hi, lo, b.cnorm);
        if ((e = hipMemsetAsync(b.sums, 0, (size_t)k * 17 * 8, st)) != hipSuccess) return e;
        // the last round only assigns (its sums tell which clusters ended up non-empty)
        const bool more = it < iterations;
        hipLaunchKernelGGL(k_km_assign, ga, blk, 0, st, vec, weights, n, hi, lo, b.cnorm, b.k_pad, d_assign, reinterpret_cast<unsigned long long*>(b.sums), dims,
                           more ? wg_worst : nullptr, packed_ok, debug_skip);
        hipLaunchKernelGGL(k_km_unpack_sums, dim3((k * 8 + 255) / 256), blk, 0, st, b.sums, k, packed_ok);
        if (more) {
            hipLaunchKernelGGL(k_km_update, gk, blk, 0, st, b.sums, k, b.cen);
            // empty clusters move onto the workgroups' worst represented vectors (weighted error descending, index ascending among equals: the key order)
            bytes = b.cub_bytes;
            if ((e = hipcub::DeviceRadixSort::SortKeysDescending(b.cub, bytes, wg_worst, wg_worst_sorted, (int)ga.x, 0, 64, st)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_km_list_empty, dim3(1), dim3(1024), 0, st, b.sums, k, b.empty, b.empty + k);
            hipLaunchKernelGGL(k_km_reseed, gk, blk, 0, st, vec, wg_worst_sorted, ga.x, b.empty, b.empty + k, b.cen, b.sums);
        }
    }
    return hipGetLastError();
}

} // namespace bu
