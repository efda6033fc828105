// etc1s_kernels.hip -- hand-written gfx950 kernels for the ETC1S frontend hot path.
//
// All kernels are integer-ALU bound (hundreds of integer ops per byte of pixel data), so the design rules are:
//   * wave64 mappings that keep all 64 lanes on the SAME kind of work -- and where a wave's blocks need different NUMBERS of steps (the per-block fit's trials), the
//     cheap part (finding the next step worth taking) loops per block while the expensive part (evaluating it) runs for all blocks at once;
//   * the 64-byte pixel tile is read once per kernel with 16-byte loads and kept in registers in the metric's separable
//     basis (etc1s_device.h: cvec), so a colour distance is 3 subtractions + 3 24-bit multiplies + shifts;
//   * wavefront reductions (DPP/ds_swizzle via __shfl_xor) pick the best intensity table / candidate, with the
//     reference's tie rules encoded in the reduction key (lowest index wins on equal error);
//   * candidate codebooks are read through the scalar/vector caches (they are KB-sized and shared by all lanes).
// Result parity with the reference CPU encoder is bit-exact; each kernel cites the code it restates.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "etc1s_device.h"
#include "etc1s_tables.inc"
#include "etc1s_kernels.h"

namespace bu {

__device__ __constant__ static unsigned int c_cluster_fit_order[165];
__device__ __constant__ static unsigned char c_inten_enable_by_spread[256];
static bool g_tables_uploaded[16] = {};

hipError_t upload_etc1s_tables(int device) {
    if (device >= 0 && device < 16 && g_tables_uploaded[device]) return hipSuccess;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_cluster_fit_order), k_cluster_fit_order, sizeof(k_cluster_fit_order));
    if (e != hipSuccess) return e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_inten_enable_by_spread), k_inten_enable_by_spread, sizeof(k_inten_enable_by_spread));
    if (e != hipSuccess) return e;
    if (device >= 0 && device < 16) g_tables_uploaded[device] = true;
    return hipSuccess;
}

// -------------------------------------------------------------------------------------------------------------------
// Shared pieces of etc1_optimizer (etc.cpp:948-1278)
// -------------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t perms_for_quality(int quality) {
    return quality == BU_Q_FAST ? 4u : quality == BU_Q_MEDIUM ? 16u : quality == BU_Q_SLOW ? 64u : 165u; // etc.cpp:792-800
}

// m_br/m_bg/m_bb (etc.cpp:1047-1049): round(avg * 31 / 255), float ops in this exact order, no contraction.
__device__ __forceinline__ int avg_to_color5(float avg) {
    const float t = avg * 31.0f;
    const float q = t / 255.0f;
    const float r = q + 0.5f;
    return min(max((int)(uint32_t)r, 0), 31);
}

// One cluster-fit trial colour (etc.cpp:958-986) from the current best solution and selector histogram `hist`.
// Returns false when all three delta sums are zero (the trial is skipped).
__device__ __forceinline__ bool cluster_fit_trial(uint32_t hist, int best_r5, int best_g5, int best_b5, int best_inten,
                                                  float avg_r, float avg_g, float avg_b, int& tr, int& tg, int& tb) {
    const int base_r = scale5(best_r5), base_g = scale5(best_g5), base_b = scale5(best_b5);
    int dr = 0, dg = 0, db = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int cnt = (int)((hist >> (8 * q)) & 255u);
        const int yd = inten_delta(best_inten, q);
        dr += cnt * (clamp255(base_r + yd) - base_r);
        dg += cnt * (clamp255(base_g + yd) - base_g);
        db += cnt * (clamp255(base_b + yd) - base_b);
    }
    if (!(dr | dg | db)) return false;
    const float fr = (float)dr / 8.0f, fg = (float)dg / 8.0f, fb = (float)db / 8.0f;
    {
        const float a = avg_r - fr; const float m = a * 31.0f; const float q = m / 255.0f; const float r = q + 0.5f;
        tr = min(max((int)r, 0), 31);
    }
    {
        const float a = avg_g - fg; const float m = a * 31.0f; const float q = m / 255.0f; const float r = q + 0.5f;
        tg = min(max((int)r, 0), 31);
    }
    {
        const float a = avg_b - fb; const float m = a * 31.0f; const float q = m / 255.0f; const float r = q + 0.5f;
        tb = min(max((int)r, 0), 31);
    }
    return true;
}

// check_for_redundant_solution (etc.cpp:1072-1089) on a 1024-bit filter stored as 32 dwords. Returns true if the colour
// is definitely new (and inserts it). Must be called by exactly one lane per filter, or by lanes that all see the same
// state and write the same value.
__device__ __forceinline__ bool bloom_test_and_set(uint32_t* filter, int r5, int g5, int b5) {
    const uint32_t kh = hash_hsieh3((uint32_t)r5, (uint32_t)g5, (uint32_t)b5);
    const uint32_t h0 = kh & 1023u, h1 = (kh >> 10) & 1023u;
    const uint32_t w0 = filter[h0 >> 5], w1 = filter[h1 >> 5];
    const uint32_t m0 = 1u << (h0 & 31), m1 = 1u << (h1 & 31);
    if ((w0 & m0) && (w1 & m1)) return false;
    if ((h0 >> 5) == (h1 >> 5)) {
        filter[h0 >> 5] = w0 | m0 | m1;
    } else {
        filter[h0 >> 5] = w0 | m0;
        filter[h1 >> 5] = w1 | m1;
    }
    return true;
}

// -------------------------------------------------------------------------------------------------------------------
// a6: init_etc1_images -- per 4x4 block etc1_optimizer (frontend.cpp:765-818; etc.cpp:776-1278)
//
// Mapping: 8 lanes per block (lane = intensity table), 8 blocks per wave, 32 blocks per 256-thread workgroup.
// Every lane keeps the block's 16 pixels in registers; a trial base colour costs each lane one pass over 16 pixels x 4
// selectors for ITS table; the best table is an 8-lane min-reduction on key (error << 3 | table), which reproduces the
// reference's ascending-table strict-< scan. Trial colours depend on the running best solution, so trials are serial.
// -------------------------------------------------------------------------------------------------------------------

template <bool PERCEPTUAL, int QUALITY>
__global__ __launch_bounds__(256) void k_encode_etc1s_blocks(const uint4* __restrict__ pixel_blocks, uint32_t n_blocks, uint2* __restrict__ out_blocks) {
    __shared__ uint32_t s_bloom[32][32];

    const uint32_t tid = threadIdx.x;
    const uint32_t table = tid & 7u;
    const uint32_t slot = tid >> 3;
    const uint32_t group_shift = tid & 56u; // where this block's 8 lanes sit in a wave-wide ballot
    const uint32_t block_raw = blockIdx.x * 32u + slot;
    const bool in_range = block_raw < n_blocks;
    const uint32_t block = in_range ? block_raw : (n_blocks - 1);

    // clear this block's Bloom filter (8 lanes x 4 dwords)
#pragma unroll
    for (int i = 0; i < 4; i++) s_bloom[slot][table * 4 + i] = 0;

    // 64-byte tile: four 16-byte loads, identical for the 8 lanes of a block (served by one cache line)
    uint32_t px[16];
    {
        const uint4* src = pixel_blocks + (size_t)block * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = src[i];
            px[i * 4 + 0] = v.x; px[i * 4 + 1] = v.y; px[i * 4 + 2] = v.z; px[i * 4 + 3] = v.w;
        }
    }

    // etc1_optimizer::init (etc.cpp:998-1070)
    cvec pc[16];
    float sum_r = 0.0f, sum_g = 0.0f, sum_b = 0.0f;
    int mn_r = 255, mn_g = 255, mn_b = 255, mx_r = 0, mx_g = 0, mx_b = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = px[i] & 255, g = (px[i] >> 8) & 255, b = (px[i] >> 16) & 255;
        mn_r = min(mn_r, r); mn_g = min(mn_g, g); mn_b = min(mn_b, b);
        mx_r = max(mx_r, r); mx_g = max(mx_g, g); mx_b = max(mx_b, b);
        sum_r += (float)r; sum_g += (float)g; sum_b += (float)b;
        pc[i] = to_cvec<PERCEPTUAL>(r, g, b);
    }
    const float avg_r = sum_r / 16.0f, avg_g = sum_g / 16.0f, avg_b = sum_b / 16.0f;
    const int spread = max(max(mx_r - mn_r, mx_g - mn_g), mx_b - mn_b);
    const bool table_enabled = (QUALITY > BU_Q_MEDIUM) ? true : (((uint32_t)c_inten_enable_by_spread[spread] >> table) & 1u) != 0; // etc.cpp:1135-1140

    uint32_t best_err = 0xFFFFFFFFu; // every real total is < 2^28
    int best_r = 0, best_g = 0, best_b = 0, best_inten = 0;

    __syncthreads(); // filters cleared

    // Trials: every block at its own pace, eight trial colours per generation (see k_encode_etc1s_blocks_by_pixel below, which explains the scheme)
    const int perms = (int)perms_for_quality(QUALITY);
    int batch_base = 0, mine_r = 0, mine_g = 0, mine_b = 0;
    bool mine_ok = false, batch_fresh = false;
    uint32_t mine_h0 = 0, mine_h1 = 0;
    int next = -1;                    // the first trial index this block has not dealt with; -1 = the average colour (etc.cpp:1047-1049)
    bool done = false;

    for (;;) {
        int pick = -1;                // lane of the block's batch whose trial it evaluates now
        if (next < 0) {
            pick = 0; next = 0;
            mine_r = avg_to_color5(avg_r); mine_g = avg_to_color5(avg_g); mine_b = avg_to_color5(avg_b);
            const uint32_t kh = hash_hsieh3((uint32_t)mine_r, (uint32_t)mine_g, (uint32_t)mine_b);
            mine_h0 = kh & 1023u; mine_h1 = (kh >> 10) & 1023u;
        } else {
            bool searching = !done;
            while (__any(searching)) {
                if (searching && !batch_fresh) {
                    batch_base = next; batch_fresh = true;
                    const int idx = batch_base + (int)table;
                    mine_ok = idx < perms &&
                              cluster_fit_trial(c_cluster_fit_order[min(idx, perms - 1)], best_r, best_g, best_b, best_inten, avg_r, avg_g, avg_b, mine_r, mine_g, mine_b);
                    const uint32_t kh = hash_hsieh3((uint32_t)mine_r, (uint32_t)mine_g, (uint32_t)mine_b);
                    mine_h0 = kh & 1023u; mine_h1 = (kh >> 10) & 1023u;
                }
                bool fresh_colour = false;
                if (searching && mine_ok && batch_base + (int)table >= next) {
                    const uint32_t w0 = s_bloom[slot][mine_h0 >> 5], w1 = s_bloom[slot][mine_h1 >> 5];
                    fresh_colour = !(((w0 >> (mine_h0 & 31u)) & 1u) && ((w1 >> (mine_h1 & 31u)) & 1u));
                }
                const uint32_t m8 = (uint32_t)(__ballot(fresh_colour) >> group_shift) & 0xFFu;
                if (searching) {
                    if (m8) {
                        pick = __ffs((int)m8) - 1;
                        next = batch_base + pick + 1;
                        searching = false;
                    } else {
                        next = batch_base + 8; batch_fresh = false;
                        if (next >= perms) { done = true; searching = false; }
                    }
                }
            }
        }
        if (__all(done)) break;
        const bool active = pick >= 0;
        const int src_lane = active ? pick : 0;
        const int tr = __shfl(mine_r, src_lane, 8), tg = __shfl(mine_g, src_lane, 8), tb = __shfl(mine_b, src_lane, 8);
        if (active) {
            // check_for_redundant_solution's insertion (etc.cpp:1072-1089): the 8 lanes write the same values
            const uint32_t h0 = (uint32_t)__shfl((int)mine_h0, src_lane, 8), h1 = (uint32_t)__shfl((int)mine_h1, src_lane, 8);
            atomicOr(&s_bloom[slot][h0 >> 5], 1u << (h0 & 31u));
            atomicOr(&s_bloom[slot][h1 >> 5], 1u << (h1 & 31u));
        }
        {
            // evaluate_solution_slow (etc.cpp:1104-1278): this lane's table only
            uint32_t total = 0x0FFFFFFFu;
            if (active && table_enabled) {
                cvec bc[4];
                block_cvecs<PERCEPTUAL>(bc, scale5(tr), scale5(tg), scale5(tb), (int)table);
                total = 0;
#pragma unroll
                for (int p = 0; p < 16; p++) total += min_err4<PERCEPTUAL>(pc[p], bc);
            }
            uint32_t key = (total << 3) | table;
            key = min(key, (uint32_t)__shfl_xor((int)key, 1, 8));
            key = min(key, (uint32_t)__shfl_xor((int)key, 2, 8));
            key = min(key, (uint32_t)__shfl_xor((int)key, 4, 8));
            const uint32_t trial_err = key >> 3;
            if (active && trial_err < best_err) {
                best_err = trial_err; best_inten = (int)(key & 7u);
                best_r = tr; best_g = tg; best_b = tb;
                batch_fresh = false;              // the trials after this one start from the new best solution
            }
        }
        if (best_err == 0 || next >= perms) done = true; // etc.cpp:955-956, 993-994
    }

    // Selectors of the winning (colour, table): each of the 8 lanes classifies 2 pixels, first-min over s (etc.cpp:1188-1219).
    cvec bc[4];
    block_cvecs<PERCEPTUAL>(bc, scale5(best_r), scale5(best_g), scale5(best_b), best_inten);
    uint32_t bits = 0;
#pragma unroll
    for (int p = 0; p < 16; p++) {
        if ((uint32_t)(p >> 1) == table) {
            const uint32_t s = best_sel4<PERCEPTUAL>(pc[p], bc);
            bits |= selector_bits((uint32_t)(p & 3), (uint32_t)(p >> 2), s);
        }
    }
    bits |= (uint32_t)__shfl_xor((int)bits, 1, 8);
    bits |= (uint32_t)__shfl_xor((int)bits, 2, 8);
    bits |= (uint32_t)__shfl_xor((int)bits, 4, 8);
    if (table == 0 && in_range) {
        const uint64_t v = etc1s_header_bits((uint32_t)best_r, (uint32_t)best_g, (uint32_t)best_b, (uint32_t)best_inten) | bits;
        const uint64_t m = bswap64(v);
        out_blocks[block] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    }
}

// The same for the perceptual metric with the lanes turned by ninety degrees: the 8 lanes of a block each own TWO PIXELS and
// walk all the enabled tables. A trial colour whose table needs no clamping then costs a lane one chroma term per pixel
// (shared by all such tables) and one luma minimum per pixel and table (etc1s_device.h, base_unclamped) instead of four full
// distances; clamped tables take the four-distance form as before. The per-table totals of the 8 lanes meet in a three-step
// exchange that leaves lane l with the complete total of one table, and from there on the reduction is the one above.
//
// Trials. Most of a block's 1 + perms trial colours are ones it has seen (check_for_redundant_solution): 4.2 of 17 are
// evaluated per block of the bench image, but WHICH ones differs from block to block, and a wave that steps its 8 blocks
// through the trial indices together evaluates the union (10.4 of 17). Here every block moves at its own pace: its 8 lanes
// make the next EIGHT trial colours from the current best solution at once (lane l: trial next + l), the first of them
// the filter does not know is evaluated, and only an evaluation that improves the best solution -- the one thing later
// trial colours depend on -- makes the lanes generate again. A trial the filter knew when it was looked at stays known
// (the filter only grows), a trial is entered into the filter when it is evaluated and not before: the sequence of
// (colour, filter state) pairs is the reference's. A wave evaluates max-over-blocks trials (5.2) instead of the union.
template <int QUALITY>
__global__ __launch_bounds__(256) void k_encode_etc1s_blocks_by_pixel(const uint4* __restrict__ pixel_blocks, uint32_t n_blocks, uint2* __restrict__ out_blocks) {
    __shared__ uint32_t s_bloom[32][32];

    const uint32_t tid = threadIdx.x;
    const uint32_t sub = tid & 7u;        // pixels 2 sub, 2 sub + 1
    const uint32_t slot = tid >> 3;
    const uint32_t group_shift = tid & 56u; // where this block's 8 lanes sit in a wave-wide ballot
    const uint32_t block_raw = blockIdx.x * 32u + slot;
    const bool in_range = block_raw < n_blocks;
    const uint32_t block = in_range ? block_raw : (n_blocks - 1);
#pragma unroll
    for (int i = 0; i < 4; i++) s_bloom[slot][sub * 4 + i] = 0;

    // etc1_optimizer::init (etc.cpp:998-1070): every lane over all 16 pixels (the float sums must run in pixel order)
    float sum_r = 0.0f, sum_g = 0.0f, sum_b = 0.0f;
    int mn_r = 255, mn_g = 255, mn_b = 255, mx_r = 0, mx_g = 0, mx_b = 0;
    {
        const uint4* src = pixel_blocks + (size_t)block * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = src[i];
            const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int r = w4[k] & 255, g = (w4[k] >> 8) & 255, b = (w4[k] >> 16) & 255;
                mn_r = min(mn_r, r); mn_g = min(mn_g, g); mn_b = min(mn_b, b);
                mx_r = max(mx_r, r); mx_g = max(mx_g, g); mx_b = max(mx_b, b);
                sum_r += (float)r; sum_g += (float)g; sum_b += (float)b;
            }
        }
    }
    const uint2 mine = reinterpret_cast<const uint2*>(pixel_blocks + (size_t)block * 4)[sub];
    const cvec pc0 = pixel_cvec<true>(mine.x), pc1 = pixel_cvec<true>(mine.y);
    const float avg_r = sum_r / 16.0f, avg_g = sum_g / 16.0f, avg_b = sum_b / 16.0f;
    const int spread = max(max(mx_r - mn_r, mx_g - mn_g), mx_b - mn_b);
    const uint32_t enable_mask = (QUALITY > BU_Q_MEDIUM) ? 0xFFu : (uint32_t)c_inten_enable_by_spread[spread]; // etc.cpp:1135-1140
    // the table this lane ends up holding the total of (see the exchange below)
    const uint32_t my_table = ((sub & 1u) << 2) | (sub & 2u) | ((sub >> 2) & 1u);

    uint32_t best_err = 0xFFFFFFFFu; // every real total is < 2^28
    int best_r = 0, best_g = 0, best_b = 0, best_inten = 0;

    __syncthreads(); // filters cleared

    const int perms = (int)perms_for_quality(QUALITY);
    // what this lane holds of the block's current batch of trials: trial batch_base + sub, made from the best solution as it was then
    int batch_base = 0, mine_r = 0, mine_g = 0, mine_b = 0;
    bool mine_ok = false, batch_fresh = false;
    uint32_t mine_h0 = 0, mine_h1 = 0;
    int next = -1;                    // the first trial index this block has not dealt with; -1 = the average colour (etc.cpp:1047-1049)
    bool done = false;

    for (;;) {
        // ---- which trial does each block evaluate next? (blocks that find none are done)
        int pick = -1;                // lane of the block's batch whose trial it is
        if (next < 0) {
            pick = 0; next = 0;
            mine_r = avg_to_color5(avg_r); mine_g = avg_to_color5(avg_g); mine_b = avg_to_color5(avg_b);
            const uint32_t kh = hash_hsieh3((uint32_t)mine_r, (uint32_t)mine_g, (uint32_t)mine_b);
            mine_h0 = kh & 1023u; mine_h1 = (kh >> 10) & 1023u;
        } else {
            bool searching = !done;
            while (__any(searching)) {
                if (searching && !batch_fresh) {
                    batch_base = next; batch_fresh = true;
                    const int idx = batch_base + (int)sub;
                    mine_ok = idx < perms &&
                              cluster_fit_trial(c_cluster_fit_order[min(idx, perms - 1)], best_r, best_g, best_b, best_inten, avg_r, avg_g, avg_b, mine_r, mine_g, mine_b);
                    const uint32_t kh = hash_hsieh3((uint32_t)mine_r, (uint32_t)mine_g, (uint32_t)mine_b);
                    mine_h0 = kh & 1023u; mine_h1 = (kh >> 10) & 1023u;
                }
                bool fresh_colour = false;
                if (searching && mine_ok && batch_base + (int)sub >= next) {
                    const uint32_t w0 = s_bloom[slot][mine_h0 >> 5], w1 = s_bloom[slot][mine_h1 >> 5];
                    fresh_colour = !(((w0 >> (mine_h0 & 31u)) & 1u) && ((w1 >> (mine_h1 & 31u)) & 1u));
                }
                const uint32_t m8 = (uint32_t)(__ballot(fresh_colour) >> group_shift) & 0xFFu;
                if (searching) {
                    if (m8) {
                        pick = __ffs((int)m8) - 1;
                        next = batch_base + pick + 1;
                        searching = false;
                    } else {
                        next = batch_base + 8; batch_fresh = false;
                        if (next >= perms) { done = true; searching = false; }
                    }
                }
            }
        }
        if (__all(done)) break;
        const bool active = pick >= 0;
        const int src_lane = active ? pick : 0;
        const int tr = __shfl(mine_r, src_lane, 8), tg = __shfl(mine_g, src_lane, 8), tb = __shfl(mine_b, src_lane, 8);
        if (active) {
            // check_for_redundant_solution's insertion (etc.cpp:1072-1089): the 8 lanes write the same values
            const uint32_t h0 = (uint32_t)__shfl((int)mine_h0, src_lane, 8), h1 = (uint32_t)__shfl((int)mine_h1, src_lane, 8);
            atomicOr(&s_bloom[slot][h0 >> 5], 1u << (h0 & 31u));
            atomicOr(&s_bloom[slot][h1 >> 5], 1u << (h1 & 31u));
        }
        // evaluate_solution_slow (etc.cpp:1104-1278): this lane's two pixels against every enabled table
        const int br = scale5(tr), bg = scale5(tg), bb = scale5(tb);
        const cvec base_cv = to_cvec<true>(br, bg, bb);
        const uint32_t todo = active ? enable_mask : 0u;
        const uint32_t ch0 = chroma_term(pc0.y - base_cv.y, pc0.z - base_cv.z), ch1 = chroma_term(pc1.y - base_cv.y, pc1.z - base_cv.z);
        const int dx0 = pc0.x - base_cv.x, dx1 = pc1.x - base_cv.x;
        const int base_mn = min(br, min(bg, bb)), base_mx = max(br, max(bg, bb));
        uint32_t tot[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            tot[t] = 0;
            if (!((todo >> t) & 1u)) continue;
            if (base_unclamped(br, bg, bb, t)) {
                tot[t] = min_luma_term(dx0, k_inten_a[t] * 64, k_inten_b[t] * 64) + ch0 + min_luma_term(dx1, k_inten_a[t] * 64, k_inten_b[t] * 64) + ch1;
            } else {
                // some of the four colours clamp: the others keep the base colour's chroma (one square each), the clamped ones take the full distance
                mixed_min m0 = { ~0u, ~0u }, m1 = { ~0u, ~0u };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int d = k == 0 ? -k_inten_b[t] : k == 1 ? -k_inten_a[t] : k == 2 ? k_inten_a[t] : k_inten_b[t];
                    const bool clamps = d < 0 ? base_mn + d < 0 : base_mx + d > 255;
                    const int e0 = dx0 - 64 * d, e1 = dx1 - 64 * d;
                    m0.luma_sq = min(m0.luma_sq, clamps ? ~0u : (uint32_t)__mul24(e0, e0));
                    m1.luma_sq = min(m1.luma_sq, clamps ? ~0u : (uint32_t)__mul24(e1, e1));
                    if (clamps) {
                        const cvec c = to_cvec<true>(clamp255(br + d), clamp255(bg + d), clamp255(bb + d));
                        m0.full = min(m0.full, cdist<true>(pc0, c));
                        m1.full = min(m1.full, cdist<true>(pc1, c));
                    }
                }
                tot[t] = mixed_min_total(m0, ch0) + mixed_min_total(m1, ch1);
            }
        }
        // exchange: after the step with partner distance d the lane keeps the half of its tables selected by its bit d
        uint32_t k4[4], k2[2], k1;
        {
            const bool up = (sub & 1u) != 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t give = up ? tot[k] : tot[4 + k];
                k4[k] = (up ? tot[4 + k] : tot[k]) + (uint32_t)__shfl_xor((int)give, 1, 8);
            }
        }
        {
            const bool up = (sub & 2u) != 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t give = up ? k4[k] : k4[2 + k];
                k2[k] = (up ? k4[2 + k] : k4[k]) + (uint32_t)__shfl_xor((int)give, 2, 8);
            }
        }
        {
            const bool up = (sub & 4u) != 0;
            const uint32_t give = up ? k2[0] : k2[1];
            k1 = (up ? k2[1] : k2[0]) + (uint32_t)__shfl_xor((int)give, 4, 8);
        }
        {
            const uint32_t total = ((enable_mask >> my_table) & 1u) ? k1 : 0x0FFFFFFFu;
            uint32_t key = (total << 3) | my_table;
            key = min(key, (uint32_t)__shfl_xor((int)key, 1, 8));
            key = min(key, (uint32_t)__shfl_xor((int)key, 2, 8));
            key = min(key, (uint32_t)__shfl_xor((int)key, 4, 8));
            const uint32_t trial_err = key >> 3;
            if (active && trial_err < best_err) {
                best_err = trial_err; best_inten = (int)(key & 7u);
                best_r = tr; best_g = tg; best_b = tb;
                batch_fresh = false;              // the trials after this one start from the new best solution
            }
        }
        if (best_err == 0 || next >= perms) done = true; // etc.cpp:955-956, 993-994
    }

    // Selectors of the winning (colour, table): first-min over s (etc.cpp:1188-1219), two pixels per lane
    cvec bc[4];
    block_cvecs<true>(bc, scale5(best_r), scale5(best_g), scale5(best_b), best_inten);
    const uint32_t p0 = sub * 2u, p1 = p0 + 1u;
    uint32_t bits = selector_bits(p0 & 3u, p0 >> 2, best_sel4<true>(pc0, bc)) | selector_bits(p1 & 3u, p1 >> 2, best_sel4<true>(pc1, bc));
    bits |= (uint32_t)__shfl_xor((int)bits, 1, 8);
    bits |= (uint32_t)__shfl_xor((int)bits, 2, 8);
    bits |= (uint32_t)__shfl_xor((int)bits, 4, 8);
    if (sub == 0 && in_range) {
        const uint64_t v = etc1s_header_bits((uint32_t)best_r, (uint32_t)best_g, (uint32_t)best_b, (uint32_t)best_inten) | bits;
        const uint64_t m = bswap64(v);
        out_blocks[block] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    }
}

// Level-0 variant: evaluate_solution_fast (etc.cpp:1280-1506). Linear metric is forced (:1313); a pixel's selector is the
// number of block-colour luma midpoints at or below twice its luma; tables are scanned 7..0 with strict <, so on equal error
// the HIGHEST table wins -> reduction key uses (7 - table).
template <bool PERCEPTUAL_UNUSED>
__global__ __launch_bounds__(256) void k_encode_etc1s_blocks_fast(const uint4* __restrict__ pixel_blocks, uint32_t n_blocks, uint2* __restrict__ out_blocks) {
    __shared__ uint32_t s_bloom[32][32];
    const uint32_t tid = threadIdx.x;
    const uint32_t table = tid & 7u;
    const uint32_t slot = tid >> 3;
    const uint32_t block_raw = blockIdx.x * 32u + slot;
    const bool in_range = block_raw < n_blocks;
    const uint32_t block = in_range ? block_raw : (n_blocks - 1);
#pragma unroll
    for (int i = 0; i < 4; i++) s_bloom[slot][table * 4 + i] = 0;

    uint32_t px[16];
    {
        const uint4* src = pixel_blocks + (size_t)block * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = src[i];
            px[i * 4 + 0] = v.x; px[i * 4 + 1] = v.y; px[i * 4 + 2] = v.z; px[i * 4 + 3] = v.w;
        }
    }
    cvec pc[16];
    uint32_t luma2[16];
    float sum_r = 0.0f, sum_g = 0.0f, sum_b = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = px[i] & 255, g = (px[i] >> 8) & 255, b = (px[i] >> 16) & 255;
        sum_r += (float)r; sum_g += (float)g; sum_b += (float)b;
        pc[i] = to_cvec<false>(r, g, b);
        luma2[i] = (uint32_t)(r + g + b) * 2u;
    }
    const float avg_r = sum_r / 16.0f, avg_g = sum_g / 16.0f, avg_b = sum_b / 16.0f;

    uint32_t best_err = 0xFFFFFFFFu;
    int best_r = 0, best_g = 0, best_b = 0, best_inten = 0;
    bool done = false;
    __syncthreads();

    for (int i = -1; i < 4; i++) {
        if (__all(done)) break;
        bool active = !done;
        int tr = 0, tg = 0, tb = 0;
        if (i < 0) {
            tr = avg_to_color5(avg_r); tg = avg_to_color5(avg_g); tb = avg_to_color5(avg_b);
        } else if (active) {
            active = cluster_fit_trial(c_cluster_fit_order[i], best_r, best_g, best_b, best_inten, avg_r, avg_g, avg_b, tr, tg, tb);
        }
        if (active) active = bloom_test_and_set(&s_bloom[slot][0], tr, tg, tb);
        if (active) {
            cvec bc[4];
            block_cvecs<false>(bc, scale5(tr), scale5(tg), scale5(tb), (int)table);
            const uint32_t i0 = (uint32_t)(bc[0].x + bc[0].y + bc[0].z), i1 = (uint32_t)(bc[1].x + bc[1].y + bc[1].z);
            const uint32_t i2 = (uint32_t)(bc[2].x + bc[2].y + bc[2].z), i3 = (uint32_t)(bc[3].x + bc[3].y + bc[3].z);
            const uint32_t m0 = i0 + i1, m1 = i1 + i2, m2 = i2 + i3;
            uint32_t total = 0;
#pragma unroll
            for (int p = 0; p < 16; p++) {
                const uint32_t s = (uint32_t)(luma2[p] >= m0) + (uint32_t)(luma2[p] >= m1) + (uint32_t)(luma2[p] >= m2);
                // midpoints are non-decreasing, so the count equals the reference's walk (etc.cpp:1368-1376)
                const cvec c = select_cvec(bc, s);
                total += cdist<false>(pc[p], c);
            }
            uint32_t key = (total << 3) | (7u - table);
            key = min(key, (uint32_t)__shfl_xor((int)key, 1, 8));
            key = min(key, (uint32_t)__shfl_xor((int)key, 2, 8));
            key = min(key, (uint32_t)__shfl_xor((int)key, 4, 8));
            const uint32_t trial_err = key >> 3;
            if (trial_err < best_err) {
                best_err = trial_err; best_inten = (int)(7u - (key & 7u));
                best_r = tr; best_g = tg; best_b = tb;
            }
        }
        if (best_err == 0) done = true;
    }

    cvec bc[4];
    block_cvecs<false>(bc, scale5(best_r), scale5(best_g), scale5(best_b), best_inten);
    const uint32_t i0 = (uint32_t)(bc[0].x + bc[0].y + bc[0].z), i1 = (uint32_t)(bc[1].x + bc[1].y + bc[1].z);
    const uint32_t i2 = (uint32_t)(bc[2].x + bc[2].y + bc[2].z), i3 = (uint32_t)(bc[3].x + bc[3].y + bc[3].z);
    const uint32_t m0 = i0 + i1, m1 = i1 + i2, m2 = i2 + i3;
    uint32_t bits = 0;
#pragma unroll
    for (int p = 0; p < 16; p++) {
        if ((uint32_t)(p >> 1) == table) {
            const uint32_t s = (uint32_t)(luma2[p] >= m0) + (uint32_t)(luma2[p] >= m1) + (uint32_t)(luma2[p] >= m2);
            bits |= selector_bits((uint32_t)(p & 3), (uint32_t)(p >> 2), s);
        }
    }
    bits |= (uint32_t)__shfl_xor((int)bits, 1, 8);
    bits |= (uint32_t)__shfl_xor((int)bits, 2, 8);
    bits |= (uint32_t)__shfl_xor((int)bits, 4, 8);
    if (table == 0 && in_range) {
        const uint64_t v = etc1s_header_bits((uint32_t)best_r, (uint32_t)best_g, (uint32_t)best_b, (uint32_t)best_inten) | bits;
        const uint64_t m = bswap64(v);
        out_blocks[block] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    }
}

// -------------------------------------------------------------------------------------------------------------------
// a7: init_endpoint_training_vectors (frontend.cpp:825-866) and a12: selector training vectors (frontend.cpp:2155-2183)
// -------------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_endpoint_training_vectors(const uint64_t* __restrict__ etc_blocks, uint32_t n_blocks, float* __restrict__ out6) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) return;
    uint32_t r5, g5, b5, inten;
    unpack_etc1s_header(etc_blocks[i], r5, g5, b5, inten);
    const int br = scale5((int)r5), bg = scale5((int)g5), bb = scale5((int)b5), d = k_inten_b[inten];
    float* o = out6 + (size_t)i * 6;
    const float k = 1.0f / 255.0f; // the reference multiplies by the rounded reciprocal (frontend.cpp:846-851)
    o[0] = (float)clamp255(br - d) * k; o[1] = (float)clamp255(bg - d) * k; o[2] = (float)clamp255(bb - d) * k;
    o[3] = (float)clamp255(br + d) * k; o[4] = (float)clamp255(bg + d) * k; o[5] = (float)clamp255(bb + d) * k;
}

template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_selector_training_vectors(const uint64_t* __restrict__ enc_blocks, uint32_t n_blocks, float* __restrict__ out16, uint64_t* __restrict__ out_w) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_blocks) return;
    const uint64_t m = enc_blocks[i];
    uint32_t r5, g5, b5, inten;
    unpack_etc1s_header(m, r5, g5, b5, inten);
    const uint32_t lo = (uint32_t)bswap64(m);
    if (out16) { // the resident frontend only needs the weights: it de-duplicates on the packed selector word
        float4* o = reinterpret_cast<float4*>(out16 + (size_t)i * 16);
#pragma unroll
        for (uint32_t y = 0; y < 4; y++) {
            float4 v;
            v.x = (float)selector_from_bits(lo, 0, y); v.y = (float)selector_from_bits(lo, 1, y);
            v.z = (float)selector_from_bits(lo, 2, y); v.w = (float)selector_from_bits(lo, 3, y);
            o[y] = v;
        }
    }
    const int br = scale5((int)r5), bg = scale5((int)g5), bb = scale5((int)b5), d = k_inten_b[inten];
    const cvec lo_c = to_cvec<PERCEPTUAL>(clamp255(br - d), clamp255(bg - d), clamp255(bb - d));
    const cvec hi_c = to_cvec<PERCEPTUAL>(clamp255(br + d), clamp255(bg + d), clamp255(bb + d));
    const uint32_t dist = cdist<PERCEPTUAL>(lo_c, hi_c);
    out_w[i] = (uint64_t)min(max(dist / 300u, 1u), 4096u);
}

// -------------------------------------------------------------------------------------------------------------------
// a9: generate_endpoint_codebook (frontend.cpp:1482-1613) -- etc1_optimizer over all pixels of an endpoint cluster.
//
// One 1024-thread workgroup per cluster (largest clusters are dispatched first). A trial is one pass over the cluster's
// pixels computing all 8 intensity-table totals at once (u64), a wave shuffle reduction and a 16-wave LDS reduction.
// Pixels are gathered straight from the resident tiles: training vector v = block*2+subblock owns the 32 contiguous
// bytes of rows 2*subblock..2*subblock+1 (flipped layout, etc.cpp:352-361).
// The float colour mean is order dependent beyond 2^24 (SURVEY hazard H4): integer channel sums <= 2^24 are provably
// identical to the reference's running float sum; otherwise three lanes replay the float accumulation in pixel order.
// -------------------------------------------------------------------------------------------------------------------

constexpr int CB_THREADS = 512;   // 1024 leaves the SIMDs 44 % idle on the bench image (barriers per trial, ~7 pixels per thread); 512: 1.14 ms against 1.80
constexpr int CB_WAVES = CB_THREADS / 64;
// The first CB_STAGE texels of a cluster are kept in LDS after the first pass over them: every trial (17 at the default quality, up to 166) re-reads the
// cluster's texels, and from memory that is two dependent loads per texel (member list, then the tile) -- 7.7x the algorithmic bytes fetched per launch in
// round 2's FETCH_SIZE pass. 8192 texels (32 KiB) hold the whole cluster for all but the largest few; the rest of a larger cluster still comes from L2.
constexpr uint32_t CB_STAGE = 8192;

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ uint32_t cluster_pixel(const uint32_t* __restrict__ pixel_words, const uint32_t* __restrict__ members, uint32_t j) {
    const uint32_t tv = members[j >> 3];
    // word index = block*16 + subblock*8 + (j & 7); tv = block*2 + subblock
    return pixel_words[(size_t)tv * 8 + (j & 7u)];
}

// forced selector of cluster pixel j (refine_block_endpoints_given_selectors, frontend.cpp:2766-2775): the selector the block's
// current encoding gives that texel; sub-block texels are the flipped layout, i.e. rows {0,1} / {2,3} in raster order
__device__ __forceinline__ uint32_t cluster_pixel_selector(const uint64_t* __restrict__ enc_blocks, const uint32_t* __restrict__ members, uint32_t j) {
    const uint32_t tv = members[j >> 3], k = j & 7u;
    const uint32_t lo32 = (uint32_t)bswap64(enc_blocks[tv >> 1]);
    return selector_from_bits(lo32, k & 3u, (tv & 1u) * 2u + (k >> 2));
}

// FORCED = the etc1_optimizer with m_pForce_selectors (etc.cpp:1188-1193): every texel is scored against the colour its current
// selector picks instead of the nearest one; the previous endpoints are never kept here, instead the cluster's CURRENT error
// (each texel against its own block's current colours) is returned in cur_err_out for the caller's "only if better" test.
template <bool PERCEPTUAL, int QUALITY, bool FORCED>
__global__ __launch_bounds__(CB_THREADS) void k_generate_endpoint_codebook(
    const uint32_t* __restrict__ pixel_words, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
    const uint32_t* __restrict__ indices, uint32_t step, uint8_t* __restrict__ params, uint64_t* __restrict__ err_out, uint8_t* __restrict__ valid,
    const uint64_t* __restrict__ enc_blocks, uint64_t* __restrict__ cur_err_out) {
    __shared__ uint64_t s_part[CB_WAVES][8];
    __shared__ uint64_t s_tot[8];
    __shared__ int s_mm[CB_WAVES][6];
    __shared__ uint32_t s_bloom[32];
    __shared__ float s_avg[3];
    __shared__ int s_spread;
    __shared__ int s_active;
    __shared__ uint32_t s_px[CB_STAGE];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ci = order[blockIdx.x];
    const uint32_t first = offsets[ci];
    const uint32_t n = (offsets[ci + 1] - first) * 8u; // pixels
    const uint32_t* members = indices + first;
    auto texel = [&](uint32_t j) -> uint32_t { return j < CB_STAGE ? s_px[j] : cluster_pixel(pixel_words, members, j); };   // valid after the init pass

    if (tid < 32) s_bloom[tid] = 0;

    // ---- init: channel sums, min/max
    {
        uint64_t sr = 0, sg = 0, sb = 0;
        int mn_r = 255, mn_g = 255, mn_b = 255, mx_r = 0, mx_g = 0, mx_b = 0;
        for (uint32_t j = tid; j < n; j += CB_THREADS) {
            const uint32_t w = cluster_pixel(pixel_words, members, j);
            if (j < CB_STAGE) s_px[j] = w;
            const int r = w & 255, g = (w >> 8) & 255, b = (w >> 16) & 255;
            sr += r; sg += g; sb += b;
            mn_r = min(mn_r, r); mn_g = min(mn_g, g); mn_b = min(mn_b, b);
            mx_r = max(mx_r, r); mx_g = max(mx_g, g); mx_b = max(mx_b, b);
        }
        sr = wave_sum_u64(sr); sg = wave_sum_u64(sg); sb = wave_sum_u64(sb);
        mn_r = wave_min_i32(mn_r); mn_g = wave_min_i32(mn_g); mn_b = wave_min_i32(mn_b);
        mx_r = wave_max_i32(mx_r); mx_g = wave_max_i32(mx_g); mx_b = wave_max_i32(mx_b);
        if (lane == 0) {
            s_part[wave][0] = sr; s_part[wave][1] = sg; s_part[wave][2] = sb;
            s_mm[wave][0] = mn_r; s_mm[wave][1] = mn_g; s_mm[wave][2] = mn_b;
            s_mm[wave][3] = mx_r; s_mm[wave][4] = mx_g; s_mm[wave][5] = mx_b;
        }
    }
    __syncthreads();
    if (tid < 3) {
        uint64_t s = 0;
        for (int w = 0; w < CB_WAVES; w++) s += s_part[w][tid];
        float fs;
        if (s <= (1ull << 24)) {
            fs = (float)s; // every partial sum of the reference's running float sum is an exactly representable integer
        } else {
            fs = 0.0f;     // replay the float accumulation in pixel order (etc.cpp:1034-1041)
            for (uint32_t j = 0; j < n; j++) fs += (float)((texel(j) >> (8 * tid)) & 255u);
        }
        s_avg[tid] = fs / (float)n;
    }
    if (tid == 0) {
        int mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
        for (int w = 0; w < CB_WAVES; w++)
            for (int c = 0; c < 3; c++) { mn[c] = min(mn[c], s_mm[w][c]); mx[c] = max(mx[c], s_mm[w][3 + c]); }
        s_spread = max(max(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
    }
    __syncthreads();
    const float avg_r = s_avg[0], avg_g = s_avg[1], avg_b = s_avg[2];
    const uint32_t enable_mask = (QUALITY > BU_Q_MEDIUM) ? 0xFFu : (uint32_t)c_inten_enable_by_spread[s_spread];

    uint64_t best_err = ~0ull;
    int best_r = 0, best_g = 0, best_b = 0, best_inten = 0;
    bool best_valid = false;

    const int perms = (int)perms_for_quality(QUALITY);
    for (int i = -1; i < perms; i++) {
        int tr = 0, tg = 0, tb = 0;
        bool active = true;
        if (i < 0) {
            tr = avg_to_color5(avg_r); tg = avg_to_color5(avg_g); tb = avg_to_color5(avg_b);
        } else {
            active = cluster_fit_trial(c_cluster_fit_order[i], best_r, best_g, best_b, best_inten, avg_r, avg_g, avg_b, tr, tg, tb);
        }
        // all threads hold identical state, so `active` is workgroup-uniform; thread 0 owns the Bloom filter
        if (tid == 0) s_active = active ? (bloom_test_and_set(s_bloom, tr, tg, tb) ? 1 : 0) : 0;
        __syncthreads();
        active = s_active != 0;
        if (active) {
            cvec bc[8][4];
#pragma unroll
            for (int t = 0; t < 8; t++) block_cvecs<PERCEPTUAL>(bc[t], scale5(tr), scale5(tg), scale5(tb), t);
            uint64_t tot[8];
#pragma unroll
            for (int t = 0; t < 8; t++) tot[t] = 0;
            uint32_t plain_mask = 0;   // workgroup-uniform
#pragma unroll
            for (int t = 0; t < 8; t++) plain_mask |= base_unclamped(scale5(tr), scale5(tg), scale5(tb), t) ? (1u << t) : 0u;
            plain_mask &= enable_mask;
            const cvec base_cv = to_cvec<PERCEPTUAL>(scale5(tr), scale5(tg), scale5(tb));
            uint32_t clamp_bits = 0;   // bit 4 t + k: colour k of table t clamps a channel (workgroup-uniform)
            {
                const int base_mn = min(scale5(tr), min(scale5(tg), scale5(tb))), base_mx = max(scale5(tr), max(scale5(tg), scale5(tb)));
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    if (base_mn - k_inten_b[t] < 0) clamp_bits |= 1u << (t * 4);
                    if (base_mn - k_inten_a[t] < 0) clamp_bits |= 2u << (t * 4);
                    if (base_mx + k_inten_a[t] > 255) clamp_bits |= 4u << (t * 4);
                    if (base_mx + k_inten_b[t] > 255) clamp_bits |= 8u << (t * 4);
                }
            }
            for (uint32_t j = tid; j < n; j += CB_THREADS) {
                const cvec p = pixel_cvec<PERCEPTUAL>(texel(j));
                if (FORCED) {
                    const uint32_t sel = cluster_pixel_selector(enc_blocks, members, j);
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        const cvec c = select_cvec(bc[t], sel);
                        tot[t] += cdist<PERCEPTUAL>(p, c);
                    }
                } else if (PERCEPTUAL) {
                    // colours that need no clamping share the pixel's chroma term (etc1s_device.h, base_unclamped / mixed_min): a table none of whose colours clamp costs
                    // two squares, in the others only the clamped colours take the full distance (which ones: workgroup-uniform, scalar branches)
                    const uint32_t ch = chroma_term(p.y - base_cv.y, p.z - base_cv.z);
                    const int dx0 = p.x - base_cv.x;
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        if (!((enable_mask >> t) & 1u)) continue;
                        if ((plain_mask >> t) & 1u) {
                            tot[t] += min_luma_term(dx0, k_inten_a[t] * 64, k_inten_b[t] * 64) + ch;
                        } else {
                            mixed_min m = { ~0u, ~0u };
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const int d = k == 0 ? -k_inten_b[t] : k == 1 ? -k_inten_a[t] : k == 2 ? k_inten_a[t] : k_inten_b[t];
                                if ((clamp_bits >> (t * 4 + k)) & 1u) {
                                    m.full = min(m.full, cdist<true>(p, bc[t][k]));
                                } else {
                                    const int e = dx0 - 64 * d;
                                    m.luma_sq = min(m.luma_sq, (uint32_t)__mul24(e, e));
                                }
                            }
                            tot[t] += mixed_min_total(m, ch);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 8; t++)
                        if ((enable_mask >> t) & 1u) tot[t] += min_err4<PERCEPTUAL>(p, bc[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const uint64_t s = wave_sum_u64(tot[t]);
                if (lane == 0) s_part[wave][t] = s;
            }
            __syncthreads();
            if (tid < 8) {
                uint64_t s = 0;
                for (int w = 0; w < CB_WAVES; w++) s += s_part[w][tid];
                s_tot[tid] = s;
            }
            __syncthreads();
            uint64_t trial_err = (uint64_t)INT64_MAX; // etc.cpp:1131
            int trial_inten = 0;
            bool trial_valid = false;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                if (!((enable_mask >> t) & 1u)) continue;
                const uint64_t s = s_tot[t];
                if (s < trial_err) { trial_err = s; trial_inten = t; trial_valid = true; }
            }
            if (trial_err < best_err) {
                best_err = trial_err; best_inten = trial_inten; best_valid = trial_valid;
                best_r = tr; best_g = tg; best_b = tb;
            }
        }
        __syncthreads(); // s_active / s_part / s_tot are reused by the next trial
        if (best_err == 0 || !best_valid) break; // etc.cpp:955-956, 993-994
    }

    if (FORCED) {
        // current error of the cluster's texels under their own blocks' present colours (frontend.cpp:2773)
        uint64_t tot = 0;
        for (uint32_t j = tid; j < n; j += CB_THREADS) {
            const uint32_t tv = members[j >> 3];
            uint32_t r5, g5, b5, inten;
            unpack_etc1s_header(enc_blocks[tv >> 1], r5, g5, b5, inten);
            cvec bc[4];
            block_cvecs<PERCEPTUAL>(bc, scale5((int)r5), scale5((int)g5), scale5((int)b5), (int)inten);
            const uint32_t sel = cluster_pixel_selector(enc_blocks, members, j);
            const cvec c = select_cvec(bc, (uint32_t)sel);
            tot += cdist<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(texel(j)), c);
        }
        tot = wave_sum_u64(tot);
        __syncthreads();
        if (lane == 0) s_part[wave][0] = tot;
        __syncthreads();
        if (tid == 0) {
            uint64_t cur = 0;
            for (int w = 0; w < CB_WAVES; w++) cur += s_part[w][0];
            cur_err_out[ci] = cur;
            params[ci * 4 + 0] = (uint8_t)best_r; params[ci * 4 + 1] = (uint8_t)best_g; params[ci * 4 + 2] = (uint8_t)best_b; params[ci * 4 + 3] = (uint8_t)best_inten;
            err_out[ci] = best_err;
            valid[ci] = best_valid ? 1 : 0;
        }
        return;
    }
    // ---- keep the previous endpoints unless the error strictly drops (frontend.cpp:1554-1605)
    bool use_new = true;
    if (step != 0 && valid[ci]) {
        const int pr = params[ci * 4 + 0], pg = params[ci * 4 + 1], pb = params[ci * 4 + 2], pi = params[ci * 4 + 3];
        cvec bc[4];
        block_cvecs<PERCEPTUAL>(bc, scale5(pr), scale5(pg), scale5(pb), pi);
        uint64_t tot = 0;
        for (uint32_t j = tid; j < n; j += CB_THREADS) tot += min_err4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(texel(j)), bc);
        tot = wave_sum_u64(tot);
        if (lane == 0) s_part[wave][0] = tot;
        __syncthreads();
        uint64_t prev = 0;
        for (int w = 0; w < CB_WAVES; w++) prev += s_part[w][0];
        use_new = prev > best_err;
    }
    if (tid == 0 && use_new) {
        params[ci * 4 + 0] = (uint8_t)best_r; params[ci * 4 + 1] = (uint8_t)best_g; params[ci * 4 + 2] = (uint8_t)best_b; params[ci * 4 + 3] = (uint8_t)best_inten;
        err_out[ci] = best_err;
        valid[ci] = 1;
    }
}

#include "etc1s_codebook_wide.inc"

// -------------------------------------------------------------------------------------------------------------------
// a10: refine_endpoint_clusterization (frontend.cpp:1772-1917)
//
// One wave per block; lanes sweep the candidate endpoint clusters (the block's parent-cluster list, or all clusters for flat
// codebooks). The block's 16 pixels are wave-uniform, the candidate's four colours are per lane. Winner = first minimum in
// list order, except that the block's current cluster wins ties at non-zero error, and a zero-error candidate ends the
// reference's scan (:1896-1904) -- encoded below from (min key, error of the current cluster).
// -------------------------------------------------------------------------------------------------------------------

template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_refine_endpoint_clusterization(
    const uint4* __restrict__ pixel_blocks, uint32_t n_blocks, const uint32_t* __restrict__ block_cluster,
    const uint32_t* __restrict__ cluster_params, uint32_t n_clusters, uint32_t n_parents,
    const uint32_t* __restrict__ cand_offsets, const uint32_t* __restrict__ cand_indices, const uint8_t* __restrict__ block_parent,
    uint32_t* __restrict__ out_best) {
    constexpr uint32_t RQ = 256;   // candidates per round
    __shared__ uint2 s_q[4][2][RQ];   // per wave: {cluster parameters, position in the list | "is the block's current cluster" << 31}
    __shared__ uint32_t s_qp[4][2][RQ];   // the partial error of a candidate that survived the first four pixels
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));   // wave-uniform, and told so (see k_refine_sorted)
    if (block >= n_blocks) return; // whole wave exits together

    cvec pc[16];
    {
        const uint4* src = pixel_blocks + (size_t)block * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = src[i];
            pc[i * 4 + 0] = pixel_cvec<PERCEPTUAL>(v.x); pc[i * 4 + 1] = pixel_cvec<PERCEPTUAL>(v.y);
            pc[i * 4 + 2] = pixel_cvec<PERCEPTUAL>(v.z); pc[i * 4 + 3] = pixel_cvec<PERCEPTUAL>(v.w);
        }
    }
    const uint32_t cur = block_cluster[block];
    const uint32_t cur_inten = (cluster_params[cur] >> 24) & 255u;

    uint32_t first = 0, total = n_clusters;
    if (n_parents) {
        const uint32_t p = block_parent[block];
        first = cand_offsets[p];
        total = cand_offsets[p + 1] - first;
    }

    // key = error << 32 | position in list; the skipped / out-of-range sentinel sorts last
    uint64_t best_key = ~0ull;
    uint32_t cur_err = 0xFFFFFFFFu;
    // The list is taken RQ candidates at a time. Each round first sorts its admissible candidates into two queues in LDS -- those whose
    // four colours need no clamping and the others -- so that the lanes are full in both sweeps (the intensity filter of :1811-1815
    // otherwise leaves holes) and the unclamped ones take the short form of the distance (etc1s_device.h, base_unclamped). The position
    // in the list travels with the candidate: the winner does not depend on the order of evaluation.
    // Pruning (exact): a candidate whose error exceeds the error of ANY member of the list can neither win nor tie. The block's current
    // cluster is a member of its own parent's list by construction, so its error -- computed here directly, one pixel per lane -- is the
    // first threshold, tightened by the running minimum after every sweep. Each sweep first takes four of the sixteen pixels (a partial
    // sum is a lower bound of the error), squeezes out the candidates that are already above the threshold, and finishes the others.
    // Should the current cluster not turn up in the list after all, everything is done again without a threshold.
    uint2 (*q)[RQ] = s_q[threadIdx.x >> 6];
    uint32_t (*qp)[RQ] = s_qp[threadIdx.x >> 6];
    uint32_t thr;
    {
        const uint32_t prm = cluster_params[cur];
        cvec bc[4];
        block_cvecs<PERCEPTUAL>(bc, scale5((int)(prm & 255u)), scale5((int)((prm >> 8) & 255u)), scale5((int)((prm >> 16) & 255u)), (int)((prm >> 24) & 7u));
        const uint32_t w = reinterpret_cast<const uint32_t*>(pixel_blocks + (size_t)block * 4)[lane & 15u];
        uint32_t e = lane < 16 ? min_err4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(w), bc) : 0u;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) e += (uint32_t)__shfl_xor((int)e, o, 64);
        thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
    }
    constexpr int FIRST_PX[4] = { 0, 5, 10, 15 };
    constexpr uint64_t REST_PX = 0xEDCB98764321ull;   // the other twelve pixel indices, one per nibble
    const uint32_t* block_words = reinterpret_cast<const uint32_t*>(pixel_blocks + (size_t)block * 4);
    auto sync_queue = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto tighten = [&]() {
        uint32_t m = (uint32_t)(best_key >> 32);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
        thr = min(thr, m);
    };
    bool seen_cur = false;
    for (int attempt = 0; attempt < 2; attempt++) {
    for (uint32_t base = 0; base < total; base += RQ) {
        uint32_t n0 = 0, n1 = 0;
#pragma unroll
        for (int i = 0; i < (int)(RQ / 64); i++) {
            const uint32_t k = base + (uint32_t)i * 64u + lane;
            bool take = k < total;
            uint32_t prm = 0, ci = 0;
            if (take) {
                ci = n_parents ? cand_indices[first + k] : k;
                prm = cluster_params[ci];
                take = ((prm >> 24) & 255u) <= cur_inten; // frontend.cpp:1811-1815
            }
            const bool plain = PERCEPTUAL && base_unclamped(scale5((int)(prm & 255u)), scale5((int)((prm >> 8) & 255u)), scale5((int)((prm >> 16) & 255u)), (int)((prm >> 24) & 7u));
            const uint64_t m0 = __ballot(take && plain), m1 = __ballot(take && !plain);
            const uint32_t r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
            const uint32_t r1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
            const uint2 e = make_uint2(prm, k | (ci == cur ? 0x80000000u : 0u));
            if (take && plain) q[0][n0 + r0] = e;
            if (take && !plain) q[1][n1 + r1] = e;
            n0 += (uint32_t)__popcll(m0); n1 += (uint32_t)__popcll(m1);
            seen_cur = seen_cur || __ballot(take && ci == cur) != 0ull;
        }
        sync_queue();
        // ---- unclamped: one chroma term per pixel, the luma term's minimum over the four offsets
        {
            uint32_t ns = 0;
            for (uint32_t j0 = 0; j0 < n0; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool have = j < n0;
                const uint2 e = q[0][have ? j : 0];
                const int inten = (int)((e.x >> 24) & 7u);
                const cvec bcv = to_cvec<true>(scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)));
                const int a64 = k_inten_a[inten] * 64, b64 = k_inten_b[inten] * 64;
                uint32_t part = 0;
#pragma unroll
                for (int f = 0; f < 4; f++) { const int p = FIRST_PX[f]; part += min_luma_term(pc[p].x - bcv.x, a64, b64) + chroma_term(pc[p].y - bcv.y, pc[p].z - bcv.z); }
                const bool keep = have && part <= thr;
                const uint64_t m = __ballot(keep);
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (keep) { q[0][ns + r] = e; qp[0][ns + r] = part; }   // in place: everything up to j0 + 63 has been read
                ns += (uint32_t)__popcll(m);
            }
            sync_queue();
            // the twelve remaining pixels of a survivor are shared by four lanes (three pixels each, fetched by index: the wave-uniform
            // copy in pc[] cannot be indexed per lane), so that a handful of survivors still fills the wave
            for (uint32_t j4 = lane; j4 < ((ns * 4u + 63u) & ~63u); j4 += 64) {
                const uint32_t j = j4 >> 2, part = j4 & 3u;
                const bool have = j < ns;
                const uint2 e = q[0][have ? j : 0];
                const int inten = (int)((e.x >> 24) & 7u);
                const cvec bcv = to_cvec<true>(scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)));
                const int a64 = k_inten_a[inten] * 64, b64 = k_inten_b[inten] * 64;
                uint32_t tot = 0;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const cvec p = pixel_cvec<true>(block_words[(REST_PX >> (4u * (part * 3u + (uint32_t)i))) & 15u]);
                    tot += min_luma_term(p.x - bcv.x, a64, b64) + chroma_term(p.y - bcv.y, p.z - bcv.z);
                }
                tot += (uint32_t)__shfl_xor((int)tot, 1, 64);
                tot += (uint32_t)__shfl_xor((int)tot, 2, 64);
                if (have && part == 0) {
                    tot += qp[0][j];
                    best_key = min(best_key, ((uint64_t)tot << 32) | (e.y & 0x7fffffffu));
                    if (e.y >> 31) cur_err = tot;
                }
            }
            if (attempt == 0) tighten();
        }
        // ---- clamped colours: the four distances
        {
            uint32_t ns = 0;
            for (uint32_t j0 = 0; j0 < n1; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool have = j < n1;
                const uint2 e = q[1][have ? j : 0];
                cvec bc[4];
                block_cvecs<PERCEPTUAL>(bc, scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)), (int)((e.x >> 24) & 7u));
                uint32_t part = 0;
#pragma unroll
                for (int f = 0; f < 4; f++) part += min_err4<PERCEPTUAL>(pc[FIRST_PX[f]], bc);
                const bool keep = have && part <= thr;
                const uint64_t m = __ballot(keep);
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (keep) { q[1][ns + r] = e; qp[1][ns + r] = part; }
                ns += (uint32_t)__popcll(m);
            }
            sync_queue();
            for (uint32_t j4 = lane; j4 < ((ns * 4u + 63u) & ~63u); j4 += 64) {
                const uint32_t j = j4 >> 2, part = j4 & 3u;
                const bool have = j < ns;
                const uint2 e = q[1][have ? j : 0];
                cvec bc[4];
                block_cvecs<PERCEPTUAL>(bc, scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)), (int)((e.x >> 24) & 7u));
                uint32_t tot = 0;
#pragma unroll
                for (int i = 0; i < 3; i++) tot += min_err4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(block_words[(REST_PX >> (4u * (part * 3u + (uint32_t)i))) & 15u]), bc);
                tot += (uint32_t)__shfl_xor((int)tot, 1, 64);
                tot += (uint32_t)__shfl_xor((int)tot, 2, 64);
                if (have && part == 0) {
                    tot += qp[1][j];
                    best_key = min(best_key, ((uint64_t)tot << 32) | (e.y & 0x7fffffffu));
                    if (e.y >> 31) cur_err = tot;
                }
            }
            if (attempt == 0) tighten();
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (seen_cur || attempt == 1) break;
    thr = 0xFFFFFFFFu; best_key = ~0ull; cur_err = 0xFFFFFFFFu;   // (not expected) the threshold was not a member's error: no pruning
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best_key, o, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), o, 64);
        best_key = min(best_key, ((uint64_t)hi << 32) | lo);
        cur_err = min(cur_err, (uint32_t)__shfl_xor((int)cur_err, o, 64));
    }
    if (lane == 0) {
        const uint32_t min_err = (uint32_t)(best_key >> 32);
        const uint32_t k = (uint32_t)best_key;
        uint32_t winner;
        if (best_key == ~0ull) winner = 0;                       // no admissible candidate: best_cluster_index stays 0 (:1787)
        else if (min_err != 0 && cur_err == min_err) winner = cur; // tie goes to the current cluster
        else winner = n_parents ? cand_indices[first + k] : k;
        out_best[block] = winner;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// a10 with pre-sorted candidate lists. k_refine_endpoint_clusterization spends a third of its instructions on finding out which of a
// list's entries a block may take at all (frontend.cpp:1811-1815: intensity table <= the block's) and which distance form they need.
// Both are properties of the (list, entry) pair, not of the block: k_refine_sort_lists (one workgroup per list, a counting sort over
// 2 classes x 8 tables in LDS) rewrites every list as [unclamped, by table][clamped, by table] with the entry's cluster parameters, its
// position in the ORIGINAL list (the tie-break key: the order of evaluation does not matter) and its cluster id, plus the 2 x 8
// cumulative counts. A block then sweeps two prefixes of that, straight from memory. Same pruning as above.
// -------------------------------------------------------------------------------------------------------------------

constexpr uint32_t RS_SEG = 18;   // per list: first entry, unclamped total, 8 cumulative unclamped counts (table <= t), 8 cumulative clamped counts

template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_refine_sort_lists(const uint32_t* __restrict__ cluster_params, uint32_t n_clusters, uint32_t n_parents,
                                                           const uint32_t* __restrict__ cand_offsets, const uint32_t* __restrict__ cand_indices,
                                                           uint2* __restrict__ items, uint32_t* __restrict__ seg) {
    __shared__ uint32_t s_cnt[16], s_pos[16];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    uint32_t first = 0, total = n_clusters;
    if (n_parents) { first = cand_offsets[p]; total = cand_offsets[p + 1] - first; }
    if (tid < 16) s_cnt[tid] = 0;
    __syncthreads();
    auto bucket_of = [&](uint32_t prm) -> uint32_t {
        const uint32_t inten = (prm >> 24) & 7u;
        const bool plain = PERCEPTUAL && base_unclamped(scale5((int)(prm & 255u)), scale5((int)((prm >> 8) & 255u)), scale5((int)((prm >> 16) & 255u)), (int)inten);
        return (plain ? 0u : 8u) + inten;
    };
    for (uint32_t k = tid; k < total; k += 256) {
        const uint32_t ci = n_parents ? cand_indices[first + k] : k;
        atomicAdd(&s_cnt[bucket_of(cluster_params[ci])], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        uint32_t* sg = seg + (size_t)p * RS_SEG;
        sg[0] = first;
        for (int b = 0; b < 16; b++) {
            s_pos[b] = run; run += s_cnt[b];
            if (b < 8) sg[2 + b] = run;                 // unclamped entries with table <= b
            else sg[10 + (b - 8)] = run - sg[9];       // clamped entries with table <= b - 8
            if (b == 7) sg[1] = run;
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < total; k += 256) {
        const uint32_t ci = n_parents ? cand_indices[first + k] : k;
        const uint32_t prm = cluster_params[ci];
        const uint32_t at = atomicAdd(&s_pos[bucket_of(prm)], 1u);
        items[first + at] = make_uint2(prm, (k << 16) | ci);   // position above the cluster id: the key order is (error, position); both fit 16 bits (caller)
    }
}

template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_refine_sorted(const uint4* __restrict__ pixel_blocks, uint32_t n_blocks, const uint32_t* __restrict__ block_cluster,
                                                       const uint32_t* __restrict__ cluster_params, uint32_t n_parents, const uint2* __restrict__ items,
                                                       const uint32_t* __restrict__ seg, const uint8_t* __restrict__ block_parent, uint32_t* __restrict__ out_best) {
    constexpr uint32_t RQ = 256;
    __shared__ uint2 s_q[4][RQ];        // per wave: the survivors of a sweep's first four pixels
    __shared__ uint32_t s_qp[4][RQ];    // and their partial errors
    const uint32_t lane = threadIdx.x & 63u;
    // (wave-uniform, and told so: the block's tile is then fetched with scalar loads and its sixteen colour vectors are made on the scalar unit, once per wave instead of per lane)
    const uint32_t block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    if (block >= n_blocks) return; // whole wave exits together

    cvec pc[16];
    {
        const uint4* src = pixel_blocks + (size_t)block * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = src[i];
            pc[i * 4 + 0] = pixel_cvec<PERCEPTUAL>(v.x); pc[i * 4 + 1] = pixel_cvec<PERCEPTUAL>(v.y);
            pc[i * 4 + 2] = pixel_cvec<PERCEPTUAL>(v.z); pc[i * 4 + 3] = pixel_cvec<PERCEPTUAL>(v.w);
        }
    }
    // the tile's chroma moments (wave-uniform: scalar unit): what the sweep's first look at an unclamped candidate bounds its sixteen chroma terms with
    chroma_moments cm = { 0, 0, 0, 0, 0, 0 };
    if (PERCEPTUAL) {
        int s1y = 0, s1z = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { s1y += pc[i].y; s1z += pc[i].z; }
        cm.my = s1y >> 4; cm.mz = s1z >> 4; cm.r1y = s1y - 16 * cm.my; cm.r1z = s1z - 16 * cm.mz;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int ry = pc[i].y - cm.my, rz = pc[i].z - cm.mz;
            cm.r2y += (int)((uint32_t)(ry * ry) >> 10); cm.r2z += (int)((uint32_t)(rz * rz) >> 10);
        }
    }
    const uint32_t cur = block_cluster[block];
    const uint32_t cur_prm = cluster_params[cur];
    const uint32_t cur_inten = (cur_prm >> 24) & 7u;
    const uint32_t* sg = seg + (size_t)(n_parents ? block_parent[block] : 0u) * RS_SEG;
    const uint32_t first = sg[0], plain_total = sg[1];
    const uint32_t n_plain = sg[2 + cur_inten], n_clamped = sg[10 + cur_inten];
    const uint2* plain_items = items + first;
    const uint2* clamped_items = items + first + plain_total;

    constexpr uint64_t REST_PX = 0xEDCB98764321ull;   // the other twelve pixel indices, one per nibble
    const uint32_t* block_words = reinterpret_cast<const uint32_t*>(pixel_blocks + (size_t)block * 4);
    uint2* q = s_q[threadIdx.x >> 6];
    uint32_t* qp = s_qp[threadIdx.x >> 6];
    uint64_t best_key = ~0ull;
    uint32_t cur_err = 0xFFFFFFFFu;
    uint32_t thr;   // the error of the block's own cluster (a list member by construction), see k_refine_endpoint_clusterization
    {
        const int cr = scale5((int)(cur_prm & 255u)), cg = scale5((int)((cur_prm >> 8) & 255u)), cb = scale5((int)((cur_prm >> 16) & 255u));
        uint32_t e = 0;
        if (PERCEPTUAL && base_unclamped(cr, cg, cb, (int)cur_inten)) {   // (wave-uniform) nine clusters in ten: one chroma term and two squares instead of four distances
            const cvec bcv = to_cvec<true>(cr, cg, cb);
            const cvec p = pixel_cvec<true>(block_words[lane & 15u]);
            if (lane < 16) e = min_luma_term(p.x - bcv.x, k_inten_a[cur_inten] * 64, k_inten_b[cur_inten] * 64) + chroma_term(p.y - bcv.y, p.z - bcv.z);
        } else {
            cvec bc[4];
            block_cvecs<PERCEPTUAL>(bc, cr, cg, cb, (int)cur_inten);
            if (lane < 16) e = min_err4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(block_words[lane & 15u]), bc);
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) e += (uint32_t)__shfl_xor((int)e, o, 64);
        thr = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
    }
    auto sync_queue = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto tighten = [&]() {
        uint32_t m = (uint32_t)(best_key >> 32);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
        thr = min(thr, m);
    };
    bool seen_cur = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        // ---- unclamped: one chroma term per pixel, the luma term's minimum over the four offsets
        for (uint32_t base = 0; base < n_plain; base += RQ) {
            const uint32_t n0 = min(RQ, n_plain - base);
            uint32_t ns = 0;
            for (uint32_t j0 = 0; j0 < n0; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool have = j < n0;
                const uint2 e = plain_items[base + (have ? j : 0)];
                const cvec bcv = to_cvec<true>(scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)));
                // a lower bound of the candidate's error: a bound of all sixteen chroma terms from the tile's moments (chroma_lower_bound: 20 instructions, and it sees the whole
                // tile). Luma terms of a few pixels on top of it were measured and cost more than they prune: with 4 / 2 / 0 pixels' luma terms the kernel takes 1.64 / 1.59 / 1.52 ms
                // (8192^2 q255: 9.2 / 8.4 / 7.8), and a bound from the tile's luma range 1.57 -- the chroma bound decides
                const uint32_t part = chroma_lower_bound(cm, bcv.y, bcv.z);
                const bool keep = have && part <= thr;
                const uint64_t m = __ballot(keep);
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (keep) q[ns + r] = e;
                ns += (uint32_t)__popcll(m);
                seen_cur = seen_cur || __ballot(have && (e.y & 0xffffu) == cur) != 0ull;
            }
            sync_queue();
            for (uint32_t j4 = lane; j4 < ((ns * 4u + 63u) & ~63u); j4 += 64) {   // four lanes per survivor, four pixels each: the exact error
                const uint32_t j = j4 >> 2, part = j4 & 3u;
                const bool have = j < ns;
                const uint2 e = q[have ? j : 0];
                const int inten = (int)((e.x >> 24) & 7u);
                const cvec bcv = to_cvec<true>(scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)));
                const int a64 = k_inten_a[inten] * 64, b64 = k_inten_b[inten] * 64;
                uint32_t tot = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const cvec p = pixel_cvec<true>(block_words[part * 4u + (uint32_t)i]);
                    tot += min_luma_term(p.x - bcv.x, a64, b64) + chroma_term(p.y - bcv.y, p.z - bcv.z);
                }
                tot += (uint32_t)__shfl_xor((int)tot, 1, 64);
                tot += (uint32_t)__shfl_xor((int)tot, 2, 64);
                if (have && part == 0) {
                    best_key = min(best_key, ((uint64_t)tot << 32) | e.y);
                    if ((e.y & 0xffffu) == cur) cur_err = tot;
                }
            }
            if (attempt == 0) tighten();
            __builtin_amdgcn_wave_barrier();
        }
        // ---- clamped colours: the four distances
        for (uint32_t base = 0; base < n_clamped; base += RQ) {
            const uint32_t n1 = min(RQ, n_clamped - base);
            uint32_t ns = 0;
            // the clamped entries are few (a tenth of a list on the bench image) and need the expensive four-distance form: FOUR lanes per entry
            // take one of the four test pixels each (pixel 5 f), so that a handful of entries costs one pass of one distance instead of one of four
            for (uint32_t j0 = 0; j0 < n1; j0 += 16) {
                const uint32_t j = j0 + (lane >> 2), f = lane & 3u;
                const bool have = j < n1;
                const uint2 e = clamped_items[base + (have ? j : 0)];
                cvec bc[4];
                block_cvecs<PERCEPTUAL>(bc, scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)), (int)((e.x >> 24) & 7u));
                uint32_t part = min_err4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(block_words[f * 5u]), bc);   // FIRST_PX[f] = 5 f
                part += (uint32_t)__shfl_xor((int)part, 1, 64);
                part += (uint32_t)__shfl_xor((int)part, 2, 64);
                const bool keep = have && f == 0 && part <= thr;
                const uint64_t m = __ballot(keep);
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (keep) { q[ns + r] = e; qp[ns + r] = part; }
                ns += (uint32_t)__popcll(m);
                seen_cur = seen_cur || __ballot(have && (e.y & 0xffffu) == cur) != 0ull;
            }
            sync_queue();
            for (uint32_t j4 = lane; j4 < ((ns * 4u + 63u) & ~63u); j4 += 64) {
                const uint32_t j = j4 >> 2, part = j4 & 3u;
                const bool have = j < ns;
                const uint2 e = q[have ? j : 0];
                cvec bc[4];
                block_cvecs<PERCEPTUAL>(bc, scale5((int)(e.x & 255u)), scale5((int)((e.x >> 8) & 255u)), scale5((int)((e.x >> 16) & 255u)), (int)((e.x >> 24) & 7u));
                uint32_t tot = 0;
#pragma unroll
                for (int i = 0; i < 3; i++) tot += min_err4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(block_words[(REST_PX >> (4u * (part * 3u + (uint32_t)i))) & 15u]), bc);
                tot += (uint32_t)__shfl_xor((int)tot, 1, 64);
                tot += (uint32_t)__shfl_xor((int)tot, 2, 64);
                if (have && part == 0) {
                    tot += qp[j];
                    best_key = min(best_key, ((uint64_t)tot << 32) | e.y);
                    if ((e.y & 0xffffu) == cur) cur_err = tot;
                }
            }
            if (attempt == 0) tighten();
            __builtin_amdgcn_wave_barrier();
        }
        if (seen_cur || attempt == 1) break;
        thr = 0xFFFFFFFFu; best_key = ~0ull; cur_err = 0xFFFFFFFFu;   // (not expected) the threshold was not a member's error: no pruning
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best_key, o, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), o, 64);
        best_key = min(best_key, ((uint64_t)hi << 32) | lo);
        cur_err = min(cur_err, (uint32_t)__shfl_xor((int)cur_err, o, 64));
    }
    if (lane == 0) {
        const uint32_t min_err = (uint32_t)(best_key >> 32);
        uint32_t winner;
        if (best_key == ~0ull) winner = 0;                       // no admissible candidate: best_cluster_index stays 0 (:1787)
        else if (min_err != 0 && cur_err == min_err) winner = cur; // tie goes to the current cluster
        else winner = (uint32_t)best_key & 0xffffu;                // the winning entry's cluster id rides below its position
        out_best[block] = winner;
    }
}

// -------------------------------------------------------------------------------------------------------------------
// a11: create_initial_packed_texture -> etc_block::determine_selectors (frontend.cpp:2058-2085, etc.h:374-436)
//
// 16 lanes per block, lane l owns pixel (x = l>>2, y = l&3) so that a wave ballot of "raw selector lsb/msb" IS the packed
// selector bit plane (bit index x*4+y). The only kernel of the path that is close to HBM-bound: 64 B in, 8 B out per block.
// -------------------------------------------------------------------------------------------------------------------

template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_determine_selectors(
    const uint32_t* __restrict__ pixel_words, uint32_t n_blocks, const uint32_t* __restrict__ color5_inten,
    const uint32_t* __restrict__ block_cluster, uint2* __restrict__ out_blocks) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t block_raw = gid >> 4;
    const bool in_range = block_raw < n_blocks;
    const uint32_t block = in_range ? block_raw : (n_blocks - 1);
    const uint32_t l = threadIdx.x & 15u;
    const uint32_t x = l >> 2, y = l & 3u;
    const uint32_t w = pixel_words[(size_t)block * 16 + y * 4 + x];
    const uint32_t prm = block_cluster ? color5_inten[block_cluster[block]] : color5_inten[block];
    const uint32_t inten = (prm >> 24) & 255u;
    cvec bc[4];
    block_cvecs<PERCEPTUAL>(bc, scale5((int)(prm & 255u)), scale5((int)((prm >> 8) & 255u)), scale5((int)((prm >> 16) & 255u)), (int)inten);
    const uint32_t s = best_sel4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(w), bc);
    const uint32_t raw = (0x4Bu >> (s * 2)) & 3u;
    const uint64_t lsb = __ballot(raw & 1u);
    const uint64_t msb = __ballot(raw >> 1);
    const uint32_t group = (threadIdx.x & 63u) >> 4;
    if (l == 0 && in_range) {
        const uint32_t bits = (uint32_t)((lsb >> (group * 16)) & 0xFFFFu) | ((uint32_t)((msb >> (group * 16)) & 0xFFFFu) << 16);
        const uint64_t v = etc1s_header_bits(prm & 255u, (prm >> 8) & 255u, (prm >> 16) & 255u, inten) | bits;
        const uint64_t m = bswap64(v);
        out_blocks[block] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    }
}

// -------------------------------------------------------------------------------------------------------------------
// a13: create_optimized_selector_codebook (frontend.cpp:2259-2354)
//
// lane = (pixel p = lane>>2, selector s = lane&3) accumulates the u64 error of "pixel p of every member block encoded with selector s" --
// exactly the reference's total_err[y][x][s] -- then a 4-lane first-min picks the pixel's selector. Clusters are very uneven (a few hold tens
// of thousands of blocks), so the accumulation is cut by POSITION in the CSR member array, not by cluster: every wave takes COSC_CHUNK
// consecutive members, finds the cluster its first member belongs to (binary search in the offsets) and walks on, flushing its partial sums
// into the cluster's 64 u64 counters with atomic adds whenever it crosses into the next cluster. Integer sums: exact in any order. A second
// kernel (one wave per cluster) turns the counters into selectors.
// -------------------------------------------------------------------------------------------------------------------

constexpr uint32_t COSC_CHUNK = 128;
constexpr uint32_t COSC_WORKGROUPS = 2048;   // of four waves: one chunk per wave for a 4096^2 image, the waves stride over the chunks of a larger one

template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_cosc_accumulate(
    const uint32_t* __restrict__ pixel_words, const uint64_t* __restrict__ enc_blocks, uint32_t n_clusters,
    const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ block_indices, unsigned long long* __restrict__ acc) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));   // wave-uniform, and told so: the search below runs on the scalar unit
    // (the launch is a fixed number of waves that stride over the chunks: how many members the offsets span is on the device only, and asking for it was a round trip)
    const uint32_t begin = offsets[0], end = offsets[n_clusters];
    for (uint64_t lo64 = (uint64_t)begin + (uint64_t)wave * COSC_CHUNK; lo64 < end; lo64 += (uint64_t)gridDim.x * 4u * COSC_CHUNK) {
    const uint32_t lo = (uint32_t)lo64, hi = end - lo > COSC_CHUNK ? lo + COSC_CHUNK : end;
    // cluster of member `lo`: the last cluster whose first member is <= lo (empty clusters in front of it share that offset and are skipped)
    uint32_t a = 0, b = n_clusters;  // invariant: offsets[a] <= lo < offsets[b]
    while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (offsets[m] <= lo) a = m; else b = m; }
    uint32_t ci = a, next = offsets[ci + 1];
    const uint32_t p = lane >> 2, s = lane & 3u;
    unsigned long long tot = 0;
    for (uint32_t k = lo; k < hi; k++) {
        while (k >= next) {  // crossed into the next (non-empty) cluster
            if (tot) atomicAdd(&acc[(size_t)ci * 64 + lane], tot);
            tot = 0;
            ci++; next = offsets[ci + 1];
        }
        const uint32_t bi = block_indices[k];
        uint32_t r5, g5, b5, inten;
        unpack_etc1s_header(enc_blocks[bi], r5, g5, b5, inten);
        const int yd = inten_delta((int)inten, (int)s);
        const cvec c = to_cvec<PERCEPTUAL>(clamp255(scale5((int)r5) + yd), clamp255(scale5((int)g5) + yd), clamp255(scale5((int)b5) + yd));
        tot += cdist<PERCEPTUAL>(c, pixel_cvec<PERCEPTUAL>(pixel_words[(size_t)bi * 16 + p]));
    }
    if (tot) atomicAdd(&acc[(size_t)ci * 64 + lane], tot);
    }
}

__global__ __launch_bounds__(256) void k_cosc_select(uint32_t n_clusters, const uint32_t* __restrict__ offsets, const unsigned long long* __restrict__ acc,
                                                     uint64_t* __restrict__ selector_blocks) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t ci = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (ci >= n_clusters) return;
    if (offsets[ci + 1] == offsets[ci]) return; // empty clusters keep their previous selectors (frontend.cpp:2282-2283)
    const uint32_t p = lane >> 2, s = lane & 3u;
    // first-min over the 4 selectors of this pixel: compare (tot, s) lexicographically
    uint64_t bt = acc[(size_t)ci * 64 + lane]; uint32_t bs = s;
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bt, o, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(bt >> 32), o, 64);
        const uint64_t ot = ((uint64_t)hi << 32) | lo;
        const uint32_t os = (uint32_t)__shfl_xor((int)bs, o, 64);
        if (ot < bt || (ot == bt && os < bs)) { bt = ot; bs = os; }
    }
    // pixel p = y*4+x
    uint32_t bits = (s == 0) ? selector_bits(p & 3u, p >> 2, bs) : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, o, 64);
    if (lane == 0) {
        const uint64_t v = (bswap64(selector_blocks[ci]) & ~0xFFFFFFFFull) | bits;
        selector_blocks[ci] = bswap64(v);
    }
}

// -------------------------------------------------------------------------------------------------------------------
// a14: find_optimal_selector_clusters_for_each_block (frontend.cpp:2534-2706)
//
// One wave per block. The 4x16 table err[s][p] of the block's endpoint is built by the 64 lanes (one entry each) into LDS;
// lanes then sweep candidate codebook entries, summing 16 table lookups each (all lanes of a step hit one of 4 banks per
// pixel -> conflict-free broadcasts). Winner = first minimum in list order; the reference's early-outs (:2640-2660) never
// change it. The "identical to the previous block of this 2048-block job" shortcut (:2557-2564) is applied by a second
// pass so that results stay identical even when equal tiles carry different endpoints.
// -------------------------------------------------------------------------------------------------------------------

// A candidate's error is a sum of 16 table entries err[selector of texel][texel]. The selector word keeps texel i's two bits at positions i and 16 + i (i = x * 4 + y), so two
// neighbouring texels' selectors are a 4-bit code (two low-plane bits, two high-plane bits) and their two entries one entry of a 16-entry PAIR table: 8 look-ups per candidate
// instead of 16, after 128 entries built once per block by the wave (integer sums: any grouping gives the same total). Blocks offered many candidates (q255: ~1,000 per block)
// go one step further, four texels = an 8-bit code into four 256-entry tables made from the pair tables: 4 look-ups per candidate.
// the low 32 bits (the selector bits) of every candidate in list order: words[j] = bits of selector_blocks[cand_indices[j]] (flat codebook: of selector_blocks[j])
__global__ __launch_bounds__(256) void k_fosc_candidate_words(const uint64_t* __restrict__ selector_blocks, uint32_t n_selectors, uint32_t n_parents,
                                                              const uint32_t* __restrict__ cand_offsets, const uint32_t* __restrict__ cand_indices, uint32_t* __restrict__ words) {
    const uint32_t total = n_parents ? cand_offsets[n_parents] : n_selectors;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < total; j += gridDim.x * 256u)
        words[j] = (uint32_t)bswap64(selector_blocks[n_parents ? cand_indices[j] : j]);
}

constexpr uint32_t FOSC_QUAD_MIN = 384;   // candidates per block from which the 1,024-entry tables pay for themselves (16 entries per lane to build)

template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_find_optimal_selector_clusters(
    const uint32_t* __restrict__ pixel_words, const uint64_t* __restrict__ enc_blocks, uint32_t n_blocks,
    const uint64_t* __restrict__ selector_blocks, uint32_t n_selectors, uint32_t n_parents,
    const uint32_t* __restrict__ cand_offsets, const uint32_t* __restrict__ cand_indices, const uint8_t* __restrict__ block_parent,
    uint32_t* __restrict__ out_idx, const uint32_t* __restrict__ cand_words) {
    __shared__ uint32_t s_err[4][64];     // [wave][s*16+p]
    __shared__ uint32_t s_pair[4][128];   // [wave][g*16 + code4]: texels with selector bits 2g, 2g + 1
    __shared__ uint32_t s_quad[4][1024];  // [wave][h*256 + code8]: texels with selector bits 4h .. 4h + 3
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t block = blockIdx.x * 4u + wave;   // wave-uniform, and told so: the block's header, parent and list bounds come through scalar loads
    if (block >= n_blocks) return;

    {
        uint32_t r5, g5, b5, inten;
        unpack_etc1s_header(enc_blocks[block], r5, g5, b5, inten);
        const uint32_t s = lane >> 4, p = lane & 15u;
        const int yd = inten_delta((int)inten, (int)s);
        const cvec c = to_cvec<PERCEPTUAL>(clamp255(scale5((int)r5) + yd), clamp255(scale5((int)g5) + yd), clamp255(scale5((int)b5) + yd));
        s_err[wave][lane] = cdist<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(pixel_words[(size_t)block * 16 + p]), c);
    }
    // same-wave producer/consumer: LDS ops of one wave are ordered
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t e = 0; e < 2; e++) {
        const uint32_t idx = lane + e * 64u, g = idx >> 4, code = idx & 15u;
        const uint32_t i0 = 2u * g, i1 = i0 + 1u;                               // selector bit positions = x * 4 + y
        const uint32_t raw0 = (code & 1u) | ((code >> 1) & 2u), raw1 = ((code >> 1) & 1u) | ((code >> 2) & 2u);
        const uint32_t s0 = (0x1Eu >> (raw0 * 2)) & 3u, s1 = (0x1Eu >> (raw1 * 2)) & 3u;   // g_etc1_to_selector_index (selector_from_bits)
        const uint32_t p0 = (i0 & 3u) * 4u + (i0 >> 2), p1 = (i1 & 3u) * 4u + (i1 >> 2);   // raster texel y * 4 + x
        s_pair[wave][idx] = s_err[wave][s0 * 16 + p0] + s_err[wave][s1 * 16 + p1];
    }
    __builtin_amdgcn_wave_barrier();

    uint32_t first = 0, total = n_selectors;
    if (n_parents) {
        const uint32_t p = block_parent[block];
        first = cand_offsets[p];
        total = cand_offsets[p + 1] - first;
    }
    uint64_t best_key = ~0ull;
    if (total >= FOSC_QUAD_MIN) {   // wave-uniform
#pragma unroll
        for (uint32_t e = 0; e < 16; e++) {
            const uint32_t idx = lane + e * 64u, h = idx >> 8, code = idx & 255u, a = code & 15u, b = code >> 4;
            const uint32_t c_lo = (a & 3u) | ((b & 3u) << 2), c_hi = (a >> 2) | (b & 12u);
            s_quad[wave][idx] = s_pair[wave][(2u * h) * 16u + c_lo] + s_pair[wave][(2u * h + 1u) * 16u + c_hi];
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = lane; k < total; k += 64) {
            // cand_words (k_fosc_candidate_words): the candidates' selector words laid out in list order -- one coalesced load instead of list entry, then codebook entry
            uint32_t lo;
            if (cand_words) lo = cand_words[first + k];
            else { const uint32_t ci = n_parents ? cand_indices[first + k] : k; lo = (uint32_t)bswap64(selector_blocks[ci]); }
            uint32_t e = 0;
#pragma unroll
            for (uint32_t h = 0; h < 4; h++) e += s_quad[wave][h * 256u + (((lo >> (4u * h)) & 15u) | (((lo >> (16u + 4u * h)) & 15u) << 4))];
            best_key = min(best_key, ((uint64_t)e << 32) | k);
        }
    } else {
        for (uint32_t k = lane; k < total; k += 64) {
            // cand_words (k_fosc_candidate_words): the candidates' selector words laid out in list order -- one coalesced load instead of list entry, then codebook entry
            uint32_t lo;
            if (cand_words) lo = cand_words[first + k];
            else { const uint32_t ci = n_parents ? cand_indices[first + k] : k; lo = (uint32_t)bswap64(selector_blocks[ci]); }
            uint32_t e = 0;
#pragma unroll
            for (uint32_t g = 0; g < 8; g++) e += s_pair[wave][g * 16u + (((lo >> (2u * g)) & 3u) | (((lo >> (16u + 2u * g)) & 3u) << 2))];
            best_key = min(best_key, ((uint64_t)e << 32) | k);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best_key, o, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(best_key >> 32), o, 64);
        best_key = min(best_key, ((uint64_t)hi << 32) | lo);
    }
    if (lane == 0) {
        const uint32_t k = (uint32_t)best_key;
        out_idx[block] = (best_key == ~0ull) ? 0u : (n_parents ? cand_indices[first + k] : k);
    }
}

// Pass 2 of a14: resolve runs of identical consecutive tiles inside each `chunk`-block job to the run head's choice
// (frontend.cpp:2557-2564), then stamp the chosen selector bits into the encoded blocks (:2688-2690).
__global__ __launch_bounds__(256) void k_fosc_resolve_and_stamp(
    const uint4* __restrict__ pixel_blocks, uint64_t* __restrict__ enc_blocks, uint32_t n_blocks, const uint64_t* __restrict__ selector_blocks,
    uint32_t chunk, const uint32_t* __restrict__ raw_idx, uint32_t* __restrict__ out_idx) {
    const uint32_t block = blockIdx.x * blockDim.x + threadIdx.x;
    if (block >= n_blocks) return;
    uint32_t head = block;
    if (chunk) {
        const uint32_t chunk_first = (block / chunk) * chunk;
        while (head > chunk_first) {
            const uint4* a = pixel_blocks + (size_t)head * 4;
            const uint4* b = pixel_blocks + (size_t)(head - 1) * 4;
            bool same = true;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 u = a[i], v = b[i];
                same = same && (u.x == v.x) && (u.y == v.y) && (u.z == v.z) && (u.w == v.w);
            }
            if (!same) break;
            head--;
        }
    }
    const uint32_t best = raw_idx[head];
    out_idx[block] = best;
    const uint64_t hdr = bswap64(enc_blocks[block]) & ~0xFFFFFFFFull;
    const uint64_t sel = bswap64(selector_blocks[best]) & 0xFFFFFFFFull;
    enc_blocks[block] = bswap64(hdr | sel);
}

// -------------------------------------------------------------------------------------------------------------------
// Launchers
// -------------------------------------------------------------------------------------------------------------------

#define BU_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_encode_etc1s_blocks(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, int quality, bool perceptual, void* d_out) {
    if (!n_blocks) return hipSuccess;
    const dim3 grid((n_blocks + 31) / 32), blk(256);
    const uint4* in = static_cast<const uint4*>(d_pixel_blocks);
    uint2* out = static_cast<uint2*>(d_out);
    if (quality == BU_Q_FAST) {
        hipLaunchKernelGGL(k_encode_etc1s_blocks_fast<false>, grid, blk, 0, st, in, n_blocks, out);
    } else if (perceptual) {
        if (quality == BU_Q_MEDIUM) hipLaunchKernelGGL((k_encode_etc1s_blocks_by_pixel<BU_Q_MEDIUM>), grid, blk, 0, st, in, n_blocks, out);
        else if (quality == BU_Q_SLOW) hipLaunchKernelGGL((k_encode_etc1s_blocks_by_pixel<BU_Q_SLOW>), grid, blk, 0, st, in, n_blocks, out);
        else hipLaunchKernelGGL((k_encode_etc1s_blocks_by_pixel<BU_Q_UBER>), grid, blk, 0, st, in, n_blocks, out);
    } else {
        if (quality == BU_Q_MEDIUM) hipLaunchKernelGGL((k_encode_etc1s_blocks<false, BU_Q_MEDIUM>), grid, blk, 0, st, in, n_blocks, out);
        else if (quality == BU_Q_SLOW) hipLaunchKernelGGL((k_encode_etc1s_blocks<false, BU_Q_SLOW>), grid, blk, 0, st, in, n_blocks, out);
        else hipLaunchKernelGGL((k_encode_etc1s_blocks<false, BU_Q_UBER>), grid, blk, 0, st, in, n_blocks, out);
    }
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_endpoint_training_vectors(hipStream_t st, const void* d_etc_blocks, uint32_t n_blocks, float* d_out6) {
    if (!n_blocks) return hipSuccess;
    hipLaunchKernelGGL(k_endpoint_training_vectors, dim3((n_blocks + 255) / 256), dim3(256), 0, st, static_cast<const uint64_t*>(d_etc_blocks), n_blocks, d_out6);
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_selector_training_vectors(hipStream_t st, const void* d_enc_blocks, uint32_t n_blocks, bool perceptual, float* d_out16, uint64_t* d_w) {
    if (!n_blocks) return hipSuccess;
    const dim3 grid((n_blocks + 255) / 256), blk(256);
    if (perceptual) hipLaunchKernelGGL(k_selector_training_vectors<true>, grid, blk, 0, st, static_cast<const uint64_t*>(d_enc_blocks), n_blocks, d_out16, d_w);
    else hipLaunchKernelGGL(k_selector_training_vectors<false>, grid, blk, 0, st, static_cast<const uint64_t*>(d_enc_blocks), n_blocks, d_out16, d_w);
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_generate_endpoint_codebook(hipStream_t st, const void* d_pixel_blocks, uint32_t n_clusters, const uint32_t* d_order,
                                             const uint32_t* d_offsets, const uint32_t* d_indices, int quality, bool perceptual, uint32_t step,
                                             uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid) {
    if (!n_clusters) return hipSuccess;
    const dim3 grid(n_clusters), blk(CB_THREADS);
    const uint32_t* pw = static_cast<const uint32_t*>(d_pixel_blocks);
    // the etc1_optimizer never runs at "fast" quality for clusters (frontend.cpp:1530-1533)
    if (quality < BU_Q_MEDIUM) quality = BU_Q_MEDIUM;
#define BU_CB(P, Q) hipLaunchKernelGGL((k_generate_endpoint_codebook<P, Q, false>), grid, blk, 0, st, pw, d_order, d_offsets, d_indices, step, d_params, d_err, d_valid, (const uint64_t*)nullptr, (uint64_t*)nullptr)
    if (perceptual) {
        if (quality == BU_Q_MEDIUM) BU_CB(true, BU_Q_MEDIUM); else if (quality == BU_Q_SLOW) BU_CB(true, BU_Q_SLOW); else BU_CB(true, BU_Q_UBER);
    } else {
        if (quality == BU_Q_MEDIUM) BU_CB(false, BU_Q_MEDIUM); else if (quality == BU_Q_SLOW) BU_CB(false, BU_Q_SLOW); else BU_CB(false, BU_Q_UBER);
    }
#undef BU_CB
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

// refine_block_endpoints_given_selectors (frontend.cpp:2718-2976) / reoptimize_remapped_endpoints (:2996-3104): cluster fit with the selectors held fixed
hipError_t launch_refit_endpoints_given_selectors(hipStream_t st, const void* d_pixel_blocks, const void* d_enc_blocks, uint32_t n_clusters, const uint32_t* d_order,
                                                  const uint32_t* d_offsets, const uint32_t* d_indices, int quality, bool perceptual, uint8_t* d_params, uint64_t* d_err,
                                                  uint8_t* d_valid, uint64_t* d_cur_err) {
    if (!n_clusters) return hipSuccess;
    const dim3 grid(n_clusters), blk(CB_THREADS);
    const uint32_t* pw = static_cast<const uint32_t*>(d_pixel_blocks);
    const uint64_t* enc = static_cast<const uint64_t*>(d_enc_blocks);
#define BU_RF(P, Q) hipLaunchKernelGGL((k_generate_endpoint_codebook<P, Q, true>), grid, blk, 0, st, pw, d_order, d_offsets, d_indices, 0u, d_params, d_err, d_valid, enc, d_cur_err)
    // uber for refine_block_endpoints_given_selectors and level 6; slow for reoptimize_remapped_endpoints below level 6 (frontend.cpp:3073-3076)
    if (perceptual) { if (quality == BU_Q_SLOW) BU_RF(true, BU_Q_SLOW); else BU_RF(true, BU_Q_UBER); }
    else { if (quality == BU_Q_SLOW) BU_RF(false, BU_Q_SLOW); else BU_RF(false, BU_Q_UBER); }
#undef BU_RF
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

// compute_endpoint_subblock_error_vec (frontend.cpp:1006-1091): error of every sub-block (training vector) under its cluster's endpoints
template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_subblock_errors(const uint32_t* __restrict__ pixel_words, uint32_t n_blocks, const uint32_t* __restrict__ block_cluster,
                                                         const uint32_t* __restrict__ cluster_params, uint64_t* __restrict__ out) {
    const uint32_t tv = blockIdx.x * 256u + threadIdx.x;
    if (tv >= n_blocks * 2u) return;
    const uint32_t prm = cluster_params[block_cluster[tv >> 1]];
    cvec bc[4];
    // NOT scale5(): the reference passes the 5-bit colour with scaled = true here (frontend.cpp:1043), so the sub-block errors that
    // rank candidates for new clusters are measured against the unscaled values; reproduced as is
    block_cvecs<PERCEPTUAL>(bc, (int)(prm & 255u), (int)((prm >> 8) & 255u), (int)((prm >> 16) & 255u), (int)(prm >> 24));
    uint64_t tot = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) tot += min_err4<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(pixel_words[(size_t)tv * 8 + k]), bc);
    out[tv] = tot;
}

hipError_t launch_subblock_errors(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, const uint32_t* d_block_cluster, const uint8_t* d_cluster_params,
                                  bool perceptual, uint64_t* d_out) {
    if (!n_blocks) return hipSuccess;
    const dim3 grid((n_blocks * 2 + 255) / 256), blk(256);
    const uint32_t* pw = static_cast<const uint32_t*>(d_pixel_blocks);
    const uint32_t* prm = reinterpret_cast<const uint32_t*>(d_cluster_params);
    if (perceptual) hipLaunchKernelGGL(k_subblock_errors<true>, grid, blk, 0, st, pw, n_blocks, d_block_cluster, prm, d_out);
    else hipLaunchKernelGGL(k_subblock_errors<false>, grid, blk, 0, st, pw, n_blocks, d_block_cluster, prm, d_out);
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

// The stateless part of basisu_backend::create_encoder_blocks (backend.cpp:406-617): for every block of a slice the error of the block as the
// frontend left it (cur_err of :507 and :841) and -- where no causal neighbour already shares its endpoints -- its error under the endpoints of its left, upper and
// upper-left neighbours with its own selectors (what :520-574 evaluates when those neighbours keep their endpoints). One thread per block, tiles and blocks resident;
// the decisions that chain from block to block stay on the host. ~0u: not applicable (edge, shared endpoints, zero error, index out of range).
template <bool PERCEPTUAL>
__global__ __launch_bounds__(256) void k_backend_block_errors(const uint4* __restrict__ pixel_blocks, const uint64_t* __restrict__ etc_blocks, const uint32_t* __restrict__ block_cluster,
                                                              const uint32_t* __restrict__ cluster_params, uint32_t first, uint32_t nbx, uint32_t nby, uint32_t n_clusters,
                                                              int with_neighbours, uint32_t* __restrict__ own, uint32_t* __restrict__ neighbour) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nbx * nby) return;
    const uint32_t b = first + i, bx = i % nbx, by = i / nbx;
    uint32_t px[16];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint4 v = pixel_blocks[(size_t)b * 4 + k]; px[k * 4] = v.x; px[k * 4 + 1] = v.y; px[k * 4 + 2] = v.z; px[k * 4 + 3] = v.w; }
    const uint64_t mem = etc_blocks[b];
    uint32_t r5, g5, b5, inten;
    unpack_etc1s_header(mem, r5, g5, b5, inten);
    const uint32_t lo32 = (uint32_t)bswap64(mem);
    auto error_under = [&](uint32_t cr, uint32_t cg, uint32_t cb, uint32_t table) {
        cvec bc[4];
        block_cvecs<PERCEPTUAL>(bc, scale5((int)cr), scale5((int)cg), scale5((int)cb), (int)table);
        uint32_t e = 0;
#pragma unroll
        for (uint32_t y = 0; y < 4; y++)
#pragma unroll
            for (uint32_t x = 0; x < 4; x++) e += cdist<PERCEPTUAL>(pixel_cvec<PERCEPTUAL>(px[y * 4 + x]), select_cvec(bc, selector_from_bits(lo32, x, y)));
        return e;
    };
    const uint32_t mine_err = error_under(r5, g5, b5, inten);
    own[b] = mine_err;
    if (!with_neighbours) return;
    const uint32_t mine = block_cluster[b];
    const int dx[3] = { -1, 0, -1 }, dy[3] = { 0, -1, -1 };   // g_endpoint_preds (backend.cpp:120-128)
    uint32_t nb[3];
    bool any_equal = false;
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const int x = (int)bx + dx[p], y = (int)by + dy[p];
        nb[p] = (x >= 0 && y >= 0) ? block_cluster[first + (uint32_t)x + (uint32_t)y * nbx] : ~0u;
        any_equal = any_equal || nb[p] == mine;
    }
#pragma unroll
    for (int p = 0; p < 3; p++) {
        uint32_t e = ~0u;
        if (mine_err && !any_equal && nb[p] != ~0u && nb[p] < n_clusters) {
            const uint32_t prm = cluster_params[nb[p]];
            e = error_under(prm & 255u, (prm >> 8) & 255u, (prm >> 16) & 255u, prm >> 24);
        }
        neighbour[(size_t)b * 3 + p] = e;
    }
}

hipError_t launch_backend_block_errors(hipStream_t st, const void* d_pixel_blocks, const void* d_etc_blocks, const uint32_t* d_block_cluster, const uint8_t* d_cluster_params,
                                       uint32_t first_block, uint32_t nbx, uint32_t nby, uint32_t n_clusters, bool perceptual, bool with_neighbours, uint32_t* d_own,
                                       uint32_t* d_neighbour) {
    if (!nbx || !nby) return hipSuccess;
    const dim3 grid((nbx * nby + 255) / 256), blk(256);
    const uint4* px = static_cast<const uint4*>(d_pixel_blocks);
    const uint64_t* enc = static_cast<const uint64_t*>(d_etc_blocks);
    const uint32_t* prm = reinterpret_cast<const uint32_t*>(d_cluster_params);
    if (perceptual) hipLaunchKernelGGL(k_backend_block_errors<true>, grid, blk, 0, st, px, enc, d_block_cluster, prm, first_block, nbx, nby, n_clusters, with_neighbours ? 1 : 0, d_own, d_neighbour);
    else hipLaunchKernelGGL(k_backend_block_errors<false>, grid, blk, 0, st, px, enc, d_block_cluster, prm, first_block, nbx, nby, n_clusters, with_neighbours ? 1 : 0, d_own, d_neighbour);
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

size_t refine_workspace_bytes(uint32_t n_clusters, uint32_t n_parents) {
    if (n_clusters > 65535u) return 0;   // positions and cluster ids share a dword in the sorted lists
    const size_t lists = n_parents ? n_parents : 1;
    return ((lists * n_clusters * sizeof(uint2) + 255) & ~(size_t)255) + lists * RS_SEG * sizeof(uint32_t);   // a cluster is at most once in a list
}

hipError_t launch_refine_endpoint_clusterization(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, const uint32_t* d_block_cluster,
                                                 const uint8_t* d_cluster_params, uint32_t n_clusters, uint32_t n_parents, const uint32_t* d_cand_offsets,
                                                 const uint32_t* d_cand_indices, const uint8_t* d_block_parent, bool perceptual, uint32_t* d_out_best, void* d_work) {
    if (!n_blocks) return hipSuccess;
    const dim3 grid((n_blocks + 3) / 4), blk(256);
    const uint4* in = static_cast<const uint4*>(d_pixel_blocks);
    const uint32_t* prm = reinterpret_cast<const uint32_t*>(d_cluster_params);
    if (d_work && refine_workspace_bytes(n_clusters, n_parents)) {
        const size_t lists = n_parents ? n_parents : 1;
        uint2* items = static_cast<uint2*>(d_work);
        uint32_t* seg = reinterpret_cast<uint32_t*>(static_cast<char*>(d_work) + ((lists * n_clusters * sizeof(uint2) + 255) & ~(size_t)255));
        if (perceptual) {
            hipLaunchKernelGGL(k_refine_sort_lists<true>, dim3((uint32_t)lists), blk, 0, st, prm, n_clusters, n_parents, d_cand_offsets, d_cand_indices, items, seg);
            hipLaunchKernelGGL(k_refine_sorted<true>, grid, blk, 0, st, in, n_blocks, d_block_cluster, prm, n_parents, items, seg, d_block_parent, d_out_best);
        } else {
            hipLaunchKernelGGL(k_refine_sort_lists<false>, dim3((uint32_t)lists), blk, 0, st, prm, n_clusters, n_parents, d_cand_offsets, d_cand_indices, items, seg);
            hipLaunchKernelGGL(k_refine_sorted<false>, grid, blk, 0, st, in, n_blocks, d_block_cluster, prm, n_parents, items, seg, d_block_parent, d_out_best);
        }
        BU_LAUNCH_CHECK();
        return hipSuccess;
    }
    if (perceptual) hipLaunchKernelGGL(k_refine_endpoint_clusterization<true>, grid, blk, 0, st, in, n_blocks, d_block_cluster, prm, n_clusters, n_parents, d_cand_offsets, d_cand_indices, d_block_parent, d_out_best);
    else hipLaunchKernelGGL(k_refine_endpoint_clusterization<false>, grid, blk, 0, st, in, n_blocks, d_block_cluster, prm, n_clusters, n_parents, d_cand_offsets, d_cand_indices, d_block_parent, d_out_best);
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_determine_selectors(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, const uint8_t* d_color5_inten,
                                      const uint32_t* d_block_cluster, bool perceptual, void* d_out) {
    if (!n_blocks) return hipSuccess;
    const dim3 grid((n_blocks + 15) / 16), blk(256);
    const uint32_t* pw = static_cast<const uint32_t*>(d_pixel_blocks);
    const uint32_t* prm = reinterpret_cast<const uint32_t*>(d_color5_inten);
    if (perceptual) hipLaunchKernelGGL(k_determine_selectors<true>, grid, blk, 0, st, pw, n_blocks, prm, d_block_cluster, static_cast<uint2*>(d_out));
    else hipLaunchKernelGGL(k_determine_selectors<false>, grid, blk, 0, st, pw, n_blocks, prm, d_block_cluster, static_cast<uint2*>(d_out));
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

size_t create_optimized_selector_codebook_workspace_bytes(uint32_t n_clusters) { return (size_t)n_clusters * 64 * 8; }

hipError_t launch_create_optimized_selector_codebook(hipStream_t st, const void* d_pixel_blocks, const void* d_enc_blocks, uint32_t n_clusters,
                                                     const uint32_t* d_offsets, const uint32_t* d_block_indices, bool perceptual,
                                                     void* d_workspace, void* d_selector_blocks) {
    if (!n_clusters) return hipSuccess;
    unsigned long long* acc = static_cast<unsigned long long*>(d_workspace);
    hipError_t e = hipMemsetAsync(acc, 0, create_optimized_selector_codebook_workspace_bytes(n_clusters), st);
    if (e != hipSuccess) return e;
    const uint32_t* pw = static_cast<const uint32_t*>(d_pixel_blocks);
    {
        const dim3 grid(COSC_WORKGROUPS), blk(256);
        if (perceptual) hipLaunchKernelGGL(k_cosc_accumulate<true>, grid, blk, 0, st, pw, static_cast<const uint64_t*>(d_enc_blocks), n_clusters, d_offsets, d_block_indices, acc);
        else hipLaunchKernelGGL(k_cosc_accumulate<false>, grid, blk, 0, st, pw, static_cast<const uint64_t*>(d_enc_blocks), n_clusters, d_offsets, d_block_indices, acc);
        BU_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_cosc_select, dim3((n_clusters + 3) / 4), dim3(256), 0, st, n_clusters, d_offsets, acc, static_cast<uint64_t*>(d_selector_blocks));
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_find_optimal_selector_clusters(hipStream_t st, const void* d_pixel_blocks, void* d_enc_blocks, uint32_t n_blocks,
                                                 const void* d_selector_blocks, uint32_t n_selectors, uint32_t n_parents, const uint32_t* d_cand_offsets,
                                                 const uint32_t* d_cand_indices, const uint8_t* d_block_parent, bool perceptual, uint32_t chunk,
                                                 uint32_t* d_scratch_idx, uint32_t* d_out_idx, uint32_t* d_cand_words, size_t cand_words_capacity) {
    if (!n_blocks) return hipSuccess;
    const dim3 grid((n_blocks + 3) / 4), blk(256);
    const uint32_t* pw = static_cast<const uint32_t*>(d_pixel_blocks);
    // (a list holds every selector at most once: n_parents x n_selectors entries bound the lists' total, which only the device knows)
    const size_t most = (size_t)(n_parents ? n_parents : 1u) * n_selectors;
    if (d_cand_words && cand_words_capacity >= most && most) {
        hipLaunchKernelGGL(k_fosc_candidate_words, dim3((uint32_t)std::min<size_t>((most + 255) / 256, 2048)), dim3(256), 0, st, static_cast<const uint64_t*>(d_selector_blocks), n_selectors, n_parents,
                           d_cand_offsets, d_cand_indices, d_cand_words);
        BU_LAUNCH_CHECK();
    } else d_cand_words = nullptr;
    if (perceptual) hipLaunchKernelGGL(k_find_optimal_selector_clusters<true>, grid, blk, 0, st, pw, static_cast<const uint64_t*>(d_enc_blocks), n_blocks, static_cast<const uint64_t*>(d_selector_blocks), n_selectors, n_parents, d_cand_offsets, d_cand_indices, d_block_parent, d_scratch_idx, d_cand_words);
    else hipLaunchKernelGGL(k_find_optimal_selector_clusters<false>, grid, blk, 0, st, pw, static_cast<const uint64_t*>(d_enc_blocks), n_blocks, static_cast<const uint64_t*>(d_selector_blocks), n_selectors, n_parents, d_cand_offsets, d_cand_indices, d_block_parent, d_scratch_idx, d_cand_words);
    BU_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fosc_resolve_and_stamp, dim3((n_blocks + 255) / 256), dim3(256), 0, st, static_cast<const uint4*>(d_pixel_blocks),
                       static_cast<uint64_t*>(d_enc_blocks), n_blocks, static_cast<const uint64_t*>(d_selector_blocks), chunk, d_scratch_idx, d_out_idx);
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

// -------------------------------------------------------------------------------------------------------------------
// Input side: basis_compressor::extract_source_blocks (comp.cpp:3207-3268) = image::extract_block_clamped
// per 4x4 block. One lane per block row: a 16-byte read of four texels (clamped at the right / bottom edges) and a 16-byte write,
// so an RGBA raster can be uploaded once and tiled where it lives.
// -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_extract_blocks(const uint8_t* __restrict__ rgba, uint32_t width, uint32_t height, uint32_t pitch,
                                                        uint32_t blocks_x, uint32_t n_blocks, uint4* __restrict__ out) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t block = t >> 2, row = t & 3u;
    if (block >= n_blocks) return;
    const uint32_t bx = block % blocks_x, by = block / blocks_x;
    const uint32_t y = min(by * 4u + row, height - 1u);
    const uint8_t* line = rgba + (size_t)y * pitch;
    uint4 v;
    if (bx * 4u + 3u < width && ((pitch | (uint32_t)(uintptr_t)rgba) & 15u) == 0) {
        v = *reinterpret_cast<const uint4*>(line + (size_t)bx * 16u);
    } else {
        uint32_t p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t x = min(bx * 4u + (uint32_t)k, width - 1u);
            const uint8_t* q = line + (size_t)x * 4u;
            p[k] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        }
        v = make_uint4(p[0], p[1], p[2], p[3]);
    }
    out[(size_t)block * 4u + row] = v;
}

hipError_t launch_extract_blocks(hipStream_t st, const void* d_rgba, uint32_t width, uint32_t height, uint32_t pitch_bytes, void* d_out_blocks) {
    if (!width || !height) return hipSuccess;
    const uint32_t bx = (width + 3) / 4, by = (height + 3) / 4, n = bx * by;
    hipLaunchKernelGGL(k_extract_blocks, dim3((n * 4 + 255) / 256), dim3(256), 0, st, static_cast<const uint8_t*>(d_rgba), width, height, pitch_bytes, bx, n,
                       static_cast<uint4*>(d_out_blocks));
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

} // namespace bu
