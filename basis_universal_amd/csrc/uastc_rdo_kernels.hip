// uastc_rdo_kernels.hip -- the UASTC rate-distortion post-pass (uastc_rdo, encoder/basisu_uastc_enc.cpp:3824-4163; SURVEY.md 8a row
// a20) on gfx950. The per-block pieces are uastc_rdo.h; this file is the GPU decomposition.
//
// The reference walks a strip of blocks in order: block i may take the selector bits of any of the previous `lz_dict_size / 16`
// blocks AS THEY ARE AFTER THEIR OWN RDO STEP, and a history map of selector fields prices the LZ match. That chain is inherently
// serial per strip, so the split is:
//   k_rdo_prepare  1 thread / block      : everything that only depends on the block itself (unpack, current UASTC+BC7 error,
//                                          smooth-block scale, skip decision) -- ~half of the reference's per-block cost, fully parallel
//   k_rdo_strips   1 workgroup / strip   : the serial walk. For block i the 256 threads evaluate up to 256 window candidates at once
//                                          (thread t takes block i-1-t: history lookup, trial decode, UASTC + BC7 error, cost); a
//                                          workgroup min-reduction on (cost, newest first) reproduces the reference's strict "<" scan
//                                          order; thread 0 writes the winner back (mode-0 endpoint refit) and updates the history.
//   k_rdo_rehint   1 thread / modified   : transcode hints of every modified block. Hints live in bits the walk never reads
//                                          (uastc_rdo.h), so they are taken off the serial path and done once, in parallel.
// Strips are the reference's own unit of parallelism (uastc_rdo's total_jobs, comp.cpp:2076-2078 passes min(4, threads)); results are
// bit-identical to the reference run with the same total_jobs. The selector history is an open-addressing table in HBM per strip
// (at most one insert per block, capacity >= 2 x strip length), written by thread 0 only, between workgroup barriers.
#include <hip/hip_runtime.h>
#include <cstring>

#include "uastc_rdo.h"
#include "uastc_kernels.h"

namespace bu {
using namespace bu_uastc;

namespace {

struct rdo_info4 { float ms_err, rms_err, scale; uint32_t mode_skip; };
struct hist_entry { uint64_t sel; uint32_t ofs1; uint32_t idx; };  // ofs1 = first selector bit + 1, 0 = empty

struct rdo_workspace {
    rdo_info4* info; hist_entry* hist; uint32_t* mod_list; uint32_t* counters;  // counters: [0] modified, [1] failed, [2] refined, [3] skipped
    size_t hist_bytes;
};
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

void strip_layout(uint32_t n, uint32_t total_jobs, uint32_t& per_job, uint32_t& n_strips, uint32_t& hist_cap) {
    per_job = total_jobs ? n / total_jobs : 0;  // uastc_rdo, uastc_enc.cpp:4103-4111
    if (total_jobs <= 1 || per_job <= 8) { per_job = 0; n_strips = 1; }
    else n_strips = (n + per_job - 1) / per_job;
    const uint32_t longest = per_job ? per_job : n;
    hist_cap = 64;
    while (hist_cap < 2 * longest) hist_cap <<= 1;
}

rdo_workspace carve(void* base, uint32_t n, uint32_t total_jobs, size_t* total) {
    uint32_t per_job, n_strips, cap;
    strip_layout(n, total_jobs, per_job, n_strips, cap);
    char* p = static_cast<char*>(base);
    size_t o = 0;
    rdo_workspace w;
    w.counters = reinterpret_cast<uint32_t*>(p + o); o += 256;
    w.hist = reinterpret_cast<hist_entry*>(p + o); w.hist_bytes = align_up((size_t)n_strips * cap * sizeof(hist_entry)); o += w.hist_bytes;
    w.info = reinterpret_cast<rdo_info4*>(p + o); o += align_up((size_t)n * sizeof(rdo_info4));
    w.mod_list = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    if (total) *total = o;
    return w;
}

__device__ inline void load_tile(const uint4* px, uint32_t b, rgba8* out) {
    uint4* o = reinterpret_cast<uint4*>(out);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = px[(size_t)b * 4 + k];
}

__global__ void __launch_bounds__(64) k_rdo_prepare(const uint4* __restrict__ blocks, const uint4* __restrict__ px, uint32_t n, rdo_params p,
                                                    rdo_info4* __restrict__ info, uint32_t* __restrict__ counters) {
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n) return;
    alignas(16) uint8_t blk[16];
    *reinterpret_cast<uint4*>(blk) = blocks[b];
    alignas(16) rgba8 t[16];
    load_tile(px, b, t);
    cand c;
    rdo_block_info bi;
    if (!rdo_prepare(blk, t, p, c, bi)) {
        atomicExch(&counters[1], 1u);
        bi.mode = 8; bi.skip = 0; bi.ms_err = bi.rms_err = 0.0f; bi.scale = 1.0f;
    }
    rdo_info4 o = { bi.ms_err, bi.rms_err, bi.scale, bi.mode | (bi.skip << 8) };
    info[b] = o;
}

__device__ inline uint32_t hist_hash(uint32_t ofs, uint64_t sel) {
    uint64_t x = (sel + ofs) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    return (uint32_t)(x >> 32);
}
__device__ inline int hist_find(const hist_entry* t, uint32_t mask, uint32_t ofs, uint64_t sel) {
    for (uint32_t h = hist_hash(ofs, sel) & mask;; h = (h + 1) & mask) {
        const hist_entry e = t[h];
        if (!e.ofs1) return -1;
        if (e.ofs1 == ofs + 1 && e.sel == sel) return (int)e.idx;
    }
}
__device__ inline void hist_put(hist_entry* t, uint32_t mask, uint32_t ofs, uint64_t sel, uint32_t idx) {
    for (uint32_t h = hist_hash(ofs, sel) & mask;; h = (h + 1) & mask) {
        const hist_entry e = t[h];
        if (!e.ofs1 || (e.ofs1 == ofs + 1 && e.sel == sel)) {
            const hist_entry w = { sel, ofs + 1, idx };
            t[h] = w;
            return;
        }
    }
}

__device__ inline uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const uint64_t o = __shfl_xor(v, s, 64);
        v = o < v ? o : v;
    }
    return v;
}

constexpr uint32_t RDO_THREADS = 256;

__global__ void __launch_bounds__(RDO_THREADS) k_rdo_strips(uint4* blocks, const uint4* __restrict__ px, uint32_t n, uint32_t per_job, rdo_params p,
                                                            const rdo_info4* __restrict__ info, hist_entry* hist_all, uint32_t hist_cap,
                                                            uint32_t* mod_list, uint32_t* counters) {
    const uint32_t tid = threadIdx.x;
    const uint32_t first = per_job ? blockIdx.x * per_job : 0;
    const uint32_t last = per_job ? (first + per_job < n ? first + per_job : n) : n;
    hist_entry* hist = hist_all + (size_t)blockIdx.x * hist_cap;
    const uint32_t mask = hist_cap - 1;
    const int window = (int)(p.lz_dict_size / 16 > 1 ? p.lz_dict_size / 16 : 1);
    __shared__ uint64_t s_key[RDO_THREADS / 64];

    for (uint32_t i = first; i < last; i++) {
        const rdo_info4 inf = info[i];
        const uint32_t mode = inf.mode_skip & 255u;
        if (mode == 8) continue;  // solid: untouched, not part of the history (uastc_enc.cpp:3842-3843)
        alignas(16) uint8_t blk[16];
        *reinterpret_cast<uint4*>(blk) = blocks[i];
        const uint32_t fsb = ku_sel_first[mode], len = ku_sel_len[mode], len_lo = len < 64 ? len : 64;
        const uint64_t cur_lo = block_bits(blk, fsb, len_lo);
        if (inf.mode_skip >> 8) {  // already too much error: only remembered (:3900-3910)
            if (tid == 0) { hist_put(hist, mask, fsb, cur_lo, i); atomicAdd(&counters[3], 1u); }
            __syncthreads();
            continue;
        }
        cand cur;
        unpack_block(blk, cur);
        alignas(16) rgba8 t[16];
        load_tile(px, i, t);
        const int seen = hist_find(hist, mask, fsb, cur_lo);
        const int cur_bits = seen < 0 ? (int)((len * p.lz_literal_cost) / 100) : (int)match_cost((i - (uint32_t)seen) * 16);
        const float t0 = inf.ms_err * inf.scale + (float)cur_bits * p.lambda;
        rdo_block_info bi;
        bi.ms_err = inf.ms_err; bi.rms_err = inf.rms_err; bi.scale = inf.scale; bi.mode = mode; bi.skip = 0;

        // newest-first scan with strict "<": the minimum cost, and among equal costs the newest block, wins
        uint64_t key = ~0ull;
        const int lo_j = (int)i - window > (int)first ? (int)i - window : (int)first;
        for (int base = (int)i - 1; base >= lo_j; base -= (int)RDO_THREADS) {
            const int j = base - (int)tid;
            if (j < lo_j) continue;
            alignas(16) uint8_t prev[16];
            *reinterpret_cast<uint4*>(prev) = blocks[j];
            const uint64_t lo = block_bits(prev, fsb, len_lo), hi = len > 64 ? block_bits(prev, fsb + 64, len - 64) : 0;
            const int hit = hist_find(hist, mask, fsb, lo);
            const int match = hit < 0 ? j : hit;
            if (match > j) continue;  // this bit pattern is tried at its newest occurrence only (:3936-3942)
            float ms;
            if (!rdo_trial(cur, lo, hi, t, bi, p, ms)) continue;
            const float cost = ms * inf.scale + (float)(int)match_cost((i - (uint32_t)match) * 16) * p.lambda;
            if (cost < t0) {
                const uint64_t k = ((uint64_t)__float_as_uint(cost) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)j);
                key = k < key ? k : key;
            }
        }
        key = wave_min_u64(key);
        if ((tid & 63u) == 0) s_key[tid >> 6] = key;
        __syncthreads();
        uint64_t best = s_key[0];
#pragma unroll
        for (uint32_t w = 1; w < RDO_THREADS / 64; w++) best = s_key[w] < best ? s_key[w] : best;

        if (tid == 0) {
            uint64_t final_lo = cur_lo;
            if (best != ~0ull) {
                const uint32_t j = 0xFFFFFFFFu - (uint32_t)best;
                alignas(16) uint8_t prev[16];
                *reinterpret_cast<uint4*>(prev) = blocks[j];
                const uint64_t lo = block_bits(prev, fsb, len_lo), hi = len > 64 ? block_bits(prev, fsb + 64, len - 64) : 0;
                alignas(16) uint8_t out[16];
                bool refined;
                rdo_write_back(cur, lo, hi, t, p, out, refined);
                blocks[i] = *reinterpret_cast<const uint4*>(out);
                mod_list[atomicAdd(&counters[0], 1u)] = i;
                if (refined) atomicAdd(&counters[2], 1u);
                final_lo = lo;
            }
            hist_put(hist, mask, fsb, final_lo, i);
            __threadfence_block();
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(64) k_rdo_rehint(uint4* blocks, const uint4* __restrict__ px, enc_cfg e, const uint32_t* __restrict__ mod_list,
                                                   const uint32_t* __restrict__ counters) {
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= counters[0]) return;
    const uint32_t b = mod_list[k];
    alignas(16) uint8_t blk[16];
    *reinterpret_cast<uint4*>(blk) = blocks[b];
    alignas(16) rgba8 t[16];
    load_tile(px, b, t);
    rdo_rehint(t, e, blk);
    blocks[b] = *reinterpret_cast<const uint4*>(blk);
}

rdo_params to_params(const float* f, const uint32_t* u) {
    rdo_params p;
    p.lambda = f[0]; p.max_allowed_rms_increase_ratio = f[1]; p.skip_block_rms_thresh = f[2]; p.max_smooth_block_std_dev = f[3];
    p.smooth_block_max_error_scale = f[4];
    p.lz_dict_size = u[0]; p.lz_literal_cost = u[1]; p.endpoint_refinement = u[2];
    return p;
}

} // namespace

size_t uastc_rdo_workspace_bytes(uint32_t n_blocks, uint32_t total_jobs) {
    size_t total = 0;
    carve(nullptr, n_blocks, total_jobs, &total);
    return total;
}

uint32_t uastc_rdo_strips(uint32_t n_blocks, uint32_t total_jobs) {
    uint32_t per_job, n_strips, cap;
    strip_layout(n_blocks, total_jobs, per_job, n_strips, cap);
    return n_strips;
}

hipError_t launch_uastc_rdo_phase(hipStream_t st, int phase, void* d_blocks, const void* d_px, uint32_t n, const float* fparams, const uint32_t* uparams,
                                  uint32_t flags, uint32_t total_jobs, void* d_ws) {
    if (!n) return hipSuccess;
    const rdo_workspace w = carve(d_ws, n, total_jobs, nullptr);
    const rdo_params p = to_params(fparams, uparams);
    uint32_t per_job, n_strips, cap;
    strip_layout(n, total_jobs, per_job, n_strips, cap);
    uint4* blocks = static_cast<uint4*>(d_blocks);
    const uint4* px = static_cast<const uint4*>(d_px);
    const uint32_t gx = (n + 63) / 64;
    switch (phase) {
    case 0: {
        const hipError_t e = hipMemsetAsync(w.counters, 0, 256 + w.hist_bytes, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_rdo_prepare, dim3(gx), dim3(64), 0, st, blocks, px, n, p, w.info, w.counters);
        break;
    }
    case 1:
        hipLaunchKernelGGL(k_rdo_strips, dim3(n_strips), dim3(RDO_THREADS), 0, st, blocks, px, n, per_job, p, w.info, w.hist, cap, w.mod_list, w.counters);
        break;
    default: {
        enc_cfg e;
        make_cfg(flags, e);
        hipLaunchKernelGGL(k_rdo_rehint, dim3(gx), dim3(64), 0, st, blocks, px, e, w.mod_list, w.counters);
        break;
    }
    }
    return hipGetLastError();
}

const void* uastc_rdo_counters(void* d_ws, uint32_t n, uint32_t total_jobs) { return carve(d_ws, n, total_jobs, nullptr).counters; }

} // namespace bu
