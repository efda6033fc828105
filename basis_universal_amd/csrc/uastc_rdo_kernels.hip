// uastc_rdo_kernels.hip -- the UASTC rate-distortion post-pass (uastc_rdo, encoder/basisu_uastc_enc.cpp:3824-4163; SURVEY.md 8a row
// a20) on gfx950. The per-block pieces are uastc_rdo.h; this file is the GPU decomposition.
//
// The reference walks a strip of blocks in order: block i may take the selector bits of any of the previous `lz_dict_size / 16`
// blocks AS THEY ARE AFTER THEIR OWN RDO STEP, and a history map of selector fields prices the LZ match. That chain is inherently
// serial per strip, so everything that is not part of the chain is taken off it:
//   k_rdo_prepare  1 thread / block      : what depends on the block alone -- unpack, current UASTC+BC7 error, smooth-block scale, skip
//                                          decision, and the block's weight-error table E[k][v] (uastc_rdo.h: a trial's error is a sum of
//                                          16/32 table entries, so the walk never decodes anything)
//   k_rdo_strips   1 workgroup / strip   : the serial walk. For block i the 256 threads score up to 256 window candidates at once (thread
//                                          t takes block i-1-t out of an LDS ring of the last blocks: history lookup in HBM, table sum out
//                                          of LDS, cost); a min-reduction on (cost, newest first) reproduces the reference's strict "<" scan
//                                          order; thread 0 stores the winner's raw bits and updates the history. The next block's table,
//                                          info and bytes are prefetched into registers one step ahead. Two barriers per block.
//   k_rdo_finish   1 thread / modified   : mode-0 endpoint refit (deferred, see uastc_rdo.h) and the transcode hints of every modified
//                                          block. Hints and -- except under modes 15/17/18, which settle their window first -- the refit
//                                          live in bits the walk never reads.
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

// what the walk needs of a block, so that it never touches a constant table: meta = mode | skip << 8 | pattern << 16 | ccs << 24,
// field = first selector bit | field length << 8 | weight bits << 16 | weight slots << 24, anchors = rdo_anchor_mask
struct rdo_info4 { float ms_err, rms_err, scale; uint32_t meta, field, anchors, pad0, pad1; };
struct hist_entry { uint64_t sel; uint32_t ofs1; uint32_t idx; };    // ofs1 = first selector bit + 1, 0 = empty
constexpr uint32_t HIST_BUCKET = 4;  // entries per bucket = one 64-byte line, fetched with one round trip

struct rdo_workspace {
    rdo_info4* info; hist_entry* hist; uint32_t* mod_list; uint32_t* counters;  // counters: [0] modified, [1] failed, [2] refined, [3] skipped
    uint32_t* strip_counts;  // modified blocks per strip
    uint32_t* strip_flags;   // per strip: != 0 when it holds an active block of a sensitive mode (15 / 17 / 18) -- set by the prepare pass, decides which walk kernel takes the strip
    uint8_t* state;    // per block: 0 untouched, 1 modified + refit pending, 2 modified
    uint32_t* table;   // per block RDO_TABLE_WORDS
    size_t zero_bytes; // counters + hist + state, contiguous
};
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

void strip_layout(uint32_t n, uint32_t total_jobs, uint32_t& per_job, uint32_t& n_strips, uint32_t& hist_cap) {
    per_job = total_jobs ? n / total_jobs : 0;  // uastc_rdo, uastc_enc.cpp:4103-4111
    if (total_jobs <= 1 || per_job <= 8) { per_job = 0; n_strips = 1; }
    else n_strips = (n + per_job - 1) / per_job;
    const uint32_t longest = per_job ? per_job : n;
    hist_cap = 64;
    while (hist_cap < 4 * (uint64_t)longest && hist_cap < (1u << 30)) hist_cap <<= 1;  // load <= 1/4: a bucket of 4 holds one key on average
}

rdo_workspace carve(void* base, uint32_t n, uint32_t total_jobs, size_t* total) {
    uint32_t per_job, n_strips, cap;
    strip_layout(n, total_jobs, per_job, n_strips, cap);
    char* p = static_cast<char*>(base);
    size_t o = 0;
    rdo_workspace w;
    w.counters = reinterpret_cast<uint32_t*>(p + o); o += 256;
    w.strip_counts = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n_strips * 4);
    w.strip_flags = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n_strips * 4);
    w.hist = reinterpret_cast<hist_entry*>(p + o); o += align_up((size_t)n_strips * cap * sizeof(hist_entry));
    w.state = reinterpret_cast<uint8_t*>(p + o); o += align_up(n);
    w.zero_bytes = o;
    w.info = reinterpret_cast<rdo_info4*>(p + o); o += align_up((size_t)n * sizeof(rdo_info4));
    w.mod_list = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    w.table = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * RDO_TABLE_WORDS * 4);
    if (total) *total = o;
    return w;
}

__device__ inline void load_tile(const uint4* px, uint32_t b, rgba8* out) {
    uint4* o = reinterpret_cast<uint4*>(out);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = px[(size_t)b * 4 + k];
}

__global__ void __launch_bounds__(64) k_rdo_prepare(const uint4* __restrict__ blocks, const uint4* __restrict__ px, uint32_t n, rdo_params p,
                                                    rdo_info4* __restrict__ info, uint32_t* __restrict__ table, uint32_t* __restrict__ counters,
                                                    uint32_t* __restrict__ strip_flags, uint32_t per_job) {
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
        c.pattern = 0; c.ccs = 0;
    }
    rdo_info4 o = { bi.ms_err, bi.rms_err, bi.scale, bi.mode | (bi.skip << 8) | ((uint32_t)c.pattern << 16) | ((uint32_t)c.ccs << 24), 0, 0, 0, 0 };
    if (bi.mode != 8) {
        o.field = ku_sel_first[bi.mode] | ((uint32_t)ku_sel_len[bi.mode] << 8) | ((uint32_t)ku_mode_weight_bits[bi.mode] << 16) |
                  ((16u * ku_mode_planes[bi.mode]) << 24);
        o.anchors = rdo_anchor_mask(bi.mode, c.pattern);
    }
    info[b] = o;
    if (bi.mode == 8 || bi.skip) return;
    if (rdo_mode_reads_endpoint_bits(bi.mode) && p.endpoint_refinement) atomicOr(&strip_flags[per_job ? b / per_job : 0u], 1u);   // a handful per image
    texel_ends ends;
    rdo_texel_ends(c, ends);
    const uint32_t planes = ku_mode_planes[bi.mode], wbits = ku_mode_weight_bits[bi.mode];
    uint32_t* row = table + (size_t)b * RDO_TABLE_WORDS;
    for (uint32_t k = 0; k < 16 * planes; k++) {
        const uint32_t texel = k / planes;
        const uint32_t pxw = pack_px(t[texel].c);
        for (uint32_t v = 0; v < (1u << wbits); v++)
            row[(k << wbits) + v] = rdo_weight_error(bi.mode, c.ccs, k, v, ends.ul[texel], ends.uh[texel], ends.bl[texel], ends.bh[texel], pxw);
    }
}

// The history: buckets of HIST_BUCKET entries, a key lives in the first free entry of its home bucket or, when that is full, of the next
// ones. A lookup fetches the whole home bucket at once and all but never needs a second round trip.
struct hist_bucket { uint4 e0, e1, e2, e3; };  // entry = {sel lo, sel hi, ofs1, idx}; plain dwords so the bucket stays in registers
__device__ inline uint32_t hist_home(uint32_t ofs, uint64_t sel, uint32_t bucket_mask) {
    uint32_t h = (uint32_t)sel * 0x9E3779B1u ^ (uint32_t)(sel >> 32) * 0x85EBCA77u ^ ofs * 0xC2B2AE3Du;
    h ^= h >> 15;
    h *= 0x2C1B3C6Du;
    h ^= h >> 13;
    return h & bucket_mask;
}
__device__ inline hist_bucket hist_load(const hist_entry* t, uint32_t bucket) {
    const uint4* src = reinterpret_cast<const uint4*>(t) + (size_t)bucket * HIST_BUCKET;
    hist_bucket b = { src[0], src[1], src[2], src[3] };
    return b;
}
// 0: other key, 1: free, 2: this key
__device__ inline int hist_match(const uint4& e, uint32_t ofs1, uint32_t sel_lo, uint32_t sel_hi) {
    return e.z == 0 ? 1 : ((e.z == ofs1 && e.x == sel_lo && e.y == sel_hi) ? 2 : 0);
}
// finish a lookup whose home bucket `b` (index `bucket`) is loaded: index of the key's last block or -1; `slot` ends on the key's entry, or on
// the free entry it would be inserted at (nothing is inserted between a lookup and the end of the step, so the slot stays good)
__device__ inline int hist_resolve(const hist_entry* t, uint32_t bucket_mask, uint32_t ofs, uint64_t sel, uint32_t bucket, hist_bucket b, uint32_t& slot) {
    const uint32_t ofs1 = ofs + 1, sl = (uint32_t)sel, sh = (uint32_t)(sel >> 32);
    for (;;) {
        const int m0 = hist_match(b.e0, ofs1, sl, sh), m1 = hist_match(b.e1, ofs1, sl, sh), m2 = hist_match(b.e2, ofs1, sl, sh), m3 = hist_match(b.e3, ofs1, sl, sh);
        if (m0) { slot = bucket * HIST_BUCKET + 0; return m0 == 2 ? (int)b.e0.w : -1; }
        if (m1) { slot = bucket * HIST_BUCKET + 1; return m1 == 2 ? (int)b.e1.w : -1; }
        if (m2) { slot = bucket * HIST_BUCKET + 2; return m2 == 2 ? (int)b.e2.w : -1; }
        if (m3) { slot = bucket * HIST_BUCKET + 3; return m3 == 2 ? (int)b.e3.w : -1; }
        bucket = (bucket + 1) & bucket_mask;
        b = hist_load(t, bucket);
    }
}
__device__ inline void hist_store(hist_entry* t, uint32_t slot, uint32_t ofs, uint64_t sel, uint32_t idx) {
    *reinterpret_cast<uint4*>(t + slot) = make_uint4((uint32_t)sel, (uint32_t)(sel >> 32), ofs + 1, idx);
}

// the selector field [fsb, fsb + len) of a block held as four dwords: lo = its first 64 bits, hi = the rest (block_bits on registers)
__device__ inline void field_of(const uint4& v, uint32_t fsb, uint32_t len, uint64_t& lo, uint64_t& hi) {
    const uint64_t a = (uint64_t)v.x | ((uint64_t)v.y << 32), b = (uint64_t)v.z | ((uint64_t)v.w << 32);
    if (fsb >= 64) { lo = b >> (fsb - 64); hi = 0; }
    else { lo = (a >> fsb) | (b << (64 - fsb)); hi = b >> fsb; }  // fsb is never 0 (ku_sel_first)
    if (len < 64) { lo &= (1ull << len) - 1; hi = 0; }
    else hi &= (1ull << (len - 64)) - 1;
}

// rdo_trial_sum_n (uastc_rdo.h) for the GPU. With one subset (or two planes) the anchors are the first slot(s), so every weight sits at a
// compile-time bit offset of the field: one v_bfe_u32 (or v_alignbit_b32 across a dword boundary) per weight, no cursor bookkeeping.
// The modes with pattern-dependent anchors take the generic walk.
template <uint32_t WBITS, uint32_t NSLOTS, uint32_t NANCHORS>
__device__ inline uint32_t trial_sum_static(const uint32_t* tab, uint64_t lo, uint64_t hi) {
    const uint32_t f[4] = { (uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32) };
    uint32_t total = 0;
#pragma unroll
    for (uint32_t g = 0; g < NSLOTS; g += 16) {
        uint32_t idx[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t slot = g + k;
            const uint32_t nb = WBITS - (slot < NANCHORS ? 1u : 0u);
            const uint32_t ofs = slot * WBITS - (slot < NANCHORS ? slot : NANCHORS);
            const uint32_t word = ofs >> 5, sh = ofs & 31u;
            uint32_t v;
            if (nb == 0) v = 0;
            else if (sh + nb <= 32) v = (f[word] >> sh) & ((1u << nb) - 1);
            else v = __builtin_amdgcn_alignbit(f[word + 1 < 4 ? word + 1 : 3], f[word], sh) & ((1u << nb) - 1);
            idx[k] = (slot << WBITS) + v;
        }
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) total += tab[idx[k]];
    }
    return total;
}
__device__ inline uint32_t trial_sum_walk(const uint32_t* tab, uint32_t n_slots, uint32_t wbits, uint32_t amask, uint64_t lo, uint64_t hi) {
    uint32_t cur = (uint32_t)lo, nxt = (uint32_t)(lo >> 32), after = (uint32_t)hi, last = (uint32_t)(hi >> 32);
    uint32_t pos = 0, total = 0;  // pos: uniform (amask and wbits are)
    for (uint32_t g = 0; g < n_slots; g += 16) {
        uint32_t idx[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t nb = wbits - ((amask >> (g + k)) & 1u);
            idx[k] = ((g + k) << wbits) + (__builtin_amdgcn_alignbit(nxt, cur, pos) & ((1u << nb) - 1));
            pos += nb;
            if (pos >= 32) { pos -= 32; cur = nxt; nxt = after; after = last; last = 0; }
        }
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) total += tab[idx[k]];
    }
    return total;
}
__device__ inline uint32_t trial_sum_fast(const uint32_t* tab, uint32_t n_slots, uint32_t wbits, uint32_t amask, uint64_t lo, uint64_t hi) {
    if (n_slots == 16 && amask == 1u) {  // uniform dispatch
        switch (wbits) {
        case 4: return trial_sum_static<4, 16, 1>(tab, lo, hi);   // modes 0, 10, 15
        case 2: return trial_sum_static<2, 16, 1>(tab, lo, hi);   // 1, 14
        case 3: return trial_sum_static<3, 16, 1>(tab, lo, hi);   // 5, 12
        case 5: return trial_sum_static<5, 16, 1>(tab, lo, hi);   // 18
        default: break;
        }
    } else if (n_slots == 32 && amask == 3u) {
        if (wbits == 2) return trial_sum_static<2, 32, 2>(tab, lo, hi);  // 6, 11, 17
        if (wbits == 1) return trial_sum_static<1, 32, 2>(tab, lo, hi);  // 13
    }
    return trial_sum_walk(tab, n_slots, wbits, amask, lo, hi);
}

constexpr uint32_t RDO_THREADS = 256;
constexpr uint32_t RDO_RING_MAX = 2048;  // blocks of look-back kept in LDS (32 KiB); larger dictionaries read the blocks from HBM


// modes 15/17/18 read bits a pending mode-0 refit may change: settle the window first (uastc_rdo.h, "Deferred form"). Rare.
__device__ __noinline__ void settle_window(uint4* blocks, const uint4* px, uint8_t* state, uint4* ring, uint32_t ring_mask, int lo_j, int i, const rdo_params& p,
                                           uint32_t* counters) {
    for (int j = i - 1 - (int)threadIdx.x; j >= lo_j; j -= (int)RDO_THREADS) {
        if (state[j] != 1) continue;
        alignas(16) uint8_t blk[16];
        *reinterpret_cast<uint4*>(blk) = blocks[j];
        alignas(16) rgba8 t[16];
        load_tile(px, (uint32_t)j, t);
        bool refined;
        rdo_refit_block(t, p, blk, refined);
        if (refined) {
            blocks[j] = *reinterpret_cast<const uint4*>(blk);
            if (ring) ring[(uint32_t)j & ring_mask] = *reinterpret_cast<const uint4*>(blk);
            atomicAdd(&counters[2], 1u);
        }
        state[j] = 2;
    }
}

// put_field (uastc_rdo.h) on registers: the block with its selector field replaced
__device__ inline uint4 with_field(const uint4& v, uint32_t fsb, uint32_t len, uint64_t lo, uint64_t hi) {
    typedef unsigned __int128 u128;
    const u128 whole = ((u128)((uint64_t)v.z | ((uint64_t)v.w << 32)) << 64) | (u128)((uint64_t)v.x | ((uint64_t)v.y << 32));
    const u128 field = ((u128)hi << 64) | (u128)lo;
    const u128 ones = len >= 128 ? ~(u128)0 : (((u128)1 << len) - 1);
    const u128 r = (whole & ~(ones << fsb)) | ((field & ones) << fsb);
    const uint64_t a = (uint64_t)r, b = (uint64_t)(r >> 64);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): unlike __syncthreads() it does not drain the
// vector-memory loads in flight.
__device__ inline void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

#ifdef RDO_PROFILE
#define RDO_TICK(k) do { const long long now_ = clock64(); if (tid == 0) prof[k] += now_ - tick_; tick_ = now_; } while (0)
#else
#define RDO_TICK(k) do { } while (0)
#endif

struct cand_rec { uint64_t lo, hi; uint32_t slot, pad; };

// RING: the look-back window fits the LDS ring (the normal case); otherwise candidates are read back from HBM. Two instantiations, so the
// ring version has no global-load path whose wait would also drain the loads in flight.
// FAT: the walk of strips that hold a block of a sensitive mode (strip_flags, set by the prepare pass): in front of such a block the pending mode-0 refits of its
// window are settled -- the encoder's colour-cell fit, 256 registers + 34 accumulation registers + scratch that the walk itself (110 registers) has no use for; with
// it in the kernel a strip holds a whole CU at one wave per SIMD for the 20 ms of its chain. Sensitive blocks are a handful per image (alpha / luminance-alpha
// content apart), so most strips take the LEAN build -- no refit in it, four waves per SIMD, a 256-register encode wave of the next batch fits beside it -- and
// only the flagged ones the FAT one; both launches cover all strips and a workgroup whose strip belongs to the other returns at once. The launcher puts them on
// two streams, so the chain of a batch is as long as before.
template <bool RING, bool FAT>
__global__ void __launch_bounds__(RDO_THREADS) k_rdo_strips(uint4* blocks, const uint4* __restrict__ px, uint32_t n, uint32_t per_job, rdo_params p,
                                                            const rdo_info4* info, const uint32_t* __restrict__ table, hist_entry* hist_all,
                                                            uint32_t hist_cap, uint8_t* state, uint32_t* mod_list, uint32_t* strip_counts, uint32_t* counters,
                                                            uint32_t ring_slots, const uint32_t* __restrict__ strip_flags) {
    if ((strip_flags[blockIdx.x] != 0u) != FAT) return;
    extern __shared__ uint4 s_ring_mem[];
    __shared__ uint32_t s_table[RDO_TABLE_WORDS];
    __shared__ cand_rec s_cand[RDO_THREADS];
    __shared__ unsigned long long s_min;
    const uint32_t tid = threadIdx.x;
    const uint32_t first = per_job ? blockIdx.x * per_job : 0;
    const uint32_t last = per_job ? (first + per_job < n ? first + per_job : n) : n;
    hist_entry* hist = hist_all + (size_t)blockIdx.x * hist_cap;
    const uint32_t bucket_mask = hist_cap / HIST_BUCKET - 1;
    const int window = (int)(p.lz_dict_size / 16 > 1 ? p.lz_dict_size / 16 : 1);
    uint4* ring = RING ? s_ring_mem : nullptr;
    const uint32_t ring_mask = ring_slots - 1;
    // Block i + 1's info (8 dwords) and bytes (4 dwords) are fetched one step ahead as ONE dword per lane (lane & 15 picks it) and read out
    // with readlane when the step begins: a uniform load would be pulled into scalar registers -- and waited for -- right where it is issued.
    const uint32_t pre_lane = tid & 15u;
    const uint32_t* info_words = reinterpret_cast<const uint32_t*>(info);
    const uint32_t* block_words = reinterpret_cast<const uint32_t*>(blocks);
    auto prefetch = [&](uint32_t b) -> uint32_t {
        const uint32_t* src = pre_lane < 8 ? info_words + (size_t)b * 8 + pre_lane : block_words + (size_t)b * 4 + (pre_lane & 3u);
        return pre_lane < 12 ? *src : 0u;
    };
    uint32_t n_modified = 0, n_skipped = 0;  // thread 0
#ifdef RDO_PROFILE
    long long prof[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long long tick_ = clock64();
#endif

    uint32_t pre = prefetch(first);
    s_table[tid] = table[(size_t)first * RDO_TABLE_WORDS + tid];
    s_table[tid + RDO_THREADS] = table[(size_t)first * RDO_TABLE_WORDS + RDO_THREADS + tid];
    if (tid == 0) s_min = ~0ull;
    __syncthreads();

    for (uint32_t i = first; i < last; i++) {
        // one step ahead: the next block is still the encoder's output, and its table / info never change
        const bool has_next = i + 1 < last;
        rdo_info4 inf;
        inf.ms_err = __uint_as_float(__builtin_amdgcn_readlane(pre, 0)); inf.rms_err = __uint_as_float(__builtin_amdgcn_readlane(pre, 1));
        inf.scale = __uint_as_float(__builtin_amdgcn_readlane(pre, 2)); inf.meta = __builtin_amdgcn_readlane(pre, 3);
        inf.field = __builtin_amdgcn_readlane(pre, 4); inf.anchors = __builtin_amdgcn_readlane(pre, 5);
        const uint4 blkv = make_uint4(__builtin_amdgcn_readlane(pre, 8), __builtin_amdgcn_readlane(pre, 9), __builtin_amdgcn_readlane(pre, 10),
                                      __builtin_amdgcn_readlane(pre, 11));
        const uint32_t mode = inf.meta & 255u, skip = (inf.meta >> 8) & 255u;
        const bool active = mode != 8 && !skip;
        const uint32_t fsb = inf.field & 255u, len = (inf.field >> 8) & 255u;
        uint32_t seen_slot = 0;
        uint64_t cur_lo = 0, cur_hi = 0;
        if (mode != 8) field_of(blkv, fsb, len, cur_lo, cur_hi);
        RDO_TICK(0);
        if (active) {
            const int lo_j = (int)i - window > (int)first ? (int)i - window : (int)first;
            if (FAT && rdo_mode_reads_endpoint_bits(mode) && p.endpoint_refinement) {  // uniform branch; a strip with such a block is never given to the lean build
                settle_window(blocks, px, state, ring, ring_mask, lo_j, (int)i, p, counters);
                __syncthreads();
            }
            // first probes of the block's own key and of the candidate's key go out together; the table sum runs under their latency
            const uint32_t seen_home = hist_home(fsb, cur_lo, bucket_mask);
            const hist_bucket seen_b = hist_load(hist, seen_home);
            const float max_rms = inf.rms_err * p.max_allowed_rms_increase_ratio;
            const uint32_t amask = inf.anchors, wbits = (inf.field >> 16) & 255u, n_slots = inf.field >> 24;
            unsigned long long my_key = ~0ull;
            cand_rec mine = { 0, 0, 0, 0 };
            // newest-first scan with strict "<": the minimum cost, and among equal costs the newest block (smallest distance), wins. The
            // comparison against the block's own cost t0 is applied to the lane's minimum afterwards (same outcome: only the minimum matters).
            for (int base = (int)i - 1; base >= lo_j; base -= (int)RDO_THREADS) {
                const int j = base - (int)tid;
                if (j >= lo_j) {
                    uint4 pv;
                    if (RING) pv = s_ring_mem[(uint32_t)j & ring_mask];
                    else pv = blocks[j];
                    uint64_t lo, hi;
                    field_of(pv, fsb, len, lo, hi);
                    const uint32_t home = hist_home(fsb, lo, bucket_mask);
                    const hist_bucket cb = hist_load(hist, home);
                    RDO_TICK(8);
                    const uint32_t sum = trial_sum_fast(s_table, n_slots, wbits, amask, lo, hi);
                    RDO_TICK(9);
                    uint32_t slot;
                    const int hit = hist_resolve(hist, bucket_mask, fsb, lo, home, cb, slot);
                    RDO_TICK(10);
                    const int match = hit < 0 ? j : hit;
                    const float ms = (float)(uint64_t)(sum / 2) * (1.0f / 64.0f);
                    // match > j: this bit pattern is tried at its newest occurrence only (:3936-3942)
                    if (match <= j && !(sqrtf(ms) > max_rms)) {
                        const float cost = ms * inf.scale + (float)(int)match_cost((i - (uint32_t)match) * 16) * p.lambda;
                        const unsigned long long k = ((unsigned long long)__float_as_uint(cost) << 32) | (unsigned long long)(i - (uint32_t)j);
                        if (k < my_key) { my_key = k; mine.lo = lo; mine.hi = hi; mine.slot = slot; }
                    }
                }
            }
            RDO_TICK(1);
            const int seen = hist_resolve(hist, bucket_mask, fsb, cur_lo, seen_home, seen_b, seen_slot);
            const int cur_bits = seen < 0 ? (int)((len * p.lz_literal_cost) / 100) : (int)match_cost((i - (uint32_t)seen) * 16);
            const float t0 = inf.ms_err * inf.scale + (float)cur_bits * p.lambda;
            if (!(__uint_as_float((uint32_t)(my_key >> 32)) < t0)) my_key = ~0ull;  // costs are positive floats: their bit patterns order like the values
            RDO_TICK(3);
            if (my_key != ~0ull) {
                s_cand[tid] = mine;
                atomicMin(&s_min, my_key);
            }
        }
        // The next block's info, bytes and table are requested HERE, as the youngest loads of the step: vector memory returns in order, so
        // anything issued before the history lookups would have to land before they can be consumed. This way the HBM latency of the
        // 2 KiB table row runs under the barrier and thread 0's write-back.
        uint32_t npre = 0, nt0 = 0, nt1 = 0;
        if (has_next) {
            npre = prefetch(i + 1);
            nt0 = table[(size_t)(i + 1) * RDO_TABLE_WORDS + tid];
            nt1 = table[(size_t)(i + 1) * RDO_TABLE_WORDS + RDO_THREADS + tid];
        }
        RDO_TICK(4);
        lds_barrier();  // the winner is posted; nobody reads s_table any more. LDS-only: the loads just issued stay in flight
        RDO_TICK(5);
        if (tid == 0) {
            uint4 final_blk = blkv;
            if (mode != 8) {
                uint64_t final_lo = cur_lo;
                uint32_t put_slot = seen_slot;
                if (active) {
                    const unsigned long long best = s_min;
                    if (best != ~0ull) {
                        s_min = ~0ull;
                        const uint32_t dist = (uint32_t)best;                 // i - j
                        const cand_rec w = s_cand[(dist - 1) % RDO_THREADS];  // thread t scans j = i - 1 - t - 256 m
                        final_blk = with_field(blkv, fsb, len, w.lo, w.hi);
                        blocks[i] = final_blk;
                        state[i] = (p.endpoint_refinement && mode == 0) ? 1 : 2;
                        mod_list[first + n_modified++] = i;
                        final_lo = w.lo;
                        put_slot = w.slot;
                    }
                    hist_store(hist, put_slot, fsb, final_lo, i);
                } else {
                    n_skipped++;
                    const uint32_t home = hist_home(fsb, final_lo, bucket_mask);
                    hist_resolve(hist, bucket_mask, fsb, final_lo, home, hist_load(hist, home), put_slot);
                    hist_store(hist, put_slot, fsb, final_lo, i);
                }
            }
            if (RING) s_ring_mem[i & ring_mask] = final_blk;
        }
        if (has_next) { s_table[tid] = nt0; s_table[tid + RDO_THREADS] = nt1; }
        RDO_TICK(6);
        __syncthreads();
        RDO_TICK(7);
        pre = npre;
    }
    if (tid == 0) {
        strip_counts[blockIdx.x] = n_modified;
        atomicAdd(&counters[0], n_modified);
        atomicAdd(&counters[3], n_skipped);
#ifdef RDO_PROFILE
        if (blockIdx.x == 0) for (int k = 0; k < 16; k++) reinterpret_cast<unsigned long long*>(counters + 16)[k] = (unsigned long long)prof[k];
#endif
    }
}

// the modified blocks of strip s are listed in mod_list[s * per_job ...) (strip_counts[s] of them)
__global__ void __launch_bounds__(64, 2) k_rdo_finish(uint4* blocks, const uint4* __restrict__ px, uint32_t n, uint32_t per_job, enc_cfg e, rdo_params p,
                                                   const uint32_t* __restrict__ mod_list, const uint32_t* __restrict__ strip_counts,
                                                   const uint8_t* __restrict__ state, uint32_t* counters) {
    const uint32_t strip = blockIdx.y, k = blockIdx.x * 64 + threadIdx.x;  // grid.x covers the longest list
    if (k >= strip_counts[strip]) return;
    const uint32_t b = mod_list[(size_t)strip * per_job + k];
    if (b >= n) return;
    alignas(16) uint8_t blk[16];
    *reinterpret_cast<uint4*>(blk) = blocks[b];
    alignas(16) rgba8 t[16];
    load_tile(px, b, t);
    if (state[b] == 1) {
        bool refined;
        rdo_refit_block(t, p, blk, refined);
        if (refined) atomicAdd(&counters[2], 1u);
    }
    __shared__ double s_hint_err[32 * 64];       // repeats of the ETC1 bias list (uastc_core.h, hint_cache): one LDS column per lane
    __shared__ unsigned char s_hint_table[32 * 64];
    const hint_cache hc = { s_hint_err + threadIdx.x, s_hint_table + threadIdx.x, 64 };
    rdo_rehint(t, e, blk, hc);
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
        const hipError_t e = hipMemsetAsync(w.counters, 0, w.zero_bytes, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_rdo_prepare, dim3(gx), dim3(64), 0, st, blocks, px, n, p, w.info, w.table, w.counters, w.strip_flags, per_job);
        break;
    }
    case 1:     // the lean walk: strips without a sensitive block
    case 3: {   // the walk with the refit in it: the flagged strips (the caller runs it on a second stream beside phase 1)
        const uint32_t window = p.lz_dict_size / 16 > 1 ? p.lz_dict_size / 16 : 1;
        uint32_t ring = 1;
        while (ring < window) ring <<= 1;
        if (ring > RDO_RING_MAX) ring = 0;
        const bool fat = phase == 3;
        if (ring && fat)
            hipLaunchKernelGGL((k_rdo_strips<true, true>), dim3(n_strips), dim3(RDO_THREADS), (size_t)ring * 16, st, blocks, px, n, per_job, p, w.info, w.table, w.hist,
                               cap, w.state, w.mod_list, w.strip_counts, w.counters, ring, w.strip_flags);
        else if (ring)
            hipLaunchKernelGGL((k_rdo_strips<true, false>), dim3(n_strips), dim3(RDO_THREADS), (size_t)ring * 16, st, blocks, px, n, per_job, p, w.info, w.table, w.hist,
                               cap, w.state, w.mod_list, w.strip_counts, w.counters, ring, w.strip_flags);
        else if (fat)
            hipLaunchKernelGGL((k_rdo_strips<false, true>), dim3(n_strips), dim3(RDO_THREADS), 0, st, blocks, px, n, per_job, p, w.info, w.table, w.hist, cap, w.state,
                               w.mod_list, w.strip_counts, w.counters, 1u, w.strip_flags);
        else
            hipLaunchKernelGGL((k_rdo_strips<false, false>), dim3(n_strips), dim3(RDO_THREADS), 0, st, blocks, px, n, per_job, p, w.info, w.table, w.hist, cap, w.state,
                               w.mod_list, w.strip_counts, w.counters, 1u, w.strip_flags);
        break;
    }
    default: {
        // phase 2 takes the longest per-strip list length in `total_jobs`' place holder: see launch_uastc_rdo_finish
        return hipErrorInvalidValue;
    }
    }
    return hipGetLastError();
}

hipError_t launch_uastc_rdo_finish(hipStream_t st, void* d_blocks, const void* d_px, uint32_t n, const float* fparams, const uint32_t* uparams, uint32_t flags,
                                   uint32_t total_jobs, void* d_ws, uint32_t longest_list) {
    if (!n || !longest_list) return hipSuccess;
    const rdo_workspace w = carve(d_ws, n, total_jobs, nullptr);
    uint32_t per_job, n_strips, cap;
    strip_layout(n, total_jobs, per_job, n_strips, cap);
    enc_cfg e;
    make_cfg(flags, e);
    hipLaunchKernelGGL(k_rdo_finish, dim3((longest_list + 63) / 64, n_strips), dim3(64), 0, st, static_cast<uint4*>(d_blocks), static_cast<const uint4*>(d_px), n,
                       per_job, e, to_params(fparams, uparams), w.mod_list, w.strip_counts, w.state, w.counters);
    return hipGetLastError();
}

const void* uastc_rdo_counters(void* d_ws, uint32_t n, uint32_t total_jobs) { return carve(d_ws, n, total_jobs, nullptr).counters; }
const void* uastc_rdo_strip_counts(void* d_ws, uint32_t n, uint32_t total_jobs) { return carve(d_ws, n, total_jobs, nullptr).strip_counts; }

} // namespace bu
