// uastc_kernels.hip -- UASTC LDR 4x4 block encoding on gfx950 (boundary 8b "encode_uastc batch op").
//
// The per-block algorithm is uastc_core.h; this file is the GPU decomposition. The reference encodes a block in one long serial
// call (19 mode generators, then a choice, then three hint searches). Here the same work is cut along its natural seams so that
// every wave runs ONE code path over 64 different blocks (no intra-wave mode divergence) and the chip sees
// n_blocks x n_jobs independent work items instead of n_blocks:
//   phase 0  classify      1 thread / block : solid / alpha / luminance-alpha class byte; solid blocks are packed right here
//   phase 1  candidates    1 thread / (block, job), grid.y = job : one mode (or one pattern / rotation of it) -> 64 B slots
//   phase 2  score         1 thread / (block, slot), grid.y = slot : UASTC decode + BC7 round trip -> {overall error, rms}
//   phase 3  finish        1 thread / block : first-wins choice over the slots, BC1 / EAC / ETC1 hints, bit packing. The blocks are taken in an ORDER
//                          that groups them by the head room of their colours (k_uastc_order_*: a counting sort on a class byte from phase 0), so that the 64
//                          blocks of a wave agree on which form of the ETC1 hint search applies (uastc_core.h, etc1_fit_subblock); results do not depend on it
// Slots are stored [slot][block] so a wave's 64 records are contiguous (coalesced 64 B per lane), pixels are read as 4 x 16 B per
// lane. Between phases everything stays in HBM: at level 2 that is 27 x 64 B per block, i.e. ~1.8 GB for a 4096^2 image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "uastc_core.h"
#include "uastc_kernels.h"

namespace bu {
using namespace bu_uastc;

namespace {

struct uastc_job { uint8_t mode, first_variant, n_variants, pad; uint32_t slot; };
struct uastc_plan {
    enc_cfg e;
    uint32_t n_slots, n_jobs;
    uint8_t slot_mode[MAX_SLOTS];
    uastc_job jobs[MAX_SLOTS];
};

void build_plan(uint32_t flags, uastc_plan& p) {
    std::memset(&p, 0, sizeof(p));
    make_cfg(flags, p.e);
    p.n_slots = total_slots(p.e);
    for (uint32_t i = 0; i < 18; i++) {
        const uint32_t m = mode_order(i), nv = mode_variants(m, p.e);
        if (!nv) continue;
        const uint32_t base = slot_base(m, p.e);
        for (uint32_t v = 0; v < nv; v++) p.slot_mode[base + v] = (uint8_t)m;
        // modes whose variants share an estimated pattern list stay one job; everything else is one job per variant
        const bool shared = p.e.estimate_partition && (m == 9 || m == 16);
        if (shared) p.jobs[p.n_jobs++] = { (uint8_t)m, 0, (uint8_t)nv, 0, base };
        else for (uint32_t v = 0; v < nv; v++) p.jobs[p.n_jobs++] = { (uint8_t)m, (uint8_t)v, 1, 0, base + v };
    }
}

constexpr uint32_t ORDER_KEYS = 2064;   // etc1_order_key's 11 bits + one class for the solid blocks, rounded up
constexpr uint32_t ORDER_SOLID = 2048;
struct workspace {
    uastc_plan* plan; uint8_t* cls; cand* cands; uint64_t* overall; float* rms;
    uint16_t* order_key; uint32_t* order; uint32_t* order_counts;   // per block: class, position -> block; [ORDER_KEYS] histogram + [ORDER_KEYS] cursors
};
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
workspace carve(void* base, uint32_t n, uint32_t n_slots, size_t* total) {
    char* p = static_cast<char*>(base);
    size_t o = 0;
    workspace w;
    w.plan = reinterpret_cast<uastc_plan*>(p + o); o += align_up(sizeof(uastc_plan));
    w.cls = reinterpret_cast<uint8_t*>(p + o); o += align_up(n);
    w.cands = reinterpret_cast<cand*>(p + o); o += align_up((size_t)n * n_slots * sizeof(cand));
    w.overall = reinterpret_cast<uint64_t*>(p + o); o += align_up((size_t)n * n_slots * 8);
    w.rms = reinterpret_cast<float*>(p + o); o += align_up((size_t)n * n_slots * 4);
    w.order_key = reinterpret_cast<uint16_t*>(p + o); o += align_up((size_t)n * 2);
    w.order = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    w.order_counts = reinterpret_cast<uint32_t*>(p + o); o += align_up(2 * ORDER_KEYS * 4);
    if (total) *total = o;
    return w;
}

__device__ inline void load_tile(const uint4* px, uint32_t b, rgba8* out) {
    uint4* o = reinterpret_cast<uint4*>(out);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = px[(size_t)b * 4 + k];
}

// What the ETC1 hint search of the finish kernel will do with a block, estimated from its source texels -- per half (left / right columns, top / bottom rows: the
// sub-blocks of the two ETC1 layouts): how many intensity tables it will try (2, 4 or 8, by the spread of the half around its mean: uastc_core.h etc1_trials) and how
// many of them stay unclamped around that mean (the search moves the mean by a bias of a quantisation step or two: hence the margin). A wave of the finish kernel
// runs as long as its most demanding lane, and takes the short form of a table only when no lane's colours clamp: the key groups blocks that agree on both.
// It only steers the ORDER the finish kernel takes the blocks in; no result depends on it.
__device__ inline uint32_t etc1_order_key(const rgba8* t) {
    int room_min = 255, room_max = 0, spread_min = 255, spread_max = 0;
    for (int half = 0; half < 4; half++) {
        int sum[3] = { 0, 0, 0 }, mn[3] = { 255, 255, 255 }, mx[3] = { 0, 0, 0 };
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = half < 2 ? ((j & 3) * 4 + half * 2 + (j >> 2)) : ((half - 2) * 8 + j);
            for (int c = 0; c < 3; c++) { const int v = t[i].c[c]; sum[c] += v; mn[c] = v < mn[c] ? v : mn[c]; mx[c] = v > mx[c] ? v : mx[c]; }
        }
        int room = 255, spread = 0;
        for (int c = 0; c < 3; c++) {
            const int m = sum[c] >> 3, r = m < 255 - m ? m : 255 - m, up = mx[c] - m, down = m - mn[c];
            room = r < room ? r : room;
            spread = up > spread ? up : spread;
            spread = down > spread ? down : spread;
        }
        room_min = room < room_min ? room : room_min; room_max = room > room_max ? room : room_max;
        spread_min = spread < spread_min ? spread : spread_min; spread_max = spread > spread_max ? spread : spread_max;
    }
    uint32_t lo = 0, hi = 0;
    for (uint32_t k = 0; k < 8; k++) { lo += etc1_inten_large(k) + 12 <= room_min ? 1u : 0u; hi += etc1_inten_large(k) + 12 <= room_max ? 1u : 0u; }
    const uint32_t tables_max = spread_max > 51 ? 2u : (spread_max >= 7 ? 1u : 0u), tables_min = spread_min > 51 ? 2u : (spread_min >= 7 ? 1u : 0u);
    return (tables_max << 9) | (lo << 5) | (tables_min << 3) | (hi < 7 ? hi : 7u);   // < 2048
}

__global__ void __launch_bounds__(64) k_uastc_classify(const uint4* __restrict__ px, uint32_t n, const uastc_plan* __restrict__ plan,
                                                       uint8_t* __restrict__ cls, uint16_t* __restrict__ order_key, uint4* __restrict__ out) {
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n) return;
    alignas(16) rgba8 t[16];
    load_tile(px, b, t);
    const uint32_t c = classify(t, plan->e);
    cls[b] = (uint8_t)c;
    order_key[b] = (uint16_t)((c & CLS_SOLID) ? ORDER_SOLID : etc1_order_key(t));
    if (c & CLS_SOLID) {
        alignas(16) uint8_t o[16];
        pack_solid(t[0].c, o);
        out[b] = *reinterpret_cast<const uint4*>(o);
    }
}

// Counting sort of the blocks by class byte: histogram, exclusive scan (one workgroup), scatter. The order inside a class is whatever the atomics give;
// nothing downstream depends on it (the finish kernel writes block b's result to out[b] wherever b sits in the order).
__global__ void __launch_bounds__(256) k_uastc_order_hist(const uint16_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_h[ORDER_KEYS];
    for (uint32_t k = threadIdx.x; k < ORDER_KEYS; k += 256) s_h[k] = 0;
    __syncthreads();
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < n; b += gridDim.x * 256) atomicAdd(&s_h[key[b]], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < ORDER_KEYS; k += 256) if (s_h[k]) atomicAdd(&counts[k], s_h[k]);
}
__global__ void __launch_bounds__(64) k_uastc_order_scan(uint32_t* __restrict__ counts) {   // counts[0..K) -> cursors[K..2K) = exclusive prefix
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < ORDER_KEYS; k++) { counts[ORDER_KEYS + k] = run; run += counts[k]; }
    }
}
__global__ void __launch_bounds__(256) k_uastc_order_scatter(const uint16_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ counts, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_h[ORDER_KEYS], s_base[ORDER_KEYS];
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x, first = blockIdx.x * per, last = first + per < n ? first + per : n;
    for (uint32_t k = threadIdx.x; k < ORDER_KEYS; k += 256) s_h[k] = 0;
    __syncthreads();
    for (uint32_t b = first + threadIdx.x; b < last; b += 256) atomicAdd(&s_h[key[b]], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < ORDER_KEYS; k += 256) { s_base[k] = s_h[k] ? atomicAdd(&counts[ORDER_KEYS + k], s_h[k]) : 0u; s_h[k] = 0; }
    __syncthreads();
    for (uint32_t b = first + threadIdx.x; b < last; b += 256) { const uint32_t k = key[b]; order[s_base[k] + atomicAdd(&s_h[k], 1u)] = b; }
}

// SCORE: the candidate is scored while it is still in registers (phase 2 folded in): the slots are written once and never read back by a scoring pass, which at level 2 was
// 27 x 64 B per block fetched again plus a second read of the tiles.
template <bool SCORE>
__global__ void __launch_bounds__(64, 2) k_uastc_candidates(const uint4* __restrict__ px, uint32_t n, const uastc_plan* __restrict__ plan,
                                                         const uint8_t* __restrict__ cls, cand* __restrict__ cands, uint32_t first_job,
                                                         uint64_t* __restrict__ overall, float* __restrict__ rms) {
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    const uastc_job job = plan->jobs[first_job + blockIdx.y];
    // the least-squares rows of this job's weight set, staged in LDS (the fit reads one 16-byte row per texel and pass, by a per-lane index)
    __shared__ float s_ls[32 * 4];
    {
        const uint32_t wbits = ku_mode_weight_bits[job.mode], rows = 1u << wbits;
        for (uint32_t i = threadIdx.x; i < rows * 4; i += 64) s_ls[i] = ku_weights_ls[(rows - 2u) * 4 + i];
    }
    __syncthreads();
    if (b >= n) return;
    const uint32_t c = cls[b];
    if ((c & CLS_SOLID) || !mode_applies(job.mode, c, plan->e)) return;
    alignas(16) rgba8 t[16];
    load_tile(px, b, t);
    cand local[4];
    run_mode(job.mode, t, plan->e, local, job.first_variant, job.n_variants, s_ls);
    for (uint32_t v = 0; v < job.n_variants; v++) {
        const uint4* s = reinterpret_cast<const uint4*>(&local[v]);
        uint4* d = reinterpret_cast<uint4*>(&cands[(size_t)(job.slot + v) * n + b]);
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = s[k];
        if (SCORE) {
            const cand_score sc = score_candidate(local[v], t, c, plan->e);
            overall[(size_t)(job.slot + v) * n + b] = sc.overall;
            rms[(size_t)(job.slot + v) * n + b] = sc.uastc_rms;
        }
    }
}

__global__ void __launch_bounds__(64) k_uastc_score(const uint4* __restrict__ px, uint32_t n, const uastc_plan* __restrict__ plan,
                                                    const uint8_t* __restrict__ cls, const cand* __restrict__ cands,
                                                    uint64_t* __restrict__ overall, float* __restrict__ rms) {
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n) return;
    const uint32_t slot = blockIdx.y, c = cls[b];
    if ((c & CLS_SOLID) || !mode_applies(plan->slot_mode[slot], c, plan->e)) return;
    alignas(16) rgba8 t[16];
    load_tile(px, b, t);
    cand r;
    {
        const uint4* s = reinterpret_cast<const uint4*>(&cands[(size_t)slot * n + b]);
        uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = s[k];
    }
    const cand_score sc = score_candidate(r, t, c, plan->e);
    overall[(size_t)slot * n + b] = sc.overall;
    rms[(size_t)slot * n + b] = sc.uastc_rms;
}

struct slot_view {
    const uastc_plan* plan; const uint64_t* overall_; const float* rms_; uint32_t n, b, cls;
    __device__ inline bool valid(uint32_t i) const { return mode_applies(plan->slot_mode[i], cls, plan->e); }
    __device__ inline uint64_t overall(uint32_t i) const { return overall_[(size_t)i * n + b]; }
    __device__ inline float rms(uint32_t i) const { return rms_[(size_t)i * n + b]; }
    __device__ inline uint32_t mode(uint32_t i) const { return plan->slot_mode[i]; }
};

__global__ void __launch_bounds__(64, 2) k_uastc_finish(const uint4* __restrict__ px, uint32_t n, const uastc_plan* __restrict__ plan,
                                                     const uint8_t* __restrict__ cls, const cand* __restrict__ cands,
                                                     const uint64_t* __restrict__ overall, const float* __restrict__ rms, const uint32_t* __restrict__ order,
                                                     uint4* __restrict__ out) {
    const uint32_t at = blockIdx.x * 64 + threadIdx.x;
    if (at >= n) return;
    const uint32_t b = order[at];
    const uint32_t c = cls[b];
    if (c & CLS_SOLID) return;
    const slot_view v = { plan, overall, rms, n, b, c };
    const uint32_t pick = choose_candidate(v, plan->n_slots, plan->e);
    alignas(16) rgba8 t[16];
    load_tile(px, b, t);
    cand r;
    {
        const uint4* s = reinterpret_cast<const uint4*>(&cands[(size_t)pick * n + b]);
        uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = s[k];
    }
    // repeats of the ETC1 bias list (uastc_core.h, hint_cache): one LDS column per lane
    __shared__ double s_hint_err[32 * 64];
    __shared__ unsigned char s_hint_table[32 * 64];
    const hint_cache hc = { s_hint_err + threadIdx.x, s_hint_table + threadIdx.x, 64 };
    alignas(16) uint8_t o[16];
    finish_block(t, plan->e, r, o, hc);
    out[b] = *reinterpret_cast<const uint4*>(o);
}

} // namespace

size_t uastc_workspace_bytes(uint32_t n_blocks, uint32_t flags) {
    uastc_plan p;
    build_plan(flags, p);
    size_t total = 0;
    carve(nullptr, n_blocks, p.n_slots, &total);
    return total;
}

static bool fused_score() { static const bool on = [] { const char* e = std::getenv("BU_UASTC_FUSED_SCORE"); return !e || e[0] != '0'; }(); return on; }

hipError_t launch_uastc_phase(hipStream_t st, int phase, const void* d_px, uint32_t n, uint32_t flags, void* d_ws, void* d_out) {
    if (!n) return hipSuccess;
    uastc_plan p;
    build_plan(flags, p);
    const workspace w = carve(d_ws, n, p.n_slots, nullptr);
    const uint4* px = static_cast<const uint4*>(d_px);
    const uint32_t gx = (n + 63) / 64;
    hipError_t e = hipSuccess;
    switch (phase) {
    case 0:
        // the plan is tiny and identical for every call with the same flags; it rides in front of the first kernel
        if ((e = hipMemcpyAsync(w.plan, &p, sizeof(p), hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(w.order_counts, 0, 2 * ORDER_KEYS * 4, st)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_uastc_classify, dim3(gx), dim3(64), 0, st, px, n, w.plan, w.cls, w.order_key, static_cast<uint4*>(d_out));
        {
            const uint32_t wg = (n + 4095) / 4096 < 1024 ? (n + 4095) / 4096 : 1024;
            hipLaunchKernelGGL(k_uastc_order_hist, dim3(wg), dim3(256), 0, st, w.order_key, n, w.order_counts);
            hipLaunchKernelGGL(k_uastc_order_scan, dim3(1), dim3(64), 0, st, w.order_counts);
            hipLaunchKernelGGL(k_uastc_order_scatter, dim3(wg), dim3(256), 0, st, w.order_key, n, w.order_counts, w.order);
        }
        break;
    case 1:
        if (std::getenv("BU_UASTC_JOB_TIMES")) {  // developer aid: one launch per job, timed with events, printed to stderr
            for (uint32_t j = 0; j < p.n_jobs; j++) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a); (void)hipEventCreate(&b);
                (void)hipEventRecord(a, st);
                hipLaunchKernelGGL(k_uastc_candidates<false>, dim3(gx, 1), dim3(64), 0, st, px, n, w.plan, w.cls, w.cands, j, w.overall, w.rms);
                (void)hipEventRecord(b, st);
                (void)hipEventSynchronize(b);
                float ms = 0;
                (void)hipEventElapsedTime(&ms, a, b);
                std::fprintf(stderr, "[uastc job %2u] mode %2u variant %u x%u: %.3f ms\n", j, p.jobs[j].mode, p.jobs[j].first_variant, p.jobs[j].n_variants, ms);
                (void)hipEventDestroy(a); (void)hipEventDestroy(b);
            }
            break;
        }
        if (fused_score()) hipLaunchKernelGGL(k_uastc_candidates<true>, dim3(gx, p.n_jobs), dim3(64), 0, st, px, n, w.plan, w.cls, w.cands, 0u, w.overall, w.rms);
        else hipLaunchKernelGGL(k_uastc_candidates<false>, dim3(gx, p.n_jobs), dim3(64), 0, st, px, n, w.plan, w.cls, w.cands, 0u, w.overall, w.rms);
        break;
    case 2:
        if (!fused_score() || std::getenv("BU_UASTC_JOB_TIMES"))
            hipLaunchKernelGGL(k_uastc_score, dim3(gx, p.n_slots), dim3(64), 0, st, px, n, w.plan, w.cls, w.cands, w.overall, w.rms);
        break;
    default:
        hipLaunchKernelGGL(k_uastc_finish, dim3(gx), dim3(64), 0, st, px, n, w.plan, w.cls, w.cands, w.overall, w.rms, w.order, static_cast<uint4*>(d_out));
        break;
    }
    return hipGetLastError();
}

hipError_t launch_encode_uastc(hipStream_t st, const void* d_px, uint32_t n, uint32_t flags, void* d_ws, void* d_out) {
    for (int phase = 0; phase < 4; phase++) {
        const hipError_t e = launch_uastc_phase(st, phase, d_px, n, flags, d_ws, d_out);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace bu
