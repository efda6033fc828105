// tsvq_kernels.hip -- device side of the codebook builder (row a8): batched TSVQ node splits, bit-exact with the reference's
// tree_vector_quant<>::split_node (encoder/basisu_enc.h:1737-2077).
//
// The reference's split is order dependent: every centroid / covariance entry is a RUNNING float (or double) sum over the
// node's members in list order, so the sums cannot be re-associated (SURVEY hazard H2). What CAN run in parallel is
//   (1) different accumulators of one pass (16..136 independent chains),  (2) the per-member work that feeds the chains
//   (projection, two double-precision centroid distances, products),      (3) different nodes of the tree.
// One 512-thread workgroup owns one node and runs the whole split inside a single launch. Each pass over the members is a
// producer/consumer pipeline through double-buffered LDS tiles: the producer waves gather the member rows K tiles ahead
// (register queue, so the HBM/L2 latency of the dependent index -> row gather is off the critical path), do the per-member
// arithmetic and lay out, per chain, the exact addend of every member; the consumer lanes (one per chain) add them in member
// order with 16-byte LDS reads. A pass therefore costs about one dependent v_add_f32 per member, which is the floor for an
// order-preserving sum; everything else hides behind that chain.
// Selector training vectors (16 values in 0..3) are kept packed in one dword per vector, so a node's working set is
// 12 bytes per member and stays in L2 across the ~10 passes of a split.
// Member lists of every node are ascending index lists (children are stable partitions of the parent).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include <cstdlib>
#include "tsvq_kernels.h"
#include "tsvq_common.h"
#include "tsvq_bufs.h"

namespace bu {

// Development aid (tools/build_tsvq_profile.sh, -DTQ_PROFILE): clock64() ticks thread 0 of every split workgroup spends between the TQ_TICK marks, summed per
// interval over all workgroups + the number of workgroups and of side passes; read and reset by tsvq_profile_read(). Never compiled into the product library.
#ifdef TQ_PROFILE
__device__ unsigned long long g_tq_prof[16];
#define TQ_TICK(k) do { if (threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&g_tq_prof[k], (unsigned long long)(now_ - tq_t_)); tq_t_ = now_; } } while (0)
#define TQ_COUNT(k) do { if (threadIdx.x == 0) atomicAdd(&g_tq_prof[k], 1ull); } while (0)
#else
#define TQ_TICK(k) do { } while (0)
#define TQ_COUNT(k) do { } while (0)
#endif

constexpr int TQ_THREADS = 512;
constexpr int TQ_TILE = 256;           // members per LDS tile
constexpr int TQ_STRIDE = TQ_TILE + 4; // float row stride: keeps 16-byte reads of different chains on different banks

struct tq_ctrl { // serial state of one split, written by thread 0 between passes
    float l_c[16], r_c[16];    // current centroids
    float axis[16];
    float cov[16][16];
    uint64_t l_w, r_w;
    uint32_t l_n, r_n;
    float l_var, r_var;
    float prev_total;
    int state;                 // 0 keep iterating, 1 converged
    int mode;
};

enum { TQ_MODE_DIST = 0, TQ_MODE_PEEL_FIRST = 1, TQ_MODE_HALF = 2, TQ_MODE_PROJ = 3 };


// ---- where a training vector comes from
template <int N>
struct float_rows {
    static constexpr bool PACKED = false;
    const float* rows;
    struct payload { float v[N]; uint64_t w; uint32_t mi; };
    __device__ __forceinline__ payload fetch(const uint64_t* __restrict__ w64, uint32_t mi) const {
        payload p; p.mi = mi; p.w = w64[mi];
        const float* r = rows + (size_t)mi * N;
#pragma unroll
        for (int k = 0; k < N; k++) p.v[k] = r[k];
        return p;
    }
    static __device__ __forceinline__ void decode(const payload& p, float (&v)[N]) {
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = p.v[k];
    }
};
struct packed16_rows { // 16 two-bit values, element 0 in the top two bits (the order the frontend's de-duplication keys use)
    static constexpr bool PACKED = true;
    const uint32_t* keys;
    struct payload { uint32_t key; uint64_t w; uint32_t mi; };
    __device__ __forceinline__ payload fetch(const uint64_t* __restrict__ w64, uint32_t mi) const {
        payload p; p.mi = mi; p.w = w64[mi]; p.key = keys[mi];
        return p;
    }
    static __device__ __forceinline__ void decode(const payload& p, float (&v)[16]) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = (float)((p.key >> (30 - 2 * k)) & 3u);
    }
};

// One pipelined pass over `count` members. FROWS float rows (stride TQ_STRIDE) + DROWS double rows (stride TQ_TILE) per LDS
// tile. Producer thread `pid` owns slot pid of every tile: emit(pos, payload, f + pid, d + pid) writes that member's column.
// The first CW waves are consumers: consume(f, d, m) is called once per tile, in tile order, with m valid members.
template <int FROWS, int DROWS, int CW, typename Src, typename Emit, typename Consume>
__device__ __forceinline__ void pipeline_pass(char* lds, const Src& src, const uint64_t* __restrict__ w64, const uint32_t* __restrict__ members,
                                              uint32_t count, Emit emit, Consume consume) {
    constexpr size_t F_BYTES = (size_t)FROWS * TQ_STRIDE * sizeof(float);
    constexpr size_t D_BYTES = (size_t)DROWS * TQ_TILE * sizeof(double);
    constexpr size_t BUF_BYTES = ((F_BYTES + D_BYTES + 15) / 16) * 16;
    static_assert(TQ_THREADS == 512 && TQ_TILE == 256 && CW >= 1 && CW <= 3, "wave roles below assume 8 waves, 4 of them producing");
    const int tid = threadIdx.x;
    // Producer waves are chosen so that they do not share a SIMD with a chain consumer where that is possible (waves are dealt to
    // the 4 SIMDs round robin): a dependent add chain issues one VALU op per ~4 cycles and every foreign op on its SIMD delays it.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and told so: the roles below are scalar branches
    const bool consumer = wave < CW;
    int pslot = -1;
    if (CW == 1) pslot = wave == 1 ? 0 : wave == 2 ? 1 : wave == 3 ? 2 : wave == 5 ? 3 : -1;        // SIMD 0 is the consumer's alone
    else if (CW == 2) pslot = wave == 2 ? 0 : wave == 3 ? 1 : wave == 6 ? 2 : wave == 7 ? 3 : -1;   // SIMD 0 / 1 reserved
    else pslot = wave == 3 ? 0 : wave == 7 ? 1 : wave == 4 ? 2 : wave == 5 ? 3 : -1;                // 136 covariance chains need three waves
    const bool producer = pslot >= 0;
    const uint32_t pid = (uint32_t)(pslot * 64 + (tid & 63));
    const uint32_t tiles = (count + TQ_TILE - 1) / TQ_TILE;
    auto fptr = [&](uint32_t t) { return reinterpret_cast<float*>(lds + (size_t)(t & 1u) * BUF_BYTES); };
    auto dptr = [&](uint32_t t) { return reinterpret_cast<double*>(lds + (size_t)(t & 1u) * BUF_BYTES + F_BYTES); };
    using payload = typename Src::payload;
    // Producer schedule. The gather is two dependent loads (member index, then that member's row and weight). Both stages are
    // issued one tile period ahead of their use, right after the top-of-iteration wait, so a conservative `s_waitcnt vmcnt(0)`
    // there (which is what the compiler emits around the predicated emit code) only ever waits for loads that have had a whole
    // tile period (about one HBM latency) to land. Loads are unconditional on a clamped position: a predicated load would be
    // sunk into the emit branch and waited for on the spot.
    const uint32_t last = count - 1;
    auto fetch_index = [&](uint32_t t) -> uint32_t {
        const uint32_t pos = min(t * TQ_TILE + pid, last);
        return members ? members[pos] : pos;
    };
    auto emit_tile = [&](uint32_t t, const payload& p) {
        const uint32_t pos = t * TQ_TILE + pid;
        if (pos < count) emit(pos, p, fptr(t) + pid, dptr(t) + pid);
    };
    if (producer) { // wave-uniform roles: each role runs its own loop, the barriers pair up one to one
        // (the run-ahead stops at the last tile: the tile count is uniform, so these are branches, not predicated loads -- a node of one or two tiles, which is
        //  most nodes of a tree's lower levels, used to wait for two rounds of loads nobody needed before it could leave the pass)
        uint32_t idx = fetch_index(0);
        payload pay = src.fetch(w64, idx);   // tile 0
        if (tiles > 1) idx = fetch_index(1);
        emit_tile(0, pay);
        if (tiles > 1) pay = src.fetch(w64, idx);           // tile 1
        if (tiles > 2) idx = fetch_index(2);                // index of tile 2
        __syncthreads();
        for (uint32_t t = 0; t < tiles; t++) {
            if (t + 1 < tiles) emit_tile(t + 1, pay);
            if (t + 2 < tiles) pay = src.fetch(w64, idx);   // tile t + 2
            if (t + 3 < tiles) idx = fetch_index(t + 3);
            __syncthreads();
        }
    } else if (consumer) {
        __syncthreads();
        for (uint32_t t = 0; t < tiles; t++) {
            consume(fptr(t), dptr(t), min((uint32_t)TQ_TILE, count - t * TQ_TILE));
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (uint32_t t = 0; t < tiles; t++) __syncthreads();
    }
}

// Order-preserving running sum of one LDS row: the adds happen strictly in member order (one dependent v_add per member is the
// floor of this algorithm), the 16-byte LDS reads of the NEXT group are issued before the current group is added so that the LDS
// latency stays off the dependent chain.
constexpr int TQ_AHEAD = 3; // LDS read groups in flight ahead of the adds (a ds_read_b128 takes ~2 groups of dependent adds to land)
__device__ __forceinline__ void chain_add_f32(float& acc, const float* src, uint32_t m) {
    if (m == (uint32_t)TQ_TILE) {
        constexpr int G = TQ_TILE / 16;
        const float4* p = reinterpret_cast<const float4*>(src);
        float4 r[TQ_AHEAD + 1][4];
#pragma unroll
        for (int g = 0; g < TQ_AHEAD; g++)
#pragma unroll
            for (int k = 0; k < 4; k++) r[g][k] = p[g * 4 + k];
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g + TQ_AHEAD < G)
#pragma unroll
                for (int k = 0; k < 4; k++) r[(g + TQ_AHEAD) % (TQ_AHEAD + 1)][k] = p[(g + TQ_AHEAD) * 4 + k];
            __builtin_amdgcn_sched_barrier(0); // keep the reads TQ_AHEAD groups ahead: the scheduler otherwise sinks them next to their use
#pragma unroll
            for (int k = 0; k < 4; k++) { const float4 c = r[g % (TQ_AHEAD + 1)][k]; acc += c.x; acc += c.y; acc += c.z; acc += c.w; }
        }
        return;
    }
    uint32_t j = 0;
    for (; j + 4 <= m; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(src + j);
        acc += a.x; acc += a.y; acc += a.z; acc += a.w;
    }
    for (; j < m; j++) acc += src[j];
}
__device__ __forceinline__ void chain_add_f64(double& acc, const double* src, uint32_t m) {
    if (m == (uint32_t)TQ_TILE) {
        constexpr int G = TQ_TILE / 8;
        const double2* p = reinterpret_cast<const double2*>(src);
        double2 r[TQ_AHEAD + 1][4];
#pragma unroll
        for (int g = 0; g < TQ_AHEAD; g++)
#pragma unroll
            for (int k = 0; k < 4; k++) r[g][k] = p[g * 4 + k];
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g + TQ_AHEAD < G)
#pragma unroll
                for (int k = 0; k < 4; k++) r[(g + TQ_AHEAD) % (TQ_AHEAD + 1)][k] = p[(g + TQ_AHEAD) * 4 + k];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; k++) { const double2 c = r[g % (TQ_AHEAD + 1)][k]; acc += c.x; acc += c.y; }
        }
        return;
    }
    uint32_t j = 0;
    for (; j + 2 <= m; j += 2) {
        const double2 a = *reinterpret_cast<const double2*>(src + j);
        acc += a.x; acc += a.y;
    }
    for (; j < m; j++) acc += src[j];
}
// covariance chain: adds dx[j] * wy[j] in member order (product rounded to float first, enc.h:1819)
__device__ __forceinline__ void chain_add_prod_f32(float& acc, const float* dx, const float* wy, uint32_t m) {
    if (m == (uint32_t)TQ_TILE) {
        constexpr int G = TQ_TILE / 8;
        const float4* pa = reinterpret_cast<const float4*>(dx);
        const float4* pb = reinterpret_cast<const float4*>(wy);
        float4 ra[TQ_AHEAD + 1][2], rb[TQ_AHEAD + 1][2];
#pragma unroll
        for (int g = 0; g < TQ_AHEAD; g++)
#pragma unroll
            for (int k = 0; k < 2; k++) { ra[g][k] = pa[g * 2 + k]; rb[g][k] = pb[g * 2 + k]; }
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (g + TQ_AHEAD < G)
#pragma unroll
                for (int k = 0; k < 2; k++) { ra[(g + TQ_AHEAD) % (TQ_AHEAD + 1)][k] = pa[(g + TQ_AHEAD) * 2 + k]; rb[(g + TQ_AHEAD) % (TQ_AHEAD + 1)][k] = pb[(g + TQ_AHEAD) * 2 + k]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const float4 a = ra[g % (TQ_AHEAD + 1)][k], b = rb[g % (TQ_AHEAD + 1)][k];
                const float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z, p3 = a.w * b.w;
                acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;
            }
        }
        return;
    }
    uint32_t j = 0;
    for (; j + 4 <= m; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(dx + j), b = *reinterpret_cast<const float4*>(wy + j);
        const float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z, p3 = a.w * b.w;
        acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;
    }
    for (; j < m; j++) { const float pp = dx[j] * wy[j]; acc = acc + pp; }
}



__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t* scratch /* TQ_THREADS/64 */) {
    v = wave_sum_u64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 63) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t s = 0;
    for (int w = 0; w < TQ_THREADS / 64; w++) s += scratch[w];
    return s;
}

// several u64 block sums behind one pair of barriers; scratch holds TQ_THREADS/64 * K entries
template <int K>
__device__ __forceinline__ void block_sum_u64xN(uint64_t (&v)[K], uint64_t* scratch) {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_sum_u64(v[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 63)
#pragma unroll
        for (int k = 0; k < K; k++) scratch[(threadIdx.x >> 6) * K + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        uint64_t t = 0;
        for (int w = 0; w < TQ_THREADS / 64; w++) t += scratch[w * K + k];
        v[k] = t;
    }
}


// End-of-round signal for a host that polls (bu_hip_tsvq_split): everything enqueued before this launch on the stream has completed, so the result records
// the split kernels wrote straight into page-locked host memory are in place; the flag is written with system scope behind them.
__global__ void k_tsvq_signal(uint32_t* flag, uint32_t value) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_tsvq_signal(hipStream_t st, uint32_t* d_flag, uint32_t value) {
    hipLaunchKernelGGL(k_tsvq_signal, dim3(1), dim3(64), 0, st, d_flag, value);
    return hipGetLastError();
}

// Small results into coherent page-locked host memory + (flag != nullptr) a sequence number behind them for the host to look at (bu_hip_api.cpp, mail_fetch): one workgroup;
// the data is fenced to system scope by every thread that wrote some of it before thread 0 releases the word.
__global__ __launch_bounds__(256) void k_mail_copy(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, uint32_t bytes, uint32_t* flag, uint32_t seq) {
    const uint32_t tid = threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
        const uint32_t n16 = bytes >> 4;
        for (uint32_t i = tid; i < n16; i += 256) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (uint32_t i = (n16 << 4) + tid; i < bytes; i += 256) dst[i] = src[i];
    } else if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3u) == 0) {
        const uint32_t n4 = bytes >> 2;
        for (uint32_t i = tid; i < n4; i += 256) reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
        for (uint32_t i = (n4 << 2) + tid; i < bytes; i += 256) dst[i] = src[i];
    } else {
        for (uint32_t i = tid; i < bytes; i += 256) dst[i] = src[i];
    }
    if (!flag) return;
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_mail_copy(hipStream_t st, void* dst, const void* src, size_t bytes, uint32_t* flag, uint32_t seq) {
    hipLaunchKernelGGL(k_mail_copy, dim3(1), dim3(256), 0, st, static_cast<unsigned char*>(dst), static_cast<const unsigned char*>(src), (uint32_t)bytes, flag, seq);
    return hipGetLastError();
}

// What a many-workgroup round starts with, as ONE launch instead of a host -> device copy and a fill (two runtime commands, each with its own ~10-20 us of
// latency in front of the round's first kernel): workgroup i brings node record i over from the page-locked records the host has just written (device-visible
// host memory, read once) and clears the node's state.
__global__ __launch_bounds__(256) void k_tsvq_wide_prologue(const tsvq_wide_node* __restrict__ src, tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* __restrict__ ctrl) {
    static_assert(sizeof(tsvq_wide_node) % 4 == 0 && sizeof(tsvq_wide_ctrl) % 4 == 0, "copied and cleared by dwords");
    const uint32_t i = blockIdx.x;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src + i);
    uint32_t* d = reinterpret_cast<uint32_t*>(nodes + i);
    if (threadIdx.x < sizeof(tsvq_wide_node) / 4) d[threadIdx.x] = __builtin_nontemporal_load(s + threadIdx.x);
    uint32_t* c = reinterpret_cast<uint32_t*>(ctrl + i);
    for (uint32_t k = threadIdx.x; k < sizeof(tsvq_wide_ctrl) / 4; k += 256) c[k] = 0;
}
hipError_t launch_tsvq_wide_prologue(hipStream_t st, const tsvq_wide_node* src, tsvq_wide_node* d_nodes, tsvq_wide_ctrl* d_ctrl, uint32_t n_nodes) {
    static_assert(sizeof(tsvq_wide_node) / 4 <= 256, "one thread per dword of a node record");
    if (!n_nodes) return hipSuccess;
    hipLaunchKernelGGL(k_tsvq_wide_prologue, dim3(n_nodes), dim3(256), 0, st, src, d_nodes, d_ctrl);
    return hipGetLastError();
}

// Deep rounds (bu_hip_tsvq_split_deep): the node records of the NEXT generation, made on the device from the result records of the generation before it, so that the
// children's splits follow their parents' on the stream without the host in between. Child t = side (t & 1) of parent t >> 1. A child is attempted when its parent's
// split succeeded (ok == 1), it has more than one member, and its variance -- after the reference's substitution of 1e-4 for a non-positive variance of a node with
// differing members (enc.h:1766-1792) -- is positive and not below the floor the host put into the FIRST generation's records (`pad`: the bits of a float; children
// inherit it). The floor only bounds the speculation: a node below it cannot be popped from the caller's queue before the leaf budget is spent. What is not attempted
// gets count = 0 (the split kernel returns at once) and ok = 3 in its result record.
__global__ __launch_bounds__(256) void k_tsvq_children(const tsvq_node_in* __restrict__ parents, const tsvq_split_out* parent_outs, uint32_t n_parents,
                                                       tsvq_node_in* __restrict__ children, tsvq_split_out* child_outs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n_parents) return;
    const uint32_t j = t >> 1, s = t & 1u;
    const tsvq_node_in pn = parents[j];
    const tsvq_split_out* po = parent_outs + j;
    tsvq_node_in ch;
    ch.buf = tsvq_child_buf(pn.buf); ch.start = pn.start; ch.count = 0; ch.pad = pn.pad; ch.weight = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) ch.origin[k] = 0.0f;
    if (pn.count != 0 && po->ok == 1u) {
        const uint32_t cnt = s ? po->r_count : po->l_count;
        float var = s ? po->r_var : po->l_var;
        if (var <= 0.0f && cnt > 1) var = 1e-4f;
        if (cnt > 1 && var > 0.0f && var >= __uint_as_float(pn.pad)) {
            ch.start = pn.start + (s ? po->l_count : 0u); ch.count = cnt; ch.weight = s ? po->r_weight : po->l_weight;
#pragma unroll
            for (int k = 0; k < 16; k++) ch.origin[k] = s ? po->r_centroid[k] : po->l_centroid[k];
        }
    }
    children[t] = ch;
    if (ch.count == 0) child_outs[t].ok = 3u;
}
hipError_t launch_tsvq_children(hipStream_t st, const tsvq_node_in* d_parents, const tsvq_split_out* d_parent_outs, uint32_t n_parents, tsvq_node_in* d_children,
                                tsvq_split_out* d_child_outs) {
    if (!n_parents) return hipSuccess;
    hipLaunchKernelGGL(k_tsvq_children, dim3((2 * n_parents + 255) / 256), dim3(256), 0, st, d_parents, d_parent_outs, n_parents, d_children, d_child_outs);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_tsvq_iota(uint32_t n, uint32_t* __restrict__ perm0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm0[i] = i;
}

// prepare_root (enc.h:1708-1735): origin sums, weight, variance of the whole training set -- or, with `nodes`, of one member span per
// workgroup (the roots of the T independent trees of the partitioned build, enc.h:2137-2152: the sub-quantizer's training set is the
// leaf's member list in list order). EX: see exact_acc.
template <int N, typename Src, bool EX>
__global__ __launch_bounds__(TQ_THREADS) void k_tsvq_root(Src src, const uint64_t* __restrict__ w64, uint32_t n, tsvq_root_out* __restrict__ out,
                                                         const uint32_t* __restrict__ perm0, const uint32_t* __restrict__ perm1, const tsvq_node_in* __restrict__ nodes) {
    extern __shared__ __align__(16) char lds[];
    const uint32_t* members = nullptr;
    if (nodes) {
        const tsvq_node_in& nd = nodes[blockIdx.x];
        members = tsvq_list(perm0, perm1, nd.buf) + nd.start;
        n = nd.count;
        out += blockIdx.x;
    }
    __shared__ float s_origin[16];
    __shared__ double s_tt;
    __shared__ uint64_t s_red[TQ_THREADS / 64 * 3];
    const int tid = threadIdx.x;
    float acc_f = 0.0f; double acc_d = 0.0;
    uint64_t wsum = 0;
    exact_acc tt;
    bool bad = false;
    pipeline_pass<N, EX ? 0 : 1, EX ? 1 : 2>(lds, src, w64, members, n, // wave 0: the N float chains, wave 1: the double chain
        [&](uint32_t, const typename Src::payload& p, float* f, double* d) {
            float v[N]; Src::decode(p, v);
            const float w = (float)p.w;
#pragma unroll
            for (int k = 0; k < N; k++) f[(size_t)k * TQ_STRIDE] = v[k] * w;
            const float t = dot_seq<N>(v, v) * w;
            if (EX) bad |= !tt.add(t);
            else d[0] = (double)t;
            wsum += p.w;
        },
        [&](const float* f, const double* d, uint32_t m) {
            if (tid < N) chain_add_f32(acc_f, f + (size_t)tid * TQ_STRIDE, m);
            else if (!EX && tid == 64) chain_add_f64(acc_d, d, m);
        });
    uint64_t sums[3] = {wsum, tt.lo, tt.hi};
    block_sum_u64xN<3>(sums, s_red);
    wsum = sums[0];
    const bool any_bad = EX && __syncthreads_or(bad ? 1 : 0) != 0;
    if (tid < N) s_origin[tid] = acc_f;
    if (!EX && tid == 64) s_tt = acc_d;
    __syncthreads();
    if (tid == 0) {
        double ttsum = EX ? 0.0 : s_tt;
        const bool ok = !EX || (!any_bad && exact_total(sums[1], sums[2], &ttsum));
        float o[N];
        for (int k = 0; k < N; k++) o[k] = s_origin[k];
        const float wfl = (float)wsum;
        const float q = dot_seq<N>(o, o) / wfl;
        out->var = (float)(ttsum - (double)q);
        const float inv = 1.0f / wfl;
        for (int k = 0; k < N; k++) out->origin[k] = o[k] * inv;
        for (int k = N; k < 16; k++) out->origin[k] = 0.0f;
        out->weight = wsum;
        out->pad = ok ? 0u : 1u; // 1: outside the exact range, re-run with the chained variant
    }
}

// split_node (enc.h:1737-1800) = prep_split (:1848-1960) + refine_split (:1962-2077) for one node per workgroup.
template <int N, typename Src, bool EX>
__device__ __forceinline__ void tsvq_split_body(Src src, const uint64_t* __restrict__ w64, uint32_t* __restrict__ perm0, uint32_t* __restrict__ perm1,
                                                uint8_t* __restrict__ side, const tsvq_node_in* __restrict__ nodes, tsvq_split_out* __restrict__ outs) {
    extern __shared__ __align__(16) char lds[];
    __shared__ tq_ctrl c;
    __shared__ float s_origin[16];
    __shared__ float s_sum[2][16];
    __shared__ double s_dsum[2];
    __shared__ uint64_t s_red[TQ_THREADS / 64 * 7];
    __shared__ uint32_t s_scan[TQ_THREADS / 64][2];
    __shared__ uint32_t s_base[2];
    __shared__ double2 s_tab[16][4]; // packed rows: {(l_c[k] - val)^2, (r_c[k] - val)^2} for val = 0..3 (see TQ_MODE_DIST)
    __shared__ uint32_t s_tot[16];   // packed rows, exact side passes: sum over the node's members of value x weight, per component
    using payload = typename Src::payload;

    const int tid = threadIdx.x;
#ifdef TQ_PROFILE
    long long tq_t_ = clock64();
    TQ_COUNT(15);
#endif
    const tsvq_node_in nd = nodes[blockIdx.x];
    tsvq_split_out* out = outs + blockIdx.x;
    const uint32_t count = nd.count;
    if (count == 0) return;   // a child record k_tsvq_children decided not to attempt (its result record already says ok = 3); uniform, before any barrier
    const uint32_t* members = tsvq_list(perm0, perm1, nd.buf) + nd.start;
    uint32_t* child_members = tsvq_child_list(perm0, perm1, nd.buf) + nd.start;
    uint8_t* node_side = side + nd.start;
    if (tid < 16) s_origin[tid] = nodes[blockIdx.x].origin[tid];   // from memory: indexing the register copy by tid would put it in scratch
    __syncthreads();

    // One classification + accumulation pass. `mode` selects how a member picks its side; float chains 0..N-1 are the left sums,
    // N..2N-1 the right sums; the two double chains are the left/right "ttsum" (in the projection / half passes: the weights).
    // Returns true (uniformly) when the exact variant met data outside its range: the caller gives the node up with ok = 2.
    // The same pass without any chain, for packed rows: the float addends are non-negative INTEGER-valued (value 0..3 times an integer weight),
    // so when a chain's total is below 2^24 every partial sum of the sequential chain is an integer below 2^24, i.e. exact, and the chain's
    // result is the plain integer total -- computed here by all 512 threads in any order. Returns 0 when every chain was exact (results
    // stored like side_pass stores them), 1 when some total reached 2^24 (nothing stored: the caller runs the chained pass), 2 when the
    // integer totals of the double accumulators left their exact range (the node is given up with ok = 2, as in side_pass).
    bool have_totals = false;   // s_tot holds the node's per-component totals (they do not depend on the classification)
    auto side_pass_exact = [&](int mode, bool write_side) -> int {
        if constexpr (!(Src::PACKED && EX && N == 16)) { return 1; } else {
        // scratch in the (otherwise idle) tile memory: per-wave partial sums, then the block's verdict
        uint32_t* s_part = reinterpret_cast<uint32_t*>(lds);                     // [8 waves][32]: right sums 0..15, totals 16..31 (first pass only)
        uint64_t* s_p64 = reinterpret_cast<uint64_t*>(lds + 1024);               // [8 waves][8]: lw, rw, ln, ex0.lo, ex0.hi, ex1.lo, ex1.hi, flags (bit 0 big, bit 1 bad)
        uint32_t* s_status = reinterpret_cast<uint32_t*>(lds + 1024 + 512);
        const bool first = !have_totals;
        uint32_t acc_r[16], acc_t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { acc_r[i] = 0; acc_t[i] = 0; }
        uint64_t lw = 0, rw = 0; uint32_t ln = 0;
        exact_acc ex[2];
        // per-thread u32 partial sums stay below 2^32 for up to 64 members of < 2^24 each; a member weighing 2^22 or more leaves the pass to the chained
        // form (value x weight is then not known to stay below 2^24, where the float product and the integer one are the same number)
        bool bad = false, big = count > 512u * 64u;
        if (mode == TQ_MODE_DIST && tid < 64) {
            const int k = tid >> 2, val = tid & 3;
            const double a = (double)c.l_c[k] - (double)(float)val, b = (double)c.r_c[k] - (double)(float)val;
            s_tab[k][val] = make_double2(a * a, b * b);
        }
        __syncthreads();
        TQ_TICK(7);
        for (uint32_t pos = (uint32_t)tid; pos < count; pos += TQ_THREADS) {
            const payload p = src.fetch(w64, members[pos]);
            const float w = (float)p.w;
            bool right;
            if (mode == TQ_MODE_DIST) {
                double dl = 0, dr = 0;
#pragma unroll
                for (int k = 0; k < N; k++) { const double2 t = s_tab[k][(p.key >> (30 - 2 * k)) & 3u]; dl += t.x; dr += t.y; }
                right = dl >= dr;
            } else {
                float dd[N];
#pragma unroll
                for (int k = 0; k < N; k++) dd[k] = (float)((p.key >> (30 - 2 * k)) & 3u) - s_origin[k];
                right = (double)dot_seq<N>(dd, c.axis) >= 0.0;
            }
            if (write_side) node_side[pos] = right ? 1 : 0;
            big |= (p.w >> 22) != 0;
            const uint32_t w32 = (uint32_t)p.w & 0x3fffffu, rmask = right ? ~0u : 0u;
            uint32_t vsq_i = 0;
#pragma unroll
            for (int k = 0; k < N; k++) {
                const uint32_t val = (p.key >> (30 - 2 * k)) & 3u;
                const uint32_t ti = val * w32;          // = (uint32_t)((float)val * (float)weight): both exact below 2^24
                acc_r[k] += ti & rmask;
                if (first) acc_t[k] += ti;
                vsq_i += val * val;                     // the float dot product of small integers, exact in any order
            }
            const float dvf = mode == TQ_MODE_PROJ ? w : w * (float)vsq_i;
            bad |= !ex[0].add_if(dvf, !right); bad |= !ex[1].add_if(dvf, right);
            rw += right ? p.w : 0ull; lw += right ? 0ull : p.w; ln += right ? 0u : 1u;
        }
        TQ_TICK(8);
        // wave totals (DPP prefix: the last lane holds the sum) -> LDS -> wave 0 adds the eight waves' parts and decides
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int i = 0; i < 16; i++) acc_r[i] = wave_sum_u32(acc_r[i]);
        if (first)
#pragma unroll
            for (int i = 0; i < 16; i++) acc_t[i] = wave_sum_u32(acc_t[i]);
        uint64_t s7[7] = {lw, rw, (uint64_t)ln, ex[0].lo, ex[0].hi, ex[1].lo, ex[1].hi};
#pragma unroll
        for (int i = 0; i < 7; i++) s7[i] = wave_sum_u64(s7[i]);
        const uint32_t wflags = (__ballot(big) != 0 ? 1u : 0u) | (__ballot(bad) != 0 ? 2u : 0u);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < 16; i++) s_part[wave * 32 + i] = acc_r[i];
            if (first)
#pragma unroll
                for (int i = 0; i < 16; i++) s_part[wave * 32 + 16 + i] = acc_t[i];
#pragma unroll
            for (int i = 0; i < 7; i++) s_p64[wave * 8 + i] = s7[i];
            s_p64[wave * 8 + 7] = wflags;
        }
        TQ_TICK(9);
        __syncthreads();
        if (wave == 0) {
            // lanes 0..15: right sum / total / left sum of component `lane`; lanes 32..39: the block's eight 64-bit sums
            uint64_t r = 0, t = 0, v64 = 0;
            if (lane < 16) {
                for (int w8 = 0; w8 < TQ_THREADS / 64; w8++) { r += s_part[w8 * 32 + lane]; if (first) t += s_part[w8 * 32 + 16 + lane]; }
                if (first) s_tot[lane] = (uint32_t)t; else t = s_tot[lane];
            } else if (lane >= 32 && lane < 40) {
                for (int w8 = 0; w8 < TQ_THREADS / 64; w8++) { const uint64_t x = s_p64[w8 * 8 + (lane - 32)]; v64 = (lane == 39) ? (v64 | x) : (v64 + x); }
            }
            const uint64_t l = t - r;   // (no 32-bit wrap can have happened if four times the node's weight fits 32 bits: checked below)
            const bool inexact = __ballot(lane < 16 && (r >= 16777216ull || l >= 16777216ull)) != 0;
            uint64_t sums[8];
#pragma unroll
            for (int i = 0; i < 8; i++) sums[i] = __shfl(v64, 32 + i, 64);
            uint32_t status = 0;
            double t0 = 0.0, t1 = 0.0;
            if ((sums[7] & 1ull) || inexact || ((sums[0] + sums[1]) >> 30) != 0) status = 1;
            else if ((sums[7] & 2ull) || !exact_total(sums[3], sums[4], &t0) || !exact_total(sums[5], sums[6], &t1)) status = 2;
            if (status == 0) {
                if (lane < 16) { s_sum[0][lane] = (float)(uint32_t)l; s_sum[1][lane] = (float)(uint32_t)r; }
                if (lane == 0) { s_dsum[0] = t0; s_dsum[1] = t1; c.l_w = sums[0]; c.r_w = sums[1]; c.l_n = (uint32_t)sums[2]; c.r_n = count - (uint32_t)sums[2]; }
            }
            if (lane == 0) *s_status = status;
        }
        __syncthreads();
        const uint32_t status = *s_status;
        TQ_TICK(10);
        // the totals are good once a pass got as far as adding them up without leaving the integer range (status 1 may be a wrap: do not trust them then)
        if (first && status != 1) have_totals = true;
        return (int)status;
        }
    };

    auto side_pass = [&](int mode, bool write_side) -> bool {
        if (mode == TQ_MODE_DIST || mode == TQ_MODE_PROJ) {
            const int r = side_pass_exact(mode, write_side);
            if (r == 0) return false;
            if (r == 2) return true;
        }
        float acc_f = 0.0f; double acc_d = 0.0;
        uint64_t lw = 0, rw = 0; uint32_t ln = 0;
        exact_acc ex[2];
        bool bad = false;
        const uint32_t first_member = members[0];
        if constexpr (Src::PACKED) {
            // a selector component takes four values only, so the squared centroid differences of enc.h:483 come from a 16 x 4 table
            // (same double operations on the same operands as the direct form below, computed once per pass instead of per member)
            if (mode == TQ_MODE_DIST && tid < 64) {
                const int k = tid >> 2, val = tid & 3;
                const double a = (double)c.l_c[k] - (double)(float)val, b = (double)c.r_c[k] - (double)(float)val;
                s_tab[k][val] = make_double2(a * a, b * b);
            }
            __syncthreads();
        }
        pipeline_pass<2 * N, EX ? 0 : 2, EX ? 1 : 2>(lds, src, w64, members, count, // wave 0: 2N float chains, wave 1: the two double chains
            [&](uint32_t pos, const payload& p, float* f, double* d) {
                float v[N]; Src::decode(p, v);
                const float w = (float)p.w;
                bool right;
                if (mode == TQ_MODE_DIST) {
                    double dl = 0, dr = 0;
                    if constexpr (Src::PACKED) {
#pragma unroll
                        for (int k = 0; k < N; k++) { const double2 t = s_tab[k][(p.key >> (30 - 2 * k)) & 3u]; dl += t.x; dr += t.y; }
                    } else {
#pragma unroll
                        for (int k = 0; k < N; k++) {
                            const double a = (double)c.l_c[k] - (double)v[k]; const double aa = a * a; dl += aa;
                            const double b = (double)c.r_c[k] - (double)v[k]; const double bb = b * b; dr += bb;
                        }
                    }
                    right = dl >= dr;                                 // enc.h:1991
                } else if (mode == TQ_MODE_PROJ) {
                    float dd[N];
#pragma unroll
                    for (int k = 0; k < N; k++) dd[k] = v[k] - s_origin[k];
                    const double t = (double)dot_seq<N>(dd, c.axis);
                    right = t >= 0.0;                                 // enc.h:1870-1871
                } else if (mode == TQ_MODE_PEEL_FIRST) {
                    right = (pos == 0) || (p.mi == first_member);     // rows are distinct: only the first member equals itself (enc.h:2030)
                } else {
                    right = pos >= count / 2;                         // enc.h:1929-1945
                }
                if (write_side) node_side[pos] = right ? 1 : 0;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const float t = v[k] * w;
                    f[(size_t)k * TQ_STRIDE] = right ? 0.0f : t;
                    f[(size_t)(N + k) * TQ_STRIDE] = right ? t : 0.0f;
                }
                // l_weight / r_weight (projection and half passes) are doubles of the float weight; otherwise ttsum addends
                const float dvf = (mode == TQ_MODE_PROJ || mode == TQ_MODE_HALF) ? w : w * dot_seq<N>(v, v);
                if (EX) {
                    bad |= !ex[0].add_if(dvf, !right); bad |= !ex[1].add_if(dvf, right);
                } else {
                    const double dv = (double)dvf;
                    d[0] = right ? 0.0 : dv;
                    d[TQ_TILE] = right ? dv : 0.0;
                }
                if (right) rw += p.w; else { lw += p.w; ln++; }
            },
            [&](const float* f, const double* d, uint32_t m) {
                if (tid < 2 * N) chain_add_f32(acc_f, f + (size_t)tid * TQ_STRIDE, m);
                else if (!EX && (tid == 64 || tid == 65)) chain_add_f64(acc_d, d + (size_t)(tid - 64) * TQ_TILE, m);
            });
        if (tid < 2 * N) s_sum[tid / N][tid % N] = acc_f;
        if (!EX && tid == 64) s_dsum[0] = acc_d;
        if (!EX && tid == 65) s_dsum[1] = acc_d;
        uint64_t sums[7] = {lw, rw, (uint64_t)ln, ex[0].lo, ex[0].hi, ex[1].lo, ex[1].hi};
        block_sum_u64xN<7>(sums, s_red);
        bool failed = false;
        if (EX) {
            double t0 = 0.0, t1 = 0.0;
            failed = __syncthreads_or(bad ? 1 : 0) != 0;
            failed |= !exact_total(sums[3], sums[4], &t0) || !exact_total(sums[5], sums[6], &t1); // same value in every thread
            if (tid == 0) { s_dsum[0] = t0; s_dsum[1] = t1; }
        }
        if (tid == 0) { c.l_w = sums[0]; c.r_w = sums[1]; c.l_n = (uint32_t)sums[2]; c.r_n = count - (uint32_t)sums[2]; }
        __syncthreads();
        return failed;
    };

    TQ_TICK(0);
    // ---------------- prep_split
    if (count == 2) {
        if (tid == 0) {
            float v0[N], v1[N];
            Src::decode(src.fetch(w64, members[0]), v0); Src::decode(src.fetch(w64, members[1]), v1);
            for (int k = 0; k < N; k++) { c.l_c[k] = v0[k]; c.r_c[k] = v1[k]; }
        }
        __syncthreads();
    } else {
        // covariance (enc.h:1810-1823): chain (x, y>=x) adds d[x]*wd[y] per member, d = v - origin, wd = weight*d
        constexpr int C = N * (N + 1) / 2;
        constexpr int CW = (C + 63) / 64;
        int cx = 0, cy = 0;
        if (tid < C) { int cc = tid; while (cc >= N - cx) { cc -= N - cx; cx++; } cy = cx + cc; }
        float cv = 0.0f;
        pipeline_pass<2 * N, 0, CW>(lds, src, w64, members, count,
            [&](uint32_t, const payload& p, float* f, double*) {
                float v[N]; Src::decode(p, v);
                const float w = (float)p.w;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const float dk = v[k] - s_origin[k];
                    f[(size_t)k * TQ_STRIDE] = dk;
                    f[(size_t)(N + k) * TQ_STRIDE] = w * dk;
                }
            },
            [&](const float* f, const double*, uint32_t m) {
                if (tid < C) chain_add_prod_f32(cv, f + (size_t)cx * TQ_STRIDE, f + (size_t)(N + cy) * TQ_STRIDE, m);
            });
        TQ_TICK(1);
        if (tid < C) {   // enc.h:1825-1834: every entry times 1 / weight, mirrored -- by the thread that summed it
            const float renorm = 1.0f / (float)nd.weight;
            const float e = cv * renorm;
            c.cov[cx][cy] = e; c.cov[cy][cx] = e;
        }
        __syncthreads();
        TQ_TICK(2);
        if (tid < 64) principal_axis_wave<N>(c.cov, c.axis);
        __syncthreads();
        TQ_TICK(3);
    }

    // ---------------- the classification passes of prep_split and refine_split, driven from ONE call site (the pass body is
    // large; four inlined copies of it would not fit the instruction cache). Every decision below is uniform across the block.
    enum { PH_PROJ, PH_HALF, PH_REFINE, PH_PEEL };
    int phase = (count == 2) ? PH_REFINE : PH_PROJ;
    int iter = 0;
    if (tid == 0) { c.prev_total = 1e+10f; c.state = 0; }
    __syncthreads();
    for (;;) {
        const int mode = phase == PH_PROJ ? TQ_MODE_PROJ : phase == PH_HALF ? TQ_MODE_HALF : phase == PH_REFINE ? TQ_MODE_DIST : TQ_MODE_PEEL_FIRST;
        if (side_pass(mode, phase == PH_REFINE || phase == PH_PEEL)) { if (tid == 0) out->ok = 2; return; }
        TQ_TICK(4); TQ_COUNT(14);
        if (phase == PH_PROJ) {
            if (tid == 0) {
                const double lw = s_dsum[0], rw = s_dsum[1];
                if (lw > 0.0 && rw > 0.0) {
                    const float ls = (float)(1.0 / lw), rs = (float)(1.0 / rw);
                    for (int k = 0; k < N; k++) { c.l_c[k] = s_sum[0][k] * ls; c.r_c[k] = s_sum[1][k] * rs; }
                    c.mode = TQ_MODE_DIST;
                } else {
                    c.mode = TQ_MODE_HALF; // degenerate projection (enc.h:1893): needs the bounding box first
                }
            }
            __syncthreads();
            if (c.mode != TQ_MODE_HALF) { phase = PH_REFINE; continue; }
            // per-dimension min/max over the members (order independent)
            float lo = 1e+20f, hi = -1e+20f;
            const int k = tid % 16, lane_group = tid / 16;
            if (k < N)
                for (uint32_t i = (uint32_t)lane_group; i < count; i += TQ_THREADS / 16) {
                    float v[N]; Src::decode(src.fetch(w64, members[i]), v);
                    float x = v[0];
#pragma unroll
                    for (int kk = 1; kk < N; kk++) x = (kk == k) ? v[kk] : x;
                    lo = fminf(lo, x); hi = fmaxf(hi, x);
                }
            float* red = reinterpret_cast<float*>(lds);
            red[tid] = lo; red[TQ_THREADS + tid] = hi;
            __syncthreads();
            if (tid < N) {
                float l = 1e+20f, h = -1e+20f;
                for (int g = 0; g < TQ_THREADS / 16; g++) { l = fminf(l, red[g * 16 + tid]); h = fmaxf(h, red[TQ_THREADS + g * 16 + tid]); }
                c.cov[0][tid] = l; c.cov[1][tid] = h; // the covariance is spent: rows 0 / 1 keep the bounding box for the half pass
            }
            __syncthreads();
            float widest = 0.0f; int widest_axis = -1;
            for (int kk = 0; kk < N; kk++) { const float r = c.cov[1][kk] - c.cov[0][kk]; if (r > widest) { widest = r; widest_axis = kk; } }
            if (widest_axis < 0) {
                if (tid == 0) out->ok = 0;
                return;
            }
            phase = PH_HALF;
            continue;
        }
        if (phase == PH_HALF) {
            if (tid == 0) {
                const double lw = s_dsum[0], rw = s_dsum[1];
                if (lw > 0.0 && rw > 0.0) {
                    const float ls = (float)(1.0 / lw), rs = (float)(1.0 / rw);
                    for (int kk = 0; kk < N; kk++) { c.l_c[kk] = s_sum[0][kk] * ls; c.r_c[kk] = s_sum[1][kk] * rs; }
                } else {
                    for (int kk = 0; kk < N; kk++) { c.l_c[kk] = c.cov[0][kk]; c.r_c[kk] = c.cov[1][kk]; }
                }
            }
            __syncthreads();
            phase = PH_REFINE;
            continue;
        }
        // refine_split (enc.h:1962-2077): up to 6 two-means iterations
        if (c.l_w == 0 || c.r_w == 0) {
            if (phase == PH_PEEL) { // peeling the first member off did not help either
                if (tid == 0) out->ok = 0;
                return;
            }
            phase = PH_PEEL;
            continue;
        }
        phase = PH_REFINE;
        if (tid == 0) {
            float nl[N], nr[N];
            for (int k = 0; k < N; k++) { nl[k] = s_sum[0][k]; nr[k] = s_sum[1][k]; }
            const float lwf = (float)c.l_w, rwf = (float)c.r_w;
            const float ql = dot_seq<N>(nl, nl) / lwf, qr = dot_seq<N>(nr, nr) / rwf;
            c.l_var = (float)(s_dsum[0] - (double)ql);
            c.r_var = (float)(s_dsum[1] - (double)qr);
            const float li = 1.0f / lwf, ri = 1.0f / rwf;
            for (int k = 0; k < N; k++) { c.l_c[k] = nl[k] * li; c.r_c[k] = nr[k] * ri; }
            const float total = c.l_var + c.r_var;
            if (total < .00001f) c.state = 1;
            else {
                const float rel = (c.prev_total - total) / total;
                if (rel < .00125f) c.state = 1;
                else c.prev_total = total;
            }
        }
        __syncthreads();
        TQ_TICK(5);
        if (c.state || ++iter == 6) break;
    }

    // ---------------- children member lists: stable partition of the (ascending) member list by the last classification
    const uint32_t l_n = c.l_n;
    if (tid == 0) { s_base[0] = 0; s_base[1] = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < count; base += TQ_THREADS) {
        const uint32_t pos = base + (uint32_t)tid;
        const bool valid = pos < count;
        const bool right = valid && node_side[pos] != 0;
        const bool left = valid && !right;
        const uint64_t mL = __ballot(left), mR = __ballot(right);
        const uint32_t lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
        const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const uint32_t pl = __popcll(mL & below), pr = __popcll(mR & below);
        if (lane == 0) { s_scan[wave][0] = __popcll(mL); s_scan[wave][1] = __popcll(mR); }
        __syncthreads();
        uint32_t ol = s_base[0], orr = s_base[1];
        for (uint32_t w = 0; w < wave; w++) { ol += s_scan[w][0]; orr += s_scan[w][1]; }
        if (left) child_members[ol + pl] = members[pos];
        if (right) child_members[l_n + orr + pr] = members[pos];
        __syncthreads();
        if (tid == 0) {
            uint32_t tl = 0, tr = 0;
            for (int w = 0; w < TQ_THREADS / 64; w++) { tl += s_scan[w][0]; tr += s_scan[w][1]; }
            s_base[0] += tl; s_base[1] += tr;
        }
        __syncthreads();
    }
    TQ_TICK(6);
    if (tid == 0) {
        out->ok = 1; out->l_count = c.l_n; out->r_count = c.r_n; out->l_weight = c.l_w; out->r_weight = c.r_w;
        out->l_var = c.l_var; out->r_var = c.r_var;
        for (int k = 0; k < 16; k++) { out->l_centroid[k] = k < N ? c.l_c[k] : 0.0f; out->r_centroid[k] = k < N ? c.r_c[k] : 0.0f; }
    }
}

template <int N, typename Src, bool EX>
__global__ __launch_bounds__(TQ_THREADS) void k_tsvq_split(Src src, const uint64_t* __restrict__ w64, uint32_t* __restrict__ perm0, uint32_t* __restrict__ perm1,
                                                          uint8_t* __restrict__ side, const tsvq_node_in* __restrict__ nodes, tsvq_split_out* __restrict__ outs) {
    tsvq_split_body<N, Src, EX>(src, w64, perm0, perm1, side, nodes, outs);
}
// The same, compiled for TWO workgroups per CU (four waves per SIMD: 128 registers, ~150 bytes of scratch per lane for the exact packed variant, which takes 173 on its
// own): for the rounds of a tree's lower levels, which have more nodes than the chip has CUs -- there the second resident workgroup is worth more than the spills cost.
template <int N, typename Src, bool EX>
__global__ __launch_bounds__(TQ_THREADS, 4) void k_tsvq_split_dense(Src src, const uint64_t* __restrict__ w64, uint32_t* __restrict__ perm0, uint32_t* __restrict__ perm1,
                                                                   uint8_t* __restrict__ side, const tsvq_node_in* __restrict__ nodes, tsvq_split_out* __restrict__ outs) {
    tsvq_split_body<N, Src, EX>(src, w64, perm0, perm1, side, nodes, outs);
}

// The covariance pass of split_node on its own, chained sums, for the many-workgroup path (tsvq_wide_kernels.hip), which uses it for nodes where 136
// order-preserving walks cost more than one pass of dependent adds. A chain lane is bound by instruction ISSUE, not by the dependent add: a wave64 instruction
// takes four cycles to issue, and the one-workgroup kernel's lane issues a multiply, an add and half an LDS read per member (11 cycles). The batches that take
// this path have few nodes, so every node is given COV_GROUPS workgroups, each streaming all members but owning a third of the chains: its four producer waves
// lay out the chains' PRODUCTS (d[x] * (w * d[y]), the operands and the rounding of enc.h:1819) and its one consumer wave only adds, 16 bytes of LDS per
// four members (5 cycles per member). The raw sums go to ctrl[].sums; k_wide_finish<WM_COV> turns them into the axis. Group 0 also lays the members out in
// list order (key, float weight) for the passes that follow.
constexpr int COV_GROUPS = 3, COV_GROUP_CHAINS = 46;
constexpr int cov_chain_x(int c) { int x = 0; while (c >= 16 - x) { c -= 16 - x; x++; } return x; }
constexpr int cov_chain_y(int c) { int x = 0; while (c >= 16 - x) { c -= 16 - x; x++; } return x + c; }

template <int C> struct cov_chain { static constexpr int x = cov_chain_x(C), y = cov_chain_y(C); };
// the products of chains FIRST + J...: the component indices are template constants, so d[] and wd[] stay in registers
template <int FIRST, int... J>
__device__ __forceinline__ void cov_products(float* f, const float (&d)[16], const float (&wd)[16], std::integer_sequence<int, J...>) {
    ((f[(size_t)J * TQ_STRIDE] = d[cov_chain<FIRST + J>::x] * wd[cov_chain<FIRST + J>::y]), ...);
}

template <int GROUP>
__device__ __forceinline__ void cov_axis_group(char* lds, const float* s_origin, packed16_rows src, const uint64_t* __restrict__ w64, const uint32_t* __restrict__ members,
                                               const tsvq_wide_node& nd, tsvq_wide_ctrl* __restrict__ ct, uint2* __restrict__ pk) {
    constexpr int N = 16, C = N * (N + 1) / 2;
    constexpr int FIRST = GROUP * COV_GROUP_CHAINS, COUNT = (C - FIRST) < COV_GROUP_CHAINS ? (C - FIRST) : COV_GROUP_CHAINS;
    const int tid = threadIdx.x;
    float cv = 0.0f;
    pipeline_pass<COV_GROUP_CHAINS, 0, 1>(lds, src, w64, members, nd.count,
        [&](uint32_t pos, const packed16_rows::payload& p, float* f, double*) {
            float v[N]; packed16_rows::decode(p, v);
            const float w = (float)p.w;
            if (GROUP == 0) pk[nd.start + pos] = make_uint2(p.key, __float_as_uint(w));
            float d[N], wd[N];
#pragma unroll
            for (int k = 0; k < N; k++) { d[k] = v[k] - s_origin[k]; wd[k] = w * d[k]; }
            cov_products<FIRST>(f, d, wd, std::make_integer_sequence<int, COUNT>{});
        },
        [&](const float* f, const double*, uint32_t m) {
            if (tid < COUNT) chain_add_f32(cv, f + (size_t)tid * TQ_STRIDE, m);
        });
    if (tid < COUNT) ct->sums[FIRST + tid] = cv;
}

__global__ __launch_bounds__(TQ_THREADS) void k_tsvq_cov_axis(packed16_rows src, const uint64_t* __restrict__ w64, const uint32_t* __restrict__ perm0,
                                                             const uint32_t* __restrict__ perm1, const tsvq_wide_node* __restrict__ nodes,
                                                             tsvq_wide_ctrl* __restrict__ ctrl, uint2* __restrict__ pk) {
    static_assert(COV_GROUPS == 3 && COV_GROUPS * COV_GROUP_CHAINS >= 136 && COV_GROUP_CHAINS <= 64, "chain groups");
    extern __shared__ __align__(16) char lds[];
    __shared__ float s_origin[16];
    const int tid = threadIdx.x;
    const tsvq_wide_node nd = nodes[blockIdx.x];
    const uint32_t* members = tsvq_list(perm0, perm1, nd.buf) + nd.start;
    if (tid < 16) s_origin[tid] = nodes[blockIdx.x].origin[tid];
    __syncthreads();
    // one instantiation per group: which components a chain multiplies must be known at compile time (run-time indices into v[] would put it in scratch)
    if (blockIdx.y == 0) cov_axis_group<0>(lds, s_origin, src, w64, members, nd, ctrl + blockIdx.x, pk);
    else if (blockIdx.y == 1) cov_axis_group<1>(lds, s_origin, src, w64, members, nd, ctrl + blockIdx.x, pk);
    else cov_axis_group<2>(lds, s_origin, src, w64, members, nd, ctrl + blockIdx.x, pk);
}

// The same for the 6-float rows of the endpoint tree (tsvq_wide6_kernels.hip): 21 chains, one workgroup per node. Signed covariance chains of these rows change
// binade too often for the parity maps to pay (305 us for a 35,000-member root against 135 us of dependent adds), so the covariance pass of that path stays chained.
// The producers also lay out, in list order, what the passes that follow add per member: va[k][start + pos] = v_k * w (the side chains' addends) and
// tta[start + pos] = (double)(w * |v|^2) (the ttsum addend) -- those passes then read coalesced arrays instead of gathering rows through the member list.
constexpr int cov6_chain_x(int c) { int x = 0; while (c >= 6 - x) { c -= 6 - x; x++; } return x; }
constexpr int cov6_chain_y(int c) { int x = 0; while (c >= 6 - x) { c -= 6 - x; x++; } return x + c; }
template <int C> struct cov6_chain { static constexpr int x = cov6_chain_x(C), y = cov6_chain_y(C); };
template <int... J>
__device__ __forceinline__ void cov6_products(float* f, const float (&d)[6], const float (&wd)[6], std::integer_sequence<int, J...>) {
    ((f[(size_t)J * TQ_STRIDE] = d[cov6_chain<J>::x] * wd[cov6_chain<J>::y]), ...);
}

__global__ __launch_bounds__(TQ_THREADS) void k_tsvq_cov_axis6(float_rows<6> src, const uint64_t* __restrict__ w64, const uint32_t* __restrict__ perm0,
                                                              const uint32_t* __restrict__ perm1, const tsvq_wide_node* __restrict__ nodes,
                                                              tsvq_wide_ctrl* __restrict__ ctrl, float* __restrict__ va, double* __restrict__ tta, uint32_t n) {
    constexpr int N = 6, C = 21;
    extern __shared__ __align__(16) char lds[];
    __shared__ float s_origin[N];
    const int tid = threadIdx.x;
    const tsvq_wide_node nd = nodes[blockIdx.x];
    const uint32_t* members = tsvq_list(perm0, perm1, nd.buf) + nd.start;
    if (tid < N) s_origin[tid] = nodes[blockIdx.x].origin[tid];
    __syncthreads();
    float cv = 0.0f;
    pipeline_pass<C, 0, 1>(lds, src, w64, members, nd.count,
        [&](uint32_t pos, const float_rows<6>::payload& p, float* f, double*) {
            float v[N]; float_rows<6>::decode(p, v);
            const float w = (float)p.w;
#pragma unroll
            for (int k = 0; k < N; k++) va[(size_t)k * n + nd.start + pos] = v[k] * w;
            tta[nd.start + pos] = (double)(w * dot_seq<N>(v, v));
            float d[N], wd[N];
#pragma unroll
            for (int k = 0; k < N; k++) { d[k] = v[k] - s_origin[k]; wd[k] = w * d[k]; }
            cov6_products(f, d, wd, std::make_integer_sequence<int, C>{});
        },
        [&](const float* f, const double*, uint32_t m) {
            if (tid < C) chain_add_f32(cv, f + (size_t)tid * TQ_STRIDE, m);
        });
    if (tid < C) ctrl[blockIdx.x].sums[tid] = cv;
}

// -------------------------------------------------------------------------------------------------------------------

static size_t tsvq_lds_bytes(int n) {
    const size_t cov = (size_t)2 * 2 * n * TQ_STRIDE * sizeof(float);
    const size_t side = 2 * ((((size_t)2 * n * TQ_STRIDE * sizeof(float) + (size_t)2 * TQ_TILE * sizeof(double)) + 15) / 16 * 16);
    const size_t red = (size_t)2 * TQ_THREADS * sizeof(float);
    return std::max(std::max(cov, side), red);
}

template <typename K> static hipError_t set_lds(K kernel, size_t lds) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t launch_tsvq_cov_axis(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1,
                                const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_packed) {
    if (!n_nodes) return hipSuccess;
    const size_t lds = (size_t)2 * COV_GROUP_CHAINS * TQ_STRIDE * sizeof(float);   // two tiles of one float row per chain of the group
    hipError_t e = set_lds(k_tsvq_cov_axis, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_tsvq_cov_axis, dim3(n_nodes, COV_GROUPS), dim3(TQ_THREADS), lds, st, packed16_rows{d_keys}, d_w64, d_perm0, d_perm1, d_nodes, d_ctrl, static_cast<uint2*>(d_packed));
    return hipGetLastError();
}

hipError_t launch_tsvq_cov_axis6(hipStream_t st, const float* d_rows, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1,
                                 const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, float* d_va, double* d_tta, uint32_t n) {
    if (!n_nodes) return hipSuccess;
    const size_t lds = (size_t)2 * 21 * TQ_STRIDE * sizeof(float);
    hipError_t e = set_lds(k_tsvq_cov_axis6, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_tsvq_cov_axis6, dim3(n_nodes), dim3(TQ_THREADS), lds, st, float_rows<6>{d_rows}, d_w64, d_perm0, d_perm1, d_nodes, d_ctrl, d_va, d_tta, n);
    return hipGetLastError();
}

hipError_t launch_tsvq_root(hipStream_t st, int dim, bool packed, bool exact, const void* d_rows, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0, tsvq_root_out* d_out) {
    if (!n) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tsvq_iota, dim3((n + 255) / 256), dim3(256), 0, st, n, d_perm0);
    const size_t lds = tsvq_lds_bytes(dim);
    hipError_t e;
#define TQ_LAUNCH_ROOT(NN, SRC, EXV, srcval) do { \
        if ((e = set_lds(k_tsvq_root<NN, SRC, EXV>, lds)) != hipSuccess) return e; \
        hipLaunchKernelGGL((k_tsvq_root<NN, SRC, EXV>), dim3(1), dim3(TQ_THREADS), lds, st, srcval, d_w64, n, d_out, nullptr, nullptr, nullptr); } while (0)
    if (dim == 16 && packed) {
        packed16_rows src{static_cast<const uint32_t*>(d_rows)};
        if (exact) TQ_LAUNCH_ROOT(16, packed16_rows, true, src); else TQ_LAUNCH_ROOT(16, packed16_rows, false, src);
    } else if (dim == 16) {
        float_rows<16> src{static_cast<const float*>(d_rows)};
        TQ_LAUNCH_ROOT(16, float_rows<16>, false, src);
    } else if (dim == 6 && !packed) {
        float_rows<6> src{static_cast<const float*>(d_rows)};
        TQ_LAUNCH_ROOT(6, float_rows<6>, false, src);
    } else return hipErrorInvalidValue;
#undef TQ_LAUNCH_ROOT
    return hipGetLastError();
}

#ifdef TQ_PROFILE
extern "C" __attribute__((visibility("default"))) int tsvq_profile_read(unsigned long long* out16) {
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tq_prof), sizeof(g_tq_prof)) != hipSuccess) return 0;
    unsigned long long zero[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tq_prof), zero, sizeof(zero)) == hipSuccess ? 1 : 0;
}
#endif

hipError_t launch_tsvq_span_roots(hipStream_t st, int dim, bool packed, bool exact, const void* d_rows, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1,
                                  const tsvq_node_in* d_nodes, uint32_t n_nodes, tsvq_root_out* d_outs) {
    if (!n_nodes) return hipSuccess;
    const size_t lds = tsvq_lds_bytes(dim);
    hipError_t e;
#define TQ_LAUNCH_ROOTS(NN, SRC, EXV, srcval) do { \
        if ((e = set_lds(k_tsvq_root<NN, SRC, EXV>, lds)) != hipSuccess) return e; \
        hipLaunchKernelGGL((k_tsvq_root<NN, SRC, EXV>), dim3(n_nodes), dim3(TQ_THREADS), lds, st, srcval, d_w64, 0u, d_outs, d_perm0, d_perm1, d_nodes); } while (0)
    if (dim == 16 && packed) {
        packed16_rows src{static_cast<const uint32_t*>(d_rows)};
        if (exact) TQ_LAUNCH_ROOTS(16, packed16_rows, true, src); else TQ_LAUNCH_ROOTS(16, packed16_rows, false, src);
    } else if (dim == 16) {
        float_rows<16> src{static_cast<const float*>(d_rows)};
        TQ_LAUNCH_ROOTS(16, float_rows<16>, false, src);
    } else if (dim == 6 && !packed) {
        float_rows<6> src{static_cast<const float*>(d_rows)};
        TQ_LAUNCH_ROOTS(6, float_rows<6>, false, src);
    } else return hipErrorInvalidValue;
#undef TQ_LAUNCH_ROOTS
    return hipGetLastError();
}

hipError_t launch_tsvq_split(hipStream_t st, int dim, bool packed, bool exact, const void* d_rows, const uint64_t* d_w64, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side,
                             const tsvq_node_in* d_nodes, uint32_t n_nodes, tsvq_split_out* d_outs, uint32_t dense_min) {
    if (!n_nodes) return hipSuccess;
    const size_t lds = tsvq_lds_bytes(dim);
    hipError_t e;
#define TQ_LAUNCH_SPLIT(NN, SRC, EXV, srcval) do { \
        if ((e = set_lds(k_tsvq_split<NN, SRC, EXV>, lds)) != hipSuccess) return e; \
        hipLaunchKernelGGL((k_tsvq_split<NN, SRC, EXV>), dim3(n_nodes), dim3(TQ_THREADS), lds, st, srcval, d_w64, d_perm0, d_perm1, d_side, d_nodes, d_outs); } while (0)
    if (dim == 16 && packed) {
        packed16_rows src{static_cast<const uint32_t*>(d_rows)};
        // dense_min (bu_hip_tuning::tsvq_dense_min, default 257): node count of a round from which the two-workgroups-per-CU build of the exact kernel is used (0: never)
        if (exact && dense_min && n_nodes >= dense_min) {
            if ((e = set_lds(k_tsvq_split_dense<16, packed16_rows, true>, lds)) != hipSuccess) return e;
            hipLaunchKernelGGL((k_tsvq_split_dense<16, packed16_rows, true>), dim3(n_nodes), dim3(TQ_THREADS), lds, st, src, d_w64, d_perm0, d_perm1, d_side, d_nodes, d_outs);
        } else
        if (exact) TQ_LAUNCH_SPLIT(16, packed16_rows, true, src); else TQ_LAUNCH_SPLIT(16, packed16_rows, false, src);
    } else if (dim == 16) {
        float_rows<16> src{static_cast<const float*>(d_rows)};
        TQ_LAUNCH_SPLIT(16, float_rows<16>, false, src);
    } else if (dim == 6 && !packed) {
        float_rows<6> src{static_cast<const float*>(d_rows)};
        TQ_LAUNCH_SPLIT(6, float_rows<6>, false, src);
    } else return hipErrorInvalidValue;
#undef TQ_LAUNCH_SPLIT
    return hipGetLastError();
}

} // namespace bu
