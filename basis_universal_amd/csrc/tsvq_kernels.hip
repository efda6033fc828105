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
#include "tsvq_kernels.h"

namespace bu {

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

template <int N> __device__ __forceinline__ float dot_seq(const float* a, const float* b) {
    float r = a[0] * b[0];
#pragma unroll
    for (int i = 1; i < N; i++) r += a[i] * b[i];
    return r;
}

// ---- where a training vector comes from
template <int N>
struct float_rows {
    const float* rows;
    struct payload { float v[N]; uint64_t w; uint32_t mi; };
#ifndef BU_TQ_DEPTH_FLOAT
#define BU_TQ_DEPTH_FLOAT 4
#endif
    static constexpr int DEPTH = BU_TQ_DEPTH_FLOAT;
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
    const uint32_t* keys;
    struct payload { uint32_t key; uint64_t w; uint32_t mi; };
// Depth 8 is NOT safe here: at -O3 the packed root kernel then returns a wrong origin in ~25% of launches on MI355X
// (tools/tsvq_root_repeat.py; -O1 at depth 8 and -O3 at depths 1/4 are stable) -- treated as a code generation hazard of the
// 8-way unrolled register queue, so the queue stays at 4 and tests/test_gpu_tsvq.py repeats the root launch to catch a relapse.
#ifndef BU_TQ_DEPTH_PACKED
#define BU_TQ_DEPTH_PACKED 4
#endif
    static constexpr int DEPTH = BU_TQ_DEPTH_PACKED;
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
    constexpr int K = Src::DEPTH;
    constexpr size_t F_BYTES = (size_t)FROWS * TQ_STRIDE * sizeof(float);
    constexpr size_t D_BYTES = (size_t)DROWS * TQ_TILE * sizeof(double);
    constexpr size_t BUF_BYTES = ((F_BYTES + D_BYTES + 15) / 16) * 16;
    static_assert(TQ_THREADS - CW * 64 >= TQ_TILE, "one producer thread per tile slot");
    const int tid = threadIdx.x;
    const bool consumer = tid < CW * 64;
    const uint32_t pid = (uint32_t)(tid - CW * 64);
    const bool producer = !consumer && pid < (uint32_t)TQ_TILE;
    const uint32_t tiles = (count + TQ_TILE - 1) / TQ_TILE;
    auto fptr = [&](uint32_t t) { return reinterpret_cast<float*>(lds + (size_t)(t & 1u) * BUF_BYTES); };
    auto dptr = [&](uint32_t t) { return reinterpret_cast<double*>(lds + (size_t)(t & 1u) * BUF_BYTES + F_BYTES); };
    using payload = typename Src::payload;
    auto fetch_tile = [&](uint32_t t) -> payload {
        payload p{};
        const uint32_t pos = t * TQ_TILE + pid;
        if (producer && t < tiles && pos < count) p = src.fetch(w64, members ? members[pos] : pos);
        return p;
    };
    auto emit_tile = [&](uint32_t t, const payload& p) {
        const uint32_t pos = t * TQ_TILE + pid;
        if (producer && pos < count) emit(pos, p, fptr(t) + pid, dptr(t) + pid);
    };
    payload q[K];
#pragma unroll
    for (int i = 0; i < K; i++) q[i] = fetch_tile((uint32_t)i);
    emit_tile(0, q[0]);
    q[0] = fetch_tile((uint32_t)K);
    __syncthreads();
    for (uint32_t t0 = 0; t0 < tiles; t0 += K) {
#pragma unroll
        for (int i = 0; i < K; i++) {
            const uint32_t t = t0 + (uint32_t)i;
            if (t >= tiles) break;
            if (!consumer) {
                if (t + 1 < tiles) {
                    emit_tile(t + 1, q[(i + 1) % K]);
                    q[(i + 1) % K] = fetch_tile(t + 1 + K);
                }
            } else {
                consume(fptr(t), dptr(t), min((uint32_t)TQ_TILE, count - t * TQ_TILE));
            }
            __syncthreads();
        }
    }
}

// Order-preserving running sum of one LDS row: the adds happen strictly in member order (one dependent v_add per member is the
// floor of this algorithm), the 16-byte LDS reads of the NEXT group are issued before the current group is added so that the LDS
// latency stays off the dependent chain.
__device__ __forceinline__ void chain_add_f32(float& acc, const float* src, uint32_t m) {
    if (m == (uint32_t)TQ_TILE) {
        const float4* p = reinterpret_cast<const float4*>(src);
        float4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
#pragma unroll
        for (int g = 0; g < TQ_TILE / 16; g++) {
            float4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
            if (g + 1 < TQ_TILE / 16) { n0 = p[g * 4 + 4]; n1 = p[g * 4 + 5]; n2 = p[g * 4 + 6]; n3 = p[g * 4 + 7]; }
            acc += c0.x; acc += c0.y; acc += c0.z; acc += c0.w; acc += c1.x; acc += c1.y; acc += c1.z; acc += c1.w;
            acc += c2.x; acc += c2.y; acc += c2.z; acc += c2.w; acc += c3.x; acc += c3.y; acc += c3.z; acc += c3.w;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
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
        const double2* p = reinterpret_cast<const double2*>(src);
        double2 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];
#pragma unroll
        for (int g = 0; g < TQ_TILE / 8; g++) {
            double2 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
            if (g + 1 < TQ_TILE / 8) { n0 = p[g * 4 + 4]; n1 = p[g * 4 + 5]; n2 = p[g * 4 + 6]; n3 = p[g * 4 + 7]; }
            acc += c0.x; acc += c0.y; acc += c1.x; acc += c1.y; acc += c2.x; acc += c2.y; acc += c3.x; acc += c3.y;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
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
        const float4* pa = reinterpret_cast<const float4*>(dx);
        const float4* pb = reinterpret_cast<const float4*>(wy);
        float4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
#pragma unroll
        for (int g = 0; g < TQ_TILE / 8; g++) {
            float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
            if (g + 1 < TQ_TILE / 8) { na0 = pa[g * 2 + 2]; na1 = pa[g * 2 + 3]; nb0 = pb[g * 2 + 2]; nb1 = pb[g * 2 + 3]; }
            const float p0 = a0.x * b0.x, p1 = a0.y * b0.y, p2 = a0.z * b0.z, p3 = a0.w * b0.w;
            const float p4 = a1.x * b1.x, p5 = a1.y * b1.y, p6 = a1.z * b1.z, p7 = a1.w * b1.w;
            acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3; acc = acc + p4; acc = acc + p5; acc = acc + p6; acc = acc + p7;
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
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

// compute_pca_from_covar (enc.h:605-648) on one thread: 8 power iterations, double row sums, float early-out.
template <int N>
__device__ __noinline__ void principal_axis(tq_ctrl& c) {
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
            for (int j = 0; j < N; j++) { const float p = c.cov[i][j] * axis[j]; sum += p; }
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
    for (int i = 0; i < N; i++) c.axis[i] = axis[i];
}

__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t* scratch /* TQ_THREADS/64 */) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    uint64_t s = 0;
    for (int w = 0; w < TQ_THREADS / 64; w++) s += scratch[w];
    return s;
}

__global__ __launch_bounds__(256) void k_tsvq_iota(uint32_t n, uint32_t* __restrict__ perm0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm0[i] = i;
}

// prepare_root (enc.h:1708-1735): origin sums, weight, variance of the whole training set
template <int N, typename Src>
__global__ __launch_bounds__(TQ_THREADS) void k_tsvq_root(Src src, const uint64_t* __restrict__ w64, uint32_t n, tsvq_root_out* __restrict__ out) {
    extern __shared__ __align__(16) char lds[];
    __shared__ float s_origin[16];
    __shared__ double s_tt;
    __shared__ uint64_t s_red[TQ_THREADS / 64];
    const int tid = threadIdx.x;
    float acc_f = 0.0f; double acc_d = 0.0;
    uint64_t wsum = 0;
    pipeline_pass<N, 1, 2>(lds, src, w64, nullptr, n, // wave 0: the N float chains, wave 1: the double chain
        [&](uint32_t, const typename Src::payload& p, float* f, double* d) {
            float v[N]; Src::decode(p, v);
            const float w = (float)p.w;
#pragma unroll
            for (int k = 0; k < N; k++) f[(size_t)k * TQ_STRIDE] = v[k] * w;
            const float t = dot_seq<N>(v, v) * w;
            d[0] = (double)t;
            wsum += p.w;
        },
        [&](const float* f, const double* d, uint32_t m) {
            if (tid < N) chain_add_f32(acc_f, f + (size_t)tid * TQ_STRIDE, m);
            else if (tid == 64) chain_add_f64(acc_d, d, m);
        });
    wsum = block_sum_u64(wsum, s_red);
    if (tid < N) s_origin[tid] = acc_f;
    if (tid == 64) s_tt = acc_d;
    __syncthreads();
    if (tid == 0) {
        float o[N];
        for (int k = 0; k < N; k++) o[k] = s_origin[k];
        const float wfl = (float)wsum;
        const float q = dot_seq<N>(o, o) / wfl;
        out->var = (float)(s_tt - (double)q);
        const float inv = 1.0f / wfl;
        for (int k = 0; k < N; k++) out->origin[k] = o[k] * inv;
        for (int k = N; k < 16; k++) out->origin[k] = 0.0f;
        out->weight = wsum;
    }
}

// split_node (enc.h:1737-1800) = prep_split (:1848-1960) + refine_split (:1962-2077) for one node per workgroup.
template <int N, typename Src>
__global__ __launch_bounds__(TQ_THREADS) void k_tsvq_split(Src src, const uint64_t* __restrict__ w64, uint32_t* __restrict__ perm0, uint32_t* __restrict__ perm1,
                                                          uint8_t* __restrict__ side, const tsvq_node_in* __restrict__ nodes, tsvq_split_out* __restrict__ outs) {
    extern __shared__ __align__(16) char lds[];
    __shared__ tq_ctrl c;
    __shared__ float s_origin[16];
    __shared__ float s_sum[2][16];
    __shared__ double s_dsum[2];
    __shared__ uint64_t s_red[TQ_THREADS / 64];
    __shared__ uint32_t s_scan[TQ_THREADS / 64][2];
    __shared__ uint32_t s_base[2];
    using payload = typename Src::payload;

    const int tid = threadIdx.x;
    const tsvq_node_in nd = nodes[blockIdx.x];
    tsvq_split_out* out = outs + blockIdx.x;
    const uint32_t count = nd.count;
    const uint32_t* members = (nd.buf ? perm1 : perm0) + nd.start;
    uint32_t* child_members = (nd.buf ? perm0 : perm1) + nd.start;
    uint8_t* node_side = side + nd.start;
    if (tid < 16) s_origin[tid] = nd.origin[tid];
    __syncthreads();

    // One classification + accumulation pass. `mode` selects how a member picks its side; float chains 0..N-1 are the left sums,
    // N..2N-1 the right sums; the two double chains are the left/right "ttsum" (in the projection / half passes: the weights).
    auto side_pass = [&](int mode, bool write_side) {
        float acc_f = 0.0f; double acc_d = 0.0;
        uint64_t lw = 0, rw = 0; uint32_t ln = 0;
        const uint32_t first_member = members[0];
        pipeline_pass<2 * N, 2, 2>(lds, src, w64, members, count, // wave 0: 2N float chains, wave 1: the two double chains
            [&](uint32_t pos, const payload& p, float* f, double* d) {
                float v[N]; Src::decode(p, v);
                const float w = (float)p.w;
                bool right;
                if (mode == TQ_MODE_DIST) {
                    double dl = 0, dr = 0;
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const double a = (double)c.l_c[k] - (double)v[k]; const double aa = a * a; dl += aa;
                        const double b = (double)c.r_c[k] - (double)v[k]; const double bb = b * b; dr += bb;
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
                double dv;
                if (mode == TQ_MODE_PROJ || mode == TQ_MODE_HALF) dv = (double)w; // l_weight / r_weight: doubles of the float weight
                else { const float tt = w * dot_seq<N>(v, v); dv = (double)tt; }
                d[0] = right ? 0.0 : dv;
                d[TQ_TILE] = right ? dv : 0.0;
                if (right) rw += p.w; else { lw += p.w; ln++; }
            },
            [&](const float* f, const double* d, uint32_t m) {
                if (tid < 2 * N) chain_add_f32(acc_f, f + (size_t)tid * TQ_STRIDE, m);
                else if (tid == 64 || tid == 65) chain_add_f64(acc_d, d + (size_t)(tid - 64) * TQ_TILE, m);
            });
        if (tid < 2 * N) s_sum[tid / N][tid % N] = acc_f;
        if (tid == 64) s_dsum[0] = acc_d;
        if (tid == 65) s_dsum[1] = acc_d;
        const uint64_t LW = block_sum_u64(lw, s_red);
        const uint64_t RW = block_sum_u64(rw, s_red);
        const uint64_t LN = block_sum_u64((uint64_t)ln, s_red);
        if (tid == 0) { c.l_w = LW; c.r_w = RW; c.l_n = (uint32_t)LN; c.r_n = count - (uint32_t)LN; }
        __syncthreads();
    };

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
        if (tid < C) c.cov[cx][cy] = cv;
        __syncthreads();
        if (tid == 0) {
            const float renorm = 1.0f / (float)nd.weight;
            for (int x = 0; x < N; x++) for (int y = x; y < N; y++) c.cov[x][y] *= renorm;
            for (int x = 0; x < N - 1; x++) for (int y = x + 1; y < N; y++) c.cov[y][x] = c.cov[x][y];
            principal_axis<N>(c);
        }
        __syncthreads();
        side_pass(TQ_MODE_PROJ, false);
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
        if (c.mode == TQ_MODE_HALF) {
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
                s_sum[0][tid] = l; s_sum[1][tid] = h;
            }
            __syncthreads();
            float bb_lo[N], bb_hi[N];
            for (int kk = 0; kk < N; kk++) { bb_lo[kk] = s_sum[0][kk]; bb_hi[kk] = s_sum[1][kk]; }
            __syncthreads();
            float widest = 0.0f; int widest_axis = -1;
            for (int kk = 0; kk < N; kk++) { const float r = bb_hi[kk] - bb_lo[kk]; if (r > widest) { widest = r; widest_axis = kk; } }
            if (widest_axis < 0) {
                if (tid == 0) out->ok = 0;
                return;
            }
            side_pass(TQ_MODE_HALF, false);
            if (tid == 0) {
                const double lw = s_dsum[0], rw = s_dsum[1];
                if (lw > 0.0 && rw > 0.0) {
                    const float ls = (float)(1.0 / lw), rs = (float)(1.0 / rw);
                    for (int kk = 0; kk < N; kk++) { c.l_c[kk] = s_sum[0][kk] * ls; c.r_c[kk] = s_sum[1][kk] * rs; }
                } else {
                    for (int kk = 0; kk < N; kk++) { c.l_c[kk] = bb_lo[kk]; c.r_c[kk] = bb_hi[kk]; }
                }
            }
            __syncthreads();
        }
    }

    // ---------------- refine_split: up to 6 two-means iterations
    if (tid == 0) { c.prev_total = 1e+10f; c.state = 0; }
    __syncthreads();
    for (int iter = 0; iter < 6; iter++) {
        side_pass(TQ_MODE_DIST, true);
        if (c.l_w == 0 || c.r_w == 0) {
            side_pass(TQ_MODE_PEEL_FIRST, true);
            if (c.l_w == 0 || c.r_w == 0) {
                if (tid == 0) out->ok = 0;
                return;
            }
        }
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
        if (c.state) break;
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
        const uint32_t lane = tid & 63, wave = tid >> 6;
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
    if (tid == 0) {
        out->ok = 1; out->l_count = c.l_n; out->r_count = c.r_n; out->l_weight = c.l_w; out->r_weight = c.r_w;
        out->l_var = c.l_var; out->r_var = c.r_var;
        for (int k = 0; k < 16; k++) { out->l_centroid[k] = k < N ? c.l_c[k] : 0.0f; out->r_centroid[k] = k < N ? c.r_c[k] : 0.0f; }
    }
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

hipError_t launch_tsvq_root(hipStream_t st, int dim, bool packed, const void* d_rows, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0, tsvq_root_out* d_out) {
    if (!n) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tsvq_iota, dim3((n + 255) / 256), dim3(256), 0, st, n, d_perm0);
    const size_t lds = tsvq_lds_bytes(dim);
    hipError_t e;
    if (dim == 16 && packed) {
        packed16_rows src{static_cast<const uint32_t*>(d_rows)};
        if ((e = set_lds(k_tsvq_root<16, packed16_rows>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_tsvq_root<16, packed16_rows>), dim3(1), dim3(TQ_THREADS), lds, st, src, d_w64, n, d_out);
    } else if (dim == 16) {
        float_rows<16> src{static_cast<const float*>(d_rows)};
        if ((e = set_lds(k_tsvq_root<16, float_rows<16>>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_tsvq_root<16, float_rows<16>>), dim3(1), dim3(TQ_THREADS), lds, st, src, d_w64, n, d_out);
    } else if (dim == 6 && !packed) {
        float_rows<6> src{static_cast<const float*>(d_rows)};
        if ((e = set_lds(k_tsvq_root<6, float_rows<6>>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_tsvq_root<6, float_rows<6>>), dim3(1), dim3(TQ_THREADS), lds, st, src, d_w64, n, d_out);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_tsvq_split(hipStream_t st, int dim, bool packed, const void* d_rows, const uint64_t* d_w64, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side,
                             const tsvq_node_in* d_nodes, uint32_t n_nodes, tsvq_split_out* d_outs) {
    if (!n_nodes) return hipSuccess;
    const size_t lds = tsvq_lds_bytes(dim);
    hipError_t e;
    if (dim == 16 && packed) {
        packed16_rows src{static_cast<const uint32_t*>(d_rows)};
        if ((e = set_lds(k_tsvq_split<16, packed16_rows>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_tsvq_split<16, packed16_rows>), dim3(n_nodes), dim3(TQ_THREADS), lds, st, src, d_w64, d_perm0, d_perm1, d_side, d_nodes, d_outs);
    } else if (dim == 16) {
        float_rows<16> src{static_cast<const float*>(d_rows)};
        if ((e = set_lds(k_tsvq_split<16, float_rows<16>>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_tsvq_split<16, float_rows<16>>), dim3(n_nodes), dim3(TQ_THREADS), lds, st, src, d_w64, d_perm0, d_perm1, d_side, d_nodes, d_outs);
    } else if (dim == 6 && !packed) {
        float_rows<6> src{static_cast<const float*>(d_rows)};
        if ((e = set_lds(k_tsvq_split<6, float_rows<6>>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_tsvq_split<6, float_rows<6>>), dim3(n_nodes), dim3(TQ_THREADS), lds, st, src, d_w64, d_perm0, d_perm1, d_side, d_nodes, d_outs);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

} // namespace bu
