// bookkeeping_kernels.hip -- the cluster bookkeeping BETWEEN the frontend's stages, kept on the device (rows a15 and the list handling
// inside a9 / a10 / a13 / a14 of SURVEY.md 8a). The reference keeps every clustering as lists of block ids (basisu_frontend.h:186-299) and
// rebuilds them on the host after every stage; here a clustering is two per-block arrays in HBM -- cluster index and position inside
// the cluster's list -- and the lists the per-cluster kernels read (CSR) are produced from them by these kernels, so that block-sized arrays
// never cross PCIe between stages. Only codebook-sized data (a few thousand entries) visits the host. All integer work.
//
//   k_blocks_from_groups     distinct-vector level results (TSVQ leaf, parent, first list position) -> per-block cluster / position / parent
//   rank_blocks              frontend.cpp:1921-1942: lists rebuilt in block order = stable sort of the block ids by cluster (hipCUB radix
//                            sort) + the rank of every block inside its cluster, + the cluster sizes
//   k_endpoint_csr_fill      the (cluster, position) map as the CSR list array of training-vector ids (2b, 2b + 1) generate_endpoint_codebook reads
//   k_remap_clusters         eliminate_redundant_or_empty_endpoint_clusters / optimize_selector_codebook applied to the per-block arrays
//   k_count_differences      how many blocks changed cluster (refine_endpoint_clusterization's return value)
//   k_membership             which clusters occur under which parent (compute_*_clusters_within_each_parent_cluster)
//   k_scatter_spans          TSVQ leaves (spans of the member buffers) -> leaf index per distinct vector
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <algorithm>

#include "bookkeeping_kernels.h"
#include "sort_pairs.h"
#include "tsvq_bufs.h"

namespace bu {

namespace {

__global__ void __launch_bounds__(256) k_blocks_from_groups(const uint32_t* __restrict__ goffs, const uint32_t* __restrict__ idx, uint32_t n, uint32_t u_total,
                                                            const uint32_t* __restrict__ leaf_of_unique, const uint32_t* __restrict__ first_pos,
                                                            const uint32_t* __restrict__ parent_of_unique, uint32_t* __restrict__ cluster, uint32_t* __restrict__ pos,
                                                            uint8_t* __restrict__ parent) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    uint32_t lo = 0, hi = u_total;   // the group holding sorted position j: last u with goffs[u] <= j
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (goffs[mid] <= j) lo = mid; else hi = mid; }
    const uint32_t u = lo, b = idx[j];
    cluster[b] = leaf_of_unique[u];
    if (pos) pos[b] = first_pos[u] + (j - goffs[u]);
    if (parent) parent[b] = parent_of_unique ? (uint8_t)parent_of_unique[u] : 0;
}

__global__ void __launch_bounds__(256) k_iota_copy(const uint32_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { keys[i] = src[i]; vals[i] = i; }
}
// offsets[c] = first position of cluster c in the sorted key array (= number of keys below c), c = 0..k; sizes[c] = offsets[c + 1] - offsets[c]
__global__ void __launch_bounds__(256) k_offsets_from_sorted(const uint32_t* __restrict__ keys_sorted, uint32_t n, uint32_t k, uint32_t* __restrict__ offsets) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c > k) return;
    uint32_t lo = 0, hi = n;   // first i with keys_sorted[i] >= c
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys_sorted[mid] < c) lo = mid + 1; else hi = mid; }
    offsets[c] = lo;
}
__global__ void __launch_bounds__(256) k_sizes_from_offsets(const uint32_t* __restrict__ offsets, uint32_t k, uint32_t* __restrict__ sizes) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c <= k) sizes[c] = c < k ? offsets[c + 1] - offsets[c] : 0u;
}
__global__ void __launch_bounds__(256) k_positions(const uint32_t* __restrict__ keys_sorted, const uint32_t* __restrict__ blocks_sorted, uint32_t n,
                                                   const uint32_t* __restrict__ offsets, uint32_t* __restrict__ pos) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pos[blocks_sorted[i]] = i - offsets[keys_sorted[i]];
}

__global__ void __launch_bounds__(256) k_endpoint_csr_fill(const uint32_t* __restrict__ cluster, const uint32_t* __restrict__ pos, uint32_t n,
                                                           const uint32_t* __restrict__ offsets, uint32_t* __restrict__ indices) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const size_t at = (size_t)offsets[cluster[b]] + 2ull * pos[b];
    indices[at] = b * 2; indices[at + 1] = b * 2 + 1;
}

__global__ void __launch_bounds__(256) k_remap_clusters(uint32_t* __restrict__ cluster, uint32_t* __restrict__ pos, uint32_t n, const uint32_t* __restrict__ new_index,
                                                        const uint32_t* __restrict__ base) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const uint32_t old = cluster[b];
    if (pos && base) pos[b] += base[old];
    cluster[b] = new_index[old];
}

__global__ void __launch_bounds__(256) k_count_differences(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t n, uint32_t* __restrict__ out) {
    __shared__ uint32_t s_c[4];
    uint32_t c = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) c += a[i] != b[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && (s_c[0] | s_c[1] | s_c[2] | s_c[3])) atomicAdd(out, s_c[0] + s_c[1] + s_c[2] + s_c[3]);
}

__global__ void __launch_bounds__(256) k_membership(const uint8_t* __restrict__ parent, const uint32_t* __restrict__ cluster, uint32_t n, uint32_t clusters,
                                                    uint8_t* __restrict__ flags) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b < n) flags[(size_t)(parent ? parent[b] : 0) * clusters + cluster[b]] = 1;
}

__global__ void __launch_bounds__(256) k_scatter_spans(const uint32_t* __restrict__ perm0, const uint32_t* __restrict__ perm1, const bk_span* __restrict__ spans,
                                                       uint32_t* __restrict__ out) {
    const bk_span s = spans[blockIdx.x];
    const uint32_t* p = tsvq_list(perm0, perm1, s.buf) + s.start;
    for (uint32_t i = threadIdx.x; i < s.count; i += 256) out[p[i]] = s.value;
}

// The same for a finished tree in one pass, with what the endpoint side needs on top: every span is a leaf (its index = blockIdx.x, its value = the parent cut it
// lies under); per member (a distinct training vector) the leaf, the parent and -- when the vectors' group offsets are given -- the position of the vector's first block
// inside its leaf's block list = the blocks of the members in front of it in the span (list order: enc.h:1573-1584 + the training-vector order of frontend.cpp:825-866);
// per leaf the size of that list.
__global__ void __launch_bounds__(256) k_finish_spans(const uint32_t* __restrict__ perm0, const uint32_t* __restrict__ perm1, const bk_span* __restrict__ spans,
                                                      uint32_t* __restrict__ leaf_of, uint32_t* __restrict__ parent_of, const uint32_t* __restrict__ goffs,
                                                      uint32_t* __restrict__ first_pos, uint32_t* __restrict__ sizes) {
    __shared__ uint32_t s_wave[4];
    const bk_span s = spans[blockIdx.x];
    const uint32_t* p = tsvq_list(perm0, perm1, s.buf) + s.start;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t run = 0;
    for (uint32_t base = 0; base < s.count; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const bool have = i < s.count;
        const uint32_t m = have ? p[i] : 0u;
        if (have) { leaf_of[m] = blockIdx.x; if (parent_of) parent_of[m] = s.value; }
        if (!goffs) continue;   // (uniform)
        const uint32_t sz = have ? goffs[m + 1] - goffs[m] : 0u;
        uint32_t incl = sz;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
        __syncthreads();
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = run;
        for (int w = 0; w < wave; w++) before += s_wave[w];
        if (have) first_pos[m] = before + incl - sz;
        run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    }
    if (goffs && threadIdx.x == 0) sizes[blockIdx.x] = run;
}

// distinct endpoint training vectors from their 48-bit keys (low r,g,b | high r,g,b): the six floats of frontend.cpp:846-851 (byte / 255 as byte * (1 / 255), the
// reference's own expression) and the weight of the vector's group (both sub-blocks of every block, weight 1 each)
__global__ void __launch_bounds__(256) k_endpoint_rows(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ goffs, uint32_t n, float* __restrict__ rows,
                                                       uint64_t* __restrict__ weights) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const uint64_t k = keys[u];
#pragma unroll
    for (int c = 5; c >= 0; c--) rows[(size_t)u * 6 + (size_t)(5 - c)] = (float)(int)((k >> (8 * c)) & 255) * (1.0f / 255.0f);
    weights[u] = 2ull * (goffs[u + 1] - goffs[u]);
}

// multi-GPU TSVQ: the child member lists of a batch of split nodes, laid end to end in batch order (staging), to and from the member buffers.
// A node's children live in the NEXT buffer after the node's (tsvq_bufs.h), at the node's own [start, start + count). dir 0: buffers -> staging for the nodes
// flagged in `take` (zero for the others), dir 1: staging -> buffers for the flagged nodes.
__global__ void __launch_bounds__(256) k_exchange_children(uint32_t* __restrict__ perm0, uint32_t* __restrict__ perm1, const bk_span* __restrict__ nodes /* value = offset in staging */,
                                                           const uint8_t* __restrict__ take, uint32_t* __restrict__ staging, int dir) {
    const bk_span nd = nodes[blockIdx.x];
    uint32_t* child = tsvq_child_list(perm0, perm1, nd.buf) + nd.start;
    uint32_t* st = staging + nd.value;
    const bool mine = take[blockIdx.x] != 0;
    for (uint32_t i = threadIdx.x; i < nd.count; i += 256) {
        if (dir == 0) st[i] = mine ? child[i] : 0u;
        else if (mine) child[i] = st[i];
    }
}

// `index` and `out` may be the same array (bu_hip_kmeans_codebook gathers in place): every thread reads and writes its own element only, so neither is __restrict__
__global__ void __launch_bounds__(256) k_gather_u32(const uint32_t* __restrict__ table, const uint32_t* index, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = table[index[i]];
}

struct rank_temp { uint32_t *keys_in, *vals_in, *keys_sorted; void* cub; size_t cub_bytes; };

rank_temp carve_rank(void* ws, uint32_t n, uint32_t k, size_t* total) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)sort_pairs<uint32_t, uint32_t>(nullptr, sort_bytes, nullptr, nullptr, nullptr, nullptr, n, 0, 32, nullptr);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)k + 1);
    rank_temp t;
    char* p = static_cast<char*>(ws);
    size_t off = 0;
    t.keys_in = reinterpret_cast<uint32_t*>(p + off); off += up((size_t)n * 4);
    t.vals_in = reinterpret_cast<uint32_t*>(p + off); off += up((size_t)n * 4);
    t.keys_sorted = reinterpret_cast<uint32_t*>(p + off); off += up((size_t)n * 4);
    t.cub = p + off; t.cub_bytes = up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes); off += t.cub_bytes;
    if (total) *total = off;
    return t;
}

} // namespace

size_t rank_blocks_workspace_bytes(uint32_t n, uint32_t k) { size_t t = 0; carve_rank(nullptr, n ? n : 1, k, &t); return t; }

hipError_t launch_blocks_from_groups(hipStream_t st, const uint32_t* d_goffs, const uint32_t* d_idx, uint32_t n, uint32_t u_total, const uint32_t* d_leaf_of_unique,
                                     const uint32_t* d_first_pos, const uint32_t* d_parent_of_unique, uint32_t* d_cluster, uint32_t* d_pos, uint8_t* d_parent) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_blocks_from_groups, dim3((n + 255) / 256), dim3(256), 0, st, d_goffs, d_idx, n, u_total, d_leaf_of_unique, d_first_pos, d_parent_of_unique, d_cluster,
                       d_pos, d_parent);
    return hipGetLastError();
}

hipError_t launch_rank_blocks(hipStream_t st, const uint32_t* d_cluster, uint32_t n, uint32_t k, void* d_ws, uint32_t* d_sizes, uint32_t* d_offsets,
                              uint32_t* d_sorted_blocks, uint32_t* d_pos) {
    if (!n) return hipSuccess;
    const rank_temp t = carve_rank(d_ws, n, k, nullptr);
    hipError_t e;
    const dim3 grid((n + 255) / 256), blk(256), gk((k + 256) / 256);
    hipLaunchKernelGGL(k_iota_copy, grid, blk, 0, st, d_cluster, n, t.keys_in, t.vals_in);
    int bits = 1;
    while (bits < 32 && (1u << bits) < k) bits++;
    size_t bytes = t.cub_bytes;
    if ((e = sort_pairs<uint32_t, uint32_t>(t.cub, bytes, t.keys_in, t.keys_sorted, t.vals_in, d_sorted_blocks, n, 0, (unsigned)bits, st)) != hipSuccess) return e;
    // sizes and offsets from the sorted keys (a histogram by atomics on a few thousand skewed bins costs more than the sort)
    hipLaunchKernelGGL(k_offsets_from_sorted, gk, blk, 0, st, t.keys_sorted, n, k, d_offsets);
    hipLaunchKernelGGL(k_sizes_from_offsets, gk, blk, 0, st, d_offsets, k, d_sizes);
    if (d_pos) hipLaunchKernelGGL(k_positions, grid, blk, 0, st, t.keys_sorted, d_sorted_blocks, n, d_offsets, d_pos);
    return hipGetLastError();
}

hipError_t launch_endpoint_csr_fill(hipStream_t st, const uint32_t* d_cluster, const uint32_t* d_pos, uint32_t n, const uint32_t* d_offsets, uint32_t* d_indices) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_endpoint_csr_fill, dim3((n + 255) / 256), dim3(256), 0, st, d_cluster, d_pos, n, d_offsets, d_indices);
    return hipGetLastError();
}

hipError_t launch_remap_clusters(hipStream_t st, uint32_t* d_cluster, uint32_t* d_pos, uint32_t n, const uint32_t* d_new_index, const uint32_t* d_base) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_remap_clusters, dim3((n + 255) / 256), dim3(256), 0, st, d_cluster, d_pos, n, d_new_index, d_base);
    return hipGetLastError();
}

hipError_t launch_count_differences(hipStream_t st, const uint32_t* d_a, const uint32_t* d_b, uint32_t n, uint32_t* d_count) {
    hipError_t e = hipMemsetAsync(d_count, 0, 4, st);
    if (e != hipSuccess || !n) return e;
    hipLaunchKernelGGL(k_count_differences, dim3(std::min<uint32_t>((n + 255) / 256, 1024u)), dim3(256), 0, st, d_a, d_b, n, d_count);
    return hipGetLastError();
}

hipError_t launch_membership(hipStream_t st, const uint8_t* d_parent, const uint32_t* d_cluster, uint32_t n, uint32_t parents, uint32_t clusters, uint8_t* d_flags) {
    hipError_t e = hipMemsetAsync(d_flags, 0, (size_t)parents * clusters, st);
    if (e != hipSuccess || !n) return e;
    hipLaunchKernelGGL(k_membership, dim3((n + 255) / 256), dim3(256), 0, st, d_parent, d_cluster, n, clusters, d_flags);
    return hipGetLastError();
}

hipError_t launch_scatter_spans(hipStream_t st, const uint32_t* d_perm0, const uint32_t* d_perm1, const bk_span* d_spans, uint32_t n_spans, uint32_t* d_out) {
    if (!n_spans) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_spans, dim3(n_spans), dim3(256), 0, st, d_perm0, d_perm1, d_spans, d_out);
    return hipGetLastError();
}

hipError_t launch_finish_spans(hipStream_t st, const uint32_t* d_perm0, const uint32_t* d_perm1, const bk_span* d_spans, uint32_t n_spans, uint32_t* d_leaf_of, uint32_t* d_parent_of,
                               const uint32_t* d_goffs, uint32_t* d_first_pos, uint32_t* d_sizes) {
    if (!n_spans) return hipSuccess;
    hipLaunchKernelGGL(k_finish_spans, dim3(n_spans), dim3(256), 0, st, d_perm0, d_perm1, d_spans, d_leaf_of, d_parent_of, d_goffs, d_first_pos, d_sizes);
    return hipGetLastError();
}

hipError_t launch_endpoint_rows(hipStream_t st, const uint64_t* d_keys, const uint32_t* d_goffs, uint32_t n, float* d_rows, uint64_t* d_weights) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_endpoint_rows, dim3((n + 255) / 256), dim3(256), 0, st, d_keys, d_goffs, n, d_rows, d_weights);
    return hipGetLastError();
}

hipError_t launch_exchange_children(hipStream_t st, uint32_t* d_perm0, uint32_t* d_perm1, const bk_span* d_nodes, const uint8_t* d_take, uint32_t n_nodes, uint32_t* d_staging, int dir) {
    if (!n_nodes) return hipSuccess;
    hipLaunchKernelGGL(k_exchange_children, dim3(n_nodes), dim3(256), 0, st, d_perm0, d_perm1, d_nodes, d_take, d_staging, dir);
    return hipGetLastError();
}

hipError_t launch_gather_u32(hipStream_t st, const uint32_t* d_table, const uint32_t* d_index, uint32_t n, uint32_t* d_out) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_gather_u32, dim3((n + 255) / 256), dim3(256), 0, st, d_table, d_index, n, d_out);
    return hipGetLastError();
}

} // namespace bu
