// unique_kernels.hip -- de-duplication of the selector training vectors on the device (row a12 + the std::map de-duplication that
// generate_hierarchical_codebook_threaded performs before the TSVQ, encoder/basisu_frontend.cpp:2140-2189, encoder/basisu_enc.h:2218-2290).
//
// A selector training vector is the block's 16 two-bit selectors; the reference's std::map<vec16F, weight> orders the distinct vectors like
// the 32-bit word that holds selector (0,0) in its top two bits ... (3,3) in its bottom two. So the de-duplication is: one key per block,
// a STABLE radix sort of (key, block) pairs, run-length encoding of the sorted keys, and the sum of the u64 block weights of every run.
// Everything is integer work, so the result does not depend on the order of evaluation: distinct keys ascending, the blocks of every
// distinct vector ascending (stable sort), exact weight sums. The sort / run-length / scan primitives are hipCUB's (rocPRIM device-wide
// algorithms); the key builder and the per-run weight sum are ours.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "unique_kernels.h"
#include "sort_pairs.h"

namespace bu {

namespace {

// key of an ETC1S block: selector_of(x, y) for i = 0..15 with x = i & 3, y = i >> 2, first value in the top two bits. The block is stored
// big-endian; its low 32 bits hold the selector LSB plane in bits 0..15 and the MSB plane in bits 16..31, pixel (x, y) at bit x*4+y
// (etc.h:232-236); raw -> selector index is {2, 3, 1, 0}.
__global__ void __launch_bounds__(256) k_selector_keys(const uint64_t* __restrict__ enc_blocks, uint32_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const uint32_t lo = (uint32_t)__builtin_bswap64(enc_blocks[b]);
    uint32_t key = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
        const uint32_t bit = (i & 3u) * 4u + (i >> 2);
        const uint32_t raw = ((lo >> bit) & 1u) | (((lo >> (16u + bit)) & 1u) << 1);
        key = (key << 2) | ((0x1Eu >> (raw * 2u)) & 3u);  // {2, 3, 1, 0}[raw]
    }
    keys[b] = key;
    idx[b] = b;
}

__global__ void __launch_bounds__(256) k_close_offsets(const uint32_t* __restrict__ n_runs, uint32_t n, uint32_t* __restrict__ offsets) {
    if (blockIdx.x == 0 && threadIdx.x == 0) offsets[*n_runs] = n;
}

// weight of block idx[j] at sorted position j: the input of the per-run sum
__global__ void __launch_bounds__(256) k_gather_weights(const uint32_t* __restrict__ idx, const uint64_t* __restrict__ weights, uint32_t n, uint64_t* __restrict__ out) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) out[j] = weights[idx[j]];
}

// key of an ETC1S block for the endpoint training vectors (frontend.cpp:825-866): its low and high block colours (selector 0 and selector 3:
// etc_block::get_block_low_high_colors, etc.h:543-570), low rgb in bits 47..24, high rgb in bits 23..0 -- the lexicographic order of the
// reference's vec6F (the floats are monotone in these bytes).
// It is made from the 18 bits that determine it: an ETC1S block's endpoint training vector is a function of (colour5, intensity table) alone: 2^18 codes, 236,235
// distinct keys (clamping makes codes coincide). So instead of sorting 48-bit keys (six radix passes over 12-byte pairs), the blocks are sorted by the RANK of
// their code's key among all possible keys -- an 18-bit number from a table made once per device -- in three passes over 8-byte pairs; the distinct keys come back
// out of a second table. Same order (ranks are monotone in the keys, equal keys share a rank), same groups, same group order (the sort is stable).
__host__ __device__ inline uint64_t endpoint_key_of_code(uint32_t code) {   // code = r5 << 13 | g5 << 8 | b5 << 3 | inten
    const int r5 = (int)((code >> 13) & 31), g5 = (int)((code >> 8) & 31), b5 = (int)((code >> 3) & 31), inten = (int)(code & 7);
    const int big[8] = { 8, 17, 29, 42, 60, 80, 106, 183 };  // the outer modifiers of g_etc1_inten_tables (etc.cpp:304-308)
    const int d = big[inten];
    const int r = (r5 << 3) | (r5 >> 2), g = (g5 << 3) | (g5 >> 2), bl = (b5 << 3) | (b5 >> 2);
    auto c8 = [](int x) -> uint64_t { return (uint64_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); };
    return (c8(r - d) << 40) | (c8(g - d) << 32) | (c8(bl - d) << 24) | (c8(r + d) << 16) | (c8(g + d) << 8) | c8(bl + d);
}
constexpr uint32_t ENDPOINT_CODES = 1u << 18;
struct endpoint_rank_tables { uint32_t* rank_of_code = nullptr; uint64_t* key_of_rank = nullptr; uint32_t n_ranks = 0; hipError_t status = hipErrorNotInitialized; };

const endpoint_rank_tables& endpoint_tables(int device) {
    static endpoint_rank_tables tables[64];
    static std::once_flag once[64];
    if (device < 0 || device >= 64) { static const endpoint_rank_tables bad; return bad; }
    std::call_once(once[device], [device] {
        endpoint_rank_tables& t = tables[device];
        std::vector<std::pair<uint64_t, uint32_t>> kc(ENDPOINT_CODES);
        for (uint32_t c = 0; c < ENDPOINT_CODES; c++) kc[c] = { endpoint_key_of_code(c), c };
        std::sort(kc.begin(), kc.end());
        std::vector<uint32_t> rank(ENDPOINT_CODES);
        std::vector<uint64_t> keys;
        keys.reserve(ENDPOINT_CODES);
        for (uint32_t i = 0; i < ENDPOINT_CODES; i++) {
            if (i == 0 || kc[i].first != kc[i - 1].first) keys.push_back(kc[i].first);
            rank[kc[i].second] = (uint32_t)keys.size() - 1;
        }
        t.n_ranks = (uint32_t)keys.size();
        // (plain device allocations that live as long as the process: two tables of 1 and 2 MB per device)
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&t.rank_of_code), ENDPOINT_CODES * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t.key_of_rank), keys.size() * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemcpy(t.rank_of_code, rank.data(), ENDPOINT_CODES * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(t.key_of_rank, keys.data(), keys.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
        t.status = e;
    });
    return tables[device];
}

__global__ void __launch_bounds__(256) k_endpoint_ranks(const uint64_t* __restrict__ etc1_blocks, uint32_t n, const uint32_t* __restrict__ rank_of_code,
                                                        uint32_t* __restrict__ ranks, uint32_t* __restrict__ idx) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    const uint64_t v = __builtin_bswap64(etc1_blocks[b]);
    const uint32_t code = (uint32_t)(((v >> 59) & 31) << 13 | ((v >> 51) & 31) << 8 | ((v >> 43) & 31) << 3 | ((v >> 37) & 7));
    ranks[b] = rank_of_code[code];
    idx[b] = b;
}
// the distinct ranks of the sorted list -> the keys they stand for (n_runs is a device-side count: every thread below it has a run)
__global__ void __launch_bounds__(256) k_keys_of_ranks(const uint32_t* __restrict__ ranks, const uint32_t* __restrict__ n_runs, const uint64_t* __restrict__ key_of_rank,
                                                       uint32_t n, uint64_t* __restrict__ keys) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u < n && u < *n_runs) keys[u] = key_of_rank[ranks[u]];
}

struct unique_temp { uint32_t *keys_in, *idx_in, *keys_sorted, *counts, *n_runs, *keys_again; uint64_t* w_sorted; void* cub; size_t cub_bytes; };
size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

size_t cub_bytes_for(uint32_t n) {
    size_t a = 0, b = 0, c = 0, d = 0;
    uint32_t* p = nullptr;
    uint64_t* w = nullptr;
    (void)sort_pairs<uint32_t, uint32_t>(nullptr, a, p, p, p, p, n, 0, 32, nullptr);
    (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, b, p, p, p, p, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, p, p, (int)n);
    (void)hipcub::DeviceReduce::ReduceByKey(nullptr, d, p, p, w, w, p, hipcub::Sum(), (int)n);
    return std::max(std::max(a, b), std::max(c, d));
}

unique_temp carve(void* base, uint32_t n, size_t* total) {
    char* p = static_cast<char*>(base);
    size_t o = 0;
    unique_temp t;
    t.keys_in = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    t.idx_in = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    t.keys_sorted = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    t.counts = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    t.n_runs = reinterpret_cast<uint32_t*>(p + o); o += 256;
    t.keys_again = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    t.w_sorted = reinterpret_cast<uint64_t*>(p + o); o += align_up((size_t)n * 8);
    t.cub_bytes = cub_bytes_for(n);
    t.cub = p + o; o += align_up(t.cub_bytes);
    if (total) *total = o;
    return t;
}

struct unique64_temp { uint64_t *keys_in, *keys_sorted; uint32_t *idx_in, *counts, *n_runs; void* cub; size_t cub_bytes; };
size_t cub_bytes_for64(uint32_t n) {
    size_t a = 0, b = 0, c = 0;
    uint64_t* k = nullptr;
    uint32_t* p = nullptr;
    (void)sort_pairs<uint64_t, uint32_t>(nullptr, a, k, k, p, p, n, 0, 64, nullptr);
    (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, b, k, k, p, p, (int)n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, p, p, (int)n);
    size_t a32 = 0, b32 = 0;   // what actually runs: the 18-bit rank sort and its run-length pass (launch_unique_endpoint_vectors)
    (void)sort_pairs<uint32_t, uint32_t>(nullptr, a32, p, p, p, p, n, 0, 18, nullptr);
    (void)hipcub::DeviceRunLengthEncode::Encode(nullptr, b32, p, p, p, p, (int)n);
    return std::max(std::max(a, a32), std::max(std::max(b, b32), c));
}
unique64_temp carve64(void* base, uint32_t n, size_t* total) {
    char* p = static_cast<char*>(base);
    size_t o = 0;
    unique64_temp t;
    t.keys_in = reinterpret_cast<uint64_t*>(p + o); o += align_up((size_t)n * 8);
    t.keys_sorted = reinterpret_cast<uint64_t*>(p + o); o += align_up((size_t)n * 8);
    t.idx_in = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    t.counts = reinterpret_cast<uint32_t*>(p + o); o += align_up((size_t)n * 4);
    t.n_runs = reinterpret_cast<uint32_t*>(p + o); o += 256;
    t.cub_bytes = cub_bytes_for64(n);
    t.cub = p + o; o += align_up(t.cub_bytes);
    if (total) *total = o;
    return t;
}

} // namespace

size_t unique_endpoint_vectors_workspace_bytes(uint32_t n_blocks) {
    size_t total = 0;
    carve64(nullptr, n_blocks ? n_blocks : 1, &total);
    return total;
}

hipError_t launch_unique_endpoint_vectors(hipStream_t st, const void* d_etc1_blocks, uint32_t n, void* d_workspace, uint32_t* d_sorted_block_idx, uint64_t* d_unique_keys,
                                          uint32_t* d_group_offsets, uint32_t** d_n_unique) {
    const unique64_temp t = carve64(d_workspace, n, nullptr);
    if (d_n_unique) *d_n_unique = t.n_runs;
    if (!n) return hipMemsetAsync(t.n_runs, 0, 4, st);
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const endpoint_rank_tables& tab = endpoint_tables(device);
    if (tab.status != hipSuccess) return tab.status;
    // the workspace's 64-bit key arrays hold the 32-bit ranks (in, sorted) and the distinct ranks
    uint32_t* ranks_in = reinterpret_cast<uint32_t*>(t.keys_in);
    uint32_t* ranks_sorted = ranks_in + n;
    uint32_t* ranks_unique = reinterpret_cast<uint32_t*>(t.keys_sorted);
    hipLaunchKernelGGL(k_endpoint_ranks, dim3((n + 255) / 256), dim3(256), 0, st, static_cast<const uint64_t*>(d_etc1_blocks), n, tab.rank_of_code, ranks_in, t.idx_in);
    size_t bytes = t.cub_bytes;
    e = sort_pairs<uint32_t, uint32_t>(t.cub, bytes, ranks_in, ranks_sorted, t.idx_in, d_sorted_block_idx, n, 0, 18, st);
    if (e != hipSuccess) return e;
    bytes = t.cub_bytes;
    e = hipcub::DeviceRunLengthEncode::Encode(t.cub, bytes, ranks_sorted, ranks_unique, t.counts, t.n_runs, (int)n, st);
    if (e != hipSuccess) return e;
    bytes = t.cub_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(t.cub, bytes, t.counts, d_group_offsets, (int)n, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_close_offsets, dim3(1), dim3(64), 0, st, t.n_runs, n, d_group_offsets);
    const uint32_t most = n < tab.n_ranks ? n : tab.n_ranks;   // there cannot be more distinct vectors than blocks, or than possible keys
    hipLaunchKernelGGL(k_keys_of_ranks, dim3((most + 255) / 256), dim3(256), 0, st, ranks_unique, t.n_runs, tab.key_of_rank, most, d_unique_keys);
    return hipGetLastError();
}

size_t unique_selector_vectors_workspace_bytes(uint32_t n_blocks) {
    size_t total = 0;
    carve(nullptr, n_blocks ? n_blocks : 1, &total);
    return total;
}

hipError_t launch_unique_selector_vectors(hipStream_t st, const void* d_enc_blocks, const uint64_t* d_weights, uint32_t n, void* d_workspace, uint32_t* d_sorted_block_idx,
                                          uint32_t* d_unique_keys, uint64_t* d_unique_weights, uint32_t* d_group_offsets, uint32_t** d_n_unique) {
    const unique_temp t = carve(d_workspace, n, nullptr);
    if (d_n_unique) *d_n_unique = t.n_runs;
    if (!n) return hipMemsetAsync(t.n_runs, 0, 4, st);
    hipLaunchKernelGGL(k_selector_keys, dim3((n + 255) / 256), dim3(256), 0, st, static_cast<const uint64_t*>(d_enc_blocks), n, t.keys_in, t.idx_in);
    size_t bytes = t.cub_bytes;
    hipError_t e = sort_pairs<uint32_t, uint32_t>(t.cub, bytes, t.keys_in, t.keys_sorted, t.idx_in, d_sorted_block_idx, n, 0, 32, st);
    if (e != hipSuccess) return e;
    bytes = t.cub_bytes;
    e = hipcub::DeviceRunLengthEncode::Encode(t.cub, bytes, t.keys_sorted, d_unique_keys, t.counts, t.n_runs, (int)n, st);
    if (e != hipSuccess) return e;
    bytes = t.cub_bytes;
    // scanning all n counts keeps the launch independent of the (device-side) run count; entries past the runs are never read
    e = hipcub::DeviceScan::ExclusiveSum(t.cub, bytes, t.counts, d_group_offsets, (int)n, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_close_offsets, dim3(1), dim3(64), 0, st, t.n_runs, n, d_group_offsets);
    // weight of a distinct vector = sum over its run: a keyed reduction of the gathered weights (runs range from one block to a large part
    // of the image, so one thread per run would leave a few threads with all the work)
    hipLaunchKernelGGL(k_gather_weights, dim3((n + 255) / 256), dim3(256), 0, st, d_sorted_block_idx, d_weights, n, t.w_sorted);
    bytes = t.cub_bytes;
    e = hipcub::DeviceReduce::ReduceByKey(t.cub, bytes, t.keys_sorted, t.keys_again, t.w_sorted, d_unique_weights, t.n_runs + 1, hipcub::Sum(), (int)n, st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

} // namespace bu
