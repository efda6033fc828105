// bookkeeping_kernels.h -- launch interface of bookkeeping_kernels.hip (internal to libbasisu_hip.so; C ABI: include/basisu_hip.h, bu_hip_k_map_*).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bu {

struct bk_span { uint32_t buf, start, count, value; };   // mirrored by bu_tsvq_span in include/basisu_hip.h

hipError_t launch_blocks_from_groups(hipStream_t st, const uint32_t* d_goffs, const uint32_t* d_idx, uint32_t n, uint32_t u_total, const uint32_t* d_leaf_of_unique,
                                     const uint32_t* d_first_pos, const uint32_t* d_parent_of_unique, uint32_t* d_cluster, uint32_t* d_pos, uint8_t* d_parent);
size_t rank_blocks_workspace_bytes(uint32_t n, uint32_t k);
// d_sizes: k + 1 entries (the last stays 0), d_offsets: k + 1 entries, d_sorted_blocks: n, d_pos: n or nullptr
hipError_t launch_rank_blocks(hipStream_t st, const uint32_t* d_cluster, uint32_t n, uint32_t k, void* d_ws, uint32_t* d_sizes, uint32_t* d_offsets,
                              uint32_t* d_sorted_blocks, uint32_t* d_pos);
hipError_t launch_endpoint_csr_fill(hipStream_t st, const uint32_t* d_cluster, const uint32_t* d_pos, uint32_t n, const uint32_t* d_offsets, uint32_t* d_indices);
hipError_t launch_remap_clusters(hipStream_t st, uint32_t* d_cluster, uint32_t* d_pos, uint32_t n, const uint32_t* d_new_index, const uint32_t* d_base);
hipError_t launch_count_differences(hipStream_t st, const uint32_t* d_a, const uint32_t* d_b, uint32_t n, uint32_t* d_count);
hipError_t launch_membership(hipStream_t st, const uint8_t* d_parent, const uint32_t* d_cluster, uint32_t n, uint32_t parents, uint32_t clusters, uint8_t* d_flags);
hipError_t launch_scatter_spans(hipStream_t st, const uint32_t* d_perm0, const uint32_t* d_perm1, const bk_span* d_spans, uint32_t n_spans, uint32_t* d_out);
hipError_t launch_finish_spans(hipStream_t st, const uint32_t* d_perm0, const uint32_t* d_perm1, const bk_span* d_spans, uint32_t n_spans, uint32_t* d_leaf_of, uint32_t* d_parent_of /* may be null */,
                               const uint32_t* d_goffs /* null: no positions / sizes */, uint32_t* d_first_pos, uint32_t* d_sizes);
hipError_t launch_endpoint_rows(hipStream_t st, const uint64_t* d_keys, const uint32_t* d_goffs, uint32_t n, float* d_rows, uint64_t* d_weights);
hipError_t launch_exchange_children(hipStream_t st, uint32_t* d_perm0, uint32_t* d_perm1, const bk_span* d_nodes, const uint8_t* d_take, uint32_t n_nodes, uint32_t* d_staging, int dir);
hipError_t launch_gather_u32(hipStream_t st, const uint32_t* d_table, const uint32_t* d_index, uint32_t n, uint32_t* d_out);

} // namespace bu
