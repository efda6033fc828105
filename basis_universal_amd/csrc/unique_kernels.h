// unique_kernels.h -- host-side launch interface of unique_kernels.hip (internal to libbasisu_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bu {

// De-duplication of the selector training vectors of n resident ETC1S blocks (see unique_kernels.hip). All outputs are device arrays of n
// (offsets: n + 1) entries; the number of distinct vectors is left in device memory (*d_n_unique points into the workspace). Stream-ordered.
size_t unique_selector_vectors_workspace_bytes(uint32_t n_blocks);
hipError_t launch_unique_selector_vectors(hipStream_t st, const void* d_enc_blocks, const uint64_t* d_weights, uint32_t n_blocks, void* d_workspace,
                                          uint32_t* d_sorted_block_idx, uint32_t* d_unique_keys, uint64_t* d_unique_weights, uint32_t* d_group_offsets,
                                          uint32_t** d_n_unique);

// The same for the endpoint training vectors (frontend.cpp:825-866): one 48-bit key per ETC1S block (low / high block colour), distinct keys
// ascending = the reference's vec6F order, blocks of every distinct vector ascending. The weight of a vector is twice its block count.
size_t unique_endpoint_vectors_workspace_bytes(uint32_t n_blocks);
hipError_t launch_unique_endpoint_vectors(hipStream_t st, const void* d_etc1_blocks, uint32_t n_blocks, void* d_workspace, uint32_t* d_sorted_block_idx,
                                          uint64_t* d_unique_keys, uint32_t* d_group_offsets, uint32_t** d_n_unique);

} // namespace bu
