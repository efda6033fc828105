// etc1s_kernels.h -- host-side launch interface of etc1s_kernels.hip (internal to libbasisu_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace bu {

enum { BU_Q_FAST = 0, BU_Q_MEDIUM = 1, BU_Q_SLOW = 2, BU_Q_UBER = 3 }; // basis_etc_quality, basisu_etc.h:794-801

hipError_t upload_etc1s_tables(int device);

hipError_t launch_encode_etc1s_blocks(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, int quality, bool perceptual, void* d_out);
hipError_t launch_endpoint_training_vectors(hipStream_t st, const void* d_etc_blocks, uint32_t n_blocks, float* d_out6);
hipError_t launch_selector_training_vectors(hipStream_t st, const void* d_enc_blocks, uint32_t n_blocks, bool perceptual, float* d_out16, uint64_t* d_w);
hipError_t launch_generate_endpoint_codebook(hipStream_t st, const void* d_pixel_blocks, uint32_t n_clusters, const uint32_t* d_order,
                                             const uint32_t* d_offsets, const uint32_t* d_indices, int quality, bool perceptual, uint32_t step,
                                             uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid);
// LARGE clusters (etc1s_codebook_wide.inc): many workgroups per cluster, a launch per pass over the texels of all large clusters of the call. h_cluster / h_first /
// h_subblocks: per large cluster its index, the position of its first member in d_indices and its member (sub-block) count. prepare() lays out the workspace
// [states | chunk table | scratch] and fills `image` with the bytes of its first two parts (the caller uploads them to d_work before launch_codebook_wide).
struct cb_wide_layout { size_t states_at, chunks_at, sum_at, total, image_bytes; uint32_t n_big, n_chunks; };
cb_wide_layout codebook_wide_prepare(const uint32_t* h_cluster, const uint32_t* h_first, const uint32_t* h_subblocks, uint32_t n_big, std::vector<unsigned char>& image);
hipError_t launch_codebook_wide_means(hipStream_t st, const void* d_pixel_blocks, const uint32_t* d_indices, void* d_work, const cb_wide_layout& L, float* d_out /* 3 per large cluster */);
hipError_t launch_codebook_wide(hipStream_t st, const void* d_pixel_blocks, const uint32_t* d_indices, void* d_work, const cb_wide_layout& L, int quality, bool perceptual, bool forced,
                                uint32_t step, const void* d_enc_blocks, uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid, uint64_t* d_cur_err);
hipError_t launch_refit_endpoints_given_selectors(hipStream_t st, const void* d_pixel_blocks, const void* d_enc_blocks, uint32_t n_clusters, const uint32_t* d_order,
                                                  const uint32_t* d_offsets, const uint32_t* d_indices, int quality, bool perceptual, uint8_t* d_params, uint64_t* d_err,
                                                  uint8_t* d_valid, uint64_t* d_cur_err);
hipError_t launch_subblock_errors(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, const uint32_t* d_block_cluster, const uint8_t* d_cluster_params,
                                  bool perceptual, uint64_t* d_out);
hipError_t launch_backend_block_errors(hipStream_t st, const void* d_pixel_blocks, const void* d_etc_blocks, const uint32_t* d_block_cluster, const uint8_t* d_cluster_params,
                                       uint32_t first_block, uint32_t nbx, uint32_t nby, uint32_t n_clusters, bool perceptual, bool with_neighbours, uint32_t* d_own,
                                       uint32_t* d_neighbour);
hipError_t launch_refine_endpoint_clusterization(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, const uint32_t* d_block_cluster,
                                                 const uint8_t* d_cluster_params, uint32_t n_clusters, uint32_t n_parents, const uint32_t* d_cand_offsets,
                                                 const uint32_t* d_cand_indices, const uint8_t* d_block_parent, bool perceptual, uint32_t* d_out_best,
                                                 void* d_work /* refine_workspace_bytes(), or null: unsorted lists */);
size_t refine_workspace_bytes(uint32_t n_clusters, uint32_t n_parents);   // 0: lists too long for the sorted form
hipError_t launch_determine_selectors(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, const uint8_t* d_color5_inten,
                                      const uint32_t* d_block_cluster, bool perceptual, void* d_out);
// d_workspace: create_optimized_selector_codebook_workspace_bytes()
size_t create_optimized_selector_codebook_workspace_bytes(uint32_t n_clusters);
hipError_t launch_create_optimized_selector_codebook(hipStream_t st, const void* d_pixel_blocks, const void* d_enc_blocks, uint32_t n_clusters,
                                                     const uint32_t* d_offsets, const uint32_t* d_block_indices, bool perceptual,
                                                     void* d_workspace, void* d_selector_blocks);
hipError_t launch_find_optimal_selector_clusters(hipStream_t st, const void* d_pixel_blocks, void* d_enc_blocks, uint32_t n_blocks,
                                                 const void* d_selector_blocks, uint32_t n_selectors, uint32_t n_parents, const uint32_t* d_cand_offsets,
                                                 const uint32_t* d_cand_indices, const uint8_t* d_block_parent, bool perceptual, uint32_t chunk,
                                                 uint32_t* d_scratch_idx, uint32_t* d_out_idx, uint32_t* d_cand_words /* scratch for the candidates' selector words, or NULL */, size_t cand_words_capacity /* entries */);

hipError_t launch_extract_blocks(hipStream_t st, const void* d_rgba, uint32_t width, uint32_t height, uint32_t pitch_bytes, void* d_out_blocks);

} // namespace bu
