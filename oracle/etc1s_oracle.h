/* oracle/etc1s_oracle.h -- TEST INFRASTRUCTURE ONLY (see etc1s_oracle.c). */
#ifndef ORACLE_ETC1S_ORACLE_H
#define ORACLE_ETC1S_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* basis_etc_quality (encoder/basisu_etc.h:794-801) */
enum { ORC_QUALITY_FAST = 0, ORC_QUALITY_MEDIUM = 1, ORC_QUALITY_SLOW = 2, ORC_QUALITY_UBER = 3 };

uint32_t orc_color_distance(int perceptual, const uint8_t* a_rgb, const uint8_t* b_rgb);
uint32_t orc_hash_hsieh3(uint8_t r, uint8_t g, uint8_t b);

/* etc1_optimizer::init + compute with m_cluster_fit (etc.cpp:776-1278), ETC1S colour555 mode.
   rgba: n pixels, 4 bytes each. out_selectors (n bytes) may be NULL. Returns 1 on success. */
int orc_etc1_optimize(const uint8_t* rgba, uint32_t n, int quality, int perceptual,
                      uint8_t out_color5[3], uint32_t* out_inten, uint64_t* out_err, uint8_t* out_selectors);

/* basisu_frontend::init_etc1_images CPU branch (frontend.cpp:765-818): pixel blocks (64 B) -> etc_block (8 B). */
void orc_encode_etc1s_blocks(const uint8_t* pixel_blocks, uint32_t n_blocks, int comp_level, int perceptual, uint8_t* out_blocks);

/* etc_block::determine_selectors (etc.h:374-436) for ETC1S blocks: colour5+inten per block (4 B: r,g,b,inten). */
void orc_determine_selectors(const uint8_t* pixel_blocks, uint32_t n_blocks, const uint8_t* color5_inten, int perceptual, uint8_t* out_blocks);

/* basisu_frontend::generate_endpoint_codebook CPU branch (frontend.cpp:1482-1613).
   Clusters are CSR lists of training-vector indices (block*2+subblock). prev_params/valid may be NULL when step==0.
   params: per cluster {r,g,b,inten}; err: per cluster u64; valid: per cluster u8 (in/out for step>0). */
void orc_generate_endpoint_codebook(const uint8_t* pixel_blocks, uint32_t n_clusters, const uint32_t* offsets, const uint32_t* indices,
                                    int comp_level, int perceptual, uint32_t step, uint8_t* params, uint64_t* err, uint8_t* valid);

/* basisu_frontend::refine_endpoint_clusterization CPU branch (frontend.cpp:1772-1917): per-block argmin.
   cand_offsets/cand_indices: CSR of candidate cluster lists per parent (n_parents+1 offsets), block_parent per block;
   pass n_parents==0 for the non-hierarchical case (all clusters in index order). */
void orc_refine_endpoint_clusterization(const uint8_t* pixel_blocks, uint32_t n_blocks, const uint32_t* block_cluster,
                                        const uint8_t* cluster_params, uint32_t n_clusters,
                                        uint32_t n_parents, const uint32_t* cand_offsets, const uint32_t* cand_indices, const uint8_t* block_parent,
                                        int perceptual, uint32_t* out_best_cluster);

/* basisu_frontend::create_optimized_selector_codebook (frontend.cpp:2259-2354). blocks: m_encoded_blocks (8 B each). */
void orc_create_optimized_selector_codebook(const uint8_t* pixel_blocks, const uint8_t* encoded_blocks, uint32_t n_clusters,
                                            const uint32_t* offsets, const uint32_t* block_indices, int perceptual,
                                            uint8_t* inout_selector_blocks /* 8 B per cluster; empty clusters untouched */);

/* basisu_frontend::find_optimal_selector_clusters_for_each_block CPU branch (frontend.cpp:2534-2706), levels >= 1.
   selector_blocks: m_optimized_cluster_selectors (8 B each). Candidate CSR as for refine. chunk = job size (2048 in the reference;
   it bounds the "same pixels as previous block" shortcut). Rewrites encoded_blocks' selector bits in place. */
void orc_find_optimal_selector_clusters(const uint8_t* pixel_blocks, uint8_t* encoded_blocks, uint32_t n_blocks,
                                        const uint8_t* selector_blocks, uint32_t n_selectors,
                                        uint32_t n_parents, const uint32_t* cand_offsets, const uint32_t* cand_indices, const uint8_t* block_parent,
                                        int perceptual, uint32_t chunk, uint32_t* out_block_selector_cluster);

/* init_endpoint_training_vectors (frontend.cpp:825-866): per block 6 floats (low rgb, high rgb)/255. */
void orc_endpoint_training_vectors(const uint8_t* etc1s_blocks, uint32_t n_blocks, float* out6);
/* generate_selector_clusters training part (frontend.cpp:2155-2183): per block 16 floats + u64 weight. */
void orc_selector_training_vectors(const uint8_t* encoded_blocks, uint32_t n_blocks, int perceptual, float* out16, uint64_t* out_weight);

#ifdef __cplusplus
}
#endif
#endif
