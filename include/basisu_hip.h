/* include/basisu_hip.h -- C ABI of libbasisu_hip.so, the MI355X (gfx950) replacement for the reference's accelerator seam.
 *
 * The reference's seam is encoder/basisu_opencl.h (namespace basisu, ten entry points, opaque per-thread context).
 * Section 1 below exports the same ten operations with the same POD layouts, ownership and error conventions, so that a
 * reference maintainer can bind them from basisu_opencl.cpp's call sites one-for-one (see INTEGRATION.md for the shim).
 * Section 2 is the device-resident layer the ETC1S frontend actually runs on (stream-ordered, device pointers, no
 * implicit copies); it also covers the stages the reference never offloaded (frontend.cpp CPU-only branches).
 *
 * Conventions (same as basisu_opencl.h):
 *   - every int-returning function returns 1 on success and 0 on failure; nothing throws, nothing aborts;
 *     on failure the outputs are untouched or unspecified and the caller falls back / reports (frontend.cpp:757-762);
 *   - host pointers are caller-owned and only used during the call; the library owns all device memory it allocates;
 *   - one context per calling thread; calls on one context are serialised on its HIP stream;
 *   - bu_hip_init() is process-global and not re-entrant (opencl.cpp:730-736).
 */
#ifndef BASISU_HIP_H
#define BASISU_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define BU_HIP_API __attribute__((visibility("default")))
#else
#define BU_HIP_API
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Section 1 -- drop-in for encoder/basisu_opencl.h
 * ------------------------------------------------------------------------------------------------------------------ */

typedef struct bu_hip_context bu_hip_context; /* = basisu::opencl_context (basisu_opencl.h:28-31) */

#pragma pack(push, 1)
typedef struct { uint8_t m_pixels[16][4]; } bu_pixel_block;          /* = cl_pixel_block, basisu_opencl.h:37-40: [y*4+x] RGBA */
typedef struct { uint8_t m_bytes[8]; } bu_etc_block;                 /* = basisu::etc_block, basisu_etc.h:91-103 (big-endian u64) */
typedef struct { uint8_t r, g, b, a; } bu_color_rgba;                /* = basisu::color_rgba, basisu_enc.h:893-907 */
typedef struct { uint64_t m_total_pixels, m_first_pixel_index; } bu_pixel_cluster;      /* = cl_pixel_cluster, basisu_opencl.h:53-57 */
typedef struct { uint16_t m_first_cluster_ofs, m_num_clusters, m_cur_cluster_index; uint8_t m_cur_cluster_etc_inten; } bu_block_info;   /* = cl_block_info_struct :73-79 */
typedef struct { bu_color_rgba m_unscaled_color; uint8_t m_etc_inten; uint16_t m_cluster_index; } bu_endpoint_cluster;               /* = cl_endpoint_cluster_struct :81-86 */
typedef struct { uint32_t m_packed_selectors; } bu_fosc_selector;    /* = fosc_selector_struct :101-104 */
typedef struct { bu_color_rgba m_etc_color5_inten; uint32_t m_first_selector, m_num_selectors; } bu_fosc_block;                       /* = fosc_block_struct :106-111 */
#pragma pack(pop)

#define BU_HIP_ENCODE_ETC1S_MAX_PERMS 165u /* = OPENCL_ENCODE_ETC1S_MAX_PERMS, basisu_opencl.h:44 */

BU_HIP_API int  bu_hip_init(int force_serialization);     /* opencl_init            basisu_opencl.h:24 */
BU_HIP_API void bu_hip_deinit(void);                      /* opencl_deinit          :25 */
BU_HIP_API int  bu_hip_is_available(void);                /* opencl_is_available    :26 */
BU_HIP_API bu_hip_context* bu_hip_create_context(void);   /* opencl_create_context  :33 (current HIP device) */
BU_HIP_API void bu_hip_destroy_context(bu_hip_context*);  /* opencl_destroy_context :34 */

/* opencl_set_pixel_blocks :46 -- uploads once; the blocks stay resident in the context */
BU_HIP_API int bu_hip_set_pixel_blocks(bu_hip_context*, size_t total_blocks, const bu_pixel_block* pixel_blocks);
/* opencl_encode_etc1s_blocks :48 -- total_perms in {4,16,64,165} selects the etc1_optimizer quality (fast/medium/slow/uber) */
BU_HIP_API int bu_hip_encode_etc1s_blocks(bu_hip_context*, bu_etc_block* output_blocks, int perceptual, uint32_t total_perms);
/* opencl_encode_etc1s_pixel_clusters :60-68 -- weighted, de-duplicated pixel lists */
BU_HIP_API int bu_hip_encode_etc1s_pixel_clusters(bu_hip_context*, bu_etc_block* output_blocks, uint32_t total_clusters,
    const bu_pixel_cluster* clusters, uint64_t total_pixels, const bu_color_rgba* pixels, const uint32_t* pixel_weights,
    int perceptual, uint32_t total_perms);
/* opencl_refine_endpoint_clusterization :89-96 */
BU_HIP_API int bu_hip_refine_endpoint_clusterization(bu_hip_context*, const bu_block_info* pixel_block_info, uint32_t total_clusters,
    const bu_endpoint_cluster* cluster_info, const uint32_t* sorted_block_indices, uint32_t* output_cluster_indices, int perceptual);
/* opencl_find_optimal_selector_clusters_for_each_block :120-127 */
BU_HIP_API int bu_hip_find_optimal_selector_clusters_for_each_block(bu_hip_context*, const bu_fosc_block* input_block_info,
    uint32_t total_input_selectors, const bu_fosc_selector* input_selectors, const uint32_t* selector_cluster_indices,
    uint32_t* output_selector_cluster_indices, int perceptual);
/* opencl_determine_selectors :137-141 */
BU_HIP_API int bu_hip_determine_selectors(bu_hip_context*, const bu_color_rgba* input_etc_color5_and_inten, bu_etc_block* output_blocks, int perceptual);

/* NEW seam (the reference has none): basisu::encode_uastc (encoder/basisu_uastc_enc.h:69, uastc_enc.cpp:3126) for every
 * resident pixel block, replacing the per-block job loop of basis_compressor::encode_slices_to_uastc_4x4_ldr (comp.cpp:2020-2033).
 * `flags` are the reference's pack flags verbatim: cPackUASTCLevel* in the low bits, cPackUASTCFavor*, cPackUASTCETC1* (uastc_enc.h:24-66).
 * Output blocks are bit-identical to the reference's for every level and flag. */
typedef struct { uint8_t m_bytes[16]; } bu_uastc_block;             /* = basist::uastc_block, transcoder/basisu_transcoder_uastc.h:198-205 */
BU_HIP_API int bu_hip_encode_uastc_blocks(bu_hip_context*, bu_uastc_block* output_blocks, uint32_t flags);

/* ------------------------------------------------------------------------------------------------------------------
 * Section 2 -- device-resident layer. All `d_` pointers are DEVICE pointers valid on the context's device; every call is
 * enqueued on the context's stream and returns without synchronising unless stated. `h_` pointers are host pointers.
 * ------------------------------------------------------------------------------------------------------------------ */

/* bu_hip_destroy_context PARKS a healthy context instead of tearing it down: its streams, workspaces and block pool (whatever it grew to -- up to 16 GiB of
 * cached device blocks) stay allocated and the next bu_hip_create_context* on that device gets it back warm (the reference's basis_parallel_compress creates and
 * destroys a context per image; a context's worth of hipMalloc / hipFree calls are device-wide synchronisations that stall every other image's stream). At most
 * BU_HIP_PARKED_CONTEXTS (default 16; 0 = destroy means destroy) are kept; bu_hip_deinit releases them all. A context whose streams report an error is never parked.
 * A process that shares the GPU with other allocators and wants the memory back at once sets BU_HIP_PARKED_CONTEXTS=0 or calls bu_hip_deinit. */
BU_HIP_API bu_hip_context* bu_hip_create_context_on(int device);
/* Objects that keep device memory of a context (a resident frontend, say) may ask to be told when it is being destroyed: fn(user) runs at the
 * start of bu_hip_destroy_context, while the context still works, so that they can let go of their buffers instead of freeing them through a
 * dead context later. bu_hip_cancel_on_destroy removes a registration (same fn and user). */
typedef void (*bu_hip_destroy_fn)(void* user);
BU_HIP_API int  bu_hip_on_destroy(bu_hip_context*, bu_hip_destroy_fn fn, void* user);
BU_HIP_API void bu_hip_cancel_on_destroy(bu_hip_context*, bu_hip_destroy_fn fn, void* user);
BU_HIP_API int   bu_hip_context_device(const bu_hip_context*);
/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the context's own; NULL restores it. */
BU_HIP_API int   bu_hip_set_stream(bu_hip_context*, void* hip_stream);
BU_HIP_API void* bu_hip_get_stream(bu_hip_context*);
BU_HIP_API int   bu_hip_sync(bu_hip_context*);
/* Tuning of the device paths that have more than one (all bit-identical: tests run every one of them). Each field has a measured default; the
 * environment variable named beside it overrides that default ONCE per process (read when the first context is created); bu_hip_set_tuning overrides both for one
 * context (trees created on it afterwards take the values; NULL restores the process defaults; parking a context restores them too). Versioned by size:
 * struct_bytes = sizeof(bu_hip_tuning) of the caller's header, fields a caller does not have keep their defaults. */
typedef struct bu_hip_tuning {
    uint32_t struct_bytes;
    uint32_t tsvq_wide_min;       /* BU_TSVQ_WIDE_MIN      8192   selector (packed) nodes of this many members and more take the many-workgroup passes; 0 (BU_TSVQ_WIDE=0) = never */
    uint32_t tsvq_wide6_min;      /* BU_TSVQ_WIDE6_MIN     8192   the same for endpoint (6-float) nodes; 0 (BU_TSVQ_WIDE6=0 or BU_TSVQ_WIDE=0) = never */
    uint32_t tsvq_wide_cov_min;   /* BU_TSVQ_WIDE_COV_MIN  98304  batches whose largest node is smaller run the covariance pass chained instead of through the maps */
    uint32_t tsvq_windows;        /* BU_TSVQ_WINDOWS       0      pre-composed 64-block windows in the walks: 0 = batches averaging >= 2048 blocks per node, 1 = always, 2 (env "0") = never */
    uint32_t tsvq_dense_min;      /* BU_TSVQ_DENSE_MIN     257    rounds of this many nodes take the 128-register (two workgroups per CU) build of the exact split kernel; 0 = never */
    uint32_t tsvq_zero_copy;      /* BU_TSVQ_ZEROCOPY      1      node / result records of a round in coherent pinned memory + a flag the host looks at; 0 = staged copies + synchronise */
    uint32_t tsvq_chained_only;   /* BU_TSVQ_CHAINED       0      1 = every float sum member by member in one workgroup per node (the slowest path; what the others are tested against) */
    uint32_t tsvq_poll;           /* BU_TSVQ_POLL          0      waiting for a round: 0 = spin when this is the process's only context, else yield / nap; 1 (spin) / 2 (yield) force it */
    uint32_t refine_unsorted;     /*                       0      1 = refine_endpoint_clusterization through the unsorted kernel (the one lists beyond 65,535 entries take anyway) */
    uint32_t debug;               /* BU_TSVQ_ROUNDS = 1 | BU_TSVQ_SERIAL = 2 | BU_TSVQ_STATS = 4: developer aids (round time line on stderr, one node per round, walk statistics) */
    uint32_t tsvq_deep_levels;    /* BU_TSVQ_DEEP          0      deep rounds: generations of descendants of a round's one-workgroup nodes split in the same round trip (bu_hip_tsvq_split_deep); 0 = none, <= 2. Measured SLOWER at 4096^2 (17.4-18.0 against 16.8-17.6 ms per step: the rounds it saves cost ~50 us each, the splits nobody pops cost device time in rounds that fill the chip), hence off */
    uint32_t uastc_walk_cus;      /* BU_UASTC_WALK_CUS     0      UASTC pipeline lanes: this many CUs (every (CUs / n)-th one) carry the strip walks of uastc_rdo and nothing else -- the lanes' other kernels are masked off them; 0 = no reservation */
    uint32_t codebook_wide_min;   /* BU_CODEBOOK_WIDE_MIN  32768  endpoint clusters of this many texels and more are fitted by many workgroups (a launch per pass over the texels of all of them) instead of one workgroup each; 0 = never */
} bu_hip_tuning;
BU_HIP_API void bu_hip_get_tuning(const bu_hip_context* /* NULL: the process defaults */, bu_hip_tuning* out, uint32_t struct_bytes);
BU_HIP_API int  bu_hip_set_tuning(bu_hip_context*, const bu_hip_tuning* /* NULL: back to the process defaults */);

/* Cooperative waiting. By default a call that needs a device result blocks its host thread (hipStreamSynchronize, or a look-loop on a zero-copy flag). A host that
 * drives SEVERAL contexts from ONE thread -- bu_frontend_pipeline_* (basisu_hip_frontend.h): every image in flight is a task with a stack of its own on the pipeline's
 * one driver thread -- installs a hook instead: wherever a call on this context would block, the stream is only queried, and fn(user) is called between the looks;
 * fn switches to another task and returns when it is this one's turn again. fn NULL removes the hook (parking a context removes it too). The reference has no
 * counterpart: its opencl_context calls all end in clFinish (opencl.cpp:972-976), one blocked host thread per image in flight (comp.cpp:5466-5559). */
typedef void (*bu_hip_wait_fn)(void* user);
BU_HIP_API int   bu_hip_set_wait_hook(bu_hip_context*, bu_hip_wait_fn fn, void* user);
BU_HIP_API const char* bu_hip_last_error(const bu_hip_context*); /* NULL context -> last global (init) error */

/* Per-kernel timing with HIP events recorded on the launch stream around every section-2 kernel. enable(1) resets the totals and times every region; enable(2) only
 * the regions that are ONE kernel launch each (per-block and per-cluster kernels) -- the regions of many launches (codebook builders' rounds, de-duplication, list
 * bookkeeping: names tsvq_*, unique_*, map_*, kmeans_*) go untimed, and the step runs as it does without instrumentation (an event between two kernels of a chain
 * costs the chain ~3 us, and a round's two kinds of nodes only share the device when nobody times them apart: 0.5-0.7 ms of a 17 ms step, tools/headline_ab.py).
 * read() synchronises the pending events and returns the number of distinct kernels (names are static strings). */
BU_HIP_API int      bu_hip_profile_enable(bu_hip_context*, int on);
BU_HIP_API uint32_t bu_hip_profile_read(bu_hip_context*, const char** names, double* total_ms, uint32_t* launches, uint32_t cap);

BU_HIP_API void* bu_hip_malloc(bu_hip_context*, size_t bytes);
BU_HIP_API void  bu_hip_free(bu_hip_context*, void* d_ptr);
BU_HIP_API int   bu_hip_memcpy_h2d(bu_hip_context*, void* d_dst, const void* h_src, size_t bytes); /* synchronises */
/* ... stream-ordered instead of blocking: the source has been copied out when the call returns (it may be released), the copy itself is ordered on the context's stream */
BU_HIP_API int bu_hip_memcpy_h2d_async(bu_hip_context*, void* d_dst, const void* h_src, size_t bytes);
BU_HIP_API int   bu_hip_memcpy_d2h(bu_hip_context*, void* h_dst, const void* d_src, size_t bytes); /* synchronises */
/* A download that nobody waits for yet: the bytes d_src holds once everything enqueued on the context's stream so far has run are brought to h_dst (pageable or pinned)
 * by the copy engine on a stream of their own, driven by a helper thread, while the caller goes on enqueueing work. bu_hip_download_wait blocks until they are there
 * (1 = arrived) and releases the handle; it must be called exactly once per handle, before h_dst is released and before anything overwrites d_src. NULL: not available
 * on this context now (a wait hook is installed, or a resource could not be had) -- use bu_hip_memcpy_d2h. */
/* Page-locked host memory (for callers that want their tiles taken by the copy engine as they are: bu_hip_k_upload_and_encode_etc1s_blocks, bu_frontend_init with host tiles). */
BU_HIP_API void* bu_hip_host_alloc(size_t bytes);
BU_HIP_API void  bu_hip_host_free(void* p);
typedef struct bu_hip_download bu_hip_download;
BU_HIP_API bu_hip_download* bu_hip_download_begin(bu_hip_context*, void* h_dst, const void* d_src, size_t bytes);
BU_HIP_API int   bu_hip_download_wait(bu_hip_download*);
BU_HIP_API int   bu_hip_memset(bu_hip_context*, void* d_dst, int value, size_t bytes);
BU_HIP_API int   bu_hip_memcpy_d2d(bu_hip_context*, void* d_dst, const void* d_src, size_t bytes); /* stream-ordered */

/* Adopt pixel blocks that are already resident (no copy, not owned). Counterpart of bu_hip_set_pixel_blocks. */
BU_HIP_API int   bu_hip_set_pixel_blocks_device(bu_hip_context*, size_t total_blocks, const void* d_pixel_blocks);
BU_HIP_API const void* bu_hip_get_pixel_blocks_device(const bu_hip_context*, size_t* total_blocks);

/* Input side: basis_compressor::extract_source_blocks (comp.cpp:3207-3268) on the device. d_rgba is a resident RGBA8 raster
 * (row pitch in bytes >= 4*width); writes ceil(w/4)*ceil(h/4) tiles in block-raster order with the right/bottom edges clamped like
 * image::extract_block_clamped. The result can be adopted with bu_hip_set_pixel_blocks_device. */
BU_HIP_API int bu_hip_k_extract_blocks(bu_hip_context*, const void* d_rgba, uint32_t width, uint32_t height, uint32_t pitch_bytes, void* d_out_pixel_blocks);

/* etc1_optimizer quality (basis_etc_quality, basisu_etc.h:794-801) */
enum { BU_ETC_QUALITY_FAST = 0, BU_ETC_QUALITY_MEDIUM = 1, BU_ETC_QUALITY_SLOW = 2, BU_ETC_QUALITY_UBER = 3 };

/* a6  basisu_frontend::init_etc1_images (frontend.cpp:733-823): per block etc1_optimizer, n = 16. out: n_blocks x 8 B. */
BU_HIP_API int bu_hip_k_encode_etc1s_blocks(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_blocks, int quality, int perceptual, void* d_out_etc_blocks);
/*     The same over tiles that are still in HOST memory, as one pipeline: the tiles go to d_pixel_blocks in 4 MiB pieces on the context's side stream (page-locked source
 *     memory as it is, pageable memory through a pinned ring filled by BU_UPLOAD_THREADS (default 4) helper threads) and piece i's kernel starts behind piece i's copy, so
 *     the transfer hides behind the kernel (SURVEY 8d figure (i): the hot path with the H2D inside). On return h_pixel_blocks may be released; the device side is ordered
 *     on the context's stream. Replaces opencl_set_pixel_blocks + opencl_encode_etc1s_blocks (encoder/basisu_opencl.h:112-116) for a caller that wants both. */
BU_HIP_API int bu_hip_k_upload_and_encode_etc1s_blocks(bu_hip_context*, void* d_pixel_blocks, const void* h_pixel_blocks, uint32_t n_blocks, int quality, int perceptual,
    void* d_out_etc_blocks);
/* a7  init_endpoint_training_vectors (frontend.cpp:825-866): per block 6 floats (low rgb, high rgb)/255. */
BU_HIP_API int bu_hip_k_endpoint_training_vectors(bu_hip_context*, const void* d_etc_blocks, uint32_t n_blocks, float* d_out_vec6);
/* a9  generate_endpoint_codebook (frontend.cpp:1214-1617), CPU semantics incl. step > 0. Clusters are CSR lists of
 *     training-vector indices (block*2+subblock); h_offsets is a HOST array of n_clusters+1 entries (used to schedule
 *     large clusters first), d_offsets its device copy. d_params: n_clusters x {r5,g5,b5,inten}; d_err: u64; d_valid: u8. */
BU_HIP_API int bu_hip_k_generate_endpoint_codebook(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_clusters,
    const uint32_t* h_offsets, const uint32_t* d_offsets, const uint32_t* d_indices, int quality, int perceptual, uint32_t step,
    uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid);
/*     Clusters of bu_hip_tuning::codebook_wide_min texels and more (a sky, a flat wall, a constant alpha plane: 10^5-10^7 texels in ONE cluster) are fitted by many
 *     workgroups each -- a launch per pass over the texels of all of them -- with the same results, the order-dependent float colour mean included (H4).
 *     Test hook: that mean (etc.cpp:1034-1041: running float sum in texel order / count), evaluated the many-workgroup way for EVERY cluster given; h_out: 3 floats each. */
BU_HIP_API int bu_hip_k_cluster_colour_means(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_clusters, const uint32_t* h_offsets, const uint32_t* d_indices, float* h_out);
/* a15 refine_block_endpoints_given_selectors (frontend.cpp:2718-2976, ETC1S levels 4-6): per endpoint cluster, the uber-quality
 *     etc1_optimizer with the selectors of d_encoded_blocks held fixed (m_pForce_selectors). Lists are CSR over training-vector indices
 *     like a9 and MAY contain duplicates (the reference's m_subblocks lists grow from iteration to iteration). Outputs per cluster:
 *     the refitted {r5,g5,b5,inten}, its error, a validity flag, and the CURRENT error of the listed sub-blocks under their blocks'
 *     present colours -- the caller applies the refit only where new error < current error (:2822). */
BU_HIP_API int bu_hip_k_refit_endpoints_given_selectors(bu_hip_context*, const void* d_pixel_blocks, const void* d_encoded_blocks, uint32_t n_clusters,
    const uint32_t* h_offsets, const uint32_t* d_offsets, const uint32_t* d_indices, int perceptual,
    uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid, uint64_t* d_current_err);
/* f2  the same fit at a chosen optimizer quality (BU_ETC_QUALITY_SLOW or _UBER), for basisu_frontend::reoptimize_remapped_endpoints
 *     (frontend.cpp:2996-3104: slow below compression level 6), the backend's call back into the frontend. */
BU_HIP_API int bu_hip_k_refit_endpoints_given_selectors_q(bu_hip_context*, const void* d_pixel_blocks, const void* d_encoded_blocks, uint32_t n_clusters,
    const uint32_t* h_offsets, const uint32_t* d_offsets, const uint32_t* d_indices, int quality, int perceptual,
    uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid, uint64_t* d_current_err);
/* a15 compute_endpoint_subblock_error_vec (frontend.cpp:1006-1091): u64 error of every training vector (block*2+subblock) under the
 *     endpoints of its block's cluster; feeds introduce_new_endpoint_clusters. d_out_err: 2*n_blocks entries. */
BU_HIP_API int bu_hip_k_subblock_errors(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_blocks, const uint32_t* d_block_cluster,
    const uint8_t* d_cluster_params, int perceptual, uint64_t* d_out_err);
/* The stateless part of basisu_backend::create_encoder_blocks (encoder/basisu_backend.cpp:406-617) for one slice of num_blocks_x x num_blocks_y blocks starting at first_block:
 * d_own_err[b] = the block's error as encoded (cur_err of :507 / :841); with_neighbours: d_neighbour_err[b * 3 + p] = its error under the endpoints of its left / upper / upper-left
 * neighbour (g_endpoint_preds order) with its own selectors, ~0u where the walk does not ask for it (edge, a neighbour already shares its endpoints, zero error). The decisions that chain
 * from block to block stay on the host (include/basisu_hip_backend.h). d_cluster_params: r5, g5, b5, intensity table per endpoint cluster. */
BU_HIP_API int bu_hip_k_backend_block_errors(bu_hip_context*, const void* d_pixel_blocks, const void* d_etc_blocks, const uint32_t* d_block_cluster, const uint8_t* d_cluster_params,
    uint32_t first_block, uint32_t num_blocks_x, uint32_t num_blocks_y, uint32_t n_clusters, int perceptual, int with_neighbours, uint32_t* d_own_err, uint32_t* d_neighbour_err);
/* a9 for one of `parts` equal shares of the clusters (multi-GPU): the clusters at positions part, part + parts, ... of the
 *     size-descending order. Entries of clusters outside the share are neither read nor written. */
BU_HIP_API int bu_hip_k_generate_endpoint_codebook_part(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_clusters,
    const uint32_t* h_offsets, const uint32_t* d_offsets, const uint32_t* d_indices, int quality, int perceptual, uint32_t step,
    uint8_t* d_params, uint64_t* d_err, uint8_t* d_valid, uint32_t part, uint32_t parts);
/* a10 refine_endpoint_clusterization (frontend.cpp:1648-1917) for BOTH hierarchical (n_parents > 0: candidates are
 *     d_cand_indices[d_cand_offsets[p] .. d_cand_offsets[p+1]) for the block's parent p) and flat codebooks (n_parents == 0). */
BU_HIP_API int bu_hip_k_refine_endpoint_clusterization(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_blocks,
    const uint32_t* d_block_cluster, const uint8_t* d_cluster_params, uint32_t n_clusters, uint32_t n_parents,
    const uint32_t* d_cand_offsets, const uint32_t* d_cand_indices, const uint8_t* d_block_parent, int perceptual, uint32_t* d_out_best_cluster);
/* a11 create_initial_packed_texture (frontend.cpp:2014-2096): d_block_cluster may be NULL, then d_color5_inten is per block. */
/* f4  One separable resampling step of mip generation on a resident RGBA8 raster (basis_compressor::generate_mipmaps -> image_resample ->
 *     Resampler, comp.cpp:2146-2230, enc.cpp:1022-1180, resampler.cpp:343-435). The contributor lists of both axes (CSR: first[n+1],
 *     pixel, weight), the pass order and the two value tables (256 floats, 8192 bytes) are HOST arrays computed by the caller
 *     (libbasisu_frontend.so: bu_generate_mipmap_level does that exactly as the reference's host code); the device applies them with the
 *     reference's float operations in the reference's order. d_src: src_w x src_h, d_dst: dst_w x dst_h, tightly packed. num_comps 3
 *     or 4 (3: destination alpha = 255). */
BU_HIP_API int bu_hip_k_resample_rgba8(bu_hip_context*, const void* d_src, uint32_t src_w, uint32_t src_h, void* d_dst, uint32_t dst_w, uint32_t dst_h,
    const uint32_t* x_first, const uint16_t* x_pixel, const float* x_weight, const uint32_t* y_first, const uint16_t* y_pixel, const float* y_weight,
    int x_after_y, int srgb, const float* srgb_to_linear_256, const uint8_t* linear_to_srgb_8192, uint32_t num_comps);
BU_HIP_API int bu_hip_k_determine_selectors(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_blocks,
    const uint8_t* d_color5_inten, const uint32_t* d_block_cluster, int perceptual, void* d_out_etc_blocks);
/* a12 generate_selector_clusters training part (frontend.cpp:2155-2183): 16 floats + u64 weight per block (d_out_vec16 may be NULL). */
BU_HIP_API int bu_hip_k_selector_training_vectors(bu_hip_context*, const void* d_encoded_blocks, uint32_t n_blocks, int perceptual, float* d_out_vec16, uint64_t* d_out_weight);
/* a13 create_optimized_selector_codebook (frontend.cpp:2259-2354): CSR lists of block indices per selector cluster;
 *     rewrites the selector bytes of d_selector_blocks[cluster] (8 B each) for non-empty clusters. */
BU_HIP_API int bu_hip_k_create_optimized_selector_codebook(bu_hip_context*, const void* d_pixel_blocks, const void* d_encoded_blocks,
    uint32_t n_clusters, const uint32_t* d_offsets, const uint32_t* d_block_indices, int perceptual, void* d_selector_blocks);
/* a14 find_optimal_selector_clusters_for_each_block (frontend.cpp:2397-2715), hierarchical or flat; `chunk` reproduces the
 *     reference's "same pixels as the previous block of this job" shortcut (2048; 0 disables). Rewrites the selector bytes
 *     of d_encoded_blocks and writes the chosen cluster per block. */
BU_HIP_API int bu_hip_k_find_optimal_selector_clusters(bu_hip_context*, const void* d_pixel_blocks, void* d_encoded_blocks, uint32_t n_blocks,
    const void* d_selector_blocks, uint32_t n_selectors, uint32_t n_parents, const uint32_t* d_cand_offsets, const uint32_t* d_cand_indices,
    const uint8_t* d_block_parent, int perceptual, uint32_t chunk, uint32_t* d_out_block_selector_cluster);

/* a16-a19 encode_uastc over n_blocks resident tiles -> n_blocks x 16 B (device). Runs four kernels (classify, candidates per
 *     (block, mode job), score per (block, candidate), finish); the candidate workspace lives in the context and is reused. */
BU_HIP_API int bu_hip_k_encode_uastc_blocks(bu_hip_context*, const void* d_pixel_blocks, uint32_t n_blocks, uint32_t flags, void* d_out_uastc_blocks);
/* Bytes of context workspace that call needs (27 candidate slots x 76 B per block at level 2; 170 at level 4). */
BU_HIP_API size_t bu_hip_uastc_workspace_bytes(uint32_t n_blocks, uint32_t flags);

/* a20 (SURVEY.md 8a "next" row f1): basisu::uastc_rdo (encoder/basisu_uastc_enc.h:139, uastc_enc.cpp:4095-4163) in place over n_blocks
 * resident UASTC blocks and the pixel blocks they were encoded from. `params` mirrors uastc_rdo_params (uastc_enc.h:94-134) field for
 * field; `flags` are the pack flags the blocks were encoded with (the hints of modified blocks are recomputed with them);
 * `total_jobs` has the reference's meaning: 0/1 = one strip, otherwise strips of n_blocks / total_jobs blocks that do not see each
 * other (comp.cpp:2076-2078 passes min(4, threads)). Output is bit-identical to the reference called with the same total_jobs.
 * Synchronises the context's stream; `out_stats` (optional) receives {modified, refined, skipped, strips}. Fails (0) like the reference
 * when a block does not unpack. */
typedef struct bu_uastc_rdo_params {
    float m_lambda;
    float m_max_allowed_rms_increase_ratio;
    float m_skip_block_rms_thresh;
    float m_max_smooth_block_std_dev;
    float m_smooth_block_max_error_scale;
    uint32_t m_lz_dict_size;
    uint32_t m_lz_literal_cost;
    uint32_t m_endpoint_refinement;
} bu_uastc_rdo_params;
BU_HIP_API void bu_hip_uastc_rdo_default_params(bu_uastc_rdo_params* out);   /* uastc_rdo_params::clear(), uastc_enc.h:101-112 */
BU_HIP_API int bu_hip_k_uastc_rdo(bu_hip_context*, void* d_uastc_blocks, const void* d_pixel_blocks, uint32_t n_blocks, const bu_uastc_rdo_params* params,
                                  uint32_t flags, uint32_t total_jobs, uint32_t out_stats[4]);
/* host-buffer form over the context's resident pixel blocks (bu_hip_set_pixel_blocks): blocks in, blocks out */
BU_HIP_API int bu_hip_uastc_rdo(bu_hip_context*, bu_uastc_block* blocks, const bu_uastc_rdo_params* params, uint32_t flags, uint32_t total_jobs,
                                uint32_t out_stats[4]);

/* encode_uastc (+ uastc_rdo) over a STREAM of images with several in flight on one GPU -- what basis_parallel_compress (comp.cpp:5466-5559) does with a job pool for
 * BASELINE configs[4]'s batch. The RDO walk is one serial chain per strip and leaves most of the chip idle (96 strips of the Kodak batch: 96 of 256 CUs for ~20 ms);
 * the pipeline owns `lanes` private streams + workspaces and ENQUEUES every submission without a host synchronisation, so the next submission's encode kernels and the
 * previous one's hint refit run beside this one's walk. Bytes out = bu_hip_k_encode_uastc_blocks followed (rdo != NULL) by bu_hip_k_uastc_rdo; `flags` as there (callers
 * add cPackUASTCFavorSimplerModes for RDO themselves, as comp.cpp:2016-2018 does).
 *   create : lanes 1..8 (3 fills an MI355X with Kodak-sized batches); max_blocks / max_total_jobs size the workspaces once (they never grow afterwards)
 *   submit : d_px (n_blocks x 64 B) must stay valid and d_out (n_blocks x 16 B) unread until the ticket is waited for; inputs may still be in flight on the context's
 *            stream (the lane waits for them on the device). Blocks only when its lane's previous submission has not finished yet.
 *   wait   : ticket 0 = everything submitted so far; out_stats (may be NULL) = {modified, refined, skipped, strips} of that ticket, as bu_hip_k_uastc_rdo reports them */
typedef struct bu_uastc_pipeline bu_uastc_pipeline;
BU_HIP_API bu_uastc_pipeline* bu_hip_uastc_pipeline_create(bu_hip_context*, uint32_t lanes, uint32_t max_blocks, uint32_t flags, uint32_t max_total_jobs);
BU_HIP_API int  bu_hip_uastc_pipeline_submit(bu_uastc_pipeline*, const void* d_pixel_blocks, uint32_t n_blocks, void* d_out_uastc_blocks, const bu_uastc_rdo_params* rdo_or_null,
                                             uint32_t flags, uint32_t total_jobs, uint64_t* out_ticket);
BU_HIP_API int  bu_hip_uastc_pipeline_wait(bu_uastc_pipeline*, uint64_t ticket, uint32_t out_stats[4]);
BU_HIP_API void bu_hip_uastc_pipeline_destroy(bu_uastc_pipeline*);

/* a15 + the list handling inside a9 / a10 / a13 / a14: cluster bookkeeping on the device (basis_universal_amd/csrc/bookkeeping_kernels.hip).
 *     A clustering is two resident per-block arrays, cluster index and position inside the cluster's list; these calls turn distinct-vector level
 *     results into them, rebuild them after a reassignment, apply codebook renumberings to them and produce the CSR lists the per-cluster
 *     kernels read, so that no block-sized array crosses PCIe between the frontend's stages. Stream-ordered; nothing synchronises.
 *   map_blocks_from_groups: groups = the output of bu_hip_k_unique_*_vectors (d_group_offsets[u_total + 1], d_sorted_block_idx[n]); every block of
 *     distinct vector u gets cluster d_leaf_of_unique[u], position d_first_pos[u] + its rank inside the group, parent d_parent_of_unique[u]
 *     (d_first_pos / d_out_pos and d_parent_of_unique / d_out_parent may be NULL).
 *   map_rank_blocks (frontend.cpp:1921-1942, lists rebuilt in block order): d_out_sizes[k + 1] (last entry 0), d_out_offsets[k + 1] (exclusive
 *     sum), d_out_sorted_blocks[n] = block ids grouped by cluster, ascending inside a cluster, d_out_pos[n] (may be NULL) = rank of every block. */
BU_HIP_API int bu_hip_k_map_blocks_from_groups(bu_hip_context*, const uint32_t* d_group_offsets, const uint32_t* d_sorted_block_idx, uint32_t n_blocks, uint32_t u_total,
    const uint32_t* d_leaf_of_unique, const uint32_t* d_first_pos, const uint32_t* d_parent_of_unique, uint32_t* d_out_cluster, uint32_t* d_out_pos, uint8_t* d_out_parent);
BU_HIP_API int bu_hip_k_map_rank_blocks(bu_hip_context*, const uint32_t* d_block_cluster, uint32_t n_blocks, uint32_t n_clusters, uint32_t* d_out_sizes,
    uint32_t* d_out_offsets, uint32_t* d_out_sorted_blocks, uint32_t* d_out_pos);
/*   map_endpoint_csr: d_out_indices[d_offsets[cluster] + 2 * pos] = 2b, 2b + 1 (d_offsets in training-vector units: 2 per block). */
BU_HIP_API int bu_hip_k_map_endpoint_csr(bu_hip_context*, const uint32_t* d_block_cluster, const uint32_t* d_block_pos, uint32_t n_blocks, const uint32_t* d_offsets,
    uint32_t* d_out_indices);
/*   map_remap: cluster[b] = d_new_index[old], pos[b] += d_base[old] (d_block_pos and d_base may be NULL). */
BU_HIP_API int bu_hip_k_map_remap(bu_hip_context*, uint32_t* d_block_cluster, uint32_t* d_block_pos, uint32_t n_blocks, const uint32_t* d_new_index, const uint32_t* d_base);
/*   map_count_differences: *d_out_count = |{ i : a[i] != b[i] }| (device word). map_membership: d_out_flags[parent * n_clusters + cluster] = 1 for
 *     every (parent, cluster) pair that occurs (d_block_parent may be NULL = one parent). map_gather: out[i] = table[index[i]]. */
BU_HIP_API int bu_hip_k_map_count_differences(bu_hip_context*, const uint32_t* d_a, const uint32_t* d_b, uint32_t n, uint32_t* d_out_count);
BU_HIP_API int bu_hip_k_map_membership(bu_hip_context*, const uint8_t* d_block_parent, const uint32_t* d_block_cluster, uint32_t n_blocks, uint32_t n_parents,
    uint32_t n_clusters, uint8_t* d_out_flags);
BU_HIP_API int bu_hip_k_map_gather(bu_hip_context*, const uint32_t* d_table, const uint32_t* d_index, uint32_t n, uint32_t* d_out);

/* f3  Codebook builder FAST MODE (basis_universal_amd/csrc/kmeans_kernels.hip): weighted k-means over the distinct training
 *     vectors with the assignment step on the matrix cores, instead of the TSVQ. NOT bit-identical to the reference (different, deterministic
 *     codebooks); off unless asked for (bu_frontend_set_fast_codebooks), held to the reference's own size / PSNR tolerances by the tests.
 *     kind 0: d_keys = uint32 packed selector vectors + d_weights; kind 1: d_keys = uint64 endpoint colour keys, weights = 2 x group sizes
 *     (d_group_offsets[n + 1]). Writes, per distinct vector, its cluster (empty clusters removed, index order kept) and -- n_parents > 0 -- the
 *     parent group of that cluster (k-means over the centroids); returns the counts. Synchronises. */
BU_HIP_API int bu_hip_kmeans_codebook(bu_hip_context*, int kind, const void* d_keys, const uint64_t* d_weights, const uint32_t* d_group_offsets, uint32_t n_vectors,
    uint32_t max_clusters, uint32_t n_parents, uint32_t iterations, uint32_t* d_out_cluster_of_vector, uint32_t* d_out_parent_of_vector, uint32_t* out_clusters,
    uint32_t* out_parents);

/* a8  tree_vector_quant (encoder/basisu_enc.h:1546-2078): the order-dependent TSVQ tree build, split by split, bit-exact.
 *     The host keeps the tree, the variance priority queue and the split order (enc.h:1616-1660); the device executes batches of
 *     independent node splits (split_node, enc.h:1737-1800) on the resident training set. Rows must be the DISTINCT training
 *     vectors in ascending order (what generate_hierarchical_codebook_threaded's std::map yields, enc.h:2233-2290), dim 6 or 16.
 *     Node member lists live in two device index buffers: the root is {buf 0, start 0, count n}; a split of {buf, start, count}
 *     leaves its children at {buf^1, start, l_count} and {buf^1, start+l_count, r_count}. Member lists are ascending. */
typedef struct bu_tsvq bu_tsvq;
/* A node's member list is the span [start, start + count) of index buffer `buf` (0 .. BU_TSVQ_BUFFERS - 1); a split leaves the children's lists (left first) over the
 * same span of buffer (buf + 1) % BU_TSVQ_BUFFERS. */
#define BU_TSVQ_BUFFERS 4u
typedef struct { float origin[16]; uint64_t weight; float var; uint32_t pad; } bu_tsvq_root;      /* prepare_root, enc.h:1708-1735 */
typedef struct { uint32_t buf, start, count, pad; uint64_t weight; float origin[16]; } bu_tsvq_node;
typedef struct { uint32_t ok, l_count, r_count, pad; uint64_t l_weight, r_weight; float l_var, r_var; float l_centroid[16], r_centroid[16]; } bu_tsvq_split;
BU_HIP_API bu_tsvq* bu_hip_tsvq_create(bu_hip_context*, uint32_t dim, const float* h_rows, const uint64_t* h_weights, uint32_t n, bu_tsvq_root* h_out_root);
/* dim 16 with every component in {0,1,2,3} (ETC1S selector vectors): one dword per vector, component 0 in the top two bits. */
BU_HIP_API bu_tsvq* bu_hip_tsvq_create_packed16(bu_hip_context*, const uint32_t* h_keys, const uint64_t* h_weights, uint32_t n, bu_tsvq_root* h_out_root);
/* the same from device arrays (e.g. the outputs of bu_hip_k_unique_selector_vectors), copied on the context's stream */
BU_HIP_API bu_tsvq* bu_hip_tsvq_create_packed16_device(bu_hip_context*, const uint32_t* d_keys, const uint64_t* d_weights, uint32_t n, bu_tsvq_root* h_out_root);
/* dim 6 from the outputs of bu_hip_k_unique_endpoint_vectors: the rows (six floats = key byte * (1 / 255), frontend.cpp:846-851) and the weights (2 per block of the
 * vector's group) are made on the device; nothing is downloaded or uploaded */
BU_HIP_API bu_tsvq* bu_hip_tsvq_create_endpoint_device(bu_hip_context*, const uint64_t* d_unique_keys, const uint32_t* d_group_offsets, uint32_t n, bu_tsvq_root* h_out_root);
/* a12 + the de-duplication in front of the selector TSVQ (frontend.cpp:2140-2189; std::map<vec16F, weight> of
 * generate_hierarchical_codebook_threaded, enc.h:2218-2290) for n resident ETC1S blocks and their u64 training weights
 * (bu_hip_k_selector_training_vectors): distinct selector vectors as packed keys in ascending order = the map's order, their summed
 * weights, and the blocks of every distinct vector (ascending) as d_sorted_block_idx[d_group_offsets[u] .. d_group_offsets[u+1]).
 * All outputs are device arrays of n_blocks entries (offsets: n_blocks + 1). Integer work: exact and order independent. Synchronises. */
/* a7 + its de-duplication (frontend.cpp:825-866, enc.h:2218-2290) for n resident ETC1S blocks: distinct (low rgb, high rgb) block-colour vectors as
 * 48-bit keys (low r,g,b in bits 47..24, high r,g,b in bits 23..0; divide the bytes by 255 for the reference's vec6F) in ascending order, and the
 * blocks of every distinct vector (ascending) as d_sorted_block_idx[d_group_offsets[u] .. d_group_offsets[u+1]). Each block stands for its two
 * sub-block training vectors (ids 2b and 2b+1, weight 1 each). Outputs: device arrays of n_blocks (offsets: n_blocks + 1) entries. Synchronises. */
BU_HIP_API int bu_hip_k_unique_endpoint_vectors(bu_hip_context*, const void* d_etc1_blocks, uint32_t n_blocks, uint32_t* d_sorted_block_idx, uint64_t* d_unique_keys,
                                                uint32_t* d_group_offsets, uint32_t* out_unique);
BU_HIP_API int bu_hip_k_unique_selector_vectors(bu_hip_context*, const void* d_enc_blocks, const uint64_t* d_weights, uint32_t n_blocks, uint32_t* d_sorted_block_idx,
                                                uint32_t* d_unique_keys, uint64_t* d_unique_weights, uint32_t* d_group_offsets, uint32_t* out_unique);
BU_HIP_API int  bu_hip_tsvq_split(bu_hip_context*, bu_tsvq*, const bu_tsvq_node* h_nodes, uint32_t n_nodes, bu_tsvq_split* h_out); /* synchronises */
/* A DEEP round: the same, and in the same round trip the splits of the children, grandchildren, ... (`levels` generations, <= BU_TSVQ_BUFFERS - 2) of every node of the batch that went
 * through the one-workgroup kernel -- a split is a pure function of its node, so the caller's replay of the reference's variance queue (enc.h:1636-1655) finds them
 * when it gets there instead of asking for another round; what it never reaches is dropped. Each generation's node records are made ON THE DEVICE from the results of
 * the one before. A descendant is attempted when its parent's split succeeded, it has more than one member, and its variance (with the reference's 1e-4 for a
 * non-positive variance, enc.h:1766-1792) is positive and not below the float whose bits the caller put into h_nodes[i].pad (0 = no floor): a bound on the
 * speculation, never on the result. h_deep: (2^(levels+1) - 2) * n_nodes records; generation g = 1..levels of batch node i, path p (the sides taken, 0 = left, the
 * first step in the top bit of g bits) at h_deep[n_nodes * (2^g - 2) + i * 2^g + p]; ok == 3 there = not attempted. Synchronises. */
BU_HIP_API int  bu_hip_tsvq_split_deep(bu_hip_context*, bu_tsvq*, const bu_tsvq_node* h_nodes, uint32_t n_nodes, bu_tsvq_split* h_out, uint32_t levels, bu_tsvq_split* h_deep);
/* prepare_root (enc.h:1708-1735) of n_nodes member spans (buf / start / count of each record; weight and origin are not read): the root records of the
 * independent trees the reference's multi-threaded codebook build runs over the leaves of its first tree (enc.h:2137-2152). Synchronises. */
BU_HIP_API int  bu_hip_tsvq_roots(bu_hip_context*, bu_tsvq*, const bu_tsvq_node* h_nodes, uint32_t n_nodes, bu_tsvq_root* h_out);
BU_HIP_API int  bu_hip_tsvq_read_members(bu_hip_context*, bu_tsvq*, uint32_t buf, uint32_t start, uint32_t count, uint32_t* h_out);
/* Leaves (or cut nodes) as spans of the member buffers -> d_out[vector] = value for every member of every span: the leaf / parent index of
 * every distinct vector without bringing the member lists to the host. h_spans: n_spans records. Stream-ordered after the splits. */
typedef struct { uint32_t buf, start, count, value; } bu_tsvq_span;
BU_HIP_API int  bu_hip_tsvq_scatter_spans(bu_hip_context*, bu_tsvq*, const bu_tsvq_span* h_spans, uint32_t n_spans, uint32_t* d_out);
/* A finished tree in one pass: span i is leaf i and `value` its parent (cut) index -> d_leaf_of[vector], d_parent_of[vector] (may be NULL); with d_group_offsets (the
 * vectors' groups of blocks, bu_hip_k_unique_endpoint_vectors) also d_first_pos[vector] = where the vector's blocks start inside its leaf's block list (the blocks of the
 * members in front of it, in list order) and d_sizes[leaf] = the length of that list. Synchronises. */
BU_HIP_API int  bu_hip_tsvq_finish_spans(bu_hip_context*, bu_tsvq*, const bu_tsvq_span* h_spans, uint32_t n_spans, uint32_t* d_leaf_of, uint32_t* d_parent_of,
                                         const uint32_t* d_group_offsets, uint32_t* d_first_pos, uint32_t* d_sizes);
/* Multi-GPU (nodes of one round split by different ranks): the child member lists and result records of a batch laid end to end in a staging buffer the
 * host application sum-reduces across ranks (all_reduce_u64 of bu_comm): entries of nodes a rank did not split are zero, so the sum is the union.
 *   exchange_pack:   children of the nodes with h_mine[i] != 0 (from the member buffers) and their h_records into the staging buffer, zero elsewhere;
 *                    returns the device pointer and its size in u64 words.   [host: all_reduce_u64(*d_staging, *n_u64)]
 *   exchange_unpack: children of the nodes with h_mine[i] == 0 from the staging buffer into the member buffers, all records into h_records. Synchronises. */
BU_HIP_API int  bu_hip_tsvq_exchange_pack(bu_hip_context*, bu_tsvq*, const bu_tsvq_node* h_nodes, const uint8_t* h_mine, const bu_tsvq_split* h_records, uint32_t n_nodes,
                                          void** d_staging, uint64_t* n_u64);
BU_HIP_API int  bu_hip_tsvq_exchange_unpack(bu_hip_context*, bu_tsvq*, const bu_tsvq_node* h_nodes, const uint8_t* h_mine, bu_tsvq_split* h_records, uint32_t n_nodes);
BU_HIP_API void bu_hip_tsvq_destroy(bu_hip_context*, bu_tsvq*);

#ifdef __cplusplus
}
#endif
#endif /* BASISU_HIP_H */
