/* include/basisu_hip_frontend.h -- flat C view of bu::etc1s_frontend (basis_universal_amd/csrc/host/etc1s_frontend.h), the
 * host-side mirror of the reference's basisu_frontend (encoder/basisu_frontend.h:44-381). Lives in libbasisu_frontend.so, which
 * calls the kernels only through include/basisu_hip.h. Used by the Python binding, bench.py and the parity tests.
 * All int-returning functions: 1 = success, 0 = failure (see bu_frontend_error).
 */
#ifndef BASISU_HIP_FRONTEND_H
#define BASISU_HIP_FRONTEND_H
#include "basisu_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bu_frontend bu_frontend;

BU_HIP_API bu_frontend* bu_frontend_create(void);
BU_HIP_API void bu_frontend_destroy(bu_frontend*);  /* must precede bu_hip_destroy_context of the context it was initialised on (its device buffers are freed through it), like the reference frontend and its opencl context */
/* basisu_frontend::init (frontend.cpp:51). Exactly one of h_blocks (host tiles, uploaded once) / d_blocks (tiles already in HBM). */
BU_HIP_API int bu_frontend_init(bu_frontend*, bu_hip_context* ctx, const bu_pixel_block* h_blocks, const void* d_blocks, uint32_t n_blocks,
                                uint32_t max_endpoint_clusters, uint32_t max_selector_clusters, uint32_t compression_level, int perceptual);
/* Multi-GPU: one process per GPU, every rank runs the same frontend on the same (replicated) tiles. With a
 * communicator set, the heavy device stages are SHARDED -- per-block stages by block-row slab (a6, a10, a14), per-cluster stages by
 * cluster subsets (a9, a13) -- and their fixed-size results exchanged through the two collectives below; the order-dependent parts
 * (TSVQ, cluster bookkeeping) run replicated, so every rank ends with the identical, single-GPU-identical state.
 * The collectives are supplied by the host application (libbasisu_rccl.so: include/basisu_hip_comm.h; torch.distributed in
 * basis_universal_amd/etc1s.py for tests):
 *   all_gather    : in place over world * bytes_per_rank bytes; rank r owns segment r
 *   all_reduce_u64: in place element-wise sum of `count` u64 (used to merge disjoint per-rank results and integer accumulators)
 * Two calling conventions, told apart by `stream_ordered`:
 *   0  blocking: the frontend synchronises the context's stream before the call, and the call returns with the result complete
 *      (what a host framework that owns its own streams needs);
 *   1  stream-ordered: the collective is ENQUEUED on the context's stream (bu_hip_get_stream) behind the kernels that produced its input and in
 *      front of everything the frontend enqueues afterwards; nobody synchronises -- the host only waits where it reads a result, as in the
 *      single-GPU path. This is what the RCCL communicator does. One host thread per communicator (RCCL's own rule for blocking-free use). */
typedef struct bu_comm {
    uint32_t rank, world;
    void* user;
    int (*all_gather)(void* user, void* d_buf, uint64_t bytes_per_rank);
    int (*all_reduce_u64)(void* user, void* d_buf, uint64_t count);
    uint32_t stream_ordered, reserved;
} bu_comm;
BU_HIP_API int bu_frontend_set_comm(bu_frontend*, const bu_comm* comm); /* NULL = single GPU; call before bu_frontend_init */
/* The same for a caller that says how large ITS bu_comm is (sizeof(bu_comm) of the header it was built against): the struct has grown at its end (stream_ordered) and may
 * again; fields the caller's version does not have are taken as 0 (= the blocking convention). Prefer this entry point from bindings that outlive a header version. */
BU_HIP_API int bu_frontend_set_comm_sized(bu_frontend*, const bu_comm* comm, uint32_t struct_bytes);
/* basisu_frontend::params::m_tex_type == cBASISTexTypeVideoFrames (frontend.cpp:219-223, 291: one more fit of the merged endpoint codebook, endpoints
 * refitted to the selectors at every level); call before bu_frontend_init. The backend's half is bu_backend_params::video. */
BU_HIP_API int bu_frontend_set_video(bu_frontend*, int video);
/* SURVEY 8f row f3, the codebook builders' fast mode: both codebooks (and their parent levels) from a weighted k-means whose assignment step runs
 * on the matrix cores (bu_hip_kmeans_codebook) instead of the order-dependent TSVQ. The result is NOT bit-identical to the reference -- other,
 * deterministic codebooks of the same sizes -- and is held to the reference's own tolerances (file size +-4.5 %, PSNR -0.3 dB, basisu_tool.cpp:
 * 6786-6793) by tests/test_gpu_fast_codebooks.py. Off by default; iterations 0 keeps the default (4 Lloyd rounds + the final assignment; more rounds changed nothing measurable). Call before bu_frontend_init. */
BU_HIP_API int bu_frontend_set_fast_codebooks(bu_frontend*, int on, uint32_t iterations);

/* The reference's multi-threaded configuration -- the tool's default. max_threads is what basisu_frontend computes from params::m_multithreaded and its job
 * pool (frontend.cpp:873-876, 2195-2198; bu_frontend_reference_max_threads below restates it): from 262,144 distinct training vectors up (enc.h:2316) a value
 * T > 1 makes generate_hierarchical_codebook_threaded_internal (enc.h:2086-2215) build a T-leaf tree and then T independent trees of ceil(K / T) leaves and
 * ceil(P / T) parents over the leaves' members, concatenated in leaf order -- a different (deterministic) codebook than the single-threaded build. 0 / 1 (default)
 * = the tool under -no_multithreading. On the device the T trees share every round of node splits. Call before bu_frontend_init. */
BU_HIP_API int bu_frontend_set_max_threads(bu_frontend*, uint32_t max_threads);
/* max_threads as the reference derives it: 0 when not multithreaded, else min(hardware threads (0 = this machine's), 8, job pool threads (0 = no pool)). */
BU_HIP_API uint32_t bu_frontend_reference_max_threads(int multithreaded, uint32_t hardware_threads, uint32_t job_pool_threads);

/* basisu_frontend::compress (frontend.cpp:159) */
BU_HIP_API int bu_frontend_compress(bu_frontend*);
/* Single-step one stage method by its reference name (tests); arg = step / iteration where the method takes one. */
BU_HIP_API int bu_frontend_call(bu_frontend*, const char* stage, uint32_t arg);
/* Serialise one piece of state into buf (returns bytes needed; copies only if cap suffices; ~0 = unknown name).
 * Names and formats match oracle/ref_harness.cpp::ref_frontend_get so both sides can be diffed directly. */
BU_HIP_API uint64_t bu_frontend_get(bu_frontend*, const char* name, void* buf, uint64_t cap);
/* basisu_frontend::reoptimize_remapped_endpoints (frontend.cpp:2996-3220), what a backend calls at compression levels above 1:
 * new_block_endpoints[total blocks]; old_to_new[clusters before the call] is filled (-1: unused); block_selector_indices may be NULL. */
BU_HIP_API int bu_frontend_reoptimize_remapped_endpoints(bu_frontend*, const uint32_t* new_block_endpoints, uint32_t total_blocks, int32_t* old_to_new,
                                                         uint32_t old_to_new_count, int optimize_final_codebook, const uint32_t* block_selector_indices);
BU_HIP_API const char* bu_frontend_error(const bu_frontend*);
/* Wall seconds of each stage of the last compress(): writes up to cap entries, returns the count; names are static strings. */
BU_HIP_API uint32_t bu_frontend_stage_times(const bu_frontend*, const char** names, double* seconds, uint32_t cap);
/* No exception leaves the library: an entry point of this header or of basisu_hip_backend.h that ran into one (std::bad_alloc, a failed thread
   start) returns its failure value (0 / NULL) and leaves the exception's text here, per calling thread. */
BU_HIP_API const char* bu_host_last_exception(void);

/* ------------------------------------------------------------------------------------------------------------------
 * N images in flight on one GPU, as library behaviour: bu_frontend_pipeline_*.
 * The reference's throughput driver is basis_parallel_compress (encoder/basisu_comp.cpp:5466-5559): one host thread + one accelerator context per image in flight,
 * every thread blocked in its calls. Here the images in flight are cooperative tasks on ONE driver thread owned by the pipeline (a stack each; the frontend code is
 * the same): wherever a task's context would block on the device it yields (bu_hip_set_wait_hook), and the thread launches another image's kernels meanwhile. Half
 * of a frontend step is a chain of small dependent launches that leaves the chip mostly idle, the other half fills it: lanes interleave on the device, and the host
 * pays one thread for all of them instead of one spinning thread per image.
 *   p = bu_frontend_pipeline_create(device, lanes);                      lanes = images in flight (1..16; 3-4 fill an MI355X at 4096^2)
 *   t = bu_frontend_pipeline_submit(p, &job, sizeof(job));               never blocks: jobs beyond `lanes` queue in submission order; 0 = refused
 *   fe = bu_frontend_pipeline_wait(p, t);                                blocks the CALLER until that job's init + compress have finished; NULL = it failed
 *   ... every getter of this header / a backend (basisu_hip_backend.h) on fe, from the caller's thread; bu_frontend_pipeline_context(p, fe) is the context it ran on ...
 *   bu_frontend_pipeline_release(p, fe);                                 destroys the frontend, hands its context back (parked, warm, for a later job)
 *   bu_frontend_pipeline_destroy(p);                                     finishes what is queued, releases what nobody collected
 * Results are bit-identical to bu_frontend_init + bu_frontend_compress on a context of one's own, whatever the number of lanes (tests/test_gpu_frontend_pipeline.py).
 * Thread safety: submit / wait / poll / release may be called from any threads. Host tiles (h_blocks) must stay valid until the job's wait has returned. */
typedef struct bu_frontend_pipeline bu_frontend_pipeline;
enum { BU_FRONTEND_JOB_VIDEO = 1 };   /* bu_frontend_set_video */
typedef struct bu_frontend_job {
    const bu_pixel_block* h_blocks;   /* exactly one of: host tiles (uploaded by the job), */
    const void* d_blocks;             /*                 tiles resident on the pipeline's device */
    uint32_t n_blocks, max_endpoint_clusters, max_selector_clusters, compression_level;
    int32_t perceptual;
    uint32_t max_threads;             /* bu_frontend_set_max_threads: the reference's codebook thread configuration, 0 / 1 = -no_multithreading */
    uint32_t flags;                   /* BU_FRONTEND_JOB_* */
    uint32_t reserved;
} bu_frontend_job;
BU_HIP_API bu_frontend_pipeline* bu_frontend_pipeline_create(int device, uint32_t lanes);
/* ... with the lanes dealt out over `driver_threads` (1..lanes) driver threads instead of one: more host CPU, less waiting of a lane for the thread when the images'
 * host work (host tiles to stage, results to hand back) is heavy. Results are the same. */
BU_HIP_API bu_frontend_pipeline* bu_frontend_pipeline_create_n(int device, uint32_t lanes, uint32_t driver_threads);
BU_HIP_API uint64_t bu_frontend_pipeline_submit(bu_frontend_pipeline*, const bu_frontend_job* job, uint32_t struct_bytes /* sizeof(bu_frontend_job) of the caller's header */);
BU_HIP_API bu_frontend* bu_frontend_pipeline_wait(bu_frontend_pipeline*, uint64_t ticket);
BU_HIP_API int bu_frontend_pipeline_poll(bu_frontend_pipeline*, uint64_t ticket);   /* 1 finished (wait will not block), 0 not yet, -1 unknown ticket */
BU_HIP_API bu_hip_context* bu_frontend_pipeline_context(bu_frontend_pipeline*, bu_frontend*);
BU_HIP_API int bu_frontend_pipeline_release(bu_frontend_pipeline*, bu_frontend*);
BU_HIP_API void bu_frontend_pipeline_destroy(bu_frontend_pipeline*);
BU_HIP_API const char* bu_frontend_pipeline_error(const bu_frontend_pipeline*);
/* {jobs finished, task switches, yields, idle naps, driver seconds spent in tasks, driver seconds spent looking at idle streams, CPU seconds of the driver thread};
 * returns 7. The driver naps
 * (BU_PIPELINE_SLEEP_US, default 10) between looks once no task has had anything to do for BU_PIPELINE_SPIN_US (default 50). */
BU_HIP_API uint32_t bu_frontend_pipeline_stats(bu_frontend_pipeline*, double* out, uint32_t cap);
/* Test hook, no GPU needed: `tasks` self-test tasks (a pattern kept on the task's own stack across `yields` yields, an exception thrown and caught inside every third
 * one, the first `failing` of them ending in an exception) through a `lanes`-lane pipeline; 1 = every stack intact at every resume, failures reported as failures. */
BU_HIP_API int bu_frontend_pipeline_selftest(uint32_t lanes, uint32_t tasks, uint32_t yields, uint32_t failing);

/* The library's own host allocations of 1 MiB and more (the per-image index arrays of the frontend and the backend) come from a recycling pool of its own instead of
 * being mapped and unmapped per image (csrc/host/block_pool.cpp): local to this library -- the process's malloc is left alone. BU_HOST_POOL_MB (read once) = most
 * megabytes kept cached, default 6144, 0 = off. out = {mappings made, blocks reused, bytes cached now, cap in bytes}. */
BU_HIP_API void bu_host_pool_stats(uint64_t out[4]);
/* Gives every block the pool has cached back to the kernel (what it keeps is resident memory of the host process: a host that encodes in bursts calls this between
 * them). Blocks in use are not touched. Returns the bytes released. */
BU_HIP_API uint64_t bu_host_pool_trim(void);

/* basis_compressor::process_frontend's quality -> codebook size mapping (comp.cpp:3325-3379). */
BU_HIP_API void bu_etc1s_quality_to_clusters(int quality_level, uint32_t total_blocks, uint32_t* max_endpoint_clusters, uint32_t* max_selector_clusters);

/* f4  Mip generation (basis_compressor::generate_mipmaps, comp.cpp:2146-2230): one level = image_resample(src, dst, srgb, filter, scale,
 *     wrapping, 0, num_comps) on a resident RGBA8 raster. The filter's contributor lists and tables are built here on the host exactly as
 *     Resampler does (basis_universal_amd/csrc/host/mipmap.h), the pixels are resampled by bu_hip_k_resample_rgba8. filter: "kaiser" (the
 *     compressor's default), "box", "tent", "bell", "mitchell", "catmullrom", "blackman", "lanczos3|4|6|12". The compressor's defaults:
 *     srgb 1, scale 1, wrapping 1, and each level made from the previous one (m_mip_fast) down to 1x1. */
BU_HIP_API int bu_generate_mipmap_level(bu_hip_context* ctx, const void* d_src, uint32_t src_w, uint32_t src_h, void* d_dst, uint32_t dst_w, uint32_t dst_h,
                                        int srgb, const char* filter, float filter_scale, int wrapping, uint32_t num_comps);
/* The sizes of the levels below w x h (comp.cpp:2153-2160): writes up to cap (w, h) pairs, returns the count. */
BU_HIP_API uint32_t bu_mipmap_level_sizes(uint32_t w, uint32_t h, uint32_t smallest_dimension, uint32_t* out_wh, uint32_t cap);
/* Test hook: the plan of one resampling step as flat arrays (counts via out_counts[4] = {x taps, y taps, x_after_y, 0}); buffers may be NULL. */
BU_HIP_API int bu_mipmap_plan(uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, int srgb, const char* filter, float filter_scale, int wrapping,
                              uint32_t* out_counts, uint32_t* x_first, uint16_t* x_pixel, float* x_weight, uint32_t* y_first, uint16_t* y_pixel, float* y_weight,
                              float* srgb_to_linear_256, uint8_t* linear_to_srgb_8192);

/* Test hook: the host TSVQ (row a8) on n DISTINCT, ascending rows of `dim` (6 or 16) floats; CSR blobs out. */
BU_HIP_API int bu_host_tsvq(uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                            uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words);

/* ... with the reference's thread count (see bu_frontend_set_max_threads); min_unique_for_threads 0 = the reference's 262,144 gate, a smaller value
 * exercises generate_hierarchical_codebook_threaded_internal (enc.h:2086-2215) on small inputs. */
BU_HIP_API int bu_host_tsvq_mt(uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                               uint32_t max_threads, uint32_t min_unique_for_threads, uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent,
                               uint64_t cap_parent_words);

/* The same input through the DEVICE TSVQ driver the frontend uses (basis_universal_amd/csrc/host/tsvq_device.h). */
BU_HIP_API int bu_device_tsvq(bu_hip_context* ctx, uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size,
                              uint32_t max_parent_codebook_size, uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent,
                              uint64_t cap_parent_words, uint32_t* stats3);
BU_HIP_API int bu_device_tsvq_mt(bu_hip_context* ctx, uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size,
                                 uint32_t max_parent_codebook_size, uint32_t max_threads, uint32_t min_unique_for_threads, uint32_t* out_codebook,
                                 uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words, uint32_t* stats3);

#ifdef __cplusplus
}
#endif
#endif
