// integration/basisu_hip_shim.cpp -- the translation unit a maintainer of the reference drops in instead of encoder/basisu_opencl.cpp
// (INTEGRATION.md): the reference's ten accelerator entry points (encoder/basisu_opencl.h:24-141) forwarded to libbasisu_hip.so
// (include/basisu_hip.h, section 1). This file is OURS and contains no reference code; it only includes the reference's header.
// oracle/Makefile links it with the untouched reference objects into oracle/_ref/basisu_hip: the reference command line tool whose
// `-opencl` switch now selects the MI355X kernels.
#include "encoder/basisu_opencl.h"
#include "basisu_hip.h"            // this repository: include/basisu_hip.h

namespace basisu {
struct opencl_context { bu_hip_context* h; };
static_assert(sizeof(cl_pixel_block) == sizeof(bu_pixel_block) && sizeof(etc_block) == sizeof(bu_etc_block), "layout");
static_assert(sizeof(cl_pixel_cluster) == sizeof(bu_pixel_cluster) && sizeof(cl_block_info_struct) == sizeof(bu_block_info), "layout");
static_assert(sizeof(cl_endpoint_cluster_struct) == sizeof(bu_endpoint_cluster) && sizeof(fosc_block_struct) == sizeof(bu_fosc_block), "layout");

bool opencl_init(bool force_serialization) { return bu_hip_init(force_serialization) != 0; }
void opencl_deinit() { bu_hip_deinit(); }
bool opencl_is_available() { return bu_hip_is_available() != 0; }
opencl_context_ptr opencl_create_context() { bu_hip_context* h = bu_hip_create_context(); return h ? new opencl_context{h} : nullptr; }
void opencl_destroy_context(opencl_context_ptr c) { if (c) { bu_hip_destroy_context(c->h); delete c; } }
bool opencl_set_pixel_blocks(opencl_context_ptr c, size_t n, const cl_pixel_block* p)
{ return bu_hip_set_pixel_blocks(c->h, n, (const bu_pixel_block*)p) != 0; }
bool opencl_encode_etc1s_blocks(opencl_context_ptr c, etc_block* out, bool perceptual, uint32_t total_perms)
{ return bu_hip_encode_etc1s_blocks(c->h, (bu_etc_block*)out, perceptual, total_perms) != 0; }
bool opencl_encode_etc1s_pixel_clusters(opencl_context_ptr c, etc_block* out, uint32_t total_clusters, const cl_pixel_cluster* cl,
    uint64_t total_pixels, const color_rgba* px, const uint32_t* w, bool perceptual, uint32_t total_perms)
{ return bu_hip_encode_etc1s_pixel_clusters(c->h, (bu_etc_block*)out, total_clusters, (const bu_pixel_cluster*)cl, total_pixels,
                                            (const bu_color_rgba*)px, w, perceptual, total_perms) != 0; }
bool opencl_refine_endpoint_clusterization(opencl_context_ptr c, const cl_block_info_struct* bi, uint32_t total_clusters,
    const cl_endpoint_cluster_struct* ci, const uint32_t* sorted, uint32_t* out, bool perceptual)
{ return bu_hip_refine_endpoint_clusterization(c->h, (const bu_block_info*)bi, total_clusters, (const bu_endpoint_cluster*)ci, sorted, out, perceptual) != 0; }
bool opencl_find_optimal_selector_clusters_for_each_block(opencl_context_ptr c, const fosc_block_struct* bi, uint32_t total_sel,
    const fosc_selector_struct* sel, const uint32_t* idx, uint32_t* out, bool perceptual)
{ return bu_hip_find_optimal_selector_clusters_for_each_block(c->h, (const bu_fosc_block*)bi, total_sel, (const bu_fosc_selector*)sel, idx, out, perceptual) != 0; }
bool opencl_determine_selectors(opencl_context_ptr c, const color_rgba* c5i, etc_block* out, bool perceptual)
{ return bu_hip_determine_selectors(c->h, (const bu_color_rgba*)c5i, (bu_etc_block*)out, perceptual) != 0; }
}  // namespace basisu
