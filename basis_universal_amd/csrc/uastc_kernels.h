// uastc_kernels.h -- host-side launch interface of uastc_kernels.hip (internal to libbasisu_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bu {

// Bytes of device workspace bu::launch_encode_uastc needs for n_blocks at the given pack flags (level in the low bits).
size_t uastc_workspace_bytes(uint32_t n_blocks, uint32_t flags);
// encode_uastc (encoder/basisu_uastc_enc.cpp:3126) over n_blocks resident 4x4 RGBA tiles -> 16 B UASTC blocks.
// Stream-ordered; d_workspace must hold uastc_workspace_bytes(); kernel_ms (optional, 4 entries) is not touched here.
hipError_t launch_encode_uastc(hipStream_t st, const void* d_pixel_blocks, uint32_t n_blocks, uint32_t flags, void* d_workspace, void* d_out_blocks);
// The four phases separately (profiling brackets in the C ABI layer).
hipError_t launch_uastc_phase(hipStream_t st, int phase, const void* d_pixel_blocks, uint32_t n_blocks, uint32_t flags, void* d_workspace, void* d_out_blocks);

// uastc_rdo (encoder/basisu_uastc_enc.h:139, uastc_enc.cpp:4095) in place over n_blocks resident UASTC blocks (uastc_rdo_kernels.hip).
// fparams: lambda, max_allowed_rms_increase_ratio, skip_block_rms_thresh, max_smooth_block_std_dev, smooth_block_max_error_scale;
// uparams: lz_dict_size, lz_literal_cost, endpoint_refinement. total_jobs splits into independent strips exactly as the reference does.
// Phases: 0 prepare (parallel), 1 strips (serial per strip); then launch_uastc_rdo_finish (refit + hints of the modified blocks) with the
// longest per-strip list length read back from uastc_rdo_strip_counts. Stream-ordered.
size_t uastc_rdo_workspace_bytes(uint32_t n_blocks, uint32_t total_jobs);
uint32_t uastc_rdo_strips(uint32_t n_blocks, uint32_t total_jobs);
hipError_t launch_uastc_rdo_phase(hipStream_t st, int phase, void* d_blocks, const void* d_pixel_blocks, uint32_t n_blocks, const float* fparams,
                                  const uint32_t* uparams, uint32_t flags, uint32_t total_jobs, void* d_workspace);
hipError_t launch_uastc_rdo_finish(hipStream_t st, void* d_blocks, const void* d_pixel_blocks, uint32_t n_blocks, const float* fparams, const uint32_t* uparams,
                                   uint32_t flags, uint32_t total_jobs, void* d_workspace, uint32_t longest_list);
// device address of uastc_rdo_strips() x uint32: modified blocks per strip, valid after phase 1
const void* uastc_rdo_strip_counts(void* d_workspace, uint32_t n_blocks, uint32_t total_jobs);
// device address of 4 x uint32 {modified, failed, refined, skipped} inside the workspace, valid after phase 1 (refined: after the finish)
const void* uastc_rdo_counters(void* d_workspace, uint32_t n_blocks, uint32_t total_jobs);

} // namespace bu
