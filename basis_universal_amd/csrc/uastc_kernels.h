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

} // namespace bu
