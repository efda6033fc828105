// mipmap_kernels.h -- launch interface of mipmap_kernels.hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bu {

// One separable resampling step on a resident RGBA8 raster. The contributor lists (CSR: first[n + 1], pixel, weight)
// and the two value tables are computed by the host (host/mipmap.h) and already in device memory; d_tmp holds
// max(dst_w * src_h, src_w * dst_h) float4. The first num_comps channels are resampled (3: alpha becomes 255).
hipError_t launch_resample_rgba8(hipStream_t st, const void* d_src, uint32_t src_w, uint32_t src_h, void* d_dst, uint32_t dst_w, uint32_t dst_h,
                                 const uint32_t* d_x_first, const uint16_t* d_x_pixel, const float* d_x_weight,
                                 const uint32_t* d_y_first, const uint16_t* d_y_pixel, const float* d_y_weight,
                                 bool x_after_y, bool srgb, const float* d_srgb_to_linear, const uint8_t* d_linear_to_srgb, uint32_t num_comps, void* d_tmp);

}  // namespace bu
