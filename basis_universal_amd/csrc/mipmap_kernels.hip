// mipmap_kernels.hip -- mip generation on the resident raster: the device half of
// basis_compressor::generate_mipmaps -> image_resample -> Resampler (encoder/basisu_comp.cpp:2146-2230, basisu_enc.cpp:1022-1180,
// basisu_resampler.cpp:343-435).
//
// The reference streams source lines through a separable filter and keeps whichever pass order costs fewer multiply-adds; every
// destination sample is a float sum of (source sample * weight) over a short contributor list, accumulated in list order. Here the two
// passes are two launches over the whole image, one thread per intermediate / destination pixel, and a thread walks the same list in the
// same order with separate multiply and add (the file is compiled with -ffp-contract=off), so the float result -- and the byte it
// quantises to -- is the reference's. Which sums start from 0 and which from the first product follows the reference as well
// (resample_x: total = 0, then +=; resample_y: the first contributor moves, the rest add).
//
// HBM bound in principle (4 B in, 16 B intermediate, 4 B out per pixel, lists and tables in cache); a mip chain is ~1/3 of one pass over
// the image, so no tiling through LDS is attempted.
#include "mipmap_kernels.h"

namespace bu {
namespace {

struct taps { const uint32_t* first; const uint16_t* pixel; const float* weight; };

__device__ __forceinline__ float to_linear(uint32_t rgba, int c, const float* __restrict__ table) {
    const uint32_t v = (rgba >> (8 * c)) & 255u;
    return c == 3 ? (float)v * (1.0f / 255.0f) : table[v];   // alpha is never gamma coded (basisu_enc.cpp:1109-1112); without sRGB the table is v/255
}

// first pass along x: tmp[y][dx] = sum_k src[y][pixel_k] * weight_k, every source row
__global__ __launch_bounds__(256) void k_first_x(const uint32_t* __restrict__ src, uint32_t src_w, uint32_t src_h, uint32_t dst_w, taps t,
                                                 const float* __restrict__ table, uint32_t num_comps, float4* __restrict__ tmp) {
    const uint32_t dx = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y;
    if (dx >= dst_w) return;
    float total[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t* row = src + (size_t)y * src_w;
    for (uint32_t k = t.first[dx]; k < t.first[dx + 1]; k++) {
        const uint32_t p = row[t.pixel[k]];
        const float w = t.weight[k];
        for (uint32_t c = 0; c < 4; c++)
            if (c < num_comps) total[c] += to_linear(p, (int)c, table) * w;
    }
    tmp[(size_t)y * dst_w + dx] = make_float4(total[0], total[1], total[2], total[3]);
}

// first pass along y (x delayed): tmp[dy][x] = src[pixel_0][x] * weight_0, then += the rest, every source column
__global__ __launch_bounds__(256) void k_first_y(const uint32_t* __restrict__ src, uint32_t src_w, uint32_t dst_h, taps t, const float* __restrict__ table,
                                                 uint32_t num_comps, float4* __restrict__ tmp) {
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, dy = blockIdx.y;
    if (x >= src_w) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t k0 = t.first[dy], k1 = t.first[dy + 1];
    for (uint32_t k = k0; k < k1; k++) {
        const uint32_t p = src[(size_t)t.pixel[k] * src_w + x];
        const float w = t.weight[k];
        for (uint32_t c = 0; c < 4; c++) {
            if (c >= num_comps) continue;
            const float term = to_linear(p, (int)c, table) * w;
            acc[c] = (k == k0) ? term : acc[c] + term;
        }
    }
    tmp[(size_t)dy * src_w + x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

__device__ __forceinline__ uint32_t quantise(float v, int c, bool srgb, const uint8_t* __restrict__ to_srgb) {
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);   // Resampler::clamp_sample with [0, 1]
    if (!srgb || c == 3) { const int j = (int)(255.0f * v + .5f); return (uint32_t)(j < 0 ? 0 : (j > 255 ? 255 : j)); }
    const int j = (int)(8191.0f * v + .5f);
    return to_srgb[j < 0 ? 0 : (j > 8191 ? 8191 : j)];
}

// second pass: along y over the x-resampled rows, or along x over the y-resampled columns; quantise and store
template <bool ALONG_Y>
__global__ __launch_bounds__(256) void k_second(const float4* __restrict__ tmp, uint32_t tmp_w, uint32_t dst_w, taps t, bool srgb, const uint8_t* __restrict__ to_srgb,
                                                uint32_t num_comps, uint32_t* __restrict__ dst) {
    const uint32_t dx = blockIdx.x * 256u + threadIdx.x, dy = blockIdx.y;
    if (dx >= dst_w) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t i = ALONG_Y ? dy : dx, k0 = t.first[i], k1 = t.first[i + 1];
    for (uint32_t k = k0; k < k1; k++) {
        const float4 s = ALONG_Y ? tmp[(size_t)t.pixel[k] * tmp_w + dx] : tmp[(size_t)dy * tmp_w + t.pixel[k]];
        const float w = t.weight[k];
        const float term[4] = {s.x * w, s.y * w, s.z * w, s.w * w};
        for (uint32_t c = 0; c < 4; c++) {
            if (c >= num_comps) continue;
            acc[c] = (ALONG_Y && k == k0) ? term[c] : acc[c] + term[c];   // along x the sum starts from 0
        }
    }
    uint32_t out = 0xFF000000u;   // channels that are not resampled keep the fresh image's (0, 0, 0, 255)
    for (uint32_t c = 0; c < 4; c++)
        if (c < num_comps) out = (out & ~(255u << (8 * c))) | (quantise(acc[c], (int)c, srgb, to_srgb) << (8 * c));
    dst[(size_t)dy * dst_w + dx] = out;
}

}  // namespace

#define BU_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_resample_rgba8(hipStream_t st, const void* d_src, uint32_t src_w, uint32_t src_h, void* d_dst, uint32_t dst_w, uint32_t dst_h,
                                 const uint32_t* d_x_first, const uint16_t* d_x_pixel, const float* d_x_weight,
                                 const uint32_t* d_y_first, const uint16_t* d_y_pixel, const float* d_y_weight,
                                 bool x_after_y, bool srgb, const float* d_srgb_to_linear, const uint8_t* d_linear_to_srgb, uint32_t num_comps, void* d_tmp) {
    const taps tx{d_x_first, d_x_pixel, d_x_weight}, ty{d_y_first, d_y_pixel, d_y_weight};
    const uint32_t* src = static_cast<const uint32_t*>(d_src);
    uint32_t* dst = static_cast<uint32_t*>(d_dst);
    float4* tmp = static_cast<float4*>(d_tmp);
    const dim3 blk(256), out_grid((dst_w + 255) / 256, dst_h);
    if (!x_after_y) {
        hipLaunchKernelGGL(k_first_x, dim3((dst_w + 255) / 256, src_h), blk, 0, st, src, src_w, src_h, dst_w, tx, d_srgb_to_linear, num_comps, tmp);
        BU_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_second<true>, out_grid, blk, 0, st, tmp, dst_w, dst_w, ty, srgb, d_linear_to_srgb, num_comps, dst);
    } else {
        hipLaunchKernelGGL(k_first_y, dim3((src_w + 255) / 256, dst_h), blk, 0, st, src, src_w, dst_h, ty, d_srgb_to_linear, num_comps, tmp);
        BU_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_second<false>, out_grid, blk, 0, st, tmp, src_w, dst_w, tx, srgb, d_linear_to_srgb, num_comps, dst);
    }
    BU_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace bu
