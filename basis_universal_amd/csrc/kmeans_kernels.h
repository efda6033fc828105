// kmeans_kernels.h -- launch interface of kmeans_kernels.hip (internal to libbasisu_hip.so; C ABI: bu_hip_kmeans_codebook in include/basisu_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bu {

struct kmeans_buffers {
    void* vec; uint64_t* weights; void* hi; void* lo; float* cnorm; float* cen; uint64_t* sums; uint32_t k_pad;
    float *err_key, *err_key_sorted; uint32_t *err_idx, *worst, *pick, *empty; uint64_t* cum; void* cub; size_t cub_bytes;
};
size_t kmeans_workspace_bytes(uint32_t n, uint32_t k);
kmeans_buffers kmeans_carve(void* ws, uint32_t n, uint32_t k);
// endpoints == 0: d_keys = uint32 packed selector vectors, d_weights their weights. endpoints != 0: d_keys = uint64 48-bit colour keys, weights
// are 2 x the group sizes (d_goffs). After the call (stream-ordered): d_assign[u] = centroid of distinct vector u, b.sums[c * 17 + 16] = weight of
// cluster c, b.cen = centroids of the last update (k x 16 floats).
hipError_t launch_kmeans(hipStream_t st, int endpoints, const void* d_keys, const uint64_t* d_weights, const uint32_t* d_goffs, uint32_t n, uint32_t k, uint32_t iterations,
                         const kmeans_buffers& b, uint32_t* d_assign);

} // namespace bu
