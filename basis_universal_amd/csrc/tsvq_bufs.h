// tsvq_bufs.h -- where the codebook builder's member lists live. A node's members are a span [start, start + count) of one of TSVQ_BUFS index buffers of n entries each,
// laid out back to back in ONE allocation (perm1 = perm0 + n, so the two pointers every kernel already takes give the stride); a split writes the children's lists --
// a stable partition of the node's -- over the same span of the NEXT buffer (cyclic). Two buffers were enough while a round split only nodes whose parents the
// replay of the reference's queue had already used: writing a node's children destroys the list of its (TSVQ_BUFS - 1)-th ancestor, which is then dead. A deep round
// (bu_hip_tsvq_split_deep) also splits descendants `levels` generations below the batch BEFORE anybody knows whether their parents will be used -- a node that is
// never used stays a leaf and its list must stay intact and in order -- so the ancestor a write destroys has to lie above the batch: TSVQ_BUFS - 1 > levels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace bu {

constexpr uint32_t TSVQ_BUFS = 4;             // = BU_TSVQ_BUFFERS (include/basisu_hip.h); a power of two
constexpr uint32_t TSVQ_MAX_DEEP_LEVELS = TSVQ_BUFS - 2;

__host__ __device__ __forceinline__ uint32_t tsvq_child_buf(uint32_t buf) { return (buf + 1u) & (TSVQ_BUFS - 1u); }
template <typename T> __host__ __device__ __forceinline__ T* tsvq_list(T* perm0, T* perm1, uint32_t buf) { return perm0 + (ptrdiff_t)buf * (perm1 - perm0); }
template <typename T> __host__ __device__ __forceinline__ T* tsvq_child_list(T* perm0, T* perm1, uint32_t buf) { return tsvq_list(perm0, perm1, tsvq_child_buf(buf)); }

} // namespace bu
