// tsvq_kernels.h -- launch interface of tsvq_kernels.hip (internal to libbasisu_hip.so). The POD layouts are mirrored in
// include/basisu_hip.h (bu_tsvq_*).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

namespace bu {

struct tsvq_root_out { float origin[16]; uint64_t weight; float var; uint32_t pad; };
struct tsvq_node_in { uint32_t buf, start, count, pad; uint64_t weight; float origin[16]; };
struct tsvq_split_out { uint32_t ok, l_count, r_count, pad; uint64_t l_weight, r_weight; float l_var, r_var; float l_centroid[16], r_centroid[16]; };

// d_rows: float[n][dim] (packed == false) or, for dim 16 only, uint32[n] holding sixteen 2-bit values, element 0 in the top bits.
// exact: use the integer-reduction variants for the double accumulators (packed rows only; see exact_acc). A root record with pad == 1 /
// a split record with ok == 2 means the data left the exact range: run that item again with exact == false.
hipError_t launch_tsvq_root(hipStream_t st, int dim, bool packed, bool exact, const void* d_rows, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0, tsvq_root_out* d_out);
hipError_t launch_tsvq_split(hipStream_t st, int dim, bool packed, bool exact, const void* d_rows, const uint64_t* d_w64, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side,
                             const tsvq_node_in* d_nodes, uint32_t n_nodes, tsvq_split_out* d_outs);

} // namespace bu
