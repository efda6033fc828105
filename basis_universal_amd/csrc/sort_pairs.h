// sort_pairs.h -- hipcub::DeviceRadixSort::SortPairs with one difference. rocPRIM's default configuration hands inputs of up to 2^20 items to a MERGE sort (a block sort +
// ~20 dependent merge launches of 5-10 us each) and only larger ones to its one-sweep radix sort; a 4096^2 image is exactly 2^20 blocks, so the three
// sorts of a frontend step (distinct endpoint / selector training vectors, blocks by cluster) all took the merge path: ~0.2 ms each on the time line for what the
// radix path does in a handful of launches. Same result either way (both are stable sorts of the same keys); the limit is lowered, nothing else.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

namespace bu {

using sort_pairs_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;

// temporary == nullptr: only writes the bytes needed to `bytes`
template <class Key, class Value>
inline hipError_t sort_pairs(void* temporary, size_t& bytes, const Key* keys_in, Key* keys_out, const Value* values_in, Value* values_out, size_t n, unsigned begin_bit, unsigned end_bit,
                             hipStream_t stream) {
    return rocprim::radix_sort_pairs<sort_pairs_config>(temporary, bytes, keys_in, keys_out, values_in, values_out, n, begin_bit, end_bit, stream);
}

}  // namespace bu
