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
// prepare_root of n_nodes member spans (buf / start / count of each record), one workgroup each: d_outs[i] for d_nodes[i]
hipError_t launch_tsvq_span_roots(hipStream_t st, int dim, bool packed, bool exact, const void* d_rows, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1,
                                  const tsvq_node_in* d_nodes, uint32_t n_nodes, tsvq_root_out* d_outs);
hipError_t launch_tsvq_split(hipStream_t st, int dim, bool packed, bool exact, const void* d_rows, const uint64_t* d_w64, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side,
                             const tsvq_node_in* d_nodes, uint32_t n_nodes, tsvq_split_out* d_outs, uint32_t dense_min);

// deep rounds: the 2 n_parents node records of the next generation from the parents' records and results (count = 0 and ok = 3 for what is not attempted; a parent's
// `pad` holds the bits of the variance floor, inherited). d_parent_outs / d_child_outs may be page-locked host memory.
hipError_t launch_tsvq_children(hipStream_t st, const tsvq_node_in* d_parents, const tsvq_split_out* d_parent_outs, uint32_t n_parents, tsvq_node_in* d_children,
                                tsvq_split_out* d_child_outs);

// one-thread launch that stores `value` to *d_flag (page-locked host memory) with system scope once everything before it on the stream is done
hipError_t launch_tsvq_signal(hipStream_t st, uint32_t* d_flag, uint32_t value);
// one workgroup: bytes from device memory into (coherent, device-visible) host memory, then -- flag != nullptr -- `seq` stored to *flag with system-scope release
hipError_t launch_mail_copy(hipStream_t st, void* dst, const void* src, size_t bytes, uint32_t* flag, uint32_t seq);

// ---- large nodes spread over many workgroups (tsvq_wide_kernels.hip; packed selector vectors only). Results are bit-identical to
// launch_tsvq_root / launch_tsvq_split; a split record with ok == 2 (degenerate projection, empty child, data outside the exact
// integer range) asks for that node to be run through launch_tsvq_split.
constexpr int TSVQ_WIDE_MAX_CHAINS = 136;
struct tsvq_wide_node { uint32_t buf, start, count, out_index, first_block, n_blocks, pad0, pad1; uint64_t weight; float origin[16]; };
struct tsvq_wide_ctrl {   // device-side state of one node across the passes of its split
    float l_c[16], r_c[16], axis[16];
    float sums[TSVQ_WIDE_MAX_CHAINS];
    // this pass's chain total stayed below 2^24 with integer addends: the running sum is exact, no walk needed. (A dword per chain, not a
    // byte: the chain index is wave-uniform, and hipcc 7.2 folds a uniform byte address into the base of the next scalar dword load, whose
    // two low address bits the hardware then drops.)
    uint32_t exact[TSVQ_WIDE_MAX_CHAINS];
    uint32_t start_block[TSVQ_WIDE_MAX_CHAINS]; float start_sum[TSVQ_WIDE_MAX_CHAINS];   // where a chain's walk starts: everything before is exact (sum <= 2^24)
    uint16_t stat_scans[TSVQ_WIDE_MAX_CHAINS], stat_raw[TSVQ_WIDE_MAX_CHAINS];   // of the LAST side pass: wave scans and blocks added member by member, per chain (BU_TSVQ_STATS)
    uint16_t stat_cov_scans[TSVQ_WIDE_MAX_CHAINS], stat_cov_raw[TSVQ_WIDE_MAX_CHAINS];   // the same of the covariance pass
    uint64_t l_w, r_w;
    double dsum[2];
    uint32_t l_n, r_n;
    float l_var, r_var, prev_total;
    int32_t iter, done;
    uint32_t ex_bad;
};
// node records from device-visible host memory into d_nodes + cleared d_ctrl, one launch (instead of a copy and a fill in front of a round); a split launched after it
// is told so (ctrl_cleared)
hipError_t launch_tsvq_wide_prologue(hipStream_t st, const tsvq_wide_node* src, tsvq_wide_node* d_nodes, tsvq_wide_ctrl* d_ctrl, uint32_t n_nodes);
// covariance chain sums of every node (chained sums, three workgroups per node, each a third of the chains) into d_ctrl[i].sums -- k_wide_finish<WM_COV> makes
// the axis of them; fills d_packed for the nodes' members
hipError_t launch_tsvq_cov_axis(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1,
                                const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_packed);
size_t tsvq_wide_workspace_bytes(uint32_t total_blocks);           // total_blocks = sum over the batch's nodes of ceil(count / 256)
hipError_t launch_tsvq_wide_root(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0 /* out: 0..n-1 */,
                                 const tsvq_wide_node* d_nodes /* one node: all n vectors */, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_root_out* d_out,
                                 int windows /* bu_hip_tuning::tsvq_windows */, bool ctrl_cleared = false);
// prepare_root of n_nodes member spans through the many-workgroup passes (nodes: buf / start / count / first_block / n_blocks / out_index); d_outs[out_index].pad == 1:
// the span's integer totals left the exact range, run it through launch_tsvq_span_roots
hipError_t launch_tsvq_wide_span_roots(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1, void* d_packed,
                                       const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_root_out* d_outs, int windows);
hipError_t launch_tsvq_wide_split(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side, void* d_packed /* 8 bytes per vector */,
                                  const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_split_out* d_outs,
                                  bool chained_covariance /* the covariance pass through launch_tsvq_cov_axis instead of 136 walks per node */,
                                  bool side_chains_exact /* 3 x the heaviest node's weight < 2^24: every projection / two-means chain total is exact, no maps needed */,
                                  int windows /* bu_hip_tuning::tsvq_windows */, bool ctrl_cleared = false /* launch_tsvq_wide_prologue went in front */);
// the same split for 6-float rows (the endpoint tree's large nodes, tsvq_wide6_kernels.hip): workspace / node / ctrl records as above (no barrier words); d_va: 6 n floats,
// d_tta: n doubles -- the list-order copies of the per-member addends the covariance pass lays out (launch_tsvq_cov_axis6: chained sums, one workgroup per node)
hipError_t launch_tsvq_cov_axis6(hipStream_t st, const float* d_rows, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1,
                                 const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, float* d_va, double* d_tta, uint32_t n);
hipError_t launch_tsvq_wide6_root(hipStream_t st, const float* d_rows, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0 /* out: 0..n-1 */, uint8_t* d_side,
                                  const tsvq_wide_node* d_nodes /* one node: all n vectors */, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_root_out* d_out,
                                  float* d_va, double* d_tta, bool ctrl_cleared = false);
hipError_t launch_tsvq_wide6_split(hipStream_t st, const float* d_rows, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side,
                                   const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_split_out* d_outs,
                                   float* d_va, double* d_tta, bool ctrl_cleared = false);

} // namespace bu
