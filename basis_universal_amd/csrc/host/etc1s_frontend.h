// etc1s_frontend.h -- host-side mirror of the reference's basisu_frontend (encoder/basisu_frontend.h:44-381) on top of the
// device-resident C ABI (include/basisu_hip.h, section 2).
//
// Same public surface (params, init, compress, the getters the backend uses) and the same stage methods in the same order as
// basisu_frontend::compress() (frontend.cpp:159-316). Every stage that touches pixels runs as a HIP kernel on the resident
// 4x4 tiles; what stays on the host is what the reference keeps serial and order-dependent: the TSVQ tree builds (tsvq.h) and
// the cluster-list bookkeeping between stages. There is no CPU implementation of any kernel stage in here: a failing device
// call fails the frontend.
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/basisu_hip.h"
#include "../../../include/basisu_hip_frontend.h"

namespace bu {

struct endpoint_params {   // = basisu_frontend::endpoint_cluster_etc_params reduced to what ETC1S uses (frontend.h:202-268)
    uint8_t r = 0, g = 0, b = 0, inten = 0;
    uint64_t color_error = 0;
    bool valid = false;
    bool color_used = false;
};

// The per-block result arrays (8 / 4 bytes x blocks) are filled by device -> host copies right after they are sized: a std::vector would zero them first -- 20 MB of
// memset per 4096^2 image on the thread that drives the GPU, while the GPU waits for its next launch. resize() of these leaves new elements uninitialised.
template <class T> struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    default_init_allocator() = default;
    template <class U> default_init_allocator(const default_init_allocator<U>&) noexcept {}
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using raw_vector = std::vector<T, default_init_allocator<T>>;

class etc1s_frontend {
public:
    enum { cMaxEndpointClusters = 16128, cMaxSelectorClusters = 16128 }; // frontend.h:66-71

    struct params {                       // = basisu_frontend::params (frontend.h:73-115)
        uint32_t m_num_source_blocks = 0;
        const bu_pixel_block* m_pSource_blocks = nullptr; // host tiles; uploaded once unless m_pDevice_blocks is given
        const void* m_pDevice_blocks = nullptr;           // optional: tiles already resident in HBM (not owned)
        uint32_t m_max_endpoint_clusters = 256;
        uint32_t m_max_selector_clusters = 256;
        uint32_t m_compression_level = 2;                 // BASISU_DEFAULT_ETC1S_COMPRESSION_LEVEL
        bool m_perceptual = true;
        bool m_validate = false;
        bool m_disable_hierarchical_endpoint_codebooks = false;
        bool m_video = false;                             // = m_tex_type == cBASISTexTypeVideoFrames: only changes the order of stages (frontend.cpp:219-223, 291)
        // both codebooks from a k-means on the matrix cores (bu_hip_kmeans_codebook) instead of the order-dependent TSVQ. NOT
        // bit-identical to the reference: different codebooks (deterministic), held to the reference's size / PSNR tolerances. Off by default.
        bool m_fast_codebooks = false;
        uint32_t m_fast_codebook_iterations = 4;
        // What the reference computes from m_multithreaded and its job pool (frontend.cpp:873-876, 2195-2198): min(hardware threads, 8, pool size), or 0
        // when not multithreaded. From 262,144 distinct training vectors up (enc.h:2316) a value T > 1 makes both codebook builders partition their
        // tree T ways (enc.h:2086-2215) -- the tool's DEFAULT output on a machine with T threads; 0 / 1 = the tool under -no_multithreading.
        uint32_t m_codebook_threads = 0;
        bu_hip_context* m_pHIP_context = nullptr;         // = m_pOpenCL_context; REQUIRED
    };

    etc1s_frontend();
    ~etc1s_frontend();
    void context_closing();   // called by the context while it is being destroyed (bu_hip_on_destroy): lets go of the device buffers
    etc1s_frontend(const etc1s_frontend&) = delete;
    etc1s_frontend& operator=(const etc1s_frontend&) = delete;

    // multi-GPU: see include/basisu_hip_frontend.h (bu_comm). The struct is copied.
    void set_comm(const bu_comm* c) { if (c) { m_comm = *c; m_has_comm = c->world > 1; } else m_has_comm = false; }
    bool init(const params& p);
    bool compress();
    const std::string& error() const { return m_error; }

    // ---- getters with the reference's names (frontend.h:119-156)
    const params& get_params() const { return m_params; }
    uint32_t get_total_output_blocks() const { return m_total_blocks; }
    const bu_etc_block& get_output_block(uint32_t i) const { ensure_encoded_host(); return m_encoded_blocks[i]; }
    const raw_vector<bu_etc_block>& get_output_blocks() const { ensure_encoded_host(); return m_encoded_blocks; }
    const bu_etc_block& get_etc1s_block(uint32_t i) const { return etc1_blocks()[i]; }
    uint32_t get_total_endpoint_clusters() const { ensure_endpoint_map(); return m_endpoint_cluster_count; }
    uint32_t get_subblock_endpoint_cluster_index(uint32_t block, uint32_t) const { ensure_endpoint_map(); return m_block_endpoint_cluster[block]; }
    const endpoint_params& get_endpoint_cluster_params(uint32_t ci) const { return m_endpoint_cluster_etc_params[ci]; }
    uint32_t get_total_selector_clusters() const { return m_selector_cluster_count; }
    uint32_t get_block_selector_cluster_index(uint32_t block) const { ensure_selector_map_host(); return m_block_selector_cluster_index[block]; }
    const bu_etc_block& get_selector_cluster_selector_bits(uint32_t ci) const { return m_optimized_cluster_selectors[ci]; }
    const std::vector<uint32_t>& get_selector_cluster_block_indices(uint32_t ci) const { return selector_cluster_block_indices()[ci]; }

    // ---- stage state, exposed for stage-by-stage parity tests
    const std::vector<bu_etc_block>& etc1_blocks() const;  // fetched from the device on first use
    const raw_vector<bu_etc_block>& orig_encoded_blocks() const { ensure_orig_encoded_host(); return m_orig_encoded_blocks; }
    const std::vector<std::vector<uint32_t>>& endpoint_clusters() const;  // built on first use from the (cluster, position) map
    const std::vector<std::vector<uint32_t>>& endpoint_parent_clusters() const;  // built on first use from the parent-of-vector map
    const std::vector<endpoint_params>& endpoint_cluster_params() const { return m_endpoint_cluster_etc_params; }
    const raw_vector<uint32_t>& block_endpoint_clusters() const { ensure_endpoint_map(); return m_block_endpoint_cluster; }
    const std::vector<std::vector<uint32_t>>& selector_cluster_block_indices() const;  // built on first use from the block -> cluster map (ascending blocks)
    const std::vector<bu_etc_block>& optimized_cluster_selectors() const { return m_optimized_cluster_selectors; }
    const raw_vector<uint32_t>& block_selector_cluster_index() const { ensure_selector_map_host(); return m_block_selector_cluster_index; }

    // ---- what the backend needs beyond the getters (etc1s_backend.h)
    const bu_pixel_block* source_blocks_host();  // get_source_pixel_block: the caller's host tiles, or a host copy of device-only tiles
    // For the backend's create_encoder_blocks (row f2): every block's error as encoded and, where asked, under its three causal neighbours' endpoints, computed on the
    // device from the resident tiles, blocks and clustering (bu_hip_k_backend_block_errors). own: n words, neighbour: 3 n words (n = blocks of the slice). false: the
    // resident state is not current (or the call failed) -- the backend then computes them on the host.
    bool backend_block_errors(uint32_t first_block, uint32_t num_blocks_x, uint32_t num_blocks_y, bool with_neighbours, uint32_t* own, uint32_t* neighbour);
    // basisu_frontend::reoptimize_remapped_endpoints (frontend.cpp:2996-3220): the backend moved blocks to other endpoint clusters
    bool reoptimize_remapped_endpoints(const std::vector<uint32_t>& new_block_endpoints, std::vector<int>& old_to_new, bool optimize_final_codebook,
                                       const std::vector<uint32_t>* block_selector_indices);

    // wall time of each stage of the last compress(), in call order (name, seconds)
    struct stage_time { const char* name; double seconds; };
    const std::vector<stage_time>& stage_times() const { return m_stage_times; }

    // ---- stage methods, same names and order as the reference (frontend.h:346-372); public so tests can single-step
    bool init_etc1_images();
    int etc1_images_quality() const;
    bool init_endpoint_training_vectors();
    bool generate_endpoint_clusters();
    bool introduce_new_endpoint_clusters();
    bool generate_endpoint_codebook(uint32_t step);
    bool refine_endpoint_clusterization(uint32_t* total_reassigned);
    bool eliminate_redundant_or_empty_endpoint_clusters();
    void generate_block_endpoint_clusters();
    void compute_endpoint_clusters_within_each_parent_cluster();
    bool create_initial_packed_texture();
    bool generate_selector_clusters();
    void compute_selector_clusters_within_each_parent_cluster();
    bool create_optimized_selector_codebook(uint32_t iter);
    bool find_optimal_selector_clusters_for_each_block();
    bool introduce_special_selector_clusters();
    bool refine_block_endpoints_given_selectors(uint32_t* total_refined);
    bool optimize_selector_codebook();
    void finalize();

private:
    struct device_state;
    void drop_device_state();
    bool fail(const char* what);

    params m_params;
    bu_comm m_comm{};
    bool m_has_comm = false;
    uint32_t comm_world() const { return m_has_comm ? m_comm.world : 1; }
    uint32_t comm_rank() const { return m_has_comm ? m_comm.rank : 0; }
    uint32_t slab_blocks() const;                           // blocks per rank (multiple of the selector job size)
    void my_slab(uint32_t& first, uint32_t& count) const;
    bool gather_blocks(void* d_buf, size_t bytes_per_block); // all-gather of a per-block device array written slab-wise
    bool merge_disjoint(void* d_buf, size_t bytes);          // sum-merge of per-rank partial results (zero where not owned)
    std::string m_error;
    device_state* m_dev = nullptr;

    uint32_t m_total_blocks = 0;
    bool m_endpoint_refinement = false;
    bool m_use_hierarchical_endpoint_codebooks = false;
    bool m_use_hierarchical_selector_codebooks = false;
    uint32_t m_num_endpoint_codebook_iterations = 1;
    uint32_t m_num_selector_codebook_iterations = 1;

    // The encoded blocks (and their pre-selector-quantisation copy) live where they were last written -- normally HBM; the other side is
    // brought up to date on demand (ensure_encoded_host / ensure_encoded_device).
    mutable raw_vector<bu_etc_block> m_encoded_blocks, m_orig_encoded_blocks;
    mutable bool m_enc_host_valid = true, m_orig_host_valid = true;
    bool m_enc_dev_valid = false;
    void ensure_encoded_host() const;
    void ensure_orig_encoded_host() const;
    bool ensure_encoded_device();
    void ensure_selector_map_host() const;
    mutable std::vector<bu_etc_block> m_etc1_blocks_etc1s;  // host mirror of the device's a6 output, see etc1_blocks()
    mutable bool m_etc1_on_host = false;
    // results on their way to the host while the later stages run (bu_hip_download_*): the endpoint map from the moment it is final (levels 0-3: after
    // eliminate_redundant_or_empty_endpoint_clusters), the encoded blocks from find_optimal_selector_clusters_for_each_block on. Whoever needs the host form waits
    // (ensure_endpoint_map / ensure_encoded_host); nothing writes the device form while one is pending.
    mutable struct bu_hip_download *m_dl_ep_cluster = nullptr, *m_dl_ep_pos = nullptr, *m_dl_enc = nullptr;
    void prefetch_endpoint_map();
    void prefetch_encoded_blocks();
    void finish_prefetches(int which = 3) const;
    // the (parent, cluster) membership table of refine_endpoint_clusterization, when it came back with the codebook fit's results (one copy instead of two round trips)
    std::vector<uint8_t> m_ep_member; size_t m_ep_member_parents = 0, m_ep_member_clusters = 0; bool m_ep_member_valid = false;
    bool m_sel_blocks_dev_valid = false;   // the device copy of the selector codebook equals m_optimized_cluster_selectors (create_optimized_selector_codebook just made both)
    bool m_etc1_made_by_init = false;   // init() uploaded host tiles and encoded them piece by piece (bu_hip_k_upload_and_encode_etc1s_blocks): init_etc1_images() has nothing left to do

    // endpoint side
    std::vector<float> m_endpoint_unique_rows;            // distinct (low rgb, high rgb)/255 vectors, ascending
    std::vector<uint64_t> m_endpoint_unique_weights;
    uint32_t m_endpoint_unique_count = 0;                     // distinct endpoint training vectors (resident: ep_ukeys / ep_goffs / ep_idx)
    bool m_endpoint_parent_dev_valid = false;                 // ep_parent_u holds the parent of every distinct vector
    mutable std::vector<uint32_t> m_endpoint_group_offsets;    // CSR offsets: the blocks behind every distinct vector ... (fetched on demand)
    mutable std::vector<uint32_t> m_endpoint_group_blocks;     // ... ascending; resident, fetched by endpoint_group_blocks_host() when a list form is asked for
    const std::vector<uint32_t>& endpoint_group_blocks_host() const;
    const std::vector<uint32_t>& endpoint_group_offsets_host() const;
    const std::vector<uint32_t>& endpoint_parent_of_unique_host() const;
    // the endpoint clustering in its two forms (etc1s_frontend.cpp, ensure_endpoint_map / ensure_endpoint_lists)
    mutable std::vector<std::vector<uint32_t>> m_endpoint_clusters;
    mutable raw_vector<uint32_t> m_block_endpoint_pos;
    mutable std::vector<uint32_t> m_endpoint_cluster_sizes;
    mutable uint32_t m_endpoint_cluster_count = 0;
    mutable bool m_endpoint_map_valid = false, m_endpoint_lists_valid = false;   // the HOST forms (per-block arrays / lists)
    bool m_ep_dev_valid = false;                                                   // the RESIDENT per-block arrays (device_state::block_cluster, ep_pos) are current
    void ensure_endpoint_map() const;
    bool ensure_endpoint_map_device();
    void ensure_endpoint_lists() const;
    void endpoint_csr(std::vector<uint32_t>& offsets, std::vector<uint32_t>& indices) const;
    mutable std::vector<std::vector<uint32_t>> m_endpoint_parent_clusters;  // lazily materialised, see endpoint_parent_clusters()
    mutable std::vector<uint32_t> m_endpoint_parent_of_unique;   // parent cluster of every distinct training vector (fetched on demand: endpoint_parent_of_unique_host)
    std::vector<uint32_t> m_selector_parent_of_unique;
    uint32_t m_endpoint_parent_count = 0, m_selector_parent_count = 0;
    std::vector<uint32_t> m_selector_group_offsets, m_selector_group_blocks;  // CSR: the blocks behind every distinct selector vector
    std::vector<uint8_t> m_block_parent_endpoint_cluster;
    std::vector<std::vector<uint32_t>> m_endpoint_clusters_within_each_parent_cluster;
    std::vector<endpoint_params> m_endpoint_cluster_etc_params;
    mutable raw_vector<uint32_t> m_block_endpoint_cluster;  // mutable: part of the lazily synchronised clustering, see ensure_endpoint_map
    std::vector<std::vector<uint32_t>> m_endpoint_cluster_subblocks;  // endpoint_cluster_etc_params::m_subblocks (never cleared, frontend.cpp:2727-2729)

    // selector side
    mutable std::vector<std::vector<uint32_t>> m_selector_cluster_block_indices;  // lazily materialised from m_block_selector_cluster_index
    mutable bool m_selector_lists_valid = false;
    std::vector<std::vector<uint32_t>> m_selector_parent_cluster_block_indices;
    std::vector<uint32_t> m_selector_leaf_of_unique;
    uint32_t m_selector_cluster_count = 0;
    void selector_csr(std::vector<uint32_t>& offsets, std::vector<uint32_t>& indices) const;
    std::vector<bu_etc_block> m_optimized_cluster_selectors;
    std::vector<uint8_t> m_block_parent_selector_cluster;
    std::vector<std::vector<uint32_t>> m_selector_clusters_within_each_parent_cluster;
    mutable raw_vector<uint32_t> m_block_selector_cluster_index;   // host form of the block -> selector cluster map ...
    mutable bool m_sel_host_valid = true;
    bool m_sel_dev_valid = false;                                    // ... and whether the resident form (device_state::sel_cluster) is current
    bool ensure_selector_map_device();

    std::vector<stage_time> m_stage_times;
    std::vector<bu_pixel_block> m_source_copy;  // see source_blocks_host()
};

} // namespace bu
