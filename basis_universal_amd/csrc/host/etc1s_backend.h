// etc1s_backend.h -- host-side mirror of the reference's basisu_backend (encoder/basisu_backend.h:153-408, basisu_backend.cpp) for
// ETC1S: turns a finished frontend (two codebooks + two indices per block) into the compressed payloads of a .basis / KTX2 file --
// the endpoint and selector palettes, the Huffman tables of the slice data and one bit stream per slice -- byte for byte what the
// reference writes. Same public surface: params, slice descriptors, init(frontend, params, slices), encode(),
// get_output().
//
// The whole stage is a raster-order walk with state carried from block to block (endpoint prediction from already coded
// neighbours, the selector history buffer, run lengths), so it is host code like the reference's; what is different is how the
// walk spends its time (etc1s_backend.cpp).
//
// Not supported: global codebooks. Video textures (conditional replenishment: a block that repeats the block at its place in the previous
// frame) are coded like the reference does; the frontend's video-specific iteration order (frontend.cpp:219-223, 291) is not built.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/basisu_hip.h"
#include "block_metric.h"

namespace bu {

class etc1s_frontend;

struct backend_params {                    // = basisu_backend_params (backend.h:153-183)
    bool m_etc1s = true;
    float m_endpoint_rdo_quality_thresh = 0.0f;
    float m_selector_rdo_quality_thresh = 0.0f;
    uint32_t m_compression_level = 0;
    bool m_used_global_codebooks = false;  // must stay false
    bool m_video = false;                  // = frontend params' m_tex_type == cBASISTexTypeVideoFrames: slices are frames, blocks may repeat the previous frame's
    bool m_validate = true;
};

struct backend_slice_desc {                // = basisu_backend_slice_desc (backend.h:185-214)
    uint32_t m_first_block_index = 0;
    uint32_t m_orig_width = 0, m_orig_height = 0;
    uint32_t m_width = 0, m_height = 0;
    uint32_t m_num_blocks_x = 0, m_num_blocks_y = 0;
    uint32_t m_num_macroblocks_x = 0, m_num_macroblocks_y = 0;
    uint32_t m_source_file_index = 0;
    uint32_t m_mip_index = 0;
    bool m_alpha = false;
    bool m_iframe = false;
};

struct backend_output {                    // = basisu_backend_output (backend.h:218-276)
    uint32_t m_tex_format = 0;             // basist::basis_tex_format: 0 = cETC1S, 1 = cUASTC_LDR_4x4
    bool m_etc1s = false;
    bool m_uses_global_codebooks = false;
    bool m_srgb = true;
    uint32_t m_num_endpoints = 0;
    uint32_t m_num_selectors = 0;
    std::vector<uint8_t> m_endpoint_palette;
    std::vector<uint8_t> m_selector_palette;
    std::vector<backend_slice_desc> m_slice_desc;
    std::vector<uint8_t> m_slice_image_tables;
    std::vector<std::vector<uint8_t>> m_slice_image_data;
    std::vector<uint16_t> m_slice_image_crcs;
    uint32_t get_output_size_estimate() const {
        size_t t = m_slice_image_tables.size() + m_endpoint_palette.size() + m_selector_palette.size();
        for (const auto& s : m_slice_image_data) t += s.size();
        return (uint32_t)t;
    }
};

// What the backend reads from a finished frontend (the getters of frontend.h:119-156 as flat arrays). All pointers stay owned by
// the caller and must outlive encode().
struct backend_source {
    uint32_t total_blocks = 0;
    bool perceptual = true;
    const bu_pixel_block* source_blocks = nullptr;   // get_source_pixel_block
    const bu_etc_block* output_blocks = nullptr;     // get_output_block
    const uint32_t* block_endpoint_index = nullptr;  // get_subblock_endpoint_cluster_index(b, 0)
    const uint32_t* block_selector_index = nullptr;  // get_block_selector_cluster_index
    uint32_t total_endpoints = 0;
    const uint8_t* endpoint_color5_inten = nullptr;  // per cluster r5, g5, b5, intensity table
    uint32_t total_selectors = 0;
    const bu_etc_block* selector_blocks = nullptr;   // get_selector_cluster_selector_bits
};

class etc1s_backend {
public:
    // The backend's one call back into the frontend (basisu_frontend::reoptimize_remapped_endpoints, frontend.cpp:2996-3220): after
    // blocks were moved to other endpoint clusters, refit the clusters and (when `final_codebook`) renumber them. Refreshes `src`.
    using reoptimize_fn = std::function<bool(const std::vector<uint32_t>& new_block_endpoints, std::vector<int>& old_to_new,
                                             bool final_codebook, const std::vector<uint32_t>* block_selector_indices, backend_source& src)>;

    void init(etc1s_frontend* frontend, const backend_params& params, const std::vector<backend_slice_desc>& slices);
    void init(const backend_source& src, const backend_params& params, const std::vector<backend_slice_desc>& slices, reoptimize_fn reoptimize = nullptr);
    void set_reoptimize(reoptimize_fn fn) { m_reoptimize = std::move(fn); }
    uint32_t total_endpoints() const { return m_src.total_endpoints; }
    uint32_t encode();                     // total compressed bytes, 0 on failure (error() says why)
    const backend_output& get_output() const { return m_output; }
    const backend_params& get_params() const { return m_params; }
    const std::string& error() const { return m_error; }

    // per-block state after encode(), for stage-level parity tests (= m_slice_encoder_blocks, flattened in block order)
    struct encoder_block { uint32_t endpoint_index, selector_index; uint8_t endpoint_predictor; int8_t selector_history_index; };
    const std::vector<encoder_block>& encoder_blocks() const { return m_blocks; }
    const std::vector<uint32_t>& endpoint_remap_old_to_new() const { return m_endpoint_old_to_new; }
    const std::vector<uint32_t>& selector_remap_new_to_old() const { return m_selector_new_to_old; }

    // the stages of encode(), same names and order as the reference (backend.cpp:1747-1776)
    void create_endpoint_palette();
    void create_selector_palette();
    bool create_encoder_blocks();
    bool encode_image();
    bool encode_endpoint_palette();
    bool encode_selector_palette();

    struct stage_time { const char* name; double seconds; };
    const std::vector<stage_time>& stage_times() const { return m_stage_times; }

private:
    bool fail(const char* what) { m_error = what; return false; }
    void sub_time(const char* name, double seconds) { m_stage_times.push_back(stage_time{name, seconds}); }   // "~parent/part": inside the stage that follows it in the list
    bool reoptimize_and_sort_endpoints_codebook(uint32_t total_remapped, std::vector<uint32_t>& all_endpoint_indices);
    void sort_selector_codebook();
    void compute_slice_crcs();
    void precompute_block_errors(bool with_neighbours);
    int find_video_frame(size_t slice, int delta) const;

    etc1s_frontend* m_frontend = nullptr;

    bool m_frontend_state_changed = false;   // set once the backend has called back into the frontend (reoptimize_remapped_endpoints)
    backend_source m_src;
    reoptimize_fn m_reoptimize;
    backend_params m_params;
    std::vector<backend_slice_desc> m_slices;
    backend_output m_output;
    std::string m_error;

    struct endpoint_entry { uint8_t r, g, b, inten; };
    std::vector<endpoint_entry> m_endpoint_palette;
    std::vector<uint32_t> m_selector_palette;      // 16 selectors, 2 bits each, pixel y*4+x at bits 2*(y*4+x) (= etc1_selector_palette_entry::get_uint32)
    std::vector<metric::pal_colors> m_palette_colors;  // the block colours of every endpoint palette entry in the metric's basis
    std::vector<metric::sel16> m_selector_bytes;       // every selector pattern, one selector per byte
    std::vector<encoder_block> m_blocks;
    std::vector<uint8_t> m_cr_target;        // video: the next frame repeats this block, so its indices must not change any more
    // stateless per-block quantities both walks need, computed by all host threads up front (precompute_block_errors)
    std::vector<uint64_t> m_own_err;         // error of the frontend's output block under its own colours and selectors
    std::vector<uint32_t> m_own_sels;        // its selectors, packed
    std::vector<uint64_t> m_neighbour_err;   // 3 per block: error with the ORIGINAL endpoints of the left / upper / upper-left neighbour (UINT64_MAX: not computed)
    std::vector<uint32_t> m_endpoint_old_to_new, m_endpoint_new_to_old;
    std::vector<uint8_t> m_new_endpoint_was_used;
    std::vector<uint32_t> m_selector_old_to_new, m_selector_new_to_old;
    std::vector<stage_time> m_stage_times;
    std::vector<uint8_t> m_fe_endpoints;           // flattened endpoint codebook when driven by a frontend
};

// The .basis container around a backend's output = basisu_file::init + get_compressed_data (encoder/basisu_basis_file.cpp:25-388; the
// layouts are basis_file_header / basis_slice_desc / basis_key_value_data_header of transcoder/basisu_file_headers.h:32-261): header,
// optional key-value block, slice descriptors, the two palettes, the Huffman tables, the slices; little-endian packed fields, CRC-16 of
// the data and of the header. Returns an empty vector when a 32-bit file field would overflow or a key is malformed.
struct basis_key_value { std::string key; std::vector<uint8_t> value; };
std::vector<uint8_t> write_basis_file(const backend_output& out, uint32_t tex_type, uint32_t userdata0, uint32_t userdata1, bool y_flipped, uint32_t us_per_frame,
                                      const std::vector<basis_key_value>& key_values = {});

// The KTX2 container around the same output = basis_compressor::create_ktx2_file + get_dfd (encoder/basisu_comp.cpp:4636-5445; structures
// transcoder/basisu_transcoder.h:1028-1072): header, level index, data format descriptor, key-values (sorted by key, each padded to 4),
// for ETC1S the BasisLZ global data (counts, one image record per level x layer x face, palettes, tables), then the levels, smallest
// first, each the concatenation of its slices. UASTC is written without supercompression (the tool's -ktx2_no_zstandard), which needs
// the key-value block padded with a dummy key so that the level data starts 16-byte aligned. has_alpha = any source image has alpha.
std::vector<uint8_t> write_ktx2_file(const backend_output& out, uint32_t tex_type, bool has_alpha, const std::vector<basis_key_value>& key_values);

// basis_compressor::encode_slices_to_uastc_4x4_ldr's output record (comp.cpp:1843-1850, 2086-2090): a UASTC file has no codebooks, a slice
// is its blocks in raster order (16 bytes each, e.g. straight from bu_hip_encode_uastc_blocks / bu_hip_uastc_rdo) plus their CRC-16.
backend_output uastc_backend_output(const std::vector<backend_slice_desc>& slices, const uint8_t* blocks16, size_t total_blocks, bool srgb);

// palette_index_reorderer (enc.cpp:1785-1915, without a distance function): orders the palette so that entries that follow each other
// in `indices` get close numbers. Returns old -> new.
std::vector<uint32_t> reorder_palette_by_adjacency(const uint32_t* indices, uint32_t num_indices, uint32_t num_syms);

uint16_t crc16_ccitt(const void* data, size_t size, uint16_t crc);  // = basist::crc16 (transcoder.cpp:340-353)

}  // namespace bu
