/* include/basisu_hip_backend.h -- flat C view of bu::etc1s_backend (basis_universal_amd/csrc/host/etc1s_backend.h), the host-side
 * mirror of the reference's basisu_backend (encoder/basisu_backend.h:278-408): a finished ETC1S frontend in,
 * the compressed payloads of a .basis / KTX2 file out (endpoint palette, selector palette, slice Huffman tables, one bit stream
 * per slice, slice CRCs), byte-identical to basisu_backend::encode() (basisu_backend.cpp:1747-1776). Host code, lives in
 * libbasisu_frontend.so. All int-returning functions: 1 = success, 0 = failure (see bu_backend_error).
 * Not supported: global codebooks.
 */
#ifndef BASISU_HIP_BACKEND_H
#define BASISU_HIP_BACKEND_H
#include "basisu_hip_frontend.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bu_backend bu_backend;

typedef struct bu_backend_params {          /* = basisu_backend_params, backend.h:153-183 */
    float endpoint_rdo_quality_thresh;      /* basis_compressor default 1.5 (comp.h: m_endpoint_rdo_thresh)  */
    float selector_rdo_quality_thresh;      /* basis_compressor default 1.25 (comp.h: m_selector_rdo_thresh) */
    uint32_t compression_level;             /* the frontend's compression level */
    uint32_t video;                         /* 1: cBASISTexTypeVideoFrames -- slices are frames (iframe flag per slice), conditional replenishment is used */
} bu_backend_params;

typedef struct bu_backend_slice_desc {      /* = basisu_backend_slice_desc, backend.h:185-214 (the fields the ETC1S backend reads) */
    uint32_t first_block_index;
    uint32_t orig_width, orig_height;       /* texels before padding to whole blocks */
    uint32_t width, height;                 /* padded: multiples of 4 */
    uint32_t num_blocks_x, num_blocks_y;
    uint32_t source_file_index;             /* = the basis image index; only the file writer reads these four */
    uint32_t mip_index;
    uint8_t alpha, iframe, reserved[2];     /* iframe: video only, 0 otherwise (comp.cpp:3016-3030) */
} bu_backend_slice_desc;

typedef struct bu_basis_key_value {         /* = basist::key_value: key = C string of 1..255 chars, value = opaque bytes */
    const char* key;
    const uint8_t* value;
    uint32_t value_size;
} bu_basis_key_value;

/* A finished frontend as flat host arrays (the getters of basisu_frontend, frontend.h:119-156): what the backend reads when it is
 * not handed a bu_frontend. Arrays stay owned by the caller and must outlive bu_backend_encode. */
typedef struct bu_backend_arrays {
    uint32_t total_blocks;
    int perceptual;
    const bu_pixel_block* source_blocks;        /* get_source_pixel_block                          */
    const bu_etc_block* output_blocks;          /* get_output_block                                */
    const uint32_t* block_endpoint_index;       /* get_subblock_endpoint_cluster_index(block, 0)   */
    const uint32_t* block_selector_index;       /* get_block_selector_cluster_index                */
    uint32_t total_endpoints;
    const uint8_t* endpoint_color5_inten;       /* 4 bytes per cluster: r5, g5, b5, intensity table */
    uint32_t total_selectors;
    const bu_etc_block* selector_blocks;        /* get_selector_cluster_selector_bits              */
} bu_backend_arrays;

/* basis_compressor's backend parameters for a quality level (comp.cpp:3381-3420, 3537-3538): the default thresholds 1.5 / 1.25, relaxed
 * above quality 128 (x0.75 from 160, x0.5 from 192, x0.25 from 223, interpolated between 129 and 160); quality_level -1 = the defaults. */
BU_HIP_API void bu_backend_default_params(int quality_level, uint32_t compression_level, bu_backend_params* out);
BU_HIP_API bu_backend* bu_backend_create(void);
BU_HIP_API void bu_backend_destroy(bu_backend*);
/* basisu_backend::init (backend.cpp:52) on a compressed bu_frontend (which must outlive the backend) ... */
BU_HIP_API int bu_backend_init(bu_backend*, bu_frontend* frontend, const bu_backend_params*, const bu_backend_slice_desc* slices, uint32_t n_slices);
/* ... or on plain arrays. Compression levels above 1 need the frontend (basisu_frontend::reoptimize_remapped_endpoints). */
BU_HIP_API int bu_backend_init_arrays(bu_backend*, const bu_backend_arrays*, const bu_backend_params*, const bu_backend_slice_desc* slices, uint32_t n_slices);
/* With bu_backend_init_arrays the frontend call-back of compression levels above 1 (basisu_frontend::reoptimize_remapped_endpoints,
 * frontend.cpp:2996-3220) can be supplied by the host application: it receives the new endpoint cluster of every block, has to refit /
 * (when final_codebook) renumber its codebook, fill old_to_new[total_endpoints before the call] (-1: unused) and describe its state afterwards
 * in *refreshed (arrays it keeps alive until the next call or bu_backend_encode returns). Return 1 on success. */
typedef int (*bu_backend_reoptimize_fn)(void* user, const uint32_t* new_block_endpoints, uint32_t total_blocks, int32_t* old_to_new, int final_codebook,
                                        const uint32_t* block_selector_indices /* or NULL */, bu_backend_arrays* refreshed);
BU_HIP_API int bu_backend_set_reoptimize_callback(bu_backend*, bu_backend_reoptimize_fn fn, void* user);  /* after bu_backend_init_arrays */
/* basisu_backend::encode (backend.cpp:1747): total compressed bytes, 0 on failure. */
BU_HIP_API uint32_t bu_backend_encode(bu_backend*);
/* One piece of basisu_backend_output (backend.h:218-276) or of the per-block state; returns the bytes needed, copies when cap
 * suffices, ~0 for an unknown name. Names: "endpoint_palette", "selector_palette", "slice_image_tables", "slice_image_data"
 * (of slice `slice`), "slice_image_crcs" (u16 per slice), "num_endpoints", "num_selectors" (u32), and for tests "encoder_blocks"
 * (u32 x 4 per block: endpoint index, endpoint predictor, selector index, selector history index + 1),
 * "endpoint_remap_old_to_new", "selector_remap_new_to_old" (u32 each). */
BU_HIP_API uint64_t bu_backend_get(bu_backend*, const char* name, uint32_t slice, void* buf, uint64_t cap);
/* The .basis file around the encoded output = basisu_file::init + get_compressed_data (encoder/basisu_basis_file.cpp:290-388): returns the
 * file size, copies when cap suffices, 0 on failure. tex_type: basist::basis_texture_type (0 = 2D); basis_compressor passes its userdata,
 * y-flip flag, microseconds per frame and key-values (comp.cpp:3563-3609). */
BU_HIP_API uint64_t bu_backend_write_basis_file(bu_backend*, uint32_t tex_type, uint32_t userdata0, uint32_t userdata1, int y_flipped, uint32_t us_per_frame,
                                                const bu_basis_key_value* key_values, uint32_t n_key_values, void* buf, uint64_t cap);
/* The KTX2 file around the encoded output = basis_compressor::create_ktx2_file (encoder/basisu_comp.cpp:4830-5445), BasisLZ supercompression
 * scheme: header, level index, DFD, key-values, global data (codebooks + tables), levels. has_alpha: any source image has alpha. */
BU_HIP_API uint64_t bu_backend_write_ktx2_file(bu_backend*, uint32_t tex_type, int has_alpha, const bu_basis_key_value* key_values, uint32_t n_key_values,
                                               void* buf, uint64_t cap);
/* ... and around UASTC LDR 4x4 blocks, without Zstandard supercompression (the tool's -ktx2_no_zstandard). */
BU_HIP_API uint64_t bu_write_ktx2_file_uastc(const uint8_t* blocks16, uint64_t total_blocks, const bu_backend_slice_desc* slices, uint32_t n_slices, int srgb,
                                             uint32_t tex_type, int has_alpha, const bu_basis_key_value* key_values, uint32_t n_key_values, void* buf, uint64_t cap);
/* The same container around UASTC LDR 4x4 blocks (basis_compressor::encode_slices_to_uastc_4x4_ldr's output record, comp.cpp:1843-1850,
 * 2086-2090, then basisu_file::init): blocks16 = total_blocks x 16 bytes in the order the slices index them, e.g. the output of
 * bu_hip_encode_uastc_blocks / bu_hip_uastc_rdo. srgb = basis_compressor_params::m_ktx2_and_basis_srgb_transfer_function (default 1). */
BU_HIP_API uint64_t bu_write_basis_file_uastc(const uint8_t* blocks16, uint64_t total_blocks, const bu_backend_slice_desc* slices, uint32_t n_slices, int srgb,
                                              uint32_t tex_type, uint32_t userdata0, uint32_t userdata1, int y_flipped, uint32_t us_per_frame,
                                              const bu_basis_key_value* key_values, uint32_t n_key_values, void* buf, uint64_t cap);
BU_HIP_API const char* bu_backend_error(const bu_backend*);
BU_HIP_API uint32_t bu_backend_stage_times(const bu_backend*, const char** names, double* seconds, uint32_t cap);

/* Test hooks: the coding tools on their own -- a length-limited Huffman table (code sizes, codes, and its serialised form followed by
 * vlc(n, 4)) from a histogram; the slice CRC; the adjacency-driven palette ordering. */
BU_HIP_API uint64_t bu_backend_test_huffman(const uint32_t* freq, uint32_t n, uint32_t max_code_size, uint8_t* out_sizes, uint16_t* out_codes, uint8_t* out_bytes, uint64_t cap);
BU_HIP_API uint32_t bu_backend_test_crc16(const uint8_t* data, uint64_t size, uint32_t crc);
BU_HIP_API void bu_backend_test_reorder(const uint32_t* indices, uint32_t num_indices, uint32_t num_syms, uint32_t* out_old_to_new);

#ifdef __cplusplus
}
#endif
#endif
