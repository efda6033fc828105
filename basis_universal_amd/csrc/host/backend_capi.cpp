// backend_capi.cpp -- extern "C" wrapper of bu::etc1s_backend (include/basisu_hip_backend.h).
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../../include/basisu_hip_backend.h"
#include "entropy.h"
#include "etc1s_backend.h"
#include "etc1s_frontend.h"
#include <exception>
#include <string>

// No exception may cross the C ABI (the callers are ctypes / cgo / JNI stubs): every entry point below is a function-try-block that turns a
// std::bad_alloc, std::system_error (thread creation) or anything else into the function's failure value, with the text in bu_host_last_exception().
extern thread_local std::string bu_last_exception_text;   // frontend_capi.cpp
#define g_last_exception bu_last_exception_text
#define BU_CATCH(fail_value) catch (const std::exception& e_) { g_last_exception = e_.what(); return fail_value; } catch (...) { g_last_exception = "unknown exception"; return fail_value; }
#define BU_CATCH_VOID catch (const std::exception& e_) { g_last_exception = e_.what(); } catch (...) { g_last_exception = "unknown exception"; }


struct bu_frontend;
bu::etc1s_frontend* bu_frontend_object(bu_frontend*);  // frontend_capi.cpp

struct bu_backend {
    bu::etc1s_backend be;
};

namespace {

template <typename T> uint64_t emit(const std::vector<T>& v, void* buf, uint64_t cap) {
    const uint64_t need = (uint64_t)v.size() * sizeof(T);
    if (buf && cap >= need && need) std::memcpy(buf, v.data(), need);
    return need;
}

void convert(const bu_backend_params* p, const bu_backend_slice_desc* s, uint32_t n, bu::backend_params& bp, std::vector<bu::backend_slice_desc>& slices) {
    bp.m_etc1s = true;
    bp.m_endpoint_rdo_quality_thresh = p->endpoint_rdo_quality_thresh;
    bp.m_selector_rdo_quality_thresh = p->selector_rdo_quality_thresh;
    bp.m_compression_level = p->compression_level;
    bp.m_video = p->video != 0;
    slices.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        bu::backend_slice_desc& d = slices[i];
        d.m_first_block_index = s[i].first_block_index;
        d.m_orig_width = s[i].orig_width; d.m_orig_height = s[i].orig_height;
        d.m_width = s[i].width; d.m_height = s[i].height;
        d.m_num_blocks_x = s[i].num_blocks_x; d.m_num_blocks_y = s[i].num_blocks_y;
        d.m_num_macroblocks_x = (d.m_num_blocks_x + 1) / 2; d.m_num_macroblocks_y = (d.m_num_blocks_y + 1) / 2;
        d.m_source_file_index = s[i].source_file_index; d.m_mip_index = s[i].mip_index;
        d.m_alpha = s[i].alpha != 0; d.m_iframe = s[i].iframe != 0;
    }
}

}  // namespace

extern "C" {

void bu_backend_default_params(int quality_level, uint32_t compression_level, bu_backend_params* out) try {
    if (!out) return;
    float scale = 1.0f;
    if (quality_level != -1) {
        const float q = quality_level / 255.0f, quality = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
        if (quality_level >= 223) scale = .25f;
        else if (quality_level >= 192) scale = .5f;
        else if (quality_level >= 160) scale = .75f;
        else if (quality_level >= 129) { const float l = (quality - 129 / 255.0f) / ((160 - 129) / 255.0f); scale = 1.0f + (.75f - 1.0f) * l; }
    }
    out->endpoint_rdo_quality_thresh = 1.5f * scale;
    out->selector_rdo_quality_thresh = 1.25f * scale;
    out->compression_level = compression_level;
    out->video = 0;
} BU_CATCH_VOID

bu_backend* bu_backend_create(void) try { return new (std::nothrow) bu_backend(); } BU_CATCH(nullptr)
void bu_backend_destroy(bu_backend* b) try { delete b; } BU_CATCH_VOID
const char* bu_backend_error(const bu_backend* b) { return b ? b->be.error().c_str() : "null backend"; }

int bu_backend_init(bu_backend* b, bu_frontend* frontend, const bu_backend_params* p, const bu_backend_slice_desc* s, uint32_t n) try {
    if (!b || !frontend || !p || (!s && n)) return 0;
    bu::backend_params bp;
    std::vector<bu::backend_slice_desc> slices;
    convert(p, s, n, bp, slices);
    b->be.init(bu_frontend_object(frontend), bp, slices);
    return 1;
} BU_CATCH(0)

int bu_backend_init_arrays(bu_backend* b, const bu_backend_arrays* a, const bu_backend_params* p, const bu_backend_slice_desc* s, uint32_t n) try {
    if (!b || !a || !p || (!s && n)) return 0;
    bu::backend_params bp;
    std::vector<bu::backend_slice_desc> slices;
    convert(p, s, n, bp, slices);
    bu::backend_source src;
    src.total_blocks = a->total_blocks;
    src.perceptual = a->perceptual != 0;
    src.source_blocks = a->source_blocks;
    src.output_blocks = a->output_blocks;
    src.block_endpoint_index = a->block_endpoint_index;
    src.block_selector_index = a->block_selector_index;
    src.total_endpoints = a->total_endpoints;
    src.endpoint_color5_inten = a->endpoint_color5_inten;
    src.total_selectors = a->total_selectors;
    src.selector_blocks = a->selector_blocks;
    b->be.init(src, bp, slices);
    return 1;
} BU_CATCH(0)

namespace {
bu::backend_source to_source(const bu_backend_arrays* a) {
    bu::backend_source src;
    src.total_blocks = a->total_blocks;
    src.perceptual = a->perceptual != 0;
    src.source_blocks = a->source_blocks;
    src.output_blocks = a->output_blocks;
    src.block_endpoint_index = a->block_endpoint_index;
    src.block_selector_index = a->block_selector_index;
    src.total_endpoints = a->total_endpoints;
    src.endpoint_color5_inten = a->endpoint_color5_inten;
    src.total_selectors = a->total_selectors;
    src.selector_blocks = a->selector_blocks;
    return src;
}
}  // namespace

int bu_backend_set_reoptimize_callback(bu_backend* b, bu_backend_reoptimize_fn fn, void* user) try {
    if (!b) return 0;
    if (!fn) { b->be.set_reoptimize(nullptr); return 1; }
    bu::etc1s_backend* be = &b->be;
    b->be.set_reoptimize([fn, user, be](const std::vector<uint32_t>& new_block_endpoints, std::vector<int>& old_to_new, bool final_codebook,
                                        const std::vector<uint32_t>* block_selector_indices, bu::backend_source& src) {
        old_to_new.assign(be->total_endpoints(), -1);
        bu_backend_arrays refreshed;
        std::memset(&refreshed, 0, sizeof(refreshed));
        if (!fn(user, new_block_endpoints.data(), (uint32_t)new_block_endpoints.size(), old_to_new.data(), final_codebook ? 1 : 0,
                block_selector_indices ? block_selector_indices->data() : nullptr, &refreshed))
            return false;
        if (!refreshed.total_blocks || !refreshed.source_blocks || !refreshed.output_blocks || !refreshed.block_endpoint_index || !refreshed.block_selector_index ||
            !refreshed.endpoint_color5_inten || !refreshed.selector_blocks)
            return false;
        src = to_source(&refreshed);
        return true;
    });
    return 1;
} BU_CATCH(0)

uint32_t bu_backend_encode(bu_backend* b) try { return b ? b->be.encode() : 0; } BU_CATCH(0)

namespace {
std::vector<bu::basis_key_value> to_kvs(const bu_basis_key_value* kvs, uint32_t n) {
    std::vector<bu::basis_key_value> kv(n);
    for (uint32_t i = 0; i < n; i++) {
        kv[i].key = kvs[i].key ? kvs[i].key : "";
        if (kvs[i].value_size) kv[i].value.assign(kvs[i].value, kvs[i].value + kvs[i].value_size);
    }
    return kv;
}
}  // namespace

uint64_t bu_backend_write_ktx2_file(bu_backend* b, uint32_t tex_type, int has_alpha, const bu_basis_key_value* kvs, uint32_t n_kvs, void* buf, uint64_t cap) try {
    if (!b || (!kvs && n_kvs)) return 0;
    return emit(bu::write_ktx2_file(b->be.get_output(), tex_type, has_alpha != 0, to_kvs(kvs, n_kvs)), buf, cap);
} BU_CATCH(0)

uint64_t bu_write_ktx2_file_uastc(const uint8_t* blocks16, uint64_t total_blocks, const bu_backend_slice_desc* s, uint32_t n, int srgb, uint32_t tex_type, int has_alpha,
                                  const bu_basis_key_value* kvs, uint32_t n_kvs, void* buf, uint64_t cap) try {
    if (!blocks16 || !s || !n || (!kvs && n_kvs)) return 0;
    bu_backend_params unused = {0, 0, 0, 0};
    bu::backend_params bp;
    std::vector<bu::backend_slice_desc> slices;
    convert(&unused, s, n, bp, slices);
    const bu::backend_output out = bu::uastc_backend_output(slices, blocks16, (size_t)total_blocks, srgb != 0);
    if (out.m_slice_desc.empty()) return 0;
    return emit(bu::write_ktx2_file(out, tex_type, has_alpha != 0, to_kvs(kvs, n_kvs)), buf, cap);
} BU_CATCH(0)

uint64_t bu_write_basis_file_uastc(const uint8_t* blocks16, uint64_t total_blocks, const bu_backend_slice_desc* s, uint32_t n, int srgb, uint32_t tex_type,
                                   uint32_t userdata0, uint32_t userdata1, int y_flipped, uint32_t us_per_frame, const bu_basis_key_value* kvs, uint32_t n_kvs,
                                   void* buf, uint64_t cap) try {
    if (!blocks16 || !s || !n || (!kvs && n_kvs)) return 0;
    bu_backend_params unused = {0, 0, 0, 0};
    bu::backend_params bp;
    std::vector<bu::backend_slice_desc> slices;
    convert(&unused, s, n, bp, slices);
    std::vector<bu::basis_key_value> kv(n_kvs);
    for (uint32_t i = 0; i < n_kvs; i++) {
        kv[i].key = kvs[i].key ? kvs[i].key : "";
        if (kvs[i].value_size) kv[i].value.assign(kvs[i].value, kvs[i].value + kvs[i].value_size);
    }
    const bu::backend_output out = bu::uastc_backend_output(slices, blocks16, (size_t)total_blocks, srgb != 0);
    if (out.m_slice_desc.empty()) return 0;
    return emit(bu::write_basis_file(out, tex_type, userdata0, userdata1, y_flipped != 0, us_per_frame, kv), buf, cap);
} BU_CATCH(0)

uint64_t bu_backend_write_basis_file(bu_backend* b, uint32_t tex_type, uint32_t userdata0, uint32_t userdata1, int y_flipped, uint32_t us_per_frame,
                                     const bu_basis_key_value* kvs, uint32_t n_kvs, void* buf, uint64_t cap) try {
    if (!b || (!kvs && n_kvs)) return 0;
    std::vector<bu::basis_key_value> kv(n_kvs);
    for (uint32_t i = 0; i < n_kvs; i++) {
        kv[i].key = kvs[i].key ? kvs[i].key : "";
        if (kvs[i].value_size) kv[i].value.assign(kvs[i].value, kvs[i].value + kvs[i].value_size);
    }
    return emit(bu::write_basis_file(b->be.get_output(), tex_type, userdata0, userdata1, y_flipped != 0, us_per_frame, kv), buf, cap);
} BU_CATCH(0)

uint64_t bu_backend_get(bu_backend* b, const char* name, uint32_t slice, void* buf, uint64_t cap) try {
    if (!b) return ~0ull;
    const std::string n(name);
    const bu::backend_output& o = b->be.get_output();
    if (n == "endpoint_palette") return emit(o.m_endpoint_palette, buf, cap);
    if (n == "selector_palette") return emit(o.m_selector_palette, buf, cap);
    if (n == "slice_image_tables") return emit(o.m_slice_image_tables, buf, cap);
    if (n == "slice_image_data") return slice < o.m_slice_image_data.size() ? emit(o.m_slice_image_data[slice], buf, cap) : ~0ull;
    if (n == "slice_image_crcs") return emit(o.m_slice_image_crcs, buf, cap);
    if (n == "num_endpoints") return emit(std::vector<uint32_t>{o.m_num_endpoints}, buf, cap);
    if (n == "num_selectors") return emit(std::vector<uint32_t>{o.m_num_selectors}, buf, cap);
    if (n == "encoder_blocks") {
        const auto& B = b->be.encoder_blocks();
        std::vector<uint32_t> v(B.size() * 4);
        for (size_t i = 0; i < B.size(); i++) { v[i * 4] = B[i].endpoint_index; v[i * 4 + 1] = B[i].endpoint_predictor; v[i * 4 + 2] = B[i].selector_index; v[i * 4 + 3] = (uint32_t)(B[i].selector_history_index + 1); }
        return emit(v, buf, cap);
    }
    if (n == "endpoint_remap_old_to_new") return emit(b->be.endpoint_remap_old_to_new(), buf, cap);
    if (n == "selector_remap_new_to_old") return emit(b->be.selector_remap_new_to_old(), buf, cap);
    return ~0ull;
} BU_CATCH(~0ull)

uint32_t bu_backend_stage_times(const bu_backend* b, const char** names, double* seconds, uint32_t cap) try {
    if (!b) return 0;
    const auto& t = b->be.stage_times();
    for (uint32_t i = 0; i < t.size() && i < cap; i++) { names[i] = t[i].name; seconds[i] = t[i].seconds; }
    return (uint32_t)t.size();
} BU_CATCH(0)

// ---- test hooks: the coding tools on their own (tests/test_backend_host.py diffs them against the reference's)
uint64_t bu_backend_test_huffman(const uint32_t* freq, uint32_t n, uint32_t max_code_size, uint8_t* out_sizes, uint16_t* out_codes, uint8_t* out_bytes, uint64_t cap) try {
    bu::huffman_table t;
    if (!t.init(freq, n, max_code_size)) return ~0ull;
    for (uint32_t i = 0; i < n; i++) { out_sizes[i] = t.sizes()[i]; out_codes[i] = t.codes()[i]; }
    bu::bit_writer w;
    w.restart();
    if (!w.put_table(t)) return ~0ull;
    w.put_vlc(n, 4);
    w.flush();
    return emit(w.bytes(), out_bytes, cap);
} BU_CATCH(0)
uint32_t bu_backend_test_crc16(const uint8_t* data, uint64_t size, uint32_t crc) try { return bu::crc16_ccitt(data, (size_t)size, (uint16_t)crc); } BU_CATCH(0)
void bu_backend_test_reorder(const uint32_t* indices, uint32_t num_indices, uint32_t num_syms, uint32_t* out_old_to_new) try {
    const std::vector<uint32_t> r = bu::reorder_palette_by_adjacency(indices, num_indices, num_syms);
    std::memcpy(out_old_to_new, r.data(), r.size() * sizeof(uint32_t));
} BU_CATCH_VOID

}  // extern "C"
