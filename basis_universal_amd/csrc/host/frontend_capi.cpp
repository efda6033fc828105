// frontend_capi.cpp -- extern "C" wrapper of bu::etc1s_frontend (include/basisu_hip_frontend.h).
#include <cmath>
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/basisu_hip_frontend.h"
#include "etc1s_frontend.h"
#include "mipmap.h"
#include "tsvq.h"
#include "tsvq_device.h"
#include <exception>
#include <string>

// No exception may cross the C ABI (the callers are ctypes / cgo / JNI stubs): every entry point below is a function-try-block that turns a
// std::bad_alloc, std::system_error (thread creation) or anything else into the function's failure value, with the text in bu_host_last_exception().
thread_local std::string bu_last_exception_text;   // shared with backend_capi.cpp
#define g_last_exception bu_last_exception_text
extern "C" __attribute__((visibility("default"))) const char* bu_host_last_exception() { return g_last_exception.c_str(); }
#define BU_CATCH(fail_value) catch (const std::exception& e_) { g_last_exception = e_.what(); return fail_value; } catch (...) { g_last_exception = "unknown exception"; return fail_value; }
#define BU_CATCH_VOID catch (const std::exception& e_) { g_last_exception = e_.what(); } catch (...) { g_last_exception = "unknown exception"; }


struct bu_frontend {
    bu::etc1s_frontend fe;
    std::string error;
    bool video = false;
    bool fast_codebooks = false;
    uint32_t fast_iterations = 4;
    uint32_t codebook_threads = 0;
};

namespace {

template <typename V> uint64_t emit(const V& v, void* buf, uint64_t cap) {   // any contiguous container of trivially copyable elements
    const uint64_t need = (uint64_t)v.size() * sizeof(typename V::value_type);
    if (buf && cap >= need && need) std::memcpy(buf, v.data(), need);
    return need;
}
std::vector<uint32_t> csr_blob(const std::vector<std::vector<uint32_t>>& lists) {
    std::vector<uint32_t> out;
    out.push_back((uint32_t)lists.size());
    uint32_t ofs = 0;
    for (const auto& l : lists) { out.push_back(ofs); ofs += (uint32_t)l.size(); }
    out.push_back(ofs);
    for (const auto& l : lists) out.insert(out.end(), l.begin(), l.end());
    return out;
}
float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
template <typename T> T clampt(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

} // namespace

bu::etc1s_frontend* bu_frontend_object(bu_frontend* f) { return f ? &f->fe : nullptr; }  // for backend_capi.cpp

extern "C" {

bu_frontend* bu_frontend_create(void) try { return new (std::nothrow) bu_frontend(); } BU_CATCH(nullptr)
void bu_frontend_destroy(bu_frontend* f) try { delete f; } BU_CATCH_VOID
const char* bu_frontend_error(const bu_frontend* f) { return f ? f->fe.error().c_str() : "null frontend"; }

int bu_frontend_init(bu_frontend* f, bu_hip_context* ctx, const bu_pixel_block* h_blocks, const void* d_blocks, uint32_t n_blocks,
                     uint32_t max_ep, uint32_t max_sel, uint32_t level, int perceptual) try {
    if (!f) return 0;
    bu::etc1s_frontend::params p;
    p.m_num_source_blocks = n_blocks;
    p.m_pSource_blocks = h_blocks;
    p.m_pDevice_blocks = d_blocks;
    p.m_max_endpoint_clusters = max_ep;
    p.m_max_selector_clusters = max_sel;
    p.m_compression_level = level;
    p.m_perceptual = perceptual != 0;
    p.m_pHIP_context = ctx;
    p.m_video = f->video;
    p.m_fast_codebooks = f->fast_codebooks;
    p.m_fast_codebook_iterations = f->fast_iterations;
    p.m_codebook_threads = f->codebook_threads;
    return f->fe.init(p) ? 1 : 0;
} BU_CATCH(0)

int bu_frontend_set_comm(bu_frontend* f, const bu_comm* comm) try {
    if (!f) return 0;
    f->fe.set_comm(comm);
    return 1;
} BU_CATCH(0)

// A caller built against an older header passes a shorter bu_comm: what it does not carry is zero (stream_ordered = 0: the blocking convention of the first version)
int bu_frontend_set_comm_sized(bu_frontend* f, const bu_comm* comm, uint32_t struct_bytes) try {
    if (!f) return 0;
    if (!comm) { f->fe.set_comm(nullptr); return 1; }
    if (struct_bytes < offsetof(bu_comm, stream_ordered)) return 0;   // not even the first version's fields
    bu_comm c;
    std::memset(&c, 0, sizeof(c));
    std::memcpy(&c, comm, std::min<size_t>(struct_bytes, sizeof(c)));
    f->fe.set_comm(&c);
    return 1;
} BU_CATCH(0)

int bu_frontend_set_video(bu_frontend* f, int video) try { if (!f) return 0; f->video = video != 0; return 1; } BU_CATCH(0)
int bu_frontend_set_fast_codebooks(bu_frontend* f, int on, uint32_t iterations) try {
    if (!f) return 0;
    f->fast_codebooks = on != 0;
    if (iterations) f->fast_iterations = iterations;
    return 1;
} BU_CATCH(0)

int bu_frontend_set_max_threads(bu_frontend* f, uint32_t max_threads) try { if (!f) return 0; f->codebook_threads = max_threads; return 1; } BU_CATCH(0)
uint32_t bu_frontend_reference_max_threads(int multithreaded, uint32_t hardware_threads, uint32_t job_pool_threads) {
    if (!multithreaded) return 0;
    if (!hardware_threads) hardware_threads = std::max(1u, std::thread::hardware_concurrency());   // get_num_hardware_threads(), enc.h
    uint32_t t = std::min<uint32_t>(hardware_threads, 8u);                                           // cMaxCodebookCreationThreads, frontend.cpp:35
    if (job_pool_threads) t = std::min(t, job_pool_threads);
    return t;
}

int bu_frontend_compress(bu_frontend* f) try { return (f && f->fe.compress()) ? 1 : 0; } BU_CATCH(0)

int bu_frontend_call(bu_frontend* f, const char* stage, uint32_t arg) try {
    if (!f) return 0;
    const std::string n(stage);
    bu::etc1s_frontend& fe = f->fe;
    if (n == "compress") return fe.compress();
    if (n == "init_etc1_images") return fe.init_etc1_images();
    if (n == "init_endpoint_training_vectors") return fe.init_endpoint_training_vectors();
    if (n == "generate_endpoint_clusters") return fe.generate_endpoint_clusters();
    if (n == "introduce_new_endpoint_clusters") return fe.introduce_new_endpoint_clusters();
    if (n == "generate_endpoint_codebook") return fe.generate_endpoint_codebook(arg);
    if (n == "refine_endpoint_clusterization") { uint32_t moved = 0; return fe.refine_endpoint_clusterization(&moved); }
    if (n == "eliminate_redundant_or_empty_endpoint_clusters") return fe.eliminate_redundant_or_empty_endpoint_clusters();
    if (n == "generate_block_endpoint_clusters") { fe.generate_block_endpoint_clusters(); return 1; }
    if (n == "create_initial_packed_texture") return fe.create_initial_packed_texture();
    if (n == "generate_selector_clusters") return fe.generate_selector_clusters();
    if (n == "compute_selector_clusters_within_each_parent_cluster") { fe.compute_selector_clusters_within_each_parent_cluster(); return 1; }
    if (n == "create_optimized_selector_codebook") return fe.create_optimized_selector_codebook(arg);
    if (n == "find_optimal_selector_clusters_for_each_block") return fe.find_optimal_selector_clusters_for_each_block();
    if (n == "introduce_special_selector_clusters") return fe.introduce_special_selector_clusters();
    if (n == "refine_block_endpoints_given_selectors") { uint32_t refined = 0; return fe.refine_block_endpoints_given_selectors(&refined); }
    if (n == "optimize_selector_codebook") return fe.optimize_selector_codebook();
    if (n == "finalize") { fe.finalize(); return 1; }
    return 0;
} BU_CATCH(0)

uint64_t bu_frontend_get(bu_frontend* f, const char* name, void* buf, uint64_t cap) try {
    if (!f) return ~0ull;
    const std::string n(name);
    const bu::etc1s_frontend& fe = f->fe;
    if (n == "etc1_blocks") return emit(fe.etc1_blocks(), buf, cap);
    if (n == "encoded_blocks") return emit(fe.get_output_blocks(), buf, cap);
    if (n == "orig_encoded_blocks") return emit(fe.orig_encoded_blocks(), buf, cap);
    if (n == "optimized_cluster_selectors") return emit(fe.optimized_cluster_selectors(), buf, cap);
    if (n == "block_selector_cluster_index") return emit(fe.block_selector_cluster_index(), buf, cap);
    if (n == "block_endpoint_clusters_indices") return emit(fe.block_endpoint_clusters(), buf, cap);
    if (n == "endpoint_clusters") return emit(csr_blob(fe.endpoint_clusters()), buf, cap);
    if (n == "endpoint_parent_clusters") return emit(csr_blob(fe.endpoint_parent_clusters()), buf, cap);
    if (n == "selector_cluster_block_indices") return emit(csr_blob(fe.selector_cluster_block_indices()), buf, cap);
    if (n == "endpoint_cluster_etc_params") {
        const auto& P = fe.endpoint_cluster_params();
        std::vector<uint8_t> v(P.size() * 16, 0);
        for (size_t i = 0; i < P.size(); i++) {
            v[i * 16] = P[i].r; v[i * 16 + 1] = P[i].g; v[i * 16 + 2] = P[i].b; v[i * 16 + 3] = P[i].inten; v[i * 16 + 4] = P[i].valid ? 1 : 0;
            std::memcpy(&v[i * 16 + 8], &P[i].color_error, 8);
        }
        return emit(v, buf, cap);
    }
    return ~0ull;
} BU_CATCH(~0ull)

int bu_frontend_reoptimize_remapped_endpoints(bu_frontend* f, const uint32_t* new_block_endpoints, uint32_t total_blocks, int32_t* old_to_new, uint32_t old_to_new_count,
                                              int optimize_final_codebook, const uint32_t* block_selector_indices) try {
    if (!f || !new_block_endpoints || !old_to_new) return 0;
    try {
        const std::vector<uint32_t> nbe(new_block_endpoints, new_block_endpoints + total_blocks);
        std::vector<uint32_t> bsi;
        if (block_selector_indices) bsi.assign(block_selector_indices, block_selector_indices + total_blocks);
        std::vector<int> o2n;
        if (!f->fe.reoptimize_remapped_endpoints(nbe, o2n, optimize_final_codebook != 0, block_selector_indices ? &bsi : nullptr)) return 0;
        if (o2n.size() != old_to_new_count) return 0;
        for (size_t i = 0; i < o2n.size(); i++) old_to_new[i] = o2n[i];
        return 1;
    } catch (...) { return 0; }
} BU_CATCH(0)

uint32_t bu_frontend_stage_times(const bu_frontend* f, const char** names, double* seconds, uint32_t cap) try {
    if (!f) return 0;
    const auto& t = f->fe.stage_times();
    const uint32_t n = (uint32_t)std::min<size_t>(t.size(), cap);
    for (uint32_t i = 0; i < n; i++) { names[i] = t[i].name; seconds[i] = t[i].seconds; }
    return (uint32_t)t.size();
} BU_CATCH(0)

// Test hook for the host TSVQ (tsvq.h): rows must be DISTINCT and ascending (the order the reference's std::map yields).
// Blobs are [n, off_0..off_n, idx...] u32, like bu_frontend_get's cluster lists. Returns 1, 0 on failure, -1 if a blob does not fit.
int bu_host_tsvq(uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                 uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words) {
    return bu_host_tsvq_mt(dim, rows, weights, n, max_codebook_size, max_parent_codebook_size, 0, 0, out_codebook, cap_codebook_words, out_parent, cap_parent_words);
}
int bu_host_tsvq_mt(uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                    uint32_t max_threads, uint32_t min_unique_for_threads, uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words) try {
    if (!min_unique_for_threads) min_unique_for_threads = bu::kThreadedCodebookMinUnique;
    std::vector<float> r(rows, rows + (size_t)n * dim);
    std::vector<uint64_t> w(weights, weights + n);
    std::vector<std::vector<uint32_t>> groups(n), codebook, parents;
    for (uint32_t i = 0; i < n; i++) groups[i].push_back(i);
    bool ok = false;
    if (dim == 6) ok = bu::hierarchical_codebook<6>(r, w, groups, max_codebook_size, max_parent_codebook_size, codebook, parents, max_threads, min_unique_for_threads);
    else if (dim == 16) ok = bu::hierarchical_codebook<16>(r, w, groups, max_codebook_size, max_parent_codebook_size, codebook, parents, max_threads, min_unique_for_threads);
    if (!ok) return 0;
    const std::vector<uint32_t> a = csr_blob(codebook), b = csr_blob(parents);
    if (a.size() > cap_codebook_words || b.size() > cap_parent_words) return -1;
    std::memcpy(out_codebook, a.data(), a.size() * 4);
    std::memcpy(out_parent, b.data(), b.size() * 4);
    return 1;
} BU_CATCH(0)

// The same through the device TSVQ (tsvq_device.h + bu_hip_tsvq_*): what the frontend actually uses. stats3 = {rounds, splits computed, splits used}.
int bu_device_tsvq(bu_hip_context* ctx, uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                   uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words, uint32_t* stats3) {
    return bu_device_tsvq_mt(ctx, dim, rows, weights, n, max_codebook_size, max_parent_codebook_size, 0, 0, out_codebook, cap_codebook_words, out_parent, cap_parent_words, stats3);
}
int bu_device_tsvq_mt(bu_hip_context* ctx, uint32_t dim, const float* rows, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                      uint32_t max_threads, uint32_t min_unique_for_threads, uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words,
                      uint32_t* stats3) try {
    if (!min_unique_for_threads) min_unique_for_threads = bu::kThreadedCodebookMinUnique;
    std::vector<float> r(rows, rows + (size_t)n * dim);
    std::vector<uint64_t> w(weights, weights + n);
    std::vector<std::vector<uint32_t>> groups(n), codebook, parents;
    for (uint32_t i = 0; i < n; i++) groups[i].push_back(i);
    bu::device_tsvq::stats st;
    bool packable = dim == 16;
    for (size_t i = 0; packable && i < r.size(); i++) packable = r[i] == 0.0f || r[i] == 1.0f || r[i] == 2.0f || r[i] == 3.0f;
    if (packable && (stats3 && stats3[0] == 0xBACCED)) { // caller asked for the packed selector path
        std::vector<uint32_t> keys(n, 0);
        for (uint32_t i = 0; i < n; i++) for (uint32_t k = 0; k < 16; k++) keys[i] = (keys[i] << 2) | (uint32_t)r[(size_t)i * 16 + k];
        if (!bu::device_tsvq::hierarchical_codebook_packed16(ctx, keys, w, bu::vec_groups{&groups}, max_codebook_size, max_parent_codebook_size, codebook, parents, &st,
                                                             max_threads, min_unique_for_threads)) return 0;
    } else if (!bu::device_tsvq::hierarchical_codebook(ctx, dim, r, w, bu::vec_groups{&groups}, max_codebook_size, max_parent_codebook_size, codebook, parents, &st,
                                                       nullptr, nullptr, nullptr, nullptr, nullptr, max_threads, min_unique_for_threads)) return 0;
    if (stats3) { stats3[0] = st.rounds; stats3[1] = st.splits_computed; stats3[2] = st.splits_used; }
    const std::vector<uint32_t> a = csr_blob(codebook), b = csr_blob(parents);
    if (a.size() > cap_codebook_words || b.size() > cap_parent_words) return -1;
    std::memcpy(out_codebook, a.data(), a.size() * 4);
    std::memcpy(out_parent, b.data(), b.size() * 4);
    return 1;
} BU_CATCH(0)

// comp.cpp:3310-3379, float arithmetic in the reference's order
int bu_generate_mipmap_level(bu_hip_context* ctx, const void* d_src, uint32_t src_w, uint32_t src_h, void* d_dst, uint32_t dst_w, uint32_t dst_h, int srgb,
                             const char* filter, float filter_scale, int wrapping, uint32_t num_comps) try {
    bu::mip::plan p;
    if (!ctx || !filter || !bu::mip::make_plan(p, src_w, src_h, dst_w, dst_h, srgb != 0, filter, filter_scale, wrapping != 0)) return 0;
    return bu_hip_k_resample_rgba8(ctx, d_src, src_w, src_h, d_dst, dst_w, dst_h, p.x.first.data(), p.x.pixel.data(), p.x.weight.data(), p.y.first.data(), p.y.pixel.data(),
                                   p.y.weight.data(), p.x_after_y, srgb, p.srgb_to_linear, p.linear_to_srgb, num_comps);
} BU_CATCH(0)

uint32_t bu_mipmap_level_sizes(uint32_t w, uint32_t h, uint32_t smallest_dimension, uint32_t* out_wh, uint32_t cap) try {
    const auto sizes = bu::mip::level_sizes(w, h, smallest_dimension ? smallest_dimension : 1);
    for (uint32_t i = 0; i < sizes.size() && i < cap; i++) { out_wh[i * 2] = sizes[i].first; out_wh[i * 2 + 1] = sizes[i].second; }
    return (uint32_t)sizes.size();
} BU_CATCH(0)

int bu_mipmap_plan(uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, int srgb, const char* filter, float filter_scale, int wrapping, uint32_t* out_counts,
                   uint32_t* x_first, uint16_t* x_pixel, float* x_weight, uint32_t* y_first, uint16_t* y_pixel, float* y_weight, float* to_linear, uint8_t* to_srgb) try {
    bu::mip::plan p;
    if (!filter || !out_counts || !bu::mip::make_plan(p, src_w, src_h, dst_w, dst_h, srgb != 0, filter, filter_scale, wrapping != 0)) return 0;
    out_counts[0] = p.x.ops(); out_counts[1] = p.y.ops(); out_counts[2] = p.x_after_y; out_counts[3] = 0;
    auto copy = [](auto* dst, const auto& v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
    copy(x_first, p.x.first); copy(x_pixel, p.x.pixel); copy(x_weight, p.x.weight); copy(y_first, p.y.first); copy(y_pixel, p.y.pixel); copy(y_weight, p.y.weight);
    if (to_linear) std::memcpy(to_linear, p.srgb_to_linear, sizeof(p.srgb_to_linear));
    if (to_srgb) std::memcpy(to_srgb, p.linear_to_srgb, sizeof(p.linear_to_srgb));
    return 1;
} BU_CATCH(0)

void bu_etc1s_quality_to_clusters(int quality_level, uint32_t total_blocks, uint32_t* out_ep, uint32_t* out_sel) try {
    const double total_texels = total_blocks * 16.0f;
    const float quality = clampf(quality_level / 255.0f, 0.0f, 1.0f);
    const float bits_per_cluster = 14.0f;
    int max_endpoints = static_cast<int>((1.0f * total_texels) / bits_per_cluster);
    const float mid = 128.0f / 255.0f;
    float q = quality;
    uint32_t endpoint_clusters;
    if (q <= mid) {
        q = 0.0f + (0.5f - 0.0f) * powf(q / mid, .65f);
        max_endpoints = clampt<int>(max_endpoints, 256, 4800);
        max_endpoints = (int)std::min<uint32_t>((uint32_t)max_endpoints, total_blocks);
        if (max_endpoints < 64) max_endpoints = 64;
        endpoint_clusters = clampt<uint32_t>((uint32_t)(.5f + (32.0f + (static_cast<float>(max_endpoints) - 32.0f) * q)), 32, 16128);
    } else {
        q = powf((q - mid) / (1.0f - mid), 1.6f);
        max_endpoints = clampt<int>(max_endpoints, 256, 8192);
        max_endpoints = (int)std::min<uint32_t>((uint32_t)max_endpoints, total_blocks);
        if (max_endpoints < 4800) max_endpoints = 4800;
        endpoint_clusters = clampt<uint32_t>((uint32_t)(.5f + (4800.0f + (static_cast<float>(max_endpoints) - 4800.0f) * q)), 32, 16128);
    }
    int max_selectors = static_cast<int>((1.0f * total_texels) / bits_per_cluster);
    max_selectors = clampt<int>(max_selectors, 256, 16128);
    max_selectors = (int)std::min<uint32_t>((uint32_t)max_selectors, total_blocks);
    const float sq = powf(quality, 2.62f);
    if (max_selectors < 96) max_selectors = 96;
    const uint32_t selector_clusters = clampt<uint32_t>((uint32_t)(.5f + (96.0f + (static_cast<float>(max_selectors) - 96.0f) * sq)), 8, 16128);
    *out_ep = endpoint_clusters;
    *out_sel = selector_clusters;
} BU_CATCH_VOID

} // extern "C"
