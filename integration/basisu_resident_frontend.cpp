// integration/basisu_resident_frontend.cpp -- the translation unit a maintainer of the reference compiles INSTEAD OF
// encoder/basisu_frontend.cpp to put the whole ETC1S frontend on the MI355X ("resident path"): the four member
// functions of basisu::basisu_frontend that other translation units call (basis_compressor: init / compress / dump_debug_image,
// comp.cpp:3441-3469; basisu_backend: reoptimize_remapped_endpoints, backend.cpp) implemented on libbasisu_frontend.so
// (include/basisu_hip_frontend.h). After compress() the object holds exactly what the reference's own compress() leaves behind
// for its getters (encoder/basisu_frontend.h:119-156), so basis_compressor and -- if it is kept -- the reference's basisu_backend
// read it unchanged. This file is OURS and contains no reference code; it only includes the reference's headers.
//
// The tiles travel to the GPU once (bu_frontend_init uploads them); every stage of basisu_frontend::compress (frontend.cpp:159-316)
// then runs device-resident. There is no CPU fallback behind this file: a failing device call fails compress().
//
// Images in flight. Under basis_parallel_compress (comp.cpp:5466-5559) N compressors run on N host threads, each with a frontend of its own, each blocked in
// its device calls. BU_RESIDENT_LANES=L (L > 0) sends them all through ONE bu_frontend_pipeline per GPU instead (include/basisu_hip_frontend.h): compress() submits
// its image and sleeps until the pipeline's single driver thread has taken it through init + compress as one of L cooperative tasks. Off by default, measured
// (16 images in flight on a 16-core box, 4096^2): 546 Mpix/s direct against 348 (L = 4) / 312 (L = 6) through the pipeline -- under this driver every image
// brings 64 MiB of host tiles to stage and hands its results back as host arrays, and that per-image host work, spread over 16 threads in the direct form, lands
// on the pipeline's one thread. The pipeline wins where it was built for: tiles resident, results collected by the caller (bench.py `pipelined`: 1,563 against
// 1,447 Mpix/s with a host thread per image, at a third of the host CPU).
#include "encoder/basisu_frontend.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "basisu_resident.h"

namespace {

struct registry {
    // One entry per live basisu_frontend object. The reference class has no destructor hook, so an entry ends in one of three ways: the same
    // object is init()ed again, init() fails (erased at once), or the backend is done with it (bu_resident_release at the end of
    // basisu_backend::encode). `blocks` is the per-init token: a basisu_frontend constructed at a recycled address whose init() never ran or
    // failed cannot resolve to the previous object's state, because find() is given the caller's own m_total_blocks.
    struct entry { bu_frontend* f; uint32_t blocks; bu_frontend_pipeline* pipe; };   // pipe != null: f belongs to that pipeline (null until compress() has collected it)
    std::mutex lock;
    std::unordered_map<const basisu::basisu_frontend*, entry> map;
    std::map<int, bu_frontend_pipeline*> pipes;   // one per device
    bu_hip_context* own_ctx = nullptr;   // used when the caller did not ask for the accelerator seam (no -opencl): the resident build always runs on the GPU
    static void drop(entry& e) {
        if (!e.f) return;
        if (e.pipe) bu_frontend_pipeline_release(e.pipe, e.f); else bu_frontend_destroy(e.f);
        e.f = nullptr;
    }
    ~registry() {
        for (auto& kv : map) drop(kv.second);
        for (auto& kv : pipes) bu_frontend_pipeline_destroy(kv.second);
        if (own_ctx) bu_hip_destroy_context(own_ctx);
    }
    static uint32_t lanes() {
        static const uint32_t n = [] { const char* e = std::getenv("BU_RESIDENT_LANES"); const long v = e ? std::atol(e) : 0; return (uint32_t)(v < 0 ? 0 : (v > 16 ? 16 : v)); }();
        return n;
    }
    bu_frontend_pipeline* pipeline(int device) {
        std::lock_guard<std::mutex> g(lock);
        auto it = pipes.find(device);
        if (it != pipes.end()) return it->second;
        bu_frontend_pipeline* p = bu_frontend_pipeline_create(device, lanes());
        if (p) pipes[device] = p;
        return p;
    }
    bu_frontend* fresh(const basisu::basisu_frontend* key, uint32_t blocks) {
        std::lock_guard<std::mutex> g(lock);
        entry& slot = map[key];
        drop(slot);
        slot.f = bu_frontend_create();
        slot.blocks = blocks; slot.pipe = nullptr;
        if (!slot.f) { map.erase(key); return nullptr; }
        return slot.f;
    }
    // pipeline mode: init() only books the object; compress() brings the frontend
    void book(const basisu::basisu_frontend* key, uint32_t blocks, bu_frontend_pipeline* pipe) {
        std::lock_guard<std::mutex> g(lock);
        entry& slot = map[key];
        drop(slot);
        slot.blocks = blocks; slot.pipe = pipe;
    }
    bu_frontend_pipeline* booked(const basisu::basisu_frontend* key, uint32_t blocks) {
        std::lock_guard<std::mutex> g(lock);
        auto it = map.find(key);
        return (it == map.end() || it->second.blocks != blocks || it->second.f) ? nullptr : it->second.pipe;
    }
    void collected(const basisu::basisu_frontend* key, bu_frontend* f) {
        std::lock_guard<std::mutex> g(lock);
        auto it = map.find(key);
        if (it != map.end()) it->second.f = f;
    }
    bu_frontend* find(const basisu::basisu_frontend* key, uint32_t blocks) {
        std::lock_guard<std::mutex> g(lock);
        auto it = map.find(key);
        return (it == map.end() || it->second.blocks != blocks) ? nullptr : it->second.f;
    }
    void release(const basisu::basisu_frontend* key) {
        std::lock_guard<std::mutex> g(lock);
        auto it = map.find(key);
        if (it == map.end()) return;
        drop(it->second);   // gives the device buffers (and, in pipeline mode, the context the image ran on) back
        map.erase(it);
    }
    bu_hip_context* context() {
        std::lock_guard<std::mutex> g(lock);
        if (!own_ctx) { bu_hip_init(0); own_ctx = bu_hip_create_context(); }
        return own_ctx;
    }
};
registry& reg() { static registry r; return r; }

template <typename T> bool fetch(bu_frontend* f, const char* name, std::vector<T>& out) {
    const uint64_t need = bu_frontend_get(f, name, nullptr, 0);
    if (need == ~0ull || need % sizeof(T)) return false;
    out.resize(need / sizeof(T));
    return need == 0 || bu_frontend_get(f, name, out.data(), need) == need;
}

} // namespace

bu_frontend* bu_resident_handle(const basisu::basisu_frontend* fe) { return fe ? reg().find(fe, fe->get_total_output_blocks()) : nullptr; }
void bu_resident_release(const basisu::basisu_frontend* fe) { if (fe) reg().release(fe); }

namespace basisu {

// the accelerator context the reference hands around is the shim's wrapper (integration/basisu_hip_shim.cpp)
struct opencl_context { bu_hip_context* h; };

bool basisu_frontend::init(const params& p) {
    if ((p.m_max_endpoint_clusters < 1) || (p.m_max_endpoint_clusters > cMaxEndpointClusters)) return false;
    if ((p.m_max_selector_clusters < 1) || (p.m_max_selector_clusters > cMaxSelectorClusters)) return false;
    if (p.m_pGlobal_codebooks) { error_printf("basisu_frontend (resident): global codebooks are not supported\n"); return false; }
    m_params = p;
    m_total_blocks = p.m_num_source_blocks;
    m_total_pixels = p.m_num_source_blocks * 16;
    m_opencl_failed = false;
    // get_source_pixel_block (frontend.h:121) serves the backend from the frontend's own copy
    m_source_blocks.resize(0);
    append_vector(m_source_blocks, p.m_pSource_blocks, p.m_num_source_blocks);
    m_encoded_blocks.resize(p.m_num_source_blocks);

    bu_hip_context* ctx = p.m_pOpenCL_context ? p.m_pOpenCL_context->h : reg().context();
    if (!ctx) { error_printf("basisu_frontend (resident): no HIP context\n"); return false; }
    if (registry::lanes()) {   // images in flight: the work happens in compress(), as one task of the device's pipeline
        bu_frontend_pipeline* pipe = reg().pipeline(bu_hip_context_device(ctx));
        if (!pipe) { error_printf("basisu_frontend (resident): no frontend pipeline: %s\n", bu_host_last_exception()); return false; }
        reg().book(this, m_total_blocks, pipe);
        return true;
    }
    bu_frontend* f = reg().fresh(this, m_total_blocks);
    if (!f) return false;
    static_assert(sizeof(pixel_block) == sizeof(bu_pixel_block) && sizeof(etc_block) == sizeof(bu_etc_block), "layout");
    bu_frontend_set_video(f, p.m_tex_type == basist::cBASISTexTypeVideoFrames);
    // the codebook builders' thread count exactly as frontend.cpp:873-876 / 2195-2198 derive it: the multi-threaded tool's files come out of T-way partitioned trees
    bu_frontend_set_max_threads(f, bu_frontend_reference_max_threads(p.m_multithreaded, get_num_hardware_threads(), p.m_pJob_pool ? (uint32_t)p.m_pJob_pool->get_total_threads() : 0));
    if (!bu_frontend_init(f, ctx, reinterpret_cast<const bu_pixel_block*>(m_source_blocks.data()), nullptr, p.m_num_source_blocks, p.m_max_endpoint_clusters,
                          p.m_max_selector_clusters, p.m_compression_level, p.m_perceptual)) {
        error_printf("basisu_frontend (resident): init failed: %s\n", bu_frontend_error(f));
        reg().release(this);   // a later compress() / backend on this object finds nothing instead of a half-initialised frontend
        return false;
    }
    return true;
}

// Copies the resident frontend's state into the members the reference's getters read: every array straight into its destination (no staging copy), the cluster lists sized from a
// count pass before they are filled (2 M push_backs into 2,400 growing vectors were 8 ms of a 4096^2 image).
template <typename V> static bool fetch_into(bu_frontend* f, const char* name, V& out, size_t elem_bytes, size_t want_elems) {
    const uint64_t need = bu_frontend_get(f, name, nullptr, 0);
    if (need == ~0ull || need != (uint64_t)want_elems * elem_bytes) return false;
    out.resize(want_elems);
    return need == 0 || bu_frontend_get(f, name, out.data(), need) == need;
}
static bool refresh_members(bu_frontend* f, uint32_t n, etc_block_vec& encoded, etc_block_vec* etc1s, basisu::vector<uint_vec>& endpoint_clusters,
                            basisu::vector<vec2U>& block_endpoints, std::vector<uint8_t>& ep_params, basisu::vector<uint_vec>& selector_lists,
                            basisu::vector<etc_block>& selector_blocks, basisu::vector<uint32_t>& block_selectors) {
    static_assert(sizeof(etc_block) == 8 && sizeof(vec2U) == 8, "layout");
    std::vector<uint8_t> raw;
    std::vector<uint32_t> u;
    if (!fetch_into(f, "encoded_blocks", encoded, 8, n)) return false;
    if (etc1s && !fetch_into(f, "etc1_blocks", *etc1s, 8, n)) return false;
    if (!fetch(f, "block_endpoint_clusters_indices", u) || u.size() != n) return false;
    block_endpoints.resize(n);
    for (uint32_t i = 0; i < n; i++) { block_endpoints[i][0] = u[i]; block_endpoints[i][1] = u[i]; }
    if (!fetch(f, "endpoint_cluster_etc_params", ep_params) || ep_params.size() % 16) return false;
    // the lists themselves (training-vector ids 2b, 2b+1 in block order is all a consumer outside the frontend can rely on)
    const uint32_t k = (uint32_t)(ep_params.size() / 16);
    std::vector<uint32_t> count(k, 0);
    for (uint32_t i = 0; i < n; i++) { if (u[i] >= k) return false; count[u[i]]++; }
    endpoint_clusters.resize(0); endpoint_clusters.resize(k);
    for (uint32_t c = 0; c < k; c++) { endpoint_clusters[c].resize(2 * count[c]); count[c] = 0; }
    for (uint32_t i = 0; i < n; i++) { uint_vec& l = endpoint_clusters[u[i]]; uint32_t& at = count[u[i]]; l[at] = i * 2; l[at + 1] = i * 2 + 1; at += 2; }
    if (!fetch(f, "optimized_cluster_selectors", raw) || raw.size() % 8) return false;
    selector_blocks.resize(raw.size() / 8); std::memcpy(selector_blocks.data(), raw.data(), raw.size());
    if (!fetch_into(f, "block_selector_cluster_index", block_selectors, 4, n)) return false;
    const uint32_t ks = (uint32_t)selector_blocks.size();
    count.assign(ks, 0);
    for (uint32_t i = 0; i < n; i++) { if (block_selectors[i] >= ks) return false; count[block_selectors[i]]++; }
    selector_lists.resize(0); selector_lists.resize(ks);
    for (uint32_t c = 0; c < ks; c++) { selector_lists[c].resize(count[c]); count[c] = 0; }
    for (uint32_t i = 0; i < n; i++) selector_lists[block_selectors[i]][count[block_selectors[i]]++] = i;
    return true;
}

bool basisu_frontend::compress() {
    bu_frontend* f = nullptr;
    if (bu_frontend_pipeline* pipe = reg().booked(this, m_total_blocks)) {
        const params& p = m_params;
        bu_frontend_job job;
        std::memset(&job, 0, sizeof(job));
        job.h_blocks = reinterpret_cast<const bu_pixel_block*>(m_source_blocks.data());
        job.n_blocks = p.m_num_source_blocks;
        job.max_endpoint_clusters = p.m_max_endpoint_clusters; job.max_selector_clusters = p.m_max_selector_clusters;
        job.compression_level = p.m_compression_level; job.perceptual = p.m_perceptual;
        job.max_threads = bu_frontend_reference_max_threads(p.m_multithreaded, get_num_hardware_threads(), p.m_pJob_pool ? (uint32_t)p.m_pJob_pool->get_total_threads() : 0);
        job.flags = p.m_tex_type == basist::cBASISTexTypeVideoFrames ? BU_FRONTEND_JOB_VIDEO : 0;
        const uint64_t ticket = bu_frontend_pipeline_submit(pipe, &job, sizeof(job));
        f = ticket ? bu_frontend_pipeline_wait(pipe, ticket) : nullptr;   // this thread sleeps; the pipeline's driver thread runs init + compress among the other images
        if (!f) {
            error_printf("basisu_frontend (resident): pipelined compress failed: %s\n", ticket ? bu_frontend_pipeline_error(pipe) : bu_host_last_exception());
            m_opencl_failed = true;
            reg().release(this);
            return false;
        }
        reg().collected(this, f);
    } else {
        f = reg().find(this, m_total_blocks);
        if (!f) return false;
        if (!bu_frontend_compress(f)) {
            error_printf("basisu_frontend (resident): compress failed: %s\n", bu_frontend_error(f));
            m_opencl_failed = true;   // basis_compressor reports it (comp.cpp:3449); there is no CPU path to continue on
            return false;
        }
    }
    if (m_params.m_debug_stats) {
        const char* names[64]; double secs[64];
        const uint32_t cnt = bu_frontend_stage_times(f, names, secs, 64);
        for (uint32_t i = 0; i < cnt && i < 64; i++) debug_printf("resident frontend: %-55s %8.3f ms\n", names[i], secs[i] * 1e3);
    }
    std::vector<uint8_t> prm;
    if (!refresh_members(f, m_total_blocks, m_encoded_blocks, &m_etc1_blocks_etc1s, m_endpoint_clusters, m_block_endpoint_clusters_indices, prm,
                         m_selector_cluster_block_indices, m_optimized_cluster_selectors, m_block_selector_cluster_index))
        return false;
    m_orig_encoded_blocks = m_encoded_blocks;
    const uint32_t k = (uint32_t)(prm.size() / 16);
    m_endpoint_cluster_etc_params.resize(0); m_endpoint_cluster_etc_params.resize(k);
    for (uint32_t i = 0; i < k; i++) {
        endpoint_cluster_etc_params& e = m_endpoint_cluster_etc_params[i];
        const uint8_t* p = &prm[(size_t)i * 16];
        e.m_color_unscaled[0].set(p[0], p[1], p[2], 255);          // what generate_endpoint_codebook stores (frontend.cpp:1586-1590)
        e.m_inten_table[0] = p[3];
        std::memcpy(&e.m_color_error[0], p + 8, 8);
        e.m_valid = p[4] != 0;
        e.m_color_used[0] = !m_endpoint_clusters[i].empty();        // finalize (frontend.cpp:2980-2992)
    }
    if (m_params.m_validate && !validate_output()) {   // frontend.cpp:305-311 (there a failed check aborts; here compress() fails and the compressor reports it)
        error_printf("basisu_frontend (resident): the state the device left does not pass validate_output\n");
        return false;
    }
    return true;
}

// m_params.m_validate (frontend.h:103, frontend.cpp:3282-3338): the delivered state must be self-consistent -- every output block is the ETC1S block its endpoint cluster's
// (colour5, table) and its selector cluster's selectors make, it is listed by that selector cluster, and the clusters it names exist and are in use. A restatement over the
// members the getters serve; the mid-run checks of the reference's CPU stages (frontend.cpp:184-256) have no counterpart, the stages run on the device.
bool basisu_frontend::validate_output() const {
    const uint32_t n_ep = (uint32_t)m_endpoint_cluster_etc_params.size(), n_sel = (uint32_t)m_optimized_cluster_selectors.size();
    if (m_encoded_blocks.size() != m_total_blocks || m_block_endpoint_clusters_indices.size() != m_total_blocks || m_block_selector_cluster_index.size() != m_total_blocks ||
        m_selector_cluster_block_indices.size() != n_sel || m_endpoint_clusters.size() != n_ep)
        return false;
    std::vector<uint8_t> listed(m_total_blocks, 0);
    for (uint32_t s = 0; s < n_sel; s++)
        for (uint32_t b : m_selector_cluster_block_indices[s]) {
            if (b >= m_total_blocks || m_block_selector_cluster_index[b] != s || listed[b]) return false;   // listed once, by the cluster the block names
            listed[b] = 1;
        }
    for (uint32_t b = 0; b < m_total_blocks; b++) {
        const etc_block& out = m_encoded_blocks[b];
        const uint32_t e0 = m_block_endpoint_clusters_indices[b][0], e1 = m_block_endpoint_clusters_indices[b][1], sc = m_block_selector_cluster_index[b];
        if (!listed[b] || e0 != e1 || e0 >= n_ep || sc >= n_sel || !out.get_flip_bit() || !out.get_diff_bit()) return false;
        const endpoint_cluster_etc_params& ep = m_endpoint_cluster_etc_params[e0];
        if (!ep.m_color_used[0]) return false;
        etc_block want;
        std::memset(&want, 0, sizeof(want));
        want.set_flip_bit(true); want.set_diff_bit(true);
        if (!want.set_block_color5_check(ep.m_color_unscaled[0], ep.m_color_unscaled[0])) return false;
        want.set_inten_table(0, ep.m_inten_table[0]); want.set_inten_table(1, ep.m_inten_table[0]);
        want.set_raw_selector_bits(m_optimized_cluster_selectors[sc].get_raw_selector_bits());
        if (std::memcmp(&want, &out, sizeof(etc_block)) != 0) return false;
    }
    return true;
}

void basisu_frontend::reoptimize_remapped_endpoints(const uint_vec& new_block_endpoints, int_vec& old_to_new_endpoint_cluster_indices, bool optimize_final_codebook,
                                                    uint_vec* pBlock_selector_indices) {
    bu_frontend* f = reg().find(this, m_total_blocks);
    const uint32_t k = (uint32_t)m_endpoint_cluster_etc_params.size();
    old_to_new_endpoint_cluster_indices.resize(k);
    if (!f || new_block_endpoints.size() != m_total_blocks ||
        !bu_frontend_reoptimize_remapped_endpoints(f, new_block_endpoints.data(), m_total_blocks, old_to_new_endpoint_cluster_indices.data(), k, optimize_final_codebook,
                                                   pBlock_selector_indices ? pBlock_selector_indices->data() : nullptr)) {
        // A device error in the middle of the backend. The signature is the reference's (void): what can be done without taking the host process down (the reference aborts
        // on internal invariants only, frontend.cpp:45-49) is to leave the frontend as it was, hand the caller the identity map -- the file it then writes is a valid one
        // whose merged endpoint clusters were not re-fitted -- and raise the flag basis_compressor / basis_parallel_compress report (comp.cpp:3449, 5516).
        error_printf("basisu_frontend (resident): reoptimize_remapped_endpoints failed: %s -- endpoints left as they were, get_opencl_failed() set\n", f ? bu_frontend_error(f) : "no frontend");
        for (uint32_t i = 0; i < k; i++) old_to_new_endpoint_cluster_indices[i] = (int)i;
        m_opencl_failed = true;
        return;
    }
    std::vector<uint8_t> prm;
    basisu::vector<uint_vec> sel_lists; basisu::vector<etc_block> sel_blocks; basisu::vector<uint32_t> block_sel;
    if (!refresh_members(f, m_total_blocks, m_encoded_blocks, nullptr, m_endpoint_clusters, m_block_endpoint_clusters_indices, prm, sel_lists, sel_blocks, block_sel)) {
        error_printf("basisu_frontend (resident): reoptimize_remapped_endpoints: the re-fitted state could not be read back: %s\n", bu_frontend_error(f));
        m_opencl_failed = true;
        return;
    }
    const uint32_t k2 = (uint32_t)(prm.size() / 16);
    m_endpoint_cluster_etc_params.resize(0); m_endpoint_cluster_etc_params.resize(k2);
    for (uint32_t i = 0; i < k2; i++) {
        endpoint_cluster_etc_params& e = m_endpoint_cluster_etc_params[i];
        const uint8_t* p = &prm[(size_t)i * 16];
        e.m_color_unscaled[0].set(p[0], p[1], p[2], 255);
        e.m_inten_table[0] = p[3];
        std::memcpy(&e.m_color_error[0], p + 8, 8);
        e.m_valid = p[4] != 0;
        e.m_color_used[0] = !m_endpoint_clusters[i].empty();
    }
}

void basisu_frontend::dump_debug_image(const char* pFilename, uint32_t, uint32_t, uint32_t, bool) {
    debug_printf("basisu_frontend (resident): dump_debug_image(%s) is not provided by the resident build\n", pFilename ? pFilename : "");
}

} // namespace basisu
