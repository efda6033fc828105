// etc1s_frontend.cpp -- see etc1s_frontend.h. Stage order and bookkeeping follow basisu_frontend (encoder/basisu_frontend.cpp);
// each method cites the reference code it mirrors.
#include "etc1s_frontend.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include "tsvq.h"
#include "tsvq_device.h"

namespace bu {

namespace {

const uint32_t kEndpointParentCodebookSize = 16;          // frontend.cpp:40
const uint32_t kSelectorParentCodebookSizeLevel01 = 32;   // frontend.cpp:41
const uint32_t kSelectorParentCodebookSizeDefault = 16;   // frontend.cpp:42
const uint32_t kMaxEndpointRefinementSteps = 3;            // BASISU_MAX_ENDPOINT_REFINEMENT_STEPS, frontend.cpp:37
const uint32_t kFoscJobSize = 2048;                        // frontend.cpp:2547

const int kIntenB[8] = {8, 17, 29, 42, 60, 80, 106, 183}; // outer entries of g_etc1_inten_tables (etc.cpp:304-308)
const int kInten[8][4] = {{-8, -2, 2, 8}, {-17, -5, 5, 17}, {-29, -9, 9, 29}, {-42, -13, 13, 42},
                          {-60, -18, 18, 60}, {-80, -24, 24, 80}, {-106, -33, 33, 106}, {-183, -47, 47, 183}};

inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
inline int scale5(int c) { return (c << 3) | (c >> 2); }

inline uint64_t load_be64(const bu_etc_block& b) { uint64_t v; std::memcpy(&v, b.m_bytes, 8); return __builtin_bswap64(v); }
inline void store_be64(bu_etc_block& b, uint64_t v) { v = __builtin_bswap64(v); std::memcpy(b.m_bytes, &v, 8); }
inline uint32_t raw_selector_bits(const bu_etc_block& b) { return (uint32_t)load_be64(b); } // low 32 bits of V; same set of bits as get_raw_selector_bits()

struct etc1s_header { uint32_t r, g, b, inten; };
inline etc1s_header header_of(const bu_etc_block& blk) {
    const uint64_t v = load_be64(blk);
    return etc1s_header{(uint32_t)(v >> 59) & 31, (uint32_t)(v >> 51) & 31, (uint32_t)(v >> 43) & 31, (uint32_t)(v >> 37) & 7};
}
// selector (index into the intensity table) of pixel (x, y): etc.h:232-236
inline uint32_t selector_of(uint32_t lo32, uint32_t x, uint32_t y) {
    const uint32_t bit = x * 4 + y;
    const uint32_t raw = ((lo32 >> bit) & 1u) | (((lo32 >> (16 + bit)) & 1u) << 1);
    static const uint8_t to_sel[4] = {2, 3, 1, 0};
    return to_sel[raw];
}
inline uint32_t flat_selector_bits(uint32_t sel) {
    static const uint8_t to_raw[4] = {3, 2, 0, 1};
    const uint32_t raw = to_raw[sel];
    return ((raw & 1u) ? 0xFFFFu : 0u) | ((raw >> 1) ? 0xFFFF0000u : 0u);
}

// color_distance (enc.h:1141-1195) -- host copy used only by introduce_special_selector_clusters' error comparison
inline uint32_t color_distance(bool perceptual, const uint8_t* a, const int* b) {
    const int dr = (int)a[0] - b[0], dg = (int)a[1] - b[1], db = (int)a[2] - b[2];
    if (!perceptual) return (uint32_t)(dr * dr + dg * dg + db * db);
    const int dl = dr * 14 + dg * 45 + db * 5, dcr = dr * 64 - dl, dcb = db * 64 - dl;
    return ((uint32_t)(dl * dl) >> 5) + ((((uint32_t)(dcr * dcr) >> 5) * 26u) >> 7) + ((((uint32_t)(dcb * dcb) >> 5) * 3u) >> 7);
}

struct csr {
    std::vector<uint32_t> offsets, indices;
    void build(const std::vector<std::vector<uint32_t>>& lists) {
        offsets.resize(lists.size() + 1);
        size_t total = 0;
        for (size_t i = 0; i < lists.size(); i++) { offsets[i] = (uint32_t)total; total += lists[i].size(); }
        offsets[lists.size()] = (uint32_t)total;
        indices.resize(total);
        for (size_t i = 0; i < lists.size(); i++)
            if (!lists[i].empty()) std::memcpy(&indices[offsets[i]], lists[i].data(), lists[i].size() * sizeof(uint32_t));
    }
};

// Plain data-parallel loop over [0, n) on a few host threads (the per-block bookkeeping between device stages).
template <class F> void parallel_for(uint32_t n, F fn) {
    const unsigned t = host_threads();
    if (n < 65536 || t == 1) { fn(0u, n); return; }
    std::vector<std::thread> th;
    const uint32_t per = (n + t - 1) / t;
    for (unsigned i = 0; i < t; i++) {
        const uint32_t a = i * per, b = a + per < n ? a + per : n;
        if (a < b) th.emplace_back([=] { fn(a, b); });
    }
    for (auto& x : th) x.join();
}

// fn(t) for t in [0, T) on T host threads
template <class F> void parallel_for_chunks(unsigned T, F fn) {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back([=] { fn(t); });
    fn(0u);
    for (auto& x : th) x.join();
}

class timer {
public:
    timer() : t0_(std::chrono::steady_clock::now()) {}
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
private:
    std::chrono::steady_clock::time_point t0_;
};

} // namespace

// Grow-only device buffers owned by the frontend (all traffic goes through the C ABI).
struct etc1s_frontend::device_state {
    bu_hip_context* ctx = nullptr;
    struct buf {
        void* p = nullptr; size_t cap = 0;
    };
    const void* d_pixels = nullptr;
    bool owns_pixels = false;
    buf etc1, enc, block_cluster, params, err, valid, offsets, indices, cand_offsets, cand_indices, block_parent, out_u32, sel_blocks, weights;
    buf sel_idx, sel_ukeys, sel_uw, sel_goffs;  // outputs of bu_hip_k_unique_selector_vectors
    buf ep_idx, ep_ukeys, ep_goffs;             // outputs of bu_hip_k_unique_endpoint_vectors
    // the clusterings as resident per-block maps (bookkeeping_kernels.hip) and the small arrays around them
    buf cb_pack;   // one fit's results + the membership table behind them: one download (generate_endpoint_codebook)
    buf ep_pos, ep_parent, ep_parent_u, sel_cluster, sel_parent, orig_enc, map_sizes, map_offs, map_sorted, map_word, tmp_a, tmp_b, tmp_c, flags;

    bool reserve(buf& b, size_t bytes) {
        if (bytes <= b.cap) return true;
        if (b.p) bu_hip_free(ctx, b.p);
        b.cap = bytes + bytes / 4 + 256;
        b.p = bu_hip_malloc(ctx, b.cap);
        if (!b.p) { b.cap = 0; return false; }
        return true;
    }
    template <typename T> bool upload(buf& b, const T* src, size_t count) {
        if (!reserve(b, count * sizeof(T))) return false;
        // stream-ordered: the source has left for the context's pinned ring when this returns, every consumer is a kernel (or a copy) enqueued on the same stream afterwards
        return count ? bu_hip_memcpy_h2d_async(ctx, b.p, src, count * sizeof(T)) != 0 : true;
    }
    template <typename T> bool download(T* dst, const buf& b, size_t count) { return bu_hip_memcpy_d2h(ctx, dst, b.p, count * sizeof(T)) != 0; }
    void release() {
        for (buf* b : {&etc1, &enc, &block_cluster, &params, &err, &valid, &offsets, &indices, &cand_offsets, &cand_indices, &block_parent, &out_u32, &sel_blocks, &weights,
                       &sel_idx, &sel_ukeys, &sel_uw, &sel_goffs, &ep_idx, &ep_ukeys, &ep_goffs, &cb_pack,
                       &ep_pos, &ep_parent, &ep_parent_u, &sel_cluster, &sel_parent, &orig_enc, &map_sizes, &map_offs, &map_sorted, &map_word, &tmp_a, &tmp_b, &tmp_c, &flags})
            if (b->p) { bu_hip_free(ctx, b->p); b->p = nullptr; b->cap = 0; }
        if (owns_pixels && d_pixels) bu_hip_free(ctx, const_cast<void*>(d_pixels));
        d_pixels = nullptr;
    }
};

etc1s_frontend::etc1s_frontend() {}
etc1s_frontend::~etc1s_frontend() { drop_device_state(); }

// The device buffers are freed through the context they came from; if that context is destroyed first it tells us (bu_hip_on_destroy)
// and the buffers are let go while it still works. The host-side results (getters) stay valid.
static void frontend_context_closing(void* user) { static_cast<etc1s_frontend*>(user)->context_closing(); }
void etc1s_frontend::context_closing() {
    finish_prefetches();
    if (m_dev) { m_dev->release(); delete m_dev; m_dev = nullptr; }
}
void etc1s_frontend::drop_device_state() {
    finish_prefetches();
    if (!m_dev) return;
    bu_hip_cancel_on_destroy(m_dev->ctx, frontend_context_closing, this);
    m_dev->release(); delete m_dev; m_dev = nullptr;
}

bool etc1s_frontend::fail(const char* what) {
    m_error = what;
    if (m_dev && m_dev->ctx) { const char* e = bu_hip_last_error(m_dev->ctx); if (e && *e) { m_error += ": "; m_error += e; } }
    return false;
}

// ---- multi-GPU helpers (include/basisu_hip_frontend.h: bu_comm)
uint32_t etc1s_frontend::slab_blocks() const {
    const uint32_t w = comm_world();
    const uint32_t per = (m_total_blocks + w - 1) / w;
    return (per + kFoscJobSize - 1) / kFoscJobSize * kFoscJobSize;  // keeps find_optimal_selector_clusters' 2048-block jobs aligned
}
void etc1s_frontend::my_slab(uint32_t& first, uint32_t& count) const {
    const uint32_t per = slab_blocks();
    first = std::min<uint64_t>((uint64_t)comm_rank() * per, m_total_blocks);
    count = std::min<uint32_t>(per, m_total_blocks - first);
}
bool etc1s_frontend::gather_blocks(void* d_buf, size_t bytes_per_block) {
    if (!m_has_comm) return true;
    if (!m_comm.stream_ordered && !bu_hip_sync(m_dev->ctx)) return fail("sync before all_gather");
    if (!m_comm.all_gather(m_comm.user, d_buf, (uint64_t)slab_blocks() * bytes_per_block)) return fail("all_gather failed");
    return true;
}
bool etc1s_frontend::merge_disjoint(void* d_buf, size_t bytes) {
    if (!m_has_comm) return true;
    if (!m_comm.stream_ordered && !bu_hip_sync(m_dev->ctx)) return fail("sync before all_reduce");
    if (!m_comm.all_reduce_u64(m_comm.user, d_buf, (uint64_t)((bytes + 7) / 8))) return fail("all_reduce failed");
    return true;
}

// basisu_frontend::init (frontend.cpp:51-157)
bool etc1s_frontend::init(const params& p) {
    if (!p.m_pHIP_context) return fail("etc1s_frontend::init: a bu_hip_context is required (there is no CPU path)");
    if (p.m_max_endpoint_clusters < 1 || p.m_max_endpoint_clusters > cMaxEndpointClusters) return fail("bad max_endpoint_clusters");
    if (p.m_max_selector_clusters < 1 || p.m_max_selector_clusters > cMaxSelectorClusters) return fail("bad max_selector_clusters");
    if (!p.m_num_source_blocks || (!p.m_pSource_blocks && !p.m_pDevice_blocks)) return fail("no source blocks");
    if (p.m_compression_level > 6) return fail("bad compression level (0..6)");
    m_params = p;
    m_total_blocks = p.m_num_source_blocks;
    m_etc1_made_by_init = false;
    m_source_copy.clear(); m_source_copy.shrink_to_fit();   // a host copy of device-only tiles belongs to the image it was made from

    drop_device_state();
    m_dev = new device_state();
    m_dev->ctx = p.m_pHIP_context;
    bu_hip_on_destroy(m_dev->ctx, frontend_context_closing, this);
    if (p.m_pDevice_blocks) {
        m_dev->d_pixels = p.m_pDevice_blocks;
    } else {
        void* d = bu_hip_malloc(m_dev->ctx, (size_t)m_total_blocks * sizeof(bu_pixel_block));
        if (!d) return fail("device allocation of the source blocks failed");
        m_dev->d_pixels = d; m_dev->owns_pixels = true;
        if (!m_has_comm) {
            // host tiles: the upload and the first stage (init_etc1_images, which needs nothing but the tiles and these parameters) as one pipeline -- piece i is encoded
            // while piece i + 1 is on the link; compress() finds the ETC1 blocks made
            if (!m_dev->reserve(m_dev->etc1, (size_t)slab_blocks() * 8)) return fail("alloc");
            if (!bu_hip_k_upload_and_encode_etc1s_blocks(m_dev->ctx, d, p.m_pSource_blocks, m_total_blocks, etc1_images_quality(), p.m_perceptual, m_dev->etc1.p))
                return fail("bu_hip_k_upload_and_encode_etc1s_blocks");
            m_etc1_made_by_init = true;
        } else if (!bu_hip_memcpy_h2d(m_dev->ctx, d, p.m_pSource_blocks, (size_t)m_total_blocks * sizeof(bu_pixel_block))) return fail("upload of the source blocks failed");
    }

    // (the encoded blocks are all-zero until create_initial_packed_texture makes them: the zeros are only written if somebody asks for them before, ensure_encoded_host)
    m_encoded_blocks.clear();
    m_orig_encoded_blocks.clear();
    m_enc_host_valid = false; m_enc_dev_valid = false; m_orig_host_valid = true;
    m_ep_dev_valid = false; m_endpoint_map_valid = false; m_endpoint_lists_valid = false;
    m_sel_dev_valid = false; m_sel_host_valid = true; m_selector_group_blocks.clear(); m_selector_group_offsets.clear();
    m_endpoint_group_blocks.clear();
    m_num_endpoint_codebook_iterations = 1;
    m_num_selector_codebook_iterations = 1;
    switch (p.m_compression_level) {  // frontend.cpp:87-148
    case 0: m_endpoint_refinement = false; m_use_hierarchical_endpoint_codebooks = true; m_use_hierarchical_selector_codebooks = true; break;
    case 1: case 2: m_endpoint_refinement = true; m_use_hierarchical_endpoint_codebooks = true; m_use_hierarchical_selector_codebooks = true; break;
    case 3: m_endpoint_refinement = true; m_use_hierarchical_endpoint_codebooks = false; m_use_hierarchical_selector_codebooks = false; break;
    case 4: m_endpoint_refinement = true; m_use_hierarchical_endpoint_codebooks = true; m_use_hierarchical_selector_codebooks = true;
            m_num_endpoint_codebook_iterations = kMaxEndpointRefinementSteps; m_num_selector_codebook_iterations = kMaxEndpointRefinementSteps; break;
    case 5: m_endpoint_refinement = true; m_use_hierarchical_endpoint_codebooks = false; m_use_hierarchical_selector_codebooks = false;
            m_num_endpoint_codebook_iterations = kMaxEndpointRefinementSteps; m_num_selector_codebook_iterations = kMaxEndpointRefinementSteps; break;
    default: m_endpoint_refinement = true; m_use_hierarchical_endpoint_codebooks = false; m_use_hierarchical_selector_codebooks = false;
            m_num_endpoint_codebook_iterations = kMaxEndpointRefinementSteps * 2; m_num_selector_codebook_iterations = kMaxEndpointRefinementSteps * 2; break;
    }
    if (p.m_disable_hierarchical_endpoint_codebooks) m_use_hierarchical_endpoint_codebooks = false;
    return true;
}

// ---- results leave for the host as soon as they are final, behind the stages that follow (the device -> host copies of a 4096^2 image are 20 MB: 0.36 ms of link time
// and as much again in round trips when they are fetched one after the other at the end)
void etc1s_frontend::finish_prefetches(int which) const {   // 1 = the endpoint map, 2 = the encoded blocks, 3 = both
    if ((which & 1) && (m_dl_ep_cluster || m_dl_ep_pos)) {
        const bool a = !m_dl_ep_cluster || bu_hip_download_wait(m_dl_ep_cluster) != 0, b = !m_dl_ep_pos || bu_hip_download_wait(m_dl_ep_pos) != 0;
        m_dl_ep_cluster = m_dl_ep_pos = nullptr;
        if (a && b && m_ep_dev_valid) m_endpoint_map_valid = true;
    }
    if ((which & 2) && m_dl_enc) {
        const bool ok = bu_hip_download_wait(m_dl_enc) != 0;
        m_dl_enc = nullptr;
        if (ok && m_enc_dev_valid) m_enc_host_valid = true;
    }
}
static bool result_prefetch_on() { static const bool on = [] { const char* e = std::getenv("BU_RESULT_PREFETCH"); return !e || e[0] != '0'; }(); return on; }   // A/B switch
void etc1s_frontend::prefetch_endpoint_map() {
    if (!result_prefetch_on()) return;
    if (!m_dev || !m_ep_dev_valid || m_endpoint_map_valid || m_dl_ep_cluster || m_dl_ep_pos) return;
    m_block_endpoint_cluster.resize(m_total_blocks); m_block_endpoint_pos.resize(m_total_blocks);
    m_dl_ep_cluster = bu_hip_download_begin(m_dev->ctx, m_block_endpoint_cluster.data(), m_dev->block_cluster.p, (size_t)m_total_blocks * 4);
    if (m_dl_ep_cluster) m_dl_ep_pos = bu_hip_download_begin(m_dev->ctx, m_block_endpoint_pos.data(), m_dev->ep_pos.p, (size_t)m_total_blocks * 4);
    if (m_dl_ep_cluster && !m_dl_ep_pos) { (void)bu_hip_download_wait(m_dl_ep_cluster); m_dl_ep_cluster = nullptr; }   // both or neither: ensure_endpoint_map fetches the pair
}
void etc1s_frontend::prefetch_encoded_blocks() {
    if (!result_prefetch_on()) return;
    if (!m_dev || !m_enc_dev_valid || m_enc_host_valid || m_dl_enc) return;
    m_encoded_blocks.resize(m_total_blocks);
    m_dl_enc = bu_hip_download_begin(m_dev->ctx, m_encoded_blocks.data(), m_dev->enc.p, (size_t)m_total_blocks * 8);
}

// basisu_frontend::compress (frontend.cpp:159-316), single endpoint/selector iteration (levels 0-3)
bool etc1s_frontend::compress() {
    if (!m_dev) return fail("etc1s_frontend::compress: not initialised, or the context it was initialised on has been destroyed");
    m_stage_times.clear();
    finish_prefetches();
    // levels 4-6 and video clips go on changing the endpoint clustering and the encoded blocks (refine_block_endpoints_given_selectors): nothing is final early there
    const bool early_results = m_params.m_compression_level <= 3 && !m_params.m_video;
#define BU_STAGE(name, call) do { timer t__; if (!(call)) return false; m_stage_times.push_back(stage_time{name, t__.seconds()}); } while (0)
#define BU_STAGE_V(name, call) do { timer t__; call; m_stage_times.push_back(stage_time{name, t__.seconds()}); } while (0)
    BU_STAGE("init_etc1_images", init_etc1_images());
    BU_STAGE("init_endpoint_training_vectors", init_endpoint_training_vectors());
    BU_STAGE("generate_endpoint_clusters", generate_endpoint_clusters());
    for (uint32_t step = 0; step < m_num_endpoint_codebook_iterations; step++) {
        if (step) BU_STAGE("introduce_new_endpoint_clusters", introduce_new_endpoint_clusters());
        BU_STAGE("generate_endpoint_codebook", generate_endpoint_codebook(step));
        bool early_out = false;
        if (m_endpoint_refinement) {
            uint32_t moved = 0;
            BU_STAGE("refine_endpoint_clusterization", refine_endpoint_clusterization(&moved));
            if (!moved) early_out = true;  // frontend.cpp:215-216
            if (m_params.m_video && !step && m_num_endpoint_codebook_iterations == 1) {  // frontend.cpp:219-223: video clips get one more fit of the merged codebook
                BU_STAGE("eliminate_redundant_or_empty_endpoint_clusters", eliminate_redundant_or_empty_endpoint_clusters());
                BU_STAGE("generate_endpoint_codebook", generate_endpoint_codebook(1));
            }
        }
        BU_STAGE("eliminate_redundant_or_empty_endpoint_clusters", eliminate_redundant_or_empty_endpoint_clusters());
        if (early_out) break;
    }
    if (early_results) prefetch_endpoint_map();
    BU_STAGE_V("generate_block_endpoint_clusters", generate_block_endpoint_clusters());
    BU_STAGE("create_initial_packed_texture", create_initial_packed_texture());
    BU_STAGE("generate_selector_clusters", generate_selector_clusters());
    // (without a communicator the membership table of the selector parents comes back with the first codebook's download instead of in a round trip of its own:
    //  create_optimized_selector_codebook; find_optimal_selector_clusters_for_each_block falls back on the pass below if the lists are missing)
    m_selector_clusters_within_each_parent_cluster.clear();
    if (m_use_hierarchical_selector_codebooks && (m_has_comm || m_params.m_compression_level == 0))
        BU_STAGE_V("compute_selector_clusters_within_each_parent_cluster", compute_selector_clusters_within_each_parent_cluster());
    const uint32_t selector_steps = m_params.m_compression_level == 0 ? 1 : m_num_selector_codebook_iterations;
    for (uint32_t it = 0; it < selector_steps; it++) {
        BU_STAGE("create_optimized_selector_codebook", create_optimized_selector_codebook(it));
        BU_STAGE("find_optimal_selector_clusters_for_each_block", find_optimal_selector_clusters_for_each_block());
        if (early_results && it + 1 == selector_steps) prefetch_encoded_blocks();
        BU_STAGE("introduce_special_selector_clusters", introduce_special_selector_clusters());
        if (m_params.m_compression_level >= 4 || m_params.m_video) {  // frontend.cpp:291
            uint32_t refined = 0;
            BU_STAGE("refine_block_endpoints_given_selectors", refine_block_endpoints_given_selectors(&refined));
            if (!refined) break;  // frontend.cpp:283-286
        }
    }
    BU_STAGE("optimize_selector_codebook", optimize_selector_codebook());
    BU_STAGE_V("finalize", finalize());
#undef BU_STAGE
#undef BU_STAGE_V
    return true;
}

// frontend.cpp:733-823
int etc1s_frontend::etc1_images_quality() const {
    return m_params.m_compression_level == 0 ? BU_ETC_QUALITY_FAST : m_params.m_compression_level == 1 ? BU_ETC_QUALITY_MEDIUM
         : m_params.m_compression_level == 6 ? BU_ETC_QUALITY_UBER : BU_ETC_QUALITY_SLOW; // frontend.cpp:783-788
}
bool etc1s_frontend::init_etc1_images() {
    const int quality = etc1_images_quality();
    device_state& d = *m_dev;
    if (m_etc1_made_by_init) {   // init() encoded the tiles as they arrived (once: a second compress() of the same frontend does the stage again)
        m_etc1_made_by_init = false;
        if (!m_has_comm && d.etc1.p) { m_etc1_blocks_etc1s.clear(); m_etc1_on_host = false; return true; }
    }
    uint32_t b0, nb;
    my_slab(b0, nb);
    if (!d.reserve(d.etc1, (size_t)comm_world() * slab_blocks() * 8)) return fail("alloc");
    if (nb && !bu_hip_k_encode_etc1s_blocks(d.ctx, (const char*)d.d_pixels + (size_t)b0 * 64, nb, quality, m_params.m_perceptual, (char*)d.etc1.p + (size_t)b0 * 8))
        return fail("bu_hip_k_encode_etc1s_blocks");
    if (!gather_blocks(d.etc1.p, 8)) return false;
    // every consumer of these blocks runs on the device (the training-vector de-duplication); the host copy is fetched when somebody asks
    m_etc1_blocks_etc1s.clear();
    m_etc1_on_host = false;
    return true;
}

// get_source_pixel_block (frontend.h:155) for the backend: the host tiles the frontend was given, else one download of the resident ones
const bu_pixel_block* etc1s_frontend::source_blocks_host() {
    if (m_params.m_pSource_blocks) return m_params.m_pSource_blocks;
    if (m_source_copy.size() != m_total_blocks) {
        m_source_copy.resize(m_total_blocks);
        if (!m_dev || !m_dev->d_pixels || !bu_hip_memcpy_d2h(m_dev->ctx, m_source_copy.data(), m_dev->d_pixels, (size_t)m_total_blocks * sizeof(bu_pixel_block))) {
            m_source_copy.clear();
            fail("download source blocks");
            return nullptr;
        }
    }
    return m_source_copy.data();
}

bool etc1s_frontend::backend_block_errors(uint32_t first_block, uint32_t nbx, uint32_t nby, bool with_neighbours, uint32_t* own, uint32_t* neighbour) {
    if (!m_dev || !m_dev->d_pixels || !m_enc_dev_valid || !m_ep_dev_valid || !m_dev->enc.p || !m_dev->block_cluster.p) return false;
    const uint64_t n = (uint64_t)nbx * nby;
    if (!n || first_block + n > m_total_blocks) return false;
    device_state& d = *m_dev;
    const uint32_t k = (uint32_t)m_endpoint_cluster_etc_params.size();
    std::vector<uint8_t> prm((size_t)k * 4);
    for (uint32_t i = 0; i < k; i++) {
        const endpoint_params& e = m_endpoint_cluster_etc_params[i];
        prm[i * 4] = e.r; prm[i * 4 + 1] = e.g; prm[i * 4 + 2] = e.b; prm[i * 4 + 3] = e.inten;
    }
    // results at slice-relative positions: the kernel indexes its outputs by absolute block, so the buffers cover blocks [0, first + n)
    const size_t upto = (size_t)first_block + n;
    if (!d.upload(d.params, prm.data(), prm.size()) || !d.reserve(d.tmp_a, upto * 4) || (with_neighbours && !d.reserve(d.tmp_c, upto * 12))) return false;
    if (!bu_hip_k_backend_block_errors(d.ctx, d.d_pixels, d.enc.p, (const uint32_t*)d.block_cluster.p, (const uint8_t*)d.params.p, first_block, nbx, nby, k, m_params.m_perceptual ? 1 : 0,
                                       with_neighbours ? 1 : 0, (uint32_t*)d.tmp_a.p, with_neighbours ? (uint32_t*)d.tmp_c.p : nullptr))
        return false;
    if (!bu_hip_memcpy_d2h(d.ctx, own, (const char*)d.tmp_a.p + (size_t)first_block * 4, n * 4)) return false;
    if (with_neighbours && !bu_hip_memcpy_d2h(d.ctx, neighbour, (const char*)d.tmp_c.p + (size_t)first_block * 12, n * 12)) return false;
    return true;
}

const std::vector<bu_etc_block>& etc1s_frontend::etc1_blocks() const {
    if (!m_etc1_on_host && m_dev && m_dev->etc1.p) {
        m_etc1_blocks_etc1s.resize(m_total_blocks);
        if (!m_dev->download(m_etc1_blocks_etc1s.data(), m_dev->etc1, m_total_blocks)) m_etc1_blocks_etc1s.clear();
        m_etc1_on_host = true;
    }
    return m_etc1_blocks_etc1s;
}

// frontend.cpp:825-866 + the de-duplication of generate_hierarchical_codebook_threaded (enc.h:2218-2290).
// A training vector is (low rgb, high rgb)/255 of the block's ETC1S colours, identical for both sub-blocks, weight 1 each. The
// reference de-duplicates them in a std::map (ascending lexicographic float order); the floats are monotone in the 8-bit colours,
// so we sort integer keys instead and materialise only the distinct vectors.
bool etc1s_frontend::init_endpoint_training_vectors() {
    const uint32_t n = m_total_blocks;
    device_state& d = *m_dev;
    // de-duplication on the device (unique_kernels.hip): 48-bit low/high colour keys of the resident blocks, stable sort, run lengths
    uint32_t u_total = 0;
    if (!d.reserve(d.ep_idx, (size_t)n * 4) || !d.reserve(d.ep_ukeys, (size_t)n * 8) || !d.reserve(d.ep_goffs, ((size_t)n + 1) * 4)) return fail("alloc");
    if (!bu_hip_k_unique_endpoint_vectors(d.ctx, d.etc1.p, n, (uint32_t*)d.ep_idx.p, (uint64_t*)d.ep_ukeys.p, (uint32_t*)d.ep_goffs.p, &u_total))
        return fail("bu_hip_k_unique_endpoint_vectors");
    // the distinct vectors and their groups stay in HBM (ep_ukeys / ep_goffs / ep_idx): the codebook builder makes its float rows from the keys there
    // (bu_hip_tsvq_create_endpoint_device); the host forms are fetched when a list getter asks (endpoint_group_offsets_host / endpoint_group_blocks_host)
    m_endpoint_unique_count = u_total;
    m_endpoint_group_blocks.clear(); m_endpoint_group_offsets.clear();
    m_endpoint_unique_rows.clear(); m_endpoint_unique_weights.clear();
    return true;
}

// frontend.cpp:868-944
bool etc1s_frontend::generate_endpoint_clusters() {
    finish_prefetches(1);   // nothing rewrites what a pending download reads
    m_ep_member_valid = false;
    const uint32_t parent_size = (m_params.m_max_endpoint_clusters >= 256) ? kEndpointParentCodebookSize : 0;
    const uint32_t n = m_total_blocks, u_total = m_endpoint_unique_count;
    const uint32_t want_parents = m_use_hierarchical_endpoint_codebooks ? parent_size : 0;
    m_endpoint_parent_clusters.clear();
    m_endpoint_parent_of_unique.clear(); m_endpoint_parent_dev_valid = false;
    device_state& d = *m_dev;
    bool sizes_pending = false;
    // Per distinct vector, resident: its leaf (tmp_a), the position of its first block in the leaf's list (tmp_b), its parent (tmp_c); per leaf the list length
    // (out_u32). A leaf lists its distinct vectors ascending and each vector's blocks ascending (enc.h:1573-1584 + the training-vector order of frontend.cpp:825-866).
    if (!d.reserve(d.tmp_a, (size_t)u_total * 4) || !d.reserve(d.tmp_b, (size_t)u_total * 4) || !d.reserve(d.tmp_c, (size_t)u_total * 4) ||
        !d.reserve(d.out_u32, ((size_t)cMaxEndpointClusters + kThreadedCodebookMaxThreads) * 4) /* T trees of ceil(K/T) leaves: up to T - 1 more than K */ || !d.reserve(d.block_cluster, (size_t)n * 4) || !d.reserve(d.ep_pos, (size_t)n * 4) || !d.reserve(d.ep_parent, n))
        return fail("alloc");
    if (m_params.m_fast_codebooks && !std::getenv("BU_FAST_SELECTORS_ONLY")) {   // row f3: k-means on the matrix cores; the per-vector results come back (a few ten thousand words) for the list offsets below
        if (!bu_hip_kmeans_codebook(d.ctx, 1, d.ep_ukeys.p, nullptr, (const uint32_t*)d.ep_goffs.p, u_total, m_params.m_max_endpoint_clusters, want_parents,
                                    m_params.m_fast_codebook_iterations, (uint32_t*)d.tmp_a.p, want_parents ? (uint32_t*)d.tmp_c.p : nullptr, &m_endpoint_cluster_count,
                                    &m_endpoint_parent_count))
            return fail("bu_hip_kmeans_codebook (endpoints)");
        std::vector<uint32_t> leaf_of_unique(u_total), first_pos(u_total);
        const std::vector<uint32_t>& goffs = endpoint_group_offsets_host();
        if (goffs.size() != (size_t)u_total + 1 || !d.download(leaf_of_unique.data(), d.tmp_a, u_total)) return fail("download");
        m_endpoint_cluster_sizes.assign(m_endpoint_cluster_count, 0);
        for (uint32_t u = 0; u < u_total; u++) {
            const uint32_t c = leaf_of_unique[u];
            first_pos[u] = m_endpoint_cluster_sizes[c];
            m_endpoint_cluster_sizes[c] += goffs[u + 1] - goffs[u];
        }
        if (!d.upload(d.tmp_b, first_pos.data(), u_total)) return fail("upload");
    } else {
        if (!device_tsvq::hierarchical_codebook_endpoint_device(d.ctx, (const uint64_t*)d.ep_ukeys.p, (const uint32_t*)d.ep_goffs.p, u_total, m_params.m_max_endpoint_clusters,
                                                                want_parents, &m_endpoint_cluster_count, &m_endpoint_parent_count, (uint32_t*)d.tmp_a.p, (uint32_t*)d.tmp_c.p,
                                                                (uint32_t*)d.tmp_b.p, (uint32_t*)d.out_u32.p, nullptr, m_has_comm ? &m_comm : nullptr, m_params.m_codebook_threads))
            return fail("endpoint TSVQ failed");
        sizes_pending = true;
    }
    const bool parents = want_parents && m_endpoint_parent_count;
    if (m_use_hierarchical_endpoint_codebooks && !m_endpoint_parent_count) m_endpoint_parent_count = 1;  // no parent level: one parent holding everything (frontend.cpp:905-911)
    if (!bu_hip_k_map_blocks_from_groups(d.ctx, (const uint32_t*)d.ep_goffs.p, (const uint32_t*)d.ep_idx.p, n, u_total, (const uint32_t*)d.tmp_a.p, (const uint32_t*)d.tmp_b.p,
                                         parents ? (const uint32_t*)d.tmp_c.p : nullptr, (uint32_t*)d.block_cluster.p, (uint32_t*)d.ep_pos.p, (uint8_t*)d.ep_parent.p))
        return fail("bu_hip_k_map_blocks_from_groups");
    if (parents) {   // the parent of every distinct vector, kept for the parent-list getter (tmp_c is everybody's scratch)
        if (!d.reserve(d.ep_parent_u, (size_t)u_total * 4) || !bu_hip_memcpy_d2d(d.ctx, d.ep_parent_u.p, d.tmp_c.p, (size_t)u_total * 4)) return fail("copy");
        m_endpoint_parent_dev_valid = true;
    }
    if (sizes_pending) {   // the leaves' list lengths: asked for behind the kernels above, which do not need them (the device works while the host waits)
        m_endpoint_cluster_sizes.resize(m_endpoint_cluster_count);
        if (!d.download(m_endpoint_cluster_sizes.data(), d.out_u32, m_endpoint_cluster_count)) return fail("download cluster sizes");
    }
    m_ep_dev_valid = true; m_endpoint_map_valid = false; m_endpoint_lists_valid = false; m_endpoint_clusters.clear();
    return true;
}

// the groups of the distinct endpoint training vectors (offsets into the sorted block array) and the parent of every distinct vector: resident, fetched when a list form is asked for
const std::vector<uint32_t>& etc1s_frontend::endpoint_group_offsets_host() const {
    if (m_endpoint_group_offsets.size() != (size_t)m_endpoint_unique_count + 1 && m_dev && m_dev->ep_goffs.p) {
        m_endpoint_group_offsets.resize((size_t)m_endpoint_unique_count + 1);
        if (!m_dev->download(m_endpoint_group_offsets.data(), m_dev->ep_goffs, (size_t)m_endpoint_unique_count + 1)) m_endpoint_group_offsets.clear();
    }
    return m_endpoint_group_offsets;
}
const std::vector<uint32_t>& etc1s_frontend::endpoint_parent_of_unique_host() const {
    if (m_endpoint_parent_of_unique.empty() && m_endpoint_parent_dev_valid && m_dev && m_dev->ep_parent_u.p) {
        m_endpoint_parent_of_unique.resize(m_endpoint_unique_count);
        if (!m_dev->download(m_endpoint_parent_of_unique.data(), m_dev->ep_parent_u, m_endpoint_unique_count)) m_endpoint_parent_of_unique.clear();
    }
    return m_endpoint_parent_of_unique;
}

// the blocks behind every distinct endpoint training vector (the device's stable sort order), fetched when a list form is asked for
const std::vector<uint32_t>& etc1s_frontend::endpoint_group_blocks_host() const {
    if (m_endpoint_group_blocks.size() != m_total_blocks && m_dev && m_dev->ep_idx.p) {
        m_endpoint_group_blocks.resize(m_total_blocks);
        if (!m_dev->download(m_endpoint_group_blocks.data(), m_dev->ep_idx, m_total_blocks)) m_endpoint_group_blocks.clear();
    }
    return m_endpoint_group_blocks;
}

// ---- the endpoint clustering has two interchangeable forms: the reference's lists of training-vector ids (block * 2 + sub-block, both
// sub-blocks of a block adjacent) and, per block, (cluster, position of the block in the cluster's list). The hot path works on the second
// one; whoever needs lists (levels 4-6 bookkeeping, the getters) gets them built, and list surgery is folded back into the map.
void etc1s_frontend::ensure_endpoint_map() const {
    if (m_dl_ep_cluster || m_dl_ep_pos) finish_prefetches(1);
    if (m_endpoint_map_valid) return;
    if (m_ep_dev_valid && m_dev) {   // the resident map is the clustering: bring it over (sizes and count are kept on the host by every stage)
        m_block_endpoint_cluster.resize(m_total_blocks); m_block_endpoint_pos.resize(m_total_blocks);
        if (m_dev->download(m_block_endpoint_cluster.data(), m_dev->block_cluster, m_total_blocks) && m_dev->download(m_block_endpoint_pos.data(), m_dev->ep_pos, m_total_blocks))
            m_endpoint_map_valid = true;
        return;
    }
    const uint32_t k = (uint32_t)m_endpoint_clusters.size();
    m_endpoint_cluster_count = k;
    m_block_endpoint_cluster.resize(m_total_blocks); m_block_endpoint_pos.resize(m_total_blocks);
    m_endpoint_cluster_sizes.assign(k, 0);
    for (uint32_t ci = 0; ci < k; ci++) {
        const std::vector<uint32_t>& l = m_endpoint_clusters[ci];
        m_endpoint_cluster_sizes[ci] = (uint32_t)(l.size() / 2);
        for (size_t i = 0; i < l.size(); i++) { m_block_endpoint_cluster[l[i] >> 1] = ci; m_block_endpoint_pos[l[i] >> 1] = (uint32_t)(i / 2); }
    }
    m_endpoint_map_valid = true;
}
bool etc1s_frontend::ensure_endpoint_map_device() {
    if (m_ep_dev_valid) return true;
    finish_prefetches(1);   // nothing rewrites what a pending download reads
    m_ep_member_valid = false;
    ensure_endpoint_map();
    device_state& d = *m_dev;
    if (!m_endpoint_map_valid || !d.upload(d.block_cluster, m_block_endpoint_cluster.data(), m_total_blocks) || !d.upload(d.ep_pos, m_block_endpoint_pos.data(), m_total_blocks))
        return fail("upload endpoint map");
    m_ep_dev_valid = true;
    return true;
}

void etc1s_frontend::endpoint_csr(std::vector<uint32_t>& offsets, std::vector<uint32_t>& indices) const {
    ensure_endpoint_map();
    const uint32_t k = m_endpoint_cluster_count, n = m_total_blocks;
    offsets.resize((size_t)k + 1);
    uint32_t run = 0;
    for (uint32_t c = 0; c < k; c++) { offsets[c] = run; run += m_endpoint_cluster_sizes[c] * 2; }
    offsets[k] = run;
    indices.resize(run);
    parallel_for(n, [&](uint32_t b0, uint32_t b1) {
        for (uint32_t b = b0; b < b1; b++) {
            const size_t at = (size_t)offsets[m_block_endpoint_cluster[b]] + 2ull * m_block_endpoint_pos[b];
            indices[at] = b * 2; indices[at + 1] = b * 2 + 1;
        }
    });
}
void etc1s_frontend::ensure_endpoint_lists() const {
    if (m_endpoint_lists_valid) return;
    std::vector<uint32_t> offsets, indices;
    endpoint_csr(offsets, indices);
    m_endpoint_clusters.assign(m_endpoint_cluster_count, {});
    for (uint32_t c = 0; c < m_endpoint_cluster_count; c++) m_endpoint_clusters[c].assign(indices.begin() + offsets[c], indices.begin() + offsets[c + 1]);
    m_endpoint_lists_valid = true;
}
const std::vector<std::vector<uint32_t>>& etc1s_frontend::endpoint_clusters() const { ensure_endpoint_lists(); return m_endpoint_clusters; }

const std::vector<std::vector<uint32_t>>& etc1s_frontend::endpoint_parent_clusters() const {
    if (m_endpoint_parent_clusters.empty() && m_use_hierarchical_endpoint_codebooks) {
        const std::vector<uint32_t>& parent_of_unique = endpoint_parent_of_unique_host();
        if (parent_of_unique.empty()) {
            m_endpoint_parent_clusters.resize(1);
            for (uint32_t i = 0; i < m_total_blocks; i++) { m_endpoint_parent_clusters[0].push_back(i * 2); m_endpoint_parent_clusters[0].push_back(i * 2 + 1); }
        } else {
            device_tsvq::expand_parents(parent_of_unique, m_endpoint_parent_count,
                                        csr_block_pair_groups{endpoint_group_offsets_host().data(), endpoint_group_blocks_host().data()}, m_endpoint_parent_clusters);
        }
    }
    return m_endpoint_parent_clusters;
}

// frontend.cpp:947-968
void etc1s_frontend::generate_block_endpoint_clusters() {   // the map is the clustering (see ensure_endpoint_map); with a prefetch under way it is arriving
    if (!(m_dl_ep_cluster || m_dl_ep_pos)) ensure_endpoint_map();
}

// frontend.cpp:971-1003. The reference collects one entry per block and then sorts + uniques each parent's list; the result is
// "the ascending set of clusters that own at least one block of this parent", which a membership table gives in O(blocks).
void etc1s_frontend::compute_endpoint_clusters_within_each_parent_cluster() {
    const size_t parents = m_endpoint_parent_count, clusters = m_endpoint_cluster_count;
    m_endpoint_clusters_within_each_parent_cluster.assign(parents, {});
    device_state& d = *m_dev;
    std::vector<uint8_t> member;
    if (m_ep_member_valid && m_ep_member_parents == parents && m_ep_member_clusters == clusters && m_ep_member.size() == parents * clusters) {
        member.swap(m_ep_member);   // came back with the codebook fit's results (generate_endpoint_codebook): the clustering has not changed since
        m_ep_member_valid = false;
    } else {
        member.assign(parents * clusters, 0);
    if (!ensure_endpoint_map_device() || !d.reserve(d.flags, parents * clusters + 8) ||
        !bu_hip_k_map_membership(d.ctx, (const uint8_t*)d.ep_parent.p, (const uint32_t*)d.block_cluster.p, m_total_blocks, (uint32_t)parents, (uint32_t)clusters, (uint8_t*)d.flags.p) ||
        !d.download(member.data(), d.flags, member.size())) {
        m_endpoint_clusters_within_each_parent_cluster.clear();   // the size check of refine_endpoint_clusterization fires on this
        fail("compute_endpoint_clusters_within_each_parent_cluster");
        return;
    }
    }
    for (size_t p = 0; p < parents; p++)
        for (size_t c = 0; c < clusters; c++)
            if (member[p * clusters + c]) m_endpoint_clusters_within_each_parent_cluster[p].push_back((uint32_t)c);
}

// frontend.cpp:1214-1617 (CPU semantics; the kernel also handles step > 0)
bool etc1s_frontend::generate_endpoint_codebook(uint32_t step) {
    csr lists;  // the members of every cluster in list order (it decides the float mean of clusters past 65k texels)
    const bool resident = !m_endpoint_lists_valid;   // no host lists in play: the list array is written on the device from the resident map
    if (!resident) lists.build(m_endpoint_clusters);
    else {
        if (!ensure_endpoint_map_device()) return false;
        lists.offsets.resize((size_t)m_endpoint_cluster_count + 1);
        uint32_t run = 0;
        for (uint32_t c = 0; c < m_endpoint_cluster_count; c++) { lists.offsets[c] = run; run += m_endpoint_cluster_sizes[c] * 2; }
        lists.offsets[m_endpoint_cluster_count] = run;
    }
    const uint32_t k = (uint32_t)lists.offsets.size() - 1;
    m_endpoint_cluster_etc_params.resize(k);
    const int quality = m_params.m_compression_level <= 1 ? BU_ETC_QUALITY_MEDIUM : m_params.m_compression_level == 6 ? BU_ETC_QUALITY_UBER : BU_ETC_QUALITY_SLOW; // :1530-1533
    std::vector<uint8_t> prm(k * 4ull), valid(k);
    std::vector<uint64_t> err(k);
    for (uint32_t i = 0; i < k; i++) {
        const endpoint_params& e = m_endpoint_cluster_etc_params[i];
        prm[i * 4] = e.r; prm[i * 4 + 1] = e.g; prm[i * 4 + 2] = e.b; prm[i * 4 + 3] = e.inten; valid[i] = e.valid; err[i] = e.color_error;
    }
    device_state& d = *m_dev;
    if (m_has_comm) {
        // this rank fits the clusters at positions rank, rank + world, ... of the size-descending order (the order the device layer
        // schedules them in); everything else is zeroed so that the sum-merge below puts the shares together
        std::vector<uint32_t> order(k);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (lists.offsets[a + 1] - lists.offsets[a]) > (lists.offsets[b + 1] - lists.offsets[b]); });
        std::vector<uint8_t> own(k, 0);
        for (uint32_t i = comm_rank(); i < k; i += comm_world()) own[order[i]] = 1;
        for (uint32_t i = 0; i < k; i++)
            if (!own[i]) { prm[i * 4] = prm[i * 4 + 1] = prm[i * 4 + 2] = prm[i * 4 + 3] = 0; valid[i] = 0; err[i] = 0; }
    }
    if (!d.upload(d.offsets, lists.offsets.data(), lists.offsets.size()) ||
        (resident ? (!d.reserve(d.indices, (size_t)m_total_blocks * 8) ||
                     !bu_hip_k_map_endpoint_csr(d.ctx, (const uint32_t*)d.block_cluster.p, (const uint32_t*)d.ep_pos.p, m_total_blocks, (const uint32_t*)d.offsets.p, (uint32_t*)d.indices.p))
                  : !d.upload(d.indices, lists.indices.data(), lists.indices.size())))
        return fail("upload endpoint clusters");
    m_ep_member_valid = false;
    if (!m_has_comm) {
        // One device buffer for the fit's three result arrays and -- when refine_endpoint_clusterization follows with parent lists -- the (parent, cluster) membership
        // table it will ask for, whose kernel goes in front of the fit: everything comes back in ONE copy instead of four blocking ones (each a round trip with the device idle).
        const size_t parents = m_endpoint_parent_count;
        const bool with_members = resident && m_endpoint_refinement && m_use_hierarchical_endpoint_codebooks && parents && parents * (size_t)k <= ((size_t)64 << 20);
        const size_t at_prm = (size_t)k * 8, at_valid = (size_t)k * 12, at_member = ((size_t)k * 13 + 15) & ~(size_t)15, total = at_member + (with_members ? parents * k : 0);
        std::vector<uint8_t> pack(total, 0);
        std::memcpy(pack.data(), err.data(), (size_t)k * 8); std::memcpy(pack.data() + at_prm, prm.data(), (size_t)k * 4); std::memcpy(pack.data() + at_valid, valid.data(), k);
        if (!d.reserve(d.cb_pack, total + 16) || !bu_hip_memcpy_h2d_async(d.ctx, d.cb_pack.p, pack.data(), at_member)) return fail("upload endpoint clusters");
        char* base = (char*)d.cb_pack.p;
        if (with_members && !bu_hip_k_map_membership(d.ctx, (const uint8_t*)d.ep_parent.p, (const uint32_t*)d.block_cluster.p, m_total_blocks, (uint32_t)parents, k, (uint8_t*)(base + at_member)))
            return fail("bu_hip_k_map_membership");
        if (!bu_hip_k_generate_endpoint_codebook_part(d.ctx, d.d_pixels, k, lists.offsets.data(), (const uint32_t*)d.offsets.p, (const uint32_t*)d.indices.p, quality,
                                                      m_params.m_perceptual, step, (uint8_t*)(base + at_prm), (uint64_t*)base, (uint8_t*)(base + at_valid), 0, 1))
            return fail("bu_hip_k_generate_endpoint_codebook");
        if (!bu_hip_memcpy_d2h(d.ctx, pack.data(), d.cb_pack.p, total)) return fail("download endpoint codebook");
        std::memcpy(err.data(), pack.data(), (size_t)k * 8); std::memcpy(prm.data(), pack.data() + at_prm, (size_t)k * 4); std::memcpy(valid.data(), pack.data() + at_valid, k);
        if (with_members) {
            m_ep_member.assign(pack.begin() + (long)at_member, pack.end());
            m_ep_member_parents = parents; m_ep_member_clusters = k; m_ep_member_valid = true;
        }
    } else {
    if (!d.reserve(d.params, (size_t)k * 4 + 8) || !d.upload(d.params, prm.data(), prm.size()) || !d.upload(d.err, err.data(), err.size()) ||
        !d.reserve(d.valid, (size_t)k + 8) || !d.upload(d.valid, valid.data(), valid.size()))
        return fail("upload endpoint clusters");
    // the u64 merge rounds the byte arrays up to whole words: clear the tail
    if (!bu_hip_memset(d.ctx, (char*)d.params.p + (size_t)k * 4, 0, 8) || !bu_hip_memset(d.ctx, (char*)d.valid.p + k, 0, 8)) return fail("memset");
    if (!bu_hip_k_generate_endpoint_codebook_part(d.ctx, d.d_pixels, k, lists.offsets.data(), (const uint32_t*)d.offsets.p, (const uint32_t*)d.indices.p, quality,
                                                  m_params.m_perceptual, step, (uint8_t*)d.params.p, (uint64_t*)d.err.p, (uint8_t*)d.valid.p, comm_rank(), comm_world()))
        return fail("bu_hip_k_generate_endpoint_codebook");
    if (!merge_disjoint(d.params.p, (size_t)k * 4) || !merge_disjoint(d.err.p, (size_t)k * 8) || !merge_disjoint(d.valid.p, k)) return false;
    if (!d.download(prm.data(), d.params, prm.size()) || !d.download(err.data(), d.err, err.size()) || !d.download(valid.data(), d.valid, valid.size()))
        return fail("download endpoint codebook");
    }
    for (uint32_t i = 0; i < k; i++) {
        endpoint_params& e = m_endpoint_cluster_etc_params[i];
        e.r = prm[i * 4]; e.g = prm[i * 4 + 1]; e.b = prm[i * 4 + 2]; e.inten = prm[i * 4 + 3]; e.valid = valid[i] != 0; e.color_error = err[i];
    }
    return true;
}

// frontend.cpp:1093-1212 (+ compute_endpoint_subblock_error_vec :1006-1091 on the device). Between codebook iterations the sub-blocks
// that their cluster represents worst are split off into new two-vector clusters until the codebook is full again.
bool etc1s_frontend::introduce_new_endpoint_clusters() {
    finish_prefetches(1);   // nothing rewrites what a pending download reads
    m_ep_member_valid = false;
    generate_block_endpoint_clusters();
    ensure_endpoint_lists();  // list surgery below; folded back into the map at the end
    int want = (int)m_params.m_max_endpoint_clusters - (int)m_endpoint_clusters.size();
    if (want <= 0) return true;
    const uint32_t n = m_total_blocks, k = (uint32_t)m_endpoint_clusters.size();
    std::vector<uint8_t> prm(k * 4ull);
    for (uint32_t i = 0; i < k; i++) {
        const endpoint_params& e = m_endpoint_cluster_etc_params[i];
        prm[i * 4] = e.r; prm[i * 4 + 1] = e.g; prm[i * 4 + 2] = e.b; prm[i * 4 + 3] = e.inten;
    }
    device_state& d = *m_dev;
    if (!ensure_endpoint_map_device() || !d.upload(d.params, prm.data(), prm.size()) || !d.reserve(d.err, (size_t)n * 2 * 8)) return fail("upload");
    if (!bu_hip_k_subblock_errors(d.ctx, d.d_pixels, n, (const uint32_t*)d.block_cluster.p, (const uint8_t*)d.params.p, m_params.m_perceptual, (uint64_t*)d.err.p))
        return fail("bu_hip_k_subblock_errors");
    std::vector<uint64_t> err((size_t)n * 2);
    if (!d.download(err.data(), d.err, err.size())) return fail("download sub-block errors");
    // The reference sorts (error, block, sub-block) ascending and takes candidates from the back (:1089, :1117-1119)
    std::vector<uint32_t> order((size_t)n * 2);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return err[a] < err[b] || (err[a] == err[b] && a < b); });
    std::vector<uint32_t> cluster_sizes(k);
    for (uint32_t i = 0; i < k; i++) cluster_sizes[i] = (uint32_t)m_endpoint_clusters[i].size();
    std::vector<uint8_t> relocated((size_t)n * 2, 0), ignore_cluster(k, 0);
    for (size_t pos = order.size(); pos-- > 0 && want > 0;) {
        const uint32_t tv = order[pos];
        const uint32_t ci = m_block_endpoint_cluster[tv >> 1];
        if (ignore_cluster[ci]) continue;
        if (cluster_sizes[ci] <= 2) continue;
        if (relocated[tv] || relocated[tv ^ 1]) continue;
        m_endpoint_clusters.push_back({tv, tv ^ 1});
        m_endpoint_cluster_etc_params.emplace_back();
        relocated[tv] = relocated[tv ^ 1] = 1;
        cluster_sizes[ci] -= 2;
        ignore_cluster[ci] = 1;
        want--;
    }
    for (uint32_t i = 0; i < k; i++) {
        std::vector<uint32_t>& l = m_endpoint_clusters[i];
        l.erase(std::remove_if(l.begin(), l.end(), [&](uint32_t tv) { return relocated[tv] != 0; }), l.end());
    }
    m_endpoint_map_valid = false; m_ep_dev_valid = false;
    generate_block_endpoint_clusters();
    return true;
}

// frontend.cpp:2718-2976. ETC1S blocks always have the differential bit set, so only the "colour5" branch of the reference exists
// here. The per-cluster sub-block lists are the reference's m_subblocks, which are appended to on every call and never cleared.
bool etc1s_frontend::refine_block_endpoints_given_selectors(uint32_t* total_refined) {
    finish_prefetches();   // nothing rewrites what a pending download reads
    m_ep_member_valid = false;
    const uint32_t n = m_total_blocks, k = (uint32_t)m_endpoint_cluster_etc_params.size();
    ensure_endpoint_map(); ensure_encoded_host();
    m_endpoint_cluster_subblocks.resize(k);
    for (uint32_t b = 0; b < n; b++) {
        std::vector<uint32_t>& l = m_endpoint_cluster_subblocks[m_block_endpoint_cluster[b]];
        l.push_back(b * 2); l.push_back(b * 2 + 1);
    }
    csr lists; lists.build(m_endpoint_cluster_subblocks);
    device_state& d = *m_dev;
    if (!d.upload(d.offsets, lists.offsets.data(), lists.offsets.size()) || !d.upload(d.indices, lists.indices.data(), lists.indices.size()) ||
        !ensure_encoded_device() || !d.reserve(d.params, (size_t)k * 4) || !d.reserve(d.err, (size_t)k * 8) || !d.reserve(d.valid, k) ||
        !d.reserve(d.weights, (size_t)k * 8))
        return fail("upload");
    if (!bu_hip_k_refit_endpoints_given_selectors(d.ctx, d.d_pixels, d.enc.p, k, lists.offsets.data(), (const uint32_t*)d.offsets.p, (const uint32_t*)d.indices.p,
                                                  m_params.m_perceptual, (uint8_t*)d.params.p, (uint64_t*)d.err.p, (uint8_t*)d.valid.p, (uint64_t*)d.weights.p))
        return fail("bu_hip_k_refit_endpoints_given_selectors");
    std::vector<uint8_t> prm(k * 4ull), valid(k);
    std::vector<uint64_t> err(k), cur(k);
    if (!d.download(prm.data(), d.params, prm.size()) || !d.download(err.data(), d.err, k) || !d.download(valid.data(), d.valid, k) || !d.download(cur.data(), d.weights, k))
        return fail("download");
    uint32_t refined = 0;
    for (uint32_t ci = 0; ci < k; ci++) {
        const std::vector<uint32_t>& subs = m_endpoint_cluster_subblocks[ci];
        if (subs.empty() || !valid[ci] || !(err[ci] < cur[ci])) continue;
        const int nr = prm[ci * 4], ng = prm[ci * 4 + 1], nb = prm[ci * 4 + 2];
        const uint32_t ninten = prm[ci * 4 + 3];
        // two passes (:2838-2921): first check that every (old, new) colour pair of every listed sub-block packs as colour5 + delta3,
        // then apply. The check is made against the block as it stands at that moment, which only matters in the second pass.
        bool all_passed = true;
        for (uint32_t pass = 0; pass < 2 && all_passed; pass++) {
            for (uint32_t tv : subs) {
                bu_etc_block& blk = m_encoded_blocks[tv >> 1];
                const uint32_t sub = tv & 1;
                uint64_t v = load_be64(blk);
                int c[2][3];
                const int dl[3] = { (int)((v >> 56) & 7), (int)((v >> 48) & 7), (int)((v >> 40) & 7) };
                const int base[3] = { (int)((v >> 59) & 31), (int)((v >> 51) & 31), (int)((v >> 43) & 31) };
                for (int ch = 0; ch < 3; ch++) {
                    c[0][ch] = base[ch];
                    c[1][ch] = base[ch] + (dl[ch] >= 4 ? dl[ch] - 8 : dl[ch]);  // unpack_color5 with delta3 (etc.cpp:413-439); in range for ETC1S blocks
                }
                c[sub][0] = nr; c[sub][1] = ng; c[sub][2] = nb;
                bool ok = true;
                for (int ch = 0; ch < 3; ch++) { const int dd = c[1][ch] - c[0][ch]; ok = ok && dd >= -4 && dd <= 3; }  // try_pack_color5_delta3
                if (!ok) { all_passed = false; break; }
                if (pass == 1) {
                    // set_block_color5 + set_inten_table(sub) (etc.h:633-646, 203-214)
                    v &= ~((0xFFull << 56) | (0xFFull << 48) | (0xFFull << 40));
                    for (int ch = 0; ch < 3; ch++) {
                        const int dd = c[1][ch] - c[0][ch];
                        v |= ((uint64_t)(((uint32_t)c[0][ch] << 3) | ((uint32_t)dd & 7u))) << (56 - 8 * ch);
                    }
                    const int shift = sub ? 34 : 37;
                    v = (v & ~(7ull << shift)) | ((uint64_t)ninten << shift);
                    store_be64(blk, v);
                    refined++;
                }
            }
        }
        if (all_passed) {
            endpoint_params& e = m_endpoint_cluster_etc_params[ci];
            e.r = (uint8_t)nr; e.g = (uint8_t)ng; e.b = (uint8_t)nb; e.inten = (uint8_t)ninten; e.color_error = err[ci];
        }
    }
    if (refined) m_enc_dev_valid = false;  // the host copy changed: the next selector pass uploads it (ensure_encoded_device)
    if (total_refined) *total_refined = refined;
    return true;
}

// frontend.cpp:2996-3220, called by the backend (etc1s_backend.cpp) after its rate-distortion passes moved blocks to other endpoint
// clusters. Every cluster of the NEW assignment is refitted with its blocks' selectors held fixed (the device fit of row a15 at the
// quality the reference uses here) and keeps the refit where it lowers the error. With optimize_final_codebook the codebook is then
// compacted (unused clusters dropped, equal ones merged), the block map and the encoded blocks follow, and old_to_new says where
// every old cluster went (-1: unused).
bool etc1s_frontend::reoptimize_remapped_endpoints(const std::vector<uint32_t>& new_block_endpoints, std::vector<int>& old_to_new, bool optimize_final_codebook,
                                                   const std::vector<uint32_t>* block_selector_indices) {
    finish_prefetches();
    m_ep_member_valid = false;
    if (!m_dev) return fail("reoptimize_remapped_endpoints: the context this frontend was initialised on has been destroyed");
    ensure_endpoint_map(); ensure_encoded_host(); ensure_selector_map_host();
    const uint32_t n = m_total_blocks, k = m_endpoint_cluster_count;
    if (new_block_endpoints.size() != n || m_endpoint_cluster_etc_params.size() != k) return fail("reoptimize_remapped_endpoints: size mismatch");
    // the blocks of every cluster under the new assignment, ascending, and the blocks as they would be coded with it
    std::vector<uint32_t> offsets((size_t)k + 1, 0), indices((size_t)n * 2), fill;
    for (uint32_t b = 0; b < n; b++) { if (new_block_endpoints[b] >= k) return fail("reoptimize_remapped_endpoints: bad index"); offsets[new_block_endpoints[b] + 1] += 2; }
    for (uint32_t c = 0; c < k; c++) offsets[c + 1] += offsets[c];
    fill.assign(offsets.begin(), offsets.end() - 1);
    std::vector<bu_etc_block> trial(n);
    for (uint32_t b = 0; b < n; b++) {
        const uint32_t c = new_block_endpoints[b];
        indices[fill[c]++] = b * 2; indices[fill[c]++] = b * 2 + 1;
        const endpoint_params& e = m_endpoint_cluster_etc_params[c];
        const uint32_t sc = block_selector_indices ? (*block_selector_indices)[b] : m_block_selector_cluster_index[b];
        store_be64(trial[b], ((uint64_t)e.r << 59) | ((uint64_t)e.g << 51) | ((uint64_t)e.b << 43) | ((uint64_t)e.inten << 37) | ((uint64_t)e.inten << 34) | (3ull << 32) |
                                 raw_selector_bits(m_optimized_cluster_selectors[sc]));
    }
    device_state& d = *m_dev;
    const int quality = m_params.m_compression_level == 6 ? BU_ETC_QUALITY_UBER : BU_ETC_QUALITY_SLOW;  // frontend.cpp:3073-3076
    if (!d.upload(d.offsets, offsets.data(), offsets.size()) || !d.upload(d.indices, indices.data(), indices.size()) || !d.upload(d.enc, trial.data(), n) ||
        !d.reserve(d.params, (size_t)k * 4) || !d.reserve(d.err, (size_t)k * 8) || !d.reserve(d.valid, k) || !d.reserve(d.weights, (size_t)k * 8))
        return fail("upload");
    m_enc_dev_valid = false;   // d.enc now holds the trial blocks
    if (!bu_hip_k_refit_endpoints_given_selectors_q(d.ctx, d.d_pixels, d.enc.p, k, offsets.data(), (const uint32_t*)d.offsets.p, (const uint32_t*)d.indices.p, quality,
                                                    m_params.m_perceptual, (uint8_t*)d.params.p, (uint64_t*)d.err.p, (uint8_t*)d.valid.p, (uint64_t*)d.weights.p))
        return fail("bu_hip_k_refit_endpoints_given_selectors_q");
    std::vector<uint8_t> prm(k * 4ull), valid(k);
    std::vector<uint64_t> err(k), cur(k);
    if (!d.download(prm.data(), d.params, prm.size()) || !d.download(err.data(), d.err, k) || !d.download(valid.data(), d.valid, k) || !d.download(cur.data(), d.weights, k))
        return fail("download");
    old_to_new.assign(k, -1);
    uint32_t kept = 0;
    for (uint32_t c = 0; c < k; c++) {
        if (offsets[c + 1] == offsets[c]) continue;
        old_to_new[c] = (int)kept++;
        if (valid[c] && err[c] < cur[c]) {
            endpoint_params& e = m_endpoint_cluster_etc_params[c];
            e.r = prm[c * 4]; e.g = prm[c * 4 + 1]; e.b = prm[c * 4 + 2]; e.inten = prm[c * 4 + 3]; e.color_error = err[c]; e.valid = true; e.color_used = true;
        }
    }
    if (!optimize_final_codebook) return true;

    // compaction in old order, then the same sort-and-merge every codebook iteration ends with (frontend.cpp:3131-3199)
    std::vector<endpoint_params> params(kept);
    std::vector<uint32_t> sizes(kept);
    for (uint32_t c = 0; c < k; c++)
        if (old_to_new[c] >= 0) { params[old_to_new[c]] = m_endpoint_cluster_etc_params[c]; sizes[old_to_new[c]] = (offsets[c + 1] - offsets[c]) / 2; }
    for (uint32_t c = 0; c < k; c++)
        for (uint32_t i = offsets[c]; i < offsets[c + 1]; i += 2) { const uint32_t b = indices[i] >> 1; m_block_endpoint_cluster[b] = (uint32_t)old_to_new[c]; m_block_endpoint_pos[b] = (i - offsets[c]) / 2; }
    m_endpoint_cluster_etc_params.swap(params);
    m_endpoint_cluster_sizes.swap(sizes);
    m_endpoint_cluster_count = kept;
    m_endpoint_map_valid = true; m_ep_dev_valid = false; m_endpoint_lists_valid = false; m_endpoint_clusters.clear();
    m_endpoint_cluster_subblocks.clear();
    if (!eliminate_redundant_or_empty_endpoint_clusters()) return false;
    for (uint32_t b = 0; b < n; b++) old_to_new[new_block_endpoints[b]] = (int)m_block_endpoint_cluster[b];
    for (uint32_t b = 0; b < n; b++) {
        const endpoint_params& e = m_endpoint_cluster_etc_params[m_block_endpoint_cluster[b]];
        const uint64_t v = load_be64(m_encoded_blocks[b]);
        const uint64_t keep = v & 0x3FFFFFFFFull;  // diff + flip bits and the selectors
        store_be64(m_encoded_blocks[b], ((uint64_t)e.r << 59) | ((uint64_t)e.g << 51) | ((uint64_t)e.b << 43) | ((uint64_t)e.inten << 37) | ((uint64_t)e.inten << 34) | keep);
    }
    return true;
}

// frontend.cpp:1648-1945
bool etc1s_frontend::refine_endpoint_clusterization(uint32_t* total_reassigned) {
    finish_prefetches(1);   // nothing rewrites what a pending download reads
    if (!ensure_endpoint_map_device()) return false;
    if (m_use_hierarchical_endpoint_codebooks) compute_endpoint_clusters_within_each_parent_cluster();
    const uint32_t n = m_total_blocks, k = m_endpoint_cluster_count;
    std::vector<uint8_t> prm(k * 4ull);
    for (uint32_t i = 0; i < k; i++) {
        const endpoint_params& e = m_endpoint_cluster_etc_params[i];
        prm[i * 4] = e.r; prm[i * 4 + 1] = e.g; prm[i * 4 + 2] = e.b; prm[i * 4 + 3] = e.inten;
    }
    device_state& d = *m_dev;
    uint32_t n_parents = 0;
    if (m_use_hierarchical_endpoint_codebooks) {
        if (m_endpoint_clusters_within_each_parent_cluster.size() != m_endpoint_parent_count) return false;   // the membership pass failed (error already set)
        csr cand; cand.build(m_endpoint_clusters_within_each_parent_cluster);
        n_parents = (uint32_t)m_endpoint_clusters_within_each_parent_cluster.size();
        if (!d.upload(d.cand_offsets, cand.offsets.data(), cand.offsets.size()) || !d.upload(d.cand_indices, cand.indices.data(), cand.indices.size()))
            return fail("upload parent lists");
    }
    const size_t padded = (size_t)comm_world() * slab_blocks();
    if (!d.upload(d.params, prm.data(), prm.size()) || !d.reserve(d.out_u32, padded * 4) || !d.reserve(d.map_sizes, ((size_t)k + 2) * 4) ||
        !d.reserve(d.map_offs, ((size_t)k + 1) * 4) || !d.reserve(d.map_sorted, (size_t)n * 4))
        return fail("upload refine inputs");
    uint32_t b0, nb;
    my_slab(b0, nb);
    if (nb && !bu_hip_k_refine_endpoint_clusterization(d.ctx, (const char*)d.d_pixels + (size_t)b0 * 64, nb, (const uint32_t*)d.block_cluster.p + b0, (const uint8_t*)d.params.p, k,
                                                       n_parents, (const uint32_t*)d.cand_offsets.p, (const uint32_t*)d.cand_indices.p,
                                                       n_parents ? (const uint8_t*)d.ep_parent.p + b0 : nullptr, m_params.m_perceptual, (uint32_t*)d.out_u32.p + b0))
        return fail("bu_hip_k_refine_endpoint_clusterization");
    if (!gather_blocks(d.out_u32.p, 4)) return false;
    // frontend.cpp:1921-1942 rebuilds the cluster lists in block order (empty clusters stay, they are removed by eliminate_...): the new
    // position of a block is its rank among its cluster's blocks -- a stable sort of the block ids by their new cluster, on the device.
    // (the number of moved blocks lands behind the cluster sizes, word k + 1: both come back in one copy)
    if (!bu_hip_k_map_count_differences(d.ctx, (const uint32_t*)d.block_cluster.p, (const uint32_t*)d.out_u32.p, n, (uint32_t*)d.map_sizes.p + k + 1) ||
        !bu_hip_k_map_rank_blocks(d.ctx, (const uint32_t*)d.out_u32.p, n, k, (uint32_t*)d.map_sizes.p, (uint32_t*)d.map_offs.p, (uint32_t*)d.map_sorted.p, (uint32_t*)d.ep_pos.p))
        return fail("bu_hip_k_map_rank_blocks");
    uint32_t moved = 0;
    m_endpoint_cluster_sizes.assign((size_t)k + 2, 0);
    if (!d.download(m_endpoint_cluster_sizes.data(), d.map_sizes, (size_t)k + 2)) return fail("download refine result");
    moved = m_endpoint_cluster_sizes[(size_t)k + 1];
    m_endpoint_cluster_sizes.resize(k);
    std::swap(d.block_cluster, d.out_u32);   // the reassignment IS the clustering now
    if (d.block_cluster.cap < (size_t)n * 4 || d.out_u32.cap < (size_t)n * 4) return fail("refine buffers");
    m_ep_dev_valid = true; m_endpoint_map_valid = false; m_endpoint_lists_valid = false; m_endpoint_clusters.clear();
    if (total_reassigned) *total_reassigned = moved;
    return true;
}

// frontend.cpp:1947-2012. The ordering comes from indirect_sort = std::sort over indices with operator< on the parameters
// (frontend.h:248-267): (r, g, b, a=255) of the colour, then the (all-zero) second colour, then inten. std::sort is not stable,
// so we call the very same algorithm with an equivalent comparator to get the same permutation among equal keys.
bool etc1s_frontend::eliminate_redundant_or_empty_endpoint_clusters() {
    finish_prefetches(1);   // nothing rewrites what a pending download reads
    m_ep_member_valid = false;
    if (!m_ep_dev_valid) ensure_endpoint_map();
    const uint32_t k = m_endpoint_cluster_count, n = m_total_blocks;
    const std::vector<endpoint_params>& P = m_endpoint_cluster_etc_params;
    auto key = [&](uint32_t i) { return ((uint32_t)P[i].r << 24) | ((uint32_t)P[i].g << 16) | ((uint32_t)P[i].b << 8) | P[i].inten; };
    // (the keys travel with the indices: what std::sort does to a sequence depends on the comparisons' outcomes only, so this is the permutation the sort of the bare
    // indices through an indirect comparator gives -- without a look-up per comparison; the device is idle while this runs)
    std::vector<std::pair<uint32_t, uint32_t>> keyed(k);
    for (uint32_t i = 0; i < k; i++) keyed[i] = std::make_pair(key(i), i);
    std::sort(keyed.begin(), keyed.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
    std::vector<uint32_t> order(k);
    for (uint32_t i = 0; i < k; i++) order[i] = keyed[i].second;

    // A run of equal parameters becomes one cluster: the first non-empty one followed by the members of the others, list after list. In map
    // form: every old cluster gets its new index and the offset its list starts at inside the merged list.
    std::vector<uint32_t> new_index(k, 0), base(k, 0), sizes;
    std::vector<endpoint_params> params;
    for (uint32_t i = 0; i < k;) {
        const uint32_t ci = order[i];
        if (!m_endpoint_cluster_sizes[ci]) { i++; continue; }
        uint32_t j = i + 1;
        while (j < k && key(order[j]) == key(ci)) j++;
        const uint32_t ni = (uint32_t)params.size();
        params.push_back(P[ci]);
        uint32_t run = 0;
        for (uint32_t t = i; t < j; t++) { new_index[order[t]] = ni; base[order[t]] = run; run += m_endpoint_cluster_sizes[order[t]]; }
        sizes.push_back(run);
        i = j;
    }
    if (m_endpoint_map_valid)
        parallel_for(n, [&](uint32_t b0, uint32_t b1) {
            for (uint32_t b = b0; b < b1; b++) {
                const uint32_t old = m_block_endpoint_cluster[b];
                m_block_endpoint_pos[b] += base[old];
                m_block_endpoint_cluster[b] = new_index[old];
            }
        });
    if (m_ep_dev_valid) {
        device_state& d = *m_dev;
        if (!d.upload(d.tmp_a, new_index.data(), k) || !d.upload(d.tmp_b, base.data(), k) ||
            !bu_hip_k_map_remap(d.ctx, (uint32_t*)d.block_cluster.p, (uint32_t*)d.ep_pos.p, n, (const uint32_t*)d.tmp_a.p, (const uint32_t*)d.tmp_b.p))
            return fail("bu_hip_k_map_remap");
    }
    m_endpoint_cluster_sizes.swap(sizes);
    m_endpoint_cluster_count = (uint32_t)params.size();
    m_endpoint_cluster_etc_params.swap(params);
    m_endpoint_lists_valid = false; m_endpoint_clusters.clear();
    return true;
}

// frontend.cpp:2014-2096
bool etc1s_frontend::create_initial_packed_texture() {
    finish_prefetches(2);   // nothing rewrites what a pending download reads
    const uint32_t n = m_total_blocks, k = (uint32_t)m_endpoint_cluster_etc_params.size();
    std::vector<uint8_t> prm(k * 4ull);
    for (uint32_t i = 0; i < k; i++) {
        const endpoint_params& e = m_endpoint_cluster_etc_params[i];
        prm[i * 4] = e.r; prm[i * 4 + 1] = e.g; prm[i * 4 + 2] = e.b; prm[i * 4 + 3] = e.inten;
    }
    device_state& d = *m_dev;
    const size_t padded = (size_t)comm_world() * slab_blocks();
    if (!ensure_endpoint_map_device() || !d.upload(d.params, prm.data(), prm.size()) || !d.reserve(d.enc, padded * 8) || !d.reserve(d.orig_enc, (size_t)n * 8)) return fail("upload");
    if (!bu_hip_k_determine_selectors(d.ctx, d.d_pixels, n, (const uint8_t*)d.params.p, (const uint32_t*)d.block_cluster.p, m_params.m_perceptual, d.enc.p))
        return fail("bu_hip_k_determine_selectors");
    // m_orig_encoded_blocks = m_encoded_blocks (frontend.cpp:2093): both stay resident, the host copies are fetched when somebody asks
    if (!bu_hip_memcpy_d2d(d.ctx, d.orig_enc.p, d.enc.p, (size_t)n * 8)) return fail("copy");
    m_enc_dev_valid = true; m_enc_host_valid = false; m_orig_host_valid = false;
    return true;
}

// ---- the encoded blocks live where they were last written; the other side is brought up to date on demand
void etc1s_frontend::ensure_encoded_host() const {
    if (m_dl_enc) finish_prefetches(2);
    if (m_enc_host_valid) return;
    if (!m_enc_dev_valid) { m_encoded_blocks.assign(m_total_blocks, bu_etc_block{}); m_enc_host_valid = true; return; }   // nothing made yet: init()'s zeros
    m_encoded_blocks.resize(m_total_blocks);
    if (m_dev && m_dev->enc.p && m_dev->download(m_encoded_blocks.data(), m_dev->enc, m_total_blocks)) m_enc_host_valid = true;
}
void etc1s_frontend::ensure_orig_encoded_host() const {
    if (m_orig_host_valid) return;
    m_orig_encoded_blocks.resize(m_total_blocks);
    if (m_dev && m_dev->orig_enc.p && m_dev->download(m_orig_encoded_blocks.data(), m_dev->orig_enc, m_total_blocks)) m_orig_host_valid = true;
}
bool etc1s_frontend::ensure_encoded_device() {
    if (m_enc_dev_valid) return true;
    finish_prefetches(2);   // nothing rewrites what a pending download reads
    device_state& d = *m_dev;
    const size_t padded = (size_t)comm_world() * slab_blocks();
    ensure_encoded_host();
    if (!m_enc_host_valid || !d.reserve(d.enc, padded * 8) || !d.upload(d.enc, m_encoded_blocks.data(), m_total_blocks)) return fail("upload encoded blocks");
    m_enc_dev_valid = true;
    return true;
}

// frontend.cpp:2140-2257: selector training vectors (16 selector values as floats, weight from the endpoint colour spread) +
// de-duplication + TSVQ. The std::map order of vec16F (enc.h:382) is the numeric order of the 32-bit word holding s(0,0) in
// its top two bits ... s(3,3) in its bottom two, so the distinct vectors come out of an LSD radix sort of those words.
bool etc1s_frontend::generate_selector_clusters() {
    const uint32_t n = m_total_blocks;
    device_state& d = *m_dev;
    timer sub;
    auto lap = [&](const char* name) { m_stage_times.push_back(stage_time{name, sub.seconds()}); sub = timer(); };
    if (!ensure_encoded_device() || !d.reserve(d.weights, (size_t)n * 8)) return fail("alloc");
    if (!bu_hip_k_selector_training_vectors(d.ctx, d.enc.p, n, m_params.m_perceptual, nullptr, (uint64_t*)d.weights.p)) return fail("bu_hip_k_selector_training_vectors");
    lap("~gsc/weights");
    // De-duplication on the device (unique_kernels.hip): keys of the resident blocks, stable sort, run lengths, exact weight sums. The host
    // only needs the grouping (which blocks share a vector) to turn TSVQ leaves back into block lists.
    uint32_t u_total = 0;
    if (!d.reserve(d.sel_idx, (size_t)n * 4) || !d.reserve(d.sel_ukeys, (size_t)n * 4) || !d.reserve(d.sel_uw, (size_t)n * 8) || !d.reserve(d.sel_goffs, ((size_t)n + 1) * 4))
        return fail("alloc");
    if (!bu_hip_k_unique_selector_vectors(d.ctx, d.enc.p, (const uint64_t*)d.weights.p, n, (uint32_t*)d.sel_idx.p, (uint32_t*)d.sel_ukeys.p, (uint64_t*)d.sel_uw.p,
                                          (uint32_t*)d.sel_goffs.p, &u_total))
        return fail("bu_hip_k_unique_selector_vectors");
    m_selector_group_blocks.clear(); m_selector_group_offsets.clear();   // the grouping stays resident (d.sel_idx / d.sel_goffs)
    const csr_groups groups{nullptr, nullptr};                           // never dereferenced: only per-vector results are asked for
    lap("~gsc/unique");
    const uint32_t parent_default = (m_params.m_compression_level <= 1) ? kSelectorParentCodebookSizeLevel01 : kSelectorParentCodebookSizeDefault;
    const uint32_t parent_size = (m_params.m_max_selector_clusters >= 256) ? parent_default : 0;
    const bool hier = m_use_hierarchical_selector_codebooks && parent_size;
    device_tsvq::stats ts;
    // leaf and parent of every distinct vector are written on the device (tmp_a / tmp_b), then spread over the vectors' blocks
    if (!d.reserve(d.tmp_a, (size_t)u_total * 4) || !d.reserve(d.tmp_b, (size_t)u_total * 4) || !d.reserve(d.sel_cluster, (size_t)comm_world() * slab_blocks() * 4) ||
        !d.reserve(d.sel_parent, n))
        return fail("alloc");
    if (m_params.m_fast_codebooks && !std::getenv("BU_FAST_ENDPOINTS_ONLY")) {   // row f3
        if (!bu_hip_kmeans_codebook(d.ctx, 0, d.sel_ukeys.p, (const uint64_t*)d.sel_uw.p, nullptr, u_total, m_params.m_max_selector_clusters, hier ? parent_size : 0,
                                    m_params.m_fast_codebook_iterations, (uint32_t*)d.tmp_a.p, hier ? (uint32_t*)d.tmp_b.p : nullptr, &m_selector_cluster_count,
                                    &m_selector_parent_count))
            return fail("bu_hip_kmeans_codebook (selectors)");
        ts.t_device = sub.seconds();
    } else
    if (!device_tsvq::hierarchical_codebook_packed16_device(d.ctx, (const uint32_t*)d.sel_ukeys.p, (const uint64_t*)d.sel_uw.p, u_total, groups, m_params.m_max_selector_clusters,
                                                            hier ? parent_size : 0, m_selector_cluster_block_indices, m_selector_parent_cluster_block_indices, &ts, nullptr,
                                                            &m_selector_parent_count, nullptr, &m_selector_cluster_count, (uint32_t*)d.tmp_a.p, (uint32_t*)d.tmp_b.p,
                                                            m_has_comm ? &m_comm : nullptr, m_params.m_codebook_threads))
        return fail("selector TSVQ failed");
    m_stage_times.push_back(stage_time{"~gsc/tsvq_create", ts.t_create});
    m_stage_times.push_back(stage_time{"~gsc/tsvq_device", ts.t_device});
    m_stage_times.push_back(stage_time{"~gsc/tsvq_replay", ts.t_replay});
    m_stage_times.push_back(stage_time{"~gsc/tsvq_expand", ts.t_expand});
    sub = timer();
    // the clustering is kept as a resident block -> cluster map (+ block -> parent); lists are built when somebody asks
    const bool parents = hier && m_selector_parent_count;
    if (!bu_hip_k_map_blocks_from_groups(d.ctx, (const uint32_t*)d.sel_goffs.p, (const uint32_t*)d.sel_idx.p, n, u_total, (const uint32_t*)d.tmp_a.p, nullptr,
                                         parents ? (const uint32_t*)d.tmp_b.p : nullptr, (uint32_t*)d.sel_cluster.p, nullptr, (uint8_t*)d.sel_parent.p))
        return fail("bu_hip_k_map_blocks_from_groups");
    if (m_use_hierarchical_selector_codebooks && !m_selector_parent_count) m_selector_parent_count = 1;  // no parent level: one parent holding everything (frontend.cpp:2230-2236)
    m_selector_cluster_block_indices.clear(); m_selector_parent_cluster_block_indices.clear();
    m_selector_lists_valid = false;
    m_sel_dev_valid = true; m_sel_host_valid = false;
    lap("~gsc/parents");
    return true;
}

// frontend.cpp:2098-2138
void etc1s_frontend::compute_selector_clusters_within_each_parent_cluster() {
    const size_t parents = m_selector_parent_count, clusters = m_selector_cluster_count;
    m_selector_clusters_within_each_parent_cluster.assign(parents, {});
    device_state& d = *m_dev;
    std::vector<uint8_t> member(parents * clusters, 0);
    if (!ensure_selector_map_device() || !d.reserve(d.flags, parents * clusters + 8) ||
        !bu_hip_k_map_membership(d.ctx, (const uint8_t*)d.sel_parent.p, (const uint32_t*)d.sel_cluster.p, m_total_blocks, (uint32_t)parents, (uint32_t)clusters, (uint8_t*)d.flags.p) ||
        !d.download(member.data(), d.flags, member.size())) {
        m_selector_clusters_within_each_parent_cluster.clear();   // find_optimal_selector_clusters_for_each_block checks the size
        fail("compute_selector_clusters_within_each_parent_cluster");
        return;
    }
    for (size_t p = 0; p < parents; p++)
        for (size_t c = 0; c < clusters; c++)
            if (member[p * clusters + c]) m_selector_clusters_within_each_parent_cluster[p].push_back((uint32_t)c);
}

// ---- the selector clustering (block -> cluster) lives where it was last written; the other side is brought up to date on demand
void etc1s_frontend::ensure_selector_map_host() const {
    if (m_sel_host_valid) return;
    m_block_selector_cluster_index.resize(m_total_blocks);
    if (m_dev && m_dev->sel_cluster.p && m_dev->download(m_block_selector_cluster_index.data(), m_dev->sel_cluster, m_total_blocks)) m_sel_host_valid = true;
}
bool etc1s_frontend::ensure_selector_map_device() {
    if (m_sel_dev_valid) return true;
    device_state& d = *m_dev;
    ensure_selector_map_host();
    if (!m_sel_host_valid || !d.reserve(d.sel_cluster, (size_t)comm_world() * slab_blocks() * 4) || !d.upload(d.sel_cluster, m_block_selector_cluster_index.data(), m_total_blocks))
        return fail("upload selector map");
    m_sel_dev_valid = true;
    return true;
}

// The blocks of every selector cluster, ascending, as one CSR array (what the per-cluster kernels read)
void etc1s_frontend::selector_csr(std::vector<uint32_t>& offsets, std::vector<uint32_t>& indices) const {
    ensure_selector_map_host();
    const uint32_t n = m_total_blocks, k = m_selector_cluster_count;
    const unsigned T = n > 65536 ? host_threads() : 1;
    const uint32_t per = (n + T - 1) / T;
    std::vector<uint32_t> hist((size_t)T * k, 0);
    offsets.assign((size_t)k + 1, 0); indices.resize(n);
    parallel_for_chunks(T, [&](unsigned t) {
        uint32_t* h = &hist[(size_t)t * k];
        const uint32_t a = t * per, b = std::min(n, a + per);
        for (uint32_t i = a; i < b; i++) h[m_block_selector_cluster_index[i]]++;
    });
    uint32_t run = 0;
    for (uint32_t c = 0; c < k; c++) {
        offsets[c] = run;
        for (unsigned t = 0; t < T; t++) { const uint32_t v = hist[(size_t)t * k + c]; hist[(size_t)t * k + c] = run; run += v; }
    }
    offsets[k] = run;
    parallel_for_chunks(T, [&](unsigned t) {
        uint32_t* cur = &hist[(size_t)t * k];
        const uint32_t a = t * per, b = std::min(n, a + per);
        for (uint32_t i = a; i < b; i++) indices[cur[m_block_selector_cluster_index[i]]++] = i;
    });
}

const std::vector<std::vector<uint32_t>>& etc1s_frontend::selector_cluster_block_indices() const {
    if (!m_selector_lists_valid) {
        std::vector<uint32_t> offsets, indices;
        selector_csr(offsets, indices);
        m_selector_cluster_block_indices.assign(m_selector_cluster_count, {});
        for (uint32_t c = 0; c < m_selector_cluster_count; c++) m_selector_cluster_block_indices[c].assign(indices.begin() + offsets[c], indices.begin() + offsets[c + 1]);
        m_selector_lists_valid = true;
    }
    return m_selector_cluster_block_indices;
}

// frontend.cpp:2259-2354
bool etc1s_frontend::create_optimized_selector_codebook(uint32_t /*iter*/) {
    const uint32_t k = m_selector_cluster_count, n = m_total_blocks;
    m_optimized_cluster_selectors.resize(k, bu_etc_block{});
    device_state& d = *m_dev;
    // the blocks of every cluster (the accumulation is integer, so their order does not matter: ascending) = a stable sort of the block
    // ids by cluster, on the device: offsets in map_offs, block ids in map_sorted
    if (!ensure_selector_map_device() || !ensure_encoded_device() || !d.reserve(d.map_sizes, ((size_t)k + 1) * 4) || !d.reserve(d.map_offs, ((size_t)k + 1) * 4) ||
        !d.reserve(d.map_sorted, (size_t)n * 4) ||
        !bu_hip_k_map_rank_blocks(d.ctx, (const uint32_t*)d.sel_cluster.p, n, k, (uint32_t*)d.map_sizes.p, (uint32_t*)d.map_offs.p, (uint32_t*)d.map_sorted.p, nullptr))
        return fail("bu_hip_k_map_rank_blocks");
    // multi-GPU: every rank takes a contiguous range of clusters holding about 1/world of the member blocks; entries it does not own
    // are uploaded as zero, so that the sum-merge below reassembles the codebook exactly
    uint32_t c0 = 0, c1 = k;
    std::vector<bu_etc_block> mine(m_optimized_cluster_selectors);
    if (m_has_comm) {
        std::vector<uint32_t> offsets((size_t)k + 1);
        if (!d.download(offsets.data(), d.map_offs, (size_t)k + 1)) return fail("download offsets");
        const uint64_t total = offsets[k], w = comm_world(), r = comm_rank();
        auto cut = [&](uint64_t part) { return (uint32_t)(std::lower_bound(offsets.begin(), offsets.end(), (uint32_t)(total * part / w)) - offsets.begin()); };
        c0 = r == 0 ? 0 : std::min(cut(r), k);
        c1 = r + 1 == w ? k : std::min(cut(r + 1), k);
        if (c1 < c0) c1 = c0;
        for (uint32_t i = 0; i < k; i++) if (i < c0 || i >= c1) mine[i] = bu_etc_block{};
    }
    if (!m_has_comm) {
        // codebook (8 bytes per cluster) and, while the parent lists are still to be made, the (parent, cluster) membership table behind it: one buffer, one download
        const size_t parents = m_selector_parent_count;
        const bool with_members = m_use_hierarchical_selector_codebooks && m_params.m_compression_level != 0 && parents && m_selector_clusters_within_each_parent_cluster.size() != parents &&
                                  parents * (size_t)k <= ((size_t)64 << 20);
        const size_t at_member = ((size_t)k * 8 + 15) & ~(size_t)15, total = at_member + (with_members ? parents * k : 0);
        if (!d.reserve(d.sel_blocks, total + 16) || !d.upload(d.sel_blocks, mine.data(), k)) return fail("upload selector clusters");
        if (with_members && !bu_hip_k_map_membership(d.ctx, (const uint8_t*)d.sel_parent.p, (const uint32_t*)d.sel_cluster.p, n, (uint32_t)parents, k, (uint8_t*)d.sel_blocks.p + at_member))
            return fail("bu_hip_k_map_membership");
        if (k && !bu_hip_k_create_optimized_selector_codebook(d.ctx, d.d_pixels, d.enc.p, k, (const uint32_t*)d.map_offs.p, (const uint32_t*)d.map_sorted.p, m_params.m_perceptual, (char*)d.sel_blocks.p))
            return fail("bu_hip_k_create_optimized_selector_codebook");
        std::vector<uint8_t> pack(total);
        if (total && !bu_hip_memcpy_d2h(d.ctx, pack.data(), d.sel_blocks.p, total)) return fail("download selector codebook");
        std::memcpy(m_optimized_cluster_selectors.data(), pack.data(), (size_t)k * 8);
        if (with_members) {
            m_selector_clusters_within_each_parent_cluster.assign(parents, {});
            for (size_t p = 0; p < parents; p++)
                for (size_t c = 0; c < k; c++)
                    if (pack[at_member + p * k + c]) m_selector_clusters_within_each_parent_cluster[p].push_back((uint32_t)c);
        }
        m_sel_blocks_dev_valid = true;   // d.sel_blocks holds what the host now holds: find_optimal_selector_clusters_for_each_block need not send it back
        return true;
    }
    if (!d.reserve(d.sel_blocks, (size_t)k * 8 + 8) || !d.upload(d.sel_blocks, mine.data(), k)) return fail("upload selector clusters");
    if (c1 > c0 && !bu_hip_k_create_optimized_selector_codebook(d.ctx, d.d_pixels, d.enc.p, c1 - c0, (const uint32_t*)d.map_offs.p + c0, (const uint32_t*)d.map_sorted.p, m_params.m_perceptual,
                                                               (char*)d.sel_blocks.p + (size_t)c0 * 8))
        return fail("bu_hip_k_create_optimized_selector_codebook");
    if (!merge_disjoint(d.sel_blocks.p, (size_t)k * 8)) return false;
    if (!d.download(m_optimized_cluster_selectors.data(), d.sel_blocks, k)) return fail("download selector codebook");
    return true;
}

// frontend.cpp:2397-2715
bool etc1s_frontend::find_optimal_selector_clusters_for_each_block() {
    finish_prefetches(2);   // nothing rewrites what a pending download reads
    const uint32_t n = m_total_blocks, k = (uint32_t)m_optimized_cluster_selectors.size();
    if (m_params.m_compression_level == 0) {
        // frontend.cpp:2420-2439: blocks stay in their TSVQ cluster and just take its optimised selectors
        ensure_encoded_host(); ensure_selector_map_host();
        for (uint32_t b = 0; b < n; b++) {
            const uint32_t bits = raw_selector_bits(m_optimized_cluster_selectors[m_block_selector_cluster_index[b]]);
            store_be64(m_encoded_blocks[b], (load_be64(m_encoded_blocks[b]) & ~0xFFFFFFFFull) | bits);
        }
        m_enc_dev_valid = false;
        return true;
    }
    device_state& d = *m_dev;
    uint32_t n_parents = 0;
    if (m_use_hierarchical_selector_codebooks) {
        if (m_selector_clusters_within_each_parent_cluster.size() != m_selector_parent_count) compute_selector_clusters_within_each_parent_cluster();
        if (m_selector_clusters_within_each_parent_cluster.size() != m_selector_parent_count) return fail("selector parent lists missing (the membership pass failed)");
        csr cand; cand.build(m_selector_clusters_within_each_parent_cluster);
        n_parents = (uint32_t)m_selector_clusters_within_each_parent_cluster.size();
        if (!d.upload(d.cand_offsets, cand.offsets.data(), cand.offsets.size()) || !d.upload(d.cand_indices, cand.indices.data(), cand.indices.size()))
            return fail("upload selector parent lists");
    }
    const size_t padded = (size_t)comm_world() * slab_blocks();
    // the encoded blocks stay resident from create_initial_packed_texture on unless the host touched them since (ensure_encoded_device)
    const bool codebook_resident = m_sel_blocks_dev_valid && d.sel_blocks.p && d.sel_blocks.cap >= (size_t)k * 8;
    m_sel_blocks_dev_valid = false;
    if (!ensure_encoded_device() || (!codebook_resident && !d.upload(d.sel_blocks, m_optimized_cluster_selectors.data(), k)) || !d.reserve(d.out_u32, padded * 4))
        return fail("upload fosc inputs");
    uint32_t b0, nb;
    my_slab(b0, nb);  // slabs start on multiples of the reference's 2048-block jobs, so the "same tile as the previous block of this job" shortcut sees the same neighbours
    if (nb && !bu_hip_k_find_optimal_selector_clusters(d.ctx, (const char*)d.d_pixels + (size_t)b0 * 64, (char*)d.enc.p + (size_t)b0 * 8, nb, d.sel_blocks.p, k, n_parents,
                                                       (const uint32_t*)d.cand_offsets.p, (const uint32_t*)d.cand_indices.p, n_parents ? (const uint8_t*)d.sel_parent.p + b0 : nullptr,
                                                       m_params.m_perceptual, kFoscJobSize, (uint32_t*)d.out_u32.p + b0))
        return fail("bu_hip_k_find_optimal_selector_clusters");
    if (!gather_blocks(d.enc.p, 8) || !gather_blocks(d.out_u32.p, 4)) return false;
    std::swap(d.sel_cluster, d.out_u32);   // the assignment IS the clustering now; both results stay resident, the host copies are fetched on demand
    m_sel_dev_valid = true; m_sel_host_valid = false;
    m_enc_host_valid = false;
    m_selector_lists_valid = false;  // frontend.cpp:2696-2708 rebuilds the lists in block order: that is what selector_cluster_block_indices() produces
    return true;
}

// frontend.cpp:554-654: guarantee the four flat selector patterns exist. Pure codebook bookkeeping, except that a block whose
// pre-quantisation selectors were flat is moved to the new entry only if that lowers its error -- evaluated here on the few
// candidate tiles (etc_block::evaluate_etc1_error, etc.cpp:640-700).
bool etc1s_frontend::introduce_special_selector_clusters() {
    const uint32_t n = m_total_blocks;
    m_selector_cluster_count = (uint32_t)m_optimized_cluster_selectors.size();
    std::vector<bu_pixel_block> tiles; // fetched lazily, only when a candidate exists and the caller gave us no host copy
    auto tile = [&](uint32_t b) -> const bu_pixel_block* {
        if (m_params.m_pSource_blocks) return &m_params.m_pSource_blocks[b];
        if (tiles.empty()) {
            tiles.resize(n);
            if (!bu_hip_memcpy_d2h(m_dev->ctx, tiles.data(), m_dev->d_pixels, (size_t)n * sizeof(bu_pixel_block))) return nullptr;
        }
        return &tiles[b];
    };
    auto block_error = [&](const bu_pixel_block& px, const endpoint_params& e, uint32_t bits) {
        int colors[4][3];
        const int r = scale5(e.r), g = scale5(e.g), bl = scale5(e.b);
        for (int s = 0; s < 4; s++) { const int yd = kInten[e.inten][s]; colors[s][0] = clamp255(r + yd); colors[s][1] = clamp255(g + yd); colors[s][2] = clamp255(bl + yd); }
        uint64_t total = 0;
        for (uint32_t y = 0; y < 4; y++)
            for (uint32_t x = 0; x < 4; x++) total += color_distance(m_params.m_perceptual, px.m_pixels[y * 4 + x], colors[selector_of(bits, x, y)]);
        return total;
    };
    uint32_t total_relocated = 0;
    for (uint32_t sel = 0; sel < 4; sel++) {
        const uint32_t flat = flat_selector_bits(sel);
        bool present = false;
        for (const bu_etc_block& s : m_optimized_cluster_selectors) if (raw_selector_bits(s) == flat) { present = true; break; }
        if (present) continue;
        const uint32_t new_index = (uint32_t)m_optimized_cluster_selectors.size();
        ensure_orig_encoded_host(); ensure_encoded_host(); ensure_endpoint_map(); ensure_selector_map_host();
        if (!m_orig_host_valid || !m_enc_host_valid || !m_endpoint_map_valid || !m_sel_host_valid) return fail("download blocks");
        bu_etc_block nb{}; store_be64(nb, flat);
        m_optimized_cluster_selectors.push_back(nb);
        m_selector_cluster_count = new_index + 1;
        m_selector_lists_valid = false;
        for (uint32_t b = 0; b < n; b++) {
            if (raw_selector_bits(m_orig_encoded_blocks[b]) != flat) continue;
            const bu_pixel_block* px = tile(b);
            if (!px) return fail("download tiles");
            const endpoint_params& e = m_endpoint_cluster_etc_params[m_block_endpoint_cluster[b]];
            const uint32_t cur_bits = raw_selector_bits(m_optimized_cluster_selectors[m_block_selector_cluster_index[b]]);
            if (block_error(*px, e, flat) >= block_error(*px, e, cur_bits)) continue;
            m_block_selector_cluster_index[b] = new_index;  // the lists (block order) follow from the map
            m_sel_dev_valid = false;
            total_relocated++;
            store_be64(m_encoded_blocks[b], (load_be64(m_encoded_blocks[b]) & ~0xFFFFFFFFull) | flat);
            m_enc_dev_valid = false;
        }
    }
    (void)total_relocated;
    return true;
}

// frontend.cpp:657-731: drop unused entries and merge entries with identical selector bits (first occurrence keeps its place)
bool etc1s_frontend::optimize_selector_codebook() {
    bool remap_failed = false;
    const uint32_t k = (uint32_t)m_optimized_cluster_selectors.size(), n = m_total_blocks;
    std::vector<uint8_t> used(k, 0);
    device_state& d = *m_dev;
    bool on_device = m_sel_dev_valid;
    if (on_device) {   // which entries own a block: one flag per entry, set by every block of the resident map (a sort of the blocks by cluster was paid for the sizes here until round 3)
        if (!(d.reserve(d.flags, k) && bu_hip_k_map_membership(d.ctx, nullptr, (const uint32_t*)d.sel_cluster.p, n, 1, k, (uint8_t*)d.flags.p) &&
              bu_hip_memcpy_d2h(d.ctx, used.data(), d.flags.p, k)))
            on_device = false;
    }
    if (!on_device) {
        ensure_selector_map_host();
        for (uint32_t b = 0; b < n; b++) used[m_block_selector_cluster_index[b]] = 1;
    }
    std::vector<int32_t> old_to_new(k, -1);
    std::vector<uint32_t> new_to_old;
    {
        // entries with identical bits collapse onto the first of them; sorted (bits, index) pairs give every entry its group's first index
        // (an insertion-sorted "seen" list is quadratic: 12 ms of host time at 15,000 entries)
        std::vector<std::pair<uint32_t, uint32_t>> by_bits;
        by_bits.reserve(k);
        for (uint32_t i = 0; i < k; i++) if (used[i]) by_bits.emplace_back(raw_selector_bits(m_optimized_cluster_selectors[i]), i);
        std::sort(by_bits.begin(), by_bits.end());
        std::vector<uint32_t> first_of(k, 0);
        for (size_t a = 0; a < by_bits.size();) {
            size_t b = a;
            while (b < by_bits.size() && by_bits[b].first == by_bits[a].first) { first_of[by_bits[b].second] = by_bits[a].second; b++; }
            a = b;
        }
        for (uint32_t i = 0; i < k; i++) {
            if (!used[i]) continue;
            if (first_of[i] == i) { old_to_new[i] = (int32_t)new_to_old.size(); new_to_old.push_back(i); }
            else old_to_new[i] = old_to_new[first_of[i]];   // the group's first entry has the smaller index: already numbered
        }
    }
    if (m_sel_host_valid)
        for (uint32_t b = 0; b < n; b++) m_block_selector_cluster_index[b] = (uint32_t)old_to_new[m_block_selector_cluster_index[b]];
    if (m_sel_dev_valid) {
        std::vector<uint32_t> remap(k);
        for (uint32_t i = 0; i < k; i++) remap[i] = (uint32_t)old_to_new[i];
        if (!d.upload(d.tmp_a, remap.data(), k) || !bu_hip_k_map_remap(d.ctx, (uint32_t*)d.sel_cluster.p, nullptr, n, (const uint32_t*)d.tmp_a.p, nullptr)) {
            fail("bu_hip_k_map_remap");
            // keep a consistent state on the host at least: a map downloaded now is still in the OLD numbering (a host copy that was
            // valid before has been renumbered above)
            if (!m_sel_host_valid) {
                ensure_selector_map_host();
                if (m_sel_host_valid)
                    for (uint32_t b = 0; b < n; b++) m_block_selector_cluster_index[b] = (uint32_t)old_to_new[m_block_selector_cluster_index[b]];
            }
            m_sel_dev_valid = false;
            remap_failed = true;
        }
    }
    std::vector<bu_etc_block> sels(new_to_old.size());
    for (size_t i = 0; i < new_to_old.size(); i++) sels[i] = m_optimized_cluster_selectors[new_to_old[i]];
    m_optimized_cluster_selectors.swap(sels);
    m_selector_cluster_count = (uint32_t)new_to_old.size();
    m_selector_lists_valid = false;
    for (auto& l : m_selector_clusters_within_each_parent_cluster)
        for (uint32_t& c : l) c = (uint32_t)old_to_new[c];
    return !remap_failed;
}

// frontend.cpp:2980-2992
void etc1s_frontend::finalize() {
    // "used" = owns at least one block; the sizes are kept current by every stage that changes the clustering
    if (m_endpoint_cluster_sizes.size() == m_endpoint_cluster_etc_params.size() && (m_ep_dev_valid || m_endpoint_map_valid)) {
        for (size_t c = 0; c < m_endpoint_cluster_sizes.size(); c++) if (m_endpoint_cluster_sizes[c]) m_endpoint_cluster_etc_params[c].color_used = true;
    } else {
        ensure_endpoint_map();
        for (uint32_t b = 0; b < m_total_blocks; b++) m_endpoint_cluster_etc_params[m_block_endpoint_cluster[b]].color_used = true;
    }
    // the frontend's results as its consumers read them (frontend.h:119-156): encoded blocks and both per-block indices on the host
    ensure_encoded_host(); ensure_endpoint_map(); ensure_selector_map_host();
}

} // namespace bu
