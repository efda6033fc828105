// etc1s_backend.cpp -- see etc1s_backend.h. Reference: encoder/basisu_backend.cpp (cited per function).
//
// How the walk differs from the reference's while producing the same bytes:
//  * one flat block array; the predictor walk as a wavefront over the rows; per slice two loops (endpoints; selectors, which build a block's distance table
//    where the history search needs one) that run as two threads, and one token stream in bit-stream order (run tokens are placeholders patched when the run ends) instead of symbol vectors that a second
//    walk over the blocks re-synchronises with; slices run concurrently; long symbol streams are coded in pieces and joined bit-wise;
//  * with a resident frontend behind it the per-block errors create_encoder_blocks starts from come from the device (k_backend_block_errors);
//  * the selector-history search looks a candidate's error up in a 4x16 table of the block's pixel-to-colour distances (64 distance
//    evaluations per block instead of up to 16 per candidate and 64 candidates), and pre-filters candidates with one SAD instruction;
//  * the error loops run 8 pixels per instruction where the CPU has AVX2 (block_metric.h), the palette's block colours are converted to
//    the metric's basis once per palette, not once per trial;
//  * the palette reordering keeps the symbol adjacency counts as sparse lists instead of a dense num_syms^2 matrix (1 GB at 16128).
#include "etc1s_backend.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <deque>

#include <atomic>
#include <cstdlib>
#include <numeric>
#include <mutex>
#include <thread>

#include "block_metric.h"
#include "entropy.h"
#include "etc1s_frontend.h"

namespace bu {
namespace {

enum : uint32_t {  // transcoder/basisu_transcoder_internal.h:256-267
    kEndpointPredSymbols = 4 * 4 * 4 * 4 + 1, kEndpointPredRepeatLast = kEndpointPredSymbols - 1, kEndpointPredMinRepeat = 3, kEndpointPredCountVlcBits = 4,
    kNumEndpointPreds = 3, kNoEndpointPred = 3,
    kSelectorHistorySize = 64, kSelectorRleThresh = 3, kSelectorRleCountBits = 6, kSelectorRleCountTotal = 1u << kSelectorRleCountBits
};
const int kColorDeltaThresh = 8, kSelDiffThreshold = 11;  // backend.cpp:719-720
const int kPredDx[3] = {-1, 0, -1}, kPredDy[3] = {0, -1, -1};  // g_endpoint_preds, backend.cpp:120-128

const int kInten[8][4] = {{-8, -2, 2, 8}, {-17, -5, 5, 17}, {-29, -9, 9, 29}, {-42, -13, 13, 42},
                          {-60, -18, 18, 60}, {-80, -24, 24, 80}, {-106, -33, 33, 106}, {-183, -47, 47, 183}};  // g_etc1_inten_tables, etc.cpp:304-308

inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
inline uint64_t load_be64(const bu_etc_block& b) { uint64_t v; std::memcpy(&v, b.m_bytes, 8); return __builtin_bswap64(v); }

struct timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// 16 selectors of an etc_block, pixel p = y*4+x in bits 2p (etc.h:232-236 for the bit positions, backend.cpp:104-117 for the order)
inline uint32_t packed_selectors(const bu_etc_block& blk) {
    // each plane byte spreads to the even bits of its eight pixels' fields: plane bit x*4+y belongs to pixel y*4+x
    static const struct spread_t {
        uint32_t v[2][256];
        spread_t() {
            for (uint32_t half = 0; half < 2; half++)
                for (uint32_t byte = 0; byte < 256; byte++) {
                    uint32_t o = 0;
                    for (uint32_t k = 0; k < 8; k++)
                        if (byte & (1u << k)) { const uint32_t bit = half * 8 + k, x = bit >> 2, y = bit & 3; o |= 1u << (2 * (y * 4 + x)); }
                    v[half][byte] = o;
                }
        }
    } S;
    const uint32_t lo32 = (uint32_t)load_be64(blk);
    const uint32_t lsb = S.v[0][lo32 & 255] | S.v[1][(lo32 >> 8) & 255], msb = S.v[0][(lo32 >> 16) & 255] | S.v[1][lo32 >> 24];
    // raw (lsb | msb << 1) -> selector {2, 3, 1, 0}: high bit = !msb, low bit = lsb ^ msb
    return ((msb ^ 0x55555555u) << 1) | (lsb ^ msb);
}

struct color5 { uint8_t r, g, b, inten; };
inline color5 header_of(const bu_etc_block& blk) {
    const uint64_t v = load_be64(blk);
    return color5{(uint8_t)((v >> 59) & 31), (uint8_t)((v >> 51) & 31), (uint8_t)((v >> 43) & 31), (uint8_t)((v >> 37) & 7)};
}

using metric::block_px;
using metric::pal_colors;
using metric::sel16;

inline void block_colors(bool perceptual, pal_colors& out, color5 c) {  // etc.h:584-602
    const int r = (c.r << 3) | (c.r >> 2), g = (c.g << 3) | (c.g >> 2), b = (c.b << 3) | (c.b >> 2);
    for (int k = 0; k < 4; k++) { const int d = kInten[c.inten][k]; metric::to_metric(perceptual, clamp255(r + d), clamp255(g + d), clamp255(b + d), out.x[k], out.y[k], out.z[k]); }
}
// basist::approx_move_to_front (transcoder_internal.h:863-929)
struct history_buffer {
    int v[kSelectorHistorySize];
    alignas(64) sel16 sel[kSelectorHistorySize];   // the pattern behind every entry, kept in step
    uint32_t rover;
    void reset(const sel16& of_zero) { std::memset(v, 0, sizeof(v)); for (sel16& s : sel) s = of_zero; rover = kSelectorHistorySize / 2; }
    void add(int x, const sel16& of_x) { v[rover] = x; sel[rover] = of_x; if (++rover == kSelectorHistorySize) rover = kSelectorHistorySize / 2; }
    void use(uint32_t i) { if (i) { std::swap(v[i / 2], v[i]); std::swap(sel[i / 2], sel[i]); } }
};

// The slices of a file (mip levels, array layers / cube faces, the alpha slice) share the codebooks and the Huffman models but no
// walking state: each is walked by its own host thread, largest first (BU_HOST_THREADS caps the count, default 8).
// BU_HOST_THREADS=1 means one thread altogether: the side jobs (selector codebook sort, slice CRCs) then run inline.
static bool host_single_threaded() {
    const char* e = std::getenv("BU_HOST_THREADS");
    return e && std::atoi(e) == 1;
}
template <class F> void for_each_slice(const std::vector<backend_slice_desc>& slices, F fn) {
    const size_t n = slices.size();
    unsigned want = 8;
    if (const char* e = std::getenv("BU_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) want = (unsigned)v; }
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t t = std::min<size_t>(std::min<unsigned>(want, hw ? hw : 1u), n);
    if (t <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    auto blocks = [&](size_t i) { return (uint64_t)slices[i].m_num_blocks_x * slices[i].m_num_blocks_y; };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return blocks(a) > blocks(b); });
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (size_t w = 0; w < t; w++)
        pool.emplace_back([&] { for (size_t k; (k = next.fetch_add(1)) < n;) fn(order[k]); });
    for (std::thread& th : pool) th.join();
}

// first index of the smallest Hamming distance to `to` among n packed patterns, stopping at the first distance <= 1 (backend.cpp:274-296)
#define BU_NEAREST_BODY(POPCOUNT) \
    uint32_t best = 100, best_j = 0; \
    for (uint32_t j = 0; j < n; j++) { \
        const uint32_t d = (uint32_t)POPCOUNT(p[j] ^ to); \
        if (d < best) { best = d; best_j = j; if (d <= 1) break; } \
    } \
    return best_j;
inline uint32_t nearest_pattern_plain(const uint32_t* p, uint32_t n, uint32_t to) { BU_NEAREST_BODY(__builtin_popcount) }
__attribute__((target("popcnt"))) inline uint32_t nearest_pattern_popcnt(const uint32_t* p, uint32_t n, uint32_t to) { BU_NEAREST_BODY(__builtin_popcount) }
#undef BU_NEAREST_BODY
// eight patterns per step: bit counts from a nibble table, and the scalar rule re-run on a group only when one of its members can improve the best
__attribute__((target("avx2"))) inline uint32_t nearest_pattern_avx2(const uint32_t* p, uint32_t n, uint32_t to) {
    const __m256i nib = _mm256_setr_epi8(0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4, 0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4);
    const __m256i low = _mm256_set1_epi8(15), key = _mm256_set1_epi32((int)to), ones8 = _mm256_set1_epi8(1), ones16 = _mm256_set1_epi16(1);
    uint32_t best = 100, best_j = 0, j = 0;
    for (; j + 8 <= n; j += 8) {
        const __m256i x = _mm256_xor_si256(_mm256_loadu_si256((const __m256i*)(p + j)), key);
        const __m256i cnt = _mm256_add_epi8(_mm256_shuffle_epi8(nib, _mm256_and_si256(x, low)), _mm256_shuffle_epi8(nib, _mm256_and_si256(_mm256_srli_epi16(x, 4), low)));
        const __m256i d = _mm256_madd_epi16(_mm256_maddubs_epi16(cnt, ones8), ones16);   // per 32-bit lane
        if (!_mm256_movemask_epi8(_mm256_cmpgt_epi32(_mm256_set1_epi32((int)best), d))) continue;
        alignas(32) uint32_t dl[8];
        _mm256_store_si256((__m256i*)dl, d);
        for (uint32_t k = 0; k < 8; k++)
            if (dl[k] < best) { best = dl[k]; best_j = j + k; if (best <= 1) return best_j; }
    }
    for (; j < n; j++) {
        const uint32_t dd = (uint32_t)__builtin_popcount(p[j] ^ to);
        if (dd < best) { best = dd; best_j = j; if (dd <= 1) break; }
    }
    return best_j;
}

// rows [0, n) of a slice over the host threads (stateless per-block work)
template <class F> void parallel_rows(uint32_t n, uint64_t work_per_row, F fn) {
    unsigned want = 8;
    if (const char* e = std::getenv("BU_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) want = (unsigned)v; }
    const unsigned hw = std::thread::hardware_concurrency();
    unsigned t = std::min(want, hw ? hw : 1u);
    if ((uint64_t)n * work_per_row < 32768) t = 1;
    t = std::min<unsigned>(t, n ? n : 1);
    if (t <= 1) { fn(0u, n); return; }
    std::vector<std::thread> pool;
    const uint32_t per = (n + t - 1) / t;
    for (unsigned i = 0; i < t; i++) {
        const uint32_t a = i * per, b = std::min(n, a + per);
        if (a < b) pool.emplace_back([=] { fn(a, b); });
    }
    for (std::thread& th : pool) th.join();
}

enum token_kind : uint8_t { T_NONE, T_PRED, T_PRED_REPEAT, T_ENDPOINT_DELTA, T_SELECTOR, T_SELECTOR_RLE };
struct token { uint32_t value; token_kind kind; };

}  // namespace

uint16_t crc16_ccitt(const void* data, size_t size, uint16_t crc) {
    // CRC-16/CCITT (polynomial 0x1021, MSB first) on the inverted register. t[0] is the usual byte table; t[k][i] is the register after byte i and
    // k zero bytes, so eight input bytes are folded in with eight independent look-ups instead of a chain of eight dependent ones (a slice's
    // ETC1 image is 8 MB at 4096^2: 13 ms a byte at a time, which the symbol coding next to it does not hide).
    static const struct table_t {
        uint16_t t[8][256];
        table_t() {
            for (uint32_t i = 0; i < 256; i++) {
                uint16_t c = (uint16_t)(i << 8);
                for (int k = 0; k < 8; k++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x1021 : (c << 1));
                t[0][i] = c;
            }
            for (int k = 1; k < 8; k++)
                for (uint32_t i = 0; i < 256; i++) t[k][i] = (uint16_t)((t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 8]);
        }
    } T;
    crc = (uint16_t)~crc;
    const uint8_t* p = static_cast<const uint8_t*>(data);
    size_t i = 0;
    for (; i + 8 <= size; i += 8)
        crc = (uint16_t)(T.t[7][p[i] ^ (crc >> 8)] ^ T.t[6][p[i + 1] ^ (crc & 255)] ^ T.t[5][p[i + 2]] ^ T.t[4][p[i + 3]] ^ T.t[3][p[i + 4]] ^ T.t[2][p[i + 5]] ^ T.t[1][p[i + 6]] ^
                         T.t[0][p[i + 7]]);
    for (; i < size; i++) crc = (uint16_t)((crc << 8) ^ T.t[0][(crc >> 8) ^ p[i]]);
    return (uint16_t)~crc;
}

std::vector<uint8_t> write_basis_file(const backend_output& out, uint32_t tex_type, uint32_t userdata0, uint32_t userdata1, bool y_flipped, uint32_t us_per_frame,
                                      const std::vector<basis_key_value>& key_values) {
    const uint32_t kHeaderSize = 77, kSliceDescSize = 23, kVersion = 0x13;  // sizeof(basis_file_header), sizeof(basis_slice_desc), BASIS_FILE_VERSION
    std::vector<uint8_t> f;
    auto put = [&f](uint64_t v, int bytes) { for (int i = 0; i < bytes; i++) f.push_back((uint8_t)(v >> (8 * i))); };
    auto poke = [&f](size_t at, uint64_t v, int bytes) { for (int i = 0; i < bytes; i++) f[at + i] = (uint8_t)(v >> (8 * i)); };
    // key-value block (basis_file.cpp:230-290): per pair the key's length (1-255, no terminator), the value's length (4 bytes), key, value
    std::vector<uint8_t> kv;
    if (!key_values.empty()) {
        std::vector<uint8_t> body;
        for (const basis_key_value& p : key_values) {
            if (p.key.empty() || p.key.size() > 255 || p.key.find('\0') != std::string::npos || p.value.size() > UINT32_MAX) return {};
            body.push_back((uint8_t)p.key.size());
            for (int i = 0; i < 4; i++) body.push_back((uint8_t)(p.value.size() >> (8 * i)));
            body.insert(body.end(), p.key.begin(), p.key.end());
            body.insert(body.end(), p.value.begin(), p.value.end());
        }
        const uint16_t crc = crc16_ccitt(body.data(), body.size(), 0);
        const uint32_t num = (uint32_t)key_values.size();
        kv = {0x4B, 0x56, (uint8_t)num, (uint8_t)(num >> 8), (uint8_t)(num >> 16), (uint8_t)(num >> 24), (uint8_t)crc, (uint8_t)(crc >> 8)};  // cBASISKVDataSig, count, crc
        kv.insert(kv.end(), body.begin(), body.end());
    }
    const size_t n_slices = out.m_slice_desc.size();
    if (out.m_slice_image_data.size() != n_slices || out.m_slice_image_crcs.size() != n_slices) return {};
    const bool etc1s = out.m_tex_format == 0;
    if (etc1s != out.m_etc1s || (!etc1s && (!out.m_endpoint_palette.empty() || !out.m_selector_palette.empty() || !out.m_slice_image_tables.empty()))) return {};
    // codebook and table offsets are 0 in files without them (basis_file.cpp:345-352)
    const uint64_t descs_ofs = kHeaderSize + kv.size(), after_descs = descs_ofs + (uint64_t)kSliceDescSize * n_slices, endpoint_ofs = etc1s ? after_descs : 0,
                   selector_ofs = etc1s ? endpoint_ofs + out.m_endpoint_palette.size() : 0, tables_ofs = etc1s ? selector_ofs + out.m_selector_palette.size() : 0,
                   first_slice_ofs = etc1s ? tables_ofs + out.m_slice_image_tables.size() : after_descs;
    uint64_t total = first_slice_ofs;
    for (const auto& d : out.m_slice_image_data) total += d.size();
    if (first_slice_ofs >= 0xFFFF0000ull || total >= 0xFFFF0000ull) return {};
    uint32_t total_images = 0, flags = etc1s ? 1u : 0u;  // cBASISHeaderFlagETC1S
    for (const backend_slice_desc& s : out.m_slice_desc) { total_images = std::max(total_images, s.m_source_file_index + 1); if (s.m_alpha) flags |= 4; }
    if (y_flipped) flags |= 2;
    if (out.m_srgb) flags |= 16;
    f.reserve((size_t)total);
    // basis_file_header
    put(0, 2); put(0, 2);                         // signature and version: written last
    put(kHeaderSize, 2); put(0, 2);               // header size, header CRC (last)
    put(total - kHeaderSize, 4); put(0, 2);       // data size, data CRC (last)
    put(n_slices, 3); put(total_images, 3);
    put(out.m_tex_format, 1);
    put(flags, 2); put(tex_type, 1); put(std::min<uint32_t>(us_per_frame, 0xFFFFFFu), 3);
    put(0, 4); put(userdata0, 4); put(userdata1, 4);
    put(out.m_num_endpoints, 2); put(endpoint_ofs, 4); put(out.m_endpoint_palette.size(), 3);
    put(out.m_num_selectors, 2); put(selector_ofs, 4); put(out.m_selector_palette.size(), 3);
    put(tables_ofs, 4); put(out.m_slice_image_tables.size(), 4);
    put(descs_ofs, 4);
    put(kv.empty() ? 0 : kHeaderSize, 4); put(kv.size(), 4);
    f.insert(f.end(), kv.begin(), kv.end());
    uint64_t slice_ofs = first_slice_ofs;
    for (size_t i = 0; i < n_slices; i++) {       // basis_slice_desc
        const backend_slice_desc& s = out.m_slice_desc[i];
        put(s.m_source_file_index, 3); put(s.m_mip_index, 1); put((s.m_alpha ? 1u : 0u) | (s.m_iframe ? 2u : 0u), 1);
        put(s.m_orig_width, 2); put(s.m_orig_height, 2); put(s.m_num_blocks_x, 2); put(s.m_num_blocks_y, 2);
        put(slice_ofs, 4); put(out.m_slice_image_data[i].size(), 4); put(out.m_slice_image_crcs[i], 2);
        slice_ofs += out.m_slice_image_data[i].size();
    }
    f.insert(f.end(), out.m_endpoint_palette.begin(), out.m_endpoint_palette.end());
    f.insert(f.end(), out.m_selector_palette.begin(), out.m_selector_palette.end());
    f.insert(f.end(), out.m_slice_image_tables.begin(), out.m_slice_image_tables.end());
    for (const auto& d : out.m_slice_image_data) f.insert(f.end(), d.begin(), d.end());
    // CRCs: of everything after the header, then of the header from the data-size field on (basis_file.cpp:200-209)
    poke(12, crc16_ccitt(f.data() + kHeaderSize, f.size() - kHeaderSize, 0), 2);
    poke(6, crc16_ccitt(f.data() + 8, kHeaderSize - 8, 0), 2);
    poke(0, ('B' << 8) | 's', 2);
    poke(2, kVersion, 2);
    return f;
}

std::vector<uint8_t> write_ktx2_file(const backend_output& out, uint32_t tex_type, bool has_alpha, const std::vector<basis_key_value>& key_values_in) {
    const bool etc1s = out.m_tex_format == 0;
    if (!etc1s && out.m_tex_format != 1) return {};
    const size_t n_slices = out.m_slice_desc.size();
    if (!n_slices || out.m_slice_image_data.size() != n_slices) return {};
    // dimensions of the texture from the slices (comp.cpp:4924-4945)
    uint32_t width = 0, height = 0, layers = 0, levels = 0, faces = 1;
    for (const backend_slice_desc& s : out.m_slice_desc) {
        if (!s.m_mip_index && !width) { width = s.m_orig_width; height = s.m_orig_height; }
        layers = std::max(layers, s.m_source_file_index + 1);
        if (!s.m_source_file_index) levels = std::max(levels, s.m_mip_index + 1);
    }
    if (tex_type == 2) { if (layers % 6) return {}; layers /= 6; faces = 6; }  // cBASISTexTypeCubemapArray
    for (const backend_slice_desc& s : out.m_slice_desc) if (s.m_mip_index >= levels) return {};
    // every level's bytes: its slices in slice order
    std::vector<std::vector<uint8_t>> level_bytes(levels);
    std::vector<size_t> offset_in_level(n_slices);
    for (size_t i = 0; i < n_slices; i++) {
        std::vector<uint8_t>& l = level_bytes[out.m_slice_desc[i].m_mip_index];
        offset_in_level[i] = l.size();
        l.insert(l.end(), out.m_slice_image_data[i].begin(), out.m_slice_image_data[i].end());
    }
    std::vector<uint8_t> f, sgd;
    auto put_to = [](std::vector<uint8_t>& v, uint64_t x, int bytes) { for (int i = 0; i < bytes; i++) v.push_back((uint8_t)(x >> (8 * i))); };
    if (etc1s) {  // BasisLZ global data (comp.cpp:5104-5151)
        put_to(sgd, out.m_num_endpoints, 2); put_to(sgd, out.m_num_selectors, 2);
        put_to(sgd, out.m_endpoint_palette.size(), 4); put_to(sgd, out.m_selector_palette.size(), 4); put_to(sgd, out.m_slice_image_tables.size(), 4); put_to(sgd, 0, 4);
        struct image_record { uint32_t flags = 0, rgb_ofs = 0, rgb_len = 0, alpha_ofs = 0, alpha_len = 0; };
        const bool video = tex_type == 3;  // cBASISTexTypeVideoFrames: frames that are not i-frames are flagged KTX2_IMAGE_IS_P_FRAME (comp.cpp:5144)
        std::vector<image_record> images((size_t)levels * layers * faces);
        for (size_t i = 0; i < n_slices; i++) {
            const backend_slice_desc& s = out.m_slice_desc[i];
            uint32_t layer = s.m_source_file_index, face = 0;
            if (tex_type == 2) { face = layer % 6; layer /= 6; }
            const size_t at = (size_t)s.m_mip_index * layers * faces + (size_t)layer * faces + face;
            if (at >= images.size()) return {};
            if (s.m_alpha) { images[at].alpha_len = (uint32_t)out.m_slice_image_data[i].size(); images[at].alpha_ofs = (uint32_t)offset_in_level[i]; }
            else { images[at].rgb_len = (uint32_t)out.m_slice_image_data[i].size(); images[at].rgb_ofs = (uint32_t)offset_in_level[i]; if (video && !s.m_iframe) images[at].flags = 2; }
        }
        for (const image_record& r : images) { put_to(sgd, r.flags, 4); put_to(sgd, r.rgb_ofs, 4); put_to(sgd, r.rgb_len, 4); put_to(sgd, r.alpha_ofs, 4); put_to(sgd, r.alpha_len, 4); }
        sgd.insert(sgd.end(), out.m_endpoint_palette.begin(), out.m_endpoint_palette.end());
        sgd.insert(sgd.end(), out.m_selector_palette.begin(), out.m_selector_palette.end());
        sgd.insert(sgd.end(), out.m_slice_image_tables.begin(), out.m_slice_image_tables.end());
    }
    // data format descriptor: one basic block, colour model ETC1S (163) / UASTC (166), BT.709 primaries, 4x4 texel block, one sample per
    // plane (ETC1S: the colour slice, plus the alpha slice as a second 8-byte plane; UASTC: one 16-byte plane, channel id RGB 0 / RGBA 3)
    std::vector<uint8_t> dfd;
    const uint32_t samples = (etc1s && has_alpha) ? 2 : 1, block_size = 24 + 16 * samples;
    put_to(dfd, 4 + block_size, 4);
    put_to(dfd, 0, 4);                                   // vendor id 0 (Khronos), descriptor type 0 (basic)
    put_to(dfd, 2, 2); put_to(dfd, block_size, 2);       // version 1.3, block size
    put_to(dfd, etc1s ? 163 : 166, 1); put_to(dfd, 1, 1); put_to(dfd, out.m_srgb ? 2 : 1, 1); put_to(dfd, 0, 1);  // model, primaries, transfer, flags
    put_to(dfd, 3, 1); put_to(dfd, 3, 1); put_to(dfd, 0, 2);                                                      // texel block 4x4x1x1, minus one
    put_to(dfd, etc1s ? 8 : 16, 1); put_to(dfd, samples == 2 ? 8 : 0, 1); put_to(dfd, 0, 6);                      // bytes per plane
    for (uint32_t k = 0; k < samples; k++) {
        put_to(dfd, k * 64, 2); put_to(dfd, etc1s ? 63 : 127, 1);                                                 // bit offset, bit length - 1
        put_to(dfd, etc1s ? (k ? 15 : 0) : (has_alpha ? 3 : 0), 1);                                               // channel: RGB / AAA; UASTC RGB / RGBA
        put_to(dfd, 0, 4); put_to(dfd, 0, 4); put_to(dfd, 0xFFFFFFFFu, 4);                                        // position, lower, upper
    }
    // key-values, sorted by key (strcmp order); without supercompression a dummy key pads the block so that the levels start 16-byte aligned
    std::vector<basis_key_value> kvs = key_values_in;
    const uint32_t kvd_ofs = 80 + 24 * levels + (uint32_t)dfd.size();
    std::vector<uint8_t> kvd;
    for (int pass = 0; pass < 2; pass++) {
        std::sort(kvs.begin(), kvs.end(), [](const basis_key_value& a, const basis_key_value& b) { return std::strcmp(a.key.c_str(), b.key.c_str()) < 0; });
        kvd.clear();
        for (const basis_key_value& p : kvs) {
            if (p.key.empty() || p.key.find('\0') != std::string::npos) return {};
            put_to(kvd, p.key.size() + 1 + p.value.size(), 4);
            kvd.insert(kvd.end(), p.key.begin(), p.key.end());
            kvd.push_back(0);
            kvd.insert(kvd.end(), p.value.begin(), p.value.end());
            while (kvd.size() & 3) kvd.push_back(0);
        }
        if (etc1s || pass) break;
        uint32_t need = (16 - ((kvd_ofs + (uint32_t)kvd.size()) & 15)) & 15;
        if (!need) break;
        if (need < 6) need += 16;
        kvs.push_back(basis_key_value{std::string(need - 6, (char)127), std::vector<uint8_t>{0}});  // 4 (length) + key + NUL + 1 value byte = need
    }
    // assemble
    f.assign(80 + 24 * (size_t)levels, 0);
    const size_t dfd_ofs = f.size();
    f.insert(f.end(), dfd.begin(), dfd.end());
    const size_t kvd_at = kvd.empty() ? 0 : f.size();
    f.insert(f.end(), kvd.begin(), kvd.end());
    size_t sgd_at = 0;
    if (!sgd.empty()) { while (f.size() & 7) f.push_back(0); sgd_at = f.size(); f.insert(f.end(), sgd.begin(), sgd.end()); }
    if (!etc1s) while (f.size() & 15) f.push_back(0);
    std::vector<uint64_t> level_ofs(levels);
    for (uint32_t l = levels; l-- > 0;) { level_ofs[l] = f.size(); f.insert(f.end(), level_bytes[l].begin(), level_bytes[l].end()); }
    auto poke = [&f](size_t at, uint64_t v, int bytes) { for (int i = 0; i < bytes; i++) f[at + i] = (uint8_t)(v >> (8 * i)); };
    static const uint8_t magic[12] = {0xAB, 0x4B, 0x54, 0x58, 0x20, 0x32, 0x30, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A};
    std::memcpy(f.data(), magic, 12);
    poke(12, 0, 4); poke(16, 1, 4);                       // VK_FORMAT_UNDEFINED, type size 1
    poke(20, width, 4); poke(24, height, 4); poke(28, 0, 4);
    poke(32, layers > 1 ? layers : 0, 4); poke(36, faces, 4); poke(40, levels, 4);
    poke(44, etc1s ? 1 : 0, 4);                           // supercompression scheme: BasisLZ / none
    poke(48, dfd_ofs, 4); poke(52, dfd.size(), 4); poke(56, kvd_at, 4); poke(60, kvd.size(), 4);
    poke(64, sgd_at, 8); poke(72, sgd.size(), 8);
    for (uint32_t l = 0; l < levels; l++) {
        poke(80 + 24 * (size_t)l, level_ofs[l], 8);
        poke(88 + 24 * (size_t)l, level_bytes[l].size(), 8);
        poke(96 + 24 * (size_t)l, etc1s ? 0 : level_bytes[l].size(), 8);  // uncompressed length: only where Zstandard could apply
    }
    return f;
}

backend_output uastc_backend_output(const std::vector<backend_slice_desc>& slices, const uint8_t* blocks16, size_t total_blocks, bool srgb) {
    backend_output o;
    o.m_tex_format = 1;  // cUASTC_LDR_4x4
    o.m_etc1s = false;
    o.m_srgb = srgb;
    o.m_slice_desc = slices;
    for (const backend_slice_desc& s : slices) {
        const size_t n = (size_t)s.m_num_blocks_x * s.m_num_blocks_y;
        if ((size_t)s.m_first_block_index + n > total_blocks) return backend_output();
        const uint8_t* p = blocks16 + (size_t)s.m_first_block_index * 16;
        o.m_slice_image_data.emplace_back(p, p + n * 16);
        o.m_slice_image_crcs.push_back(crc16_ccitt(p, n * 16, 0));
    }
    return o;
}

std::vector<uint32_t> reorder_palette_by_adjacency(const uint32_t* indices, uint32_t num_indices, uint32_t n) {
    std::vector<uint32_t> remap(n, 0);
    if (num_indices <= 1 || !n) return remap;  // enc.cpp:1796-1797
    // adjacency counts of unequal neighbours, one entry per unordered pair (enc.cpp:1832-1843): the pairs (min, max) are bucketed by min with one
    // counting pass; each bucket is then counted into a dense per-symbol array (n <= 16128 counters: first-level cache) with a bitmap of the
    // counters it touched, which gives the distinct partners back in ascending order without a sort. (A three-pass radix sort of min * n + max did the same with three scattered passes over all the keys.)
    std::vector<uint32_t> row(n + 1, 0);
    for (uint32_t i = 0; i + 1 < num_indices; i++) {
        const uint32_t a = indices[i], b = indices[i + 1];
        if (a != b) row[std::min(a, b) + 1]++;
    }
    for (uint32_t s = 0; s < n; s++) row[s + 1] += row[s];
    std::vector<uint32_t> partner(row[n]);
    {
        std::vector<uint32_t> at(row.begin(), row.end() - 1);
        for (uint32_t i = 0; i + 1 < num_indices; i++) {
            const uint32_t a = indices[i], b = indices[i + 1];
            if (a != b) partner[at[std::min(a, b)]++] = std::max(a, b);
        }
    }
    struct edge { uint32_t other, count; };
    struct pair_count { uint32_t a, b, count; };
    std::vector<uint32_t> degree(n + 1, 0);
    std::vector<pair_count> pairs;   // ascending in (a, b): row-major order of the reference's matrix
    std::vector<uint32_t> seen(n, 0);
    std::vector<uint64_t> touched((n + 63) / 64, 0);   // which counters are non-zero: read back in ascending order, no sort
    pairs.reserve(partner.size() / 4);
    for (uint32_t a = 0; a < n; a++) {
        if (row[a] == row[a + 1]) continue;
        uint32_t top = a;
        for (uint32_t k = row[a]; k < row[a + 1]; k++) { const uint32_t b = partner[k]; seen[b]++; touched[b >> 6] |= 1ull << (b & 63); top = std::max(top, b); }
        for (uint32_t w = a >> 6; w <= (top >> 6); w++) {   // partners are above a
            for (uint64_t bits = touched[w]; bits; bits &= bits - 1) {
                const uint32_t b = w * 64 + (uint32_t)__builtin_ctzll(bits);
                pairs.push_back(pair_count{a, b, seen[b]});
                degree[a]++; degree[b]++;
                seen[b] = 0;
            }
            touched[w] = 0;
        }
    }
    std::vector<uint32_t> first(n + 1, 0);
    for (uint32_t s = 0; s < n; s++) first[s + 1] = first[s] + degree[s];
    std::vector<edge> edges(first[n]);
    std::vector<uint32_t> fill(first.begin(), first.end() - 1);
    uint32_t max_count = 0, a0 = 0, b0 = 0;
    for (const pair_count& p : pairs) {
        edges[fill[p.a]++] = edge{p.b, p.count};
        edges[fill[p.b]++] = edge{p.a, p.count};
        if (p.count > max_count) { max_count = p.count; a0 = p.a; b0 = p.b; }  // first maximum in row-major order (enc.cpp:1847-1852)
    }
    std::deque<uint32_t> picked;
    std::vector<int32_t> coord(n, 0);     // position of a picked symbol on the line; the front's coordinate is `front`
    std::vector<uint8_t> done(n, 0);
    std::vector<uint32_t> to_picked(n, 0);
    int32_t front = 0;
    // the unplaced symbols by (adjacency to the placed ones, descending; number, ascending): a binary heap with one entry per symbol and the
    // symbol's position in it alongside, so that a grown count moves the entry up in place; symbols nobody is adjacent to yet are not in it.
    // (With lazy deletion instead, every one of the ~2 * pairs stale entries had to be popped through a heap of that size: 26 of this function's 43 ms.)
    std::vector<uint32_t> heap;            // symbols
    std::vector<uint32_t> where(n, UINT32_MAX);
    auto before = [&](uint32_t x, uint32_t y) { return to_picked[x] != to_picked[y] ? to_picked[x] > to_picked[y] : x < y; };
    auto sift_up = [&](uint32_t i) {
        const uint32_t s = heap[i];
        while (i) {
            const uint32_t up = (i - 1) / 2;
            if (!before(s, heap[up])) break;
            heap[i] = heap[up]; where[heap[i]] = i;
            i = up;
        }
        heap[i] = s; where[s] = i;
    };
    auto remove_top = [&]() {
        where[heap[0]] = UINT32_MAX;
        const uint32_t s = heap.back();
        heap.pop_back();
        const uint32_t size = (uint32_t)heap.size();
        if (!size) return;
        uint32_t i = 0;
        for (;;) {
            uint32_t c = 2 * i + 1;
            if (c >= size) break;
            if (c + 1 < size && before(heap[c + 1], heap[c])) c++;
            if (!before(heap[c], s)) break;
            heap[i] = heap[c]; where[heap[i]] = i;
            i = c;
        }
        heap[i] = s; where[s] = i;
    };
    auto account = [&](uint32_t moved) {
        for (uint32_t k = first[moved]; k < first[moved + 1]; k++) {
            const uint32_t o = edges[k].other;
            if (done[o]) continue;
            to_picked[o] += edges[k].count;
            if (where[o] == UINT32_MAX) { heap.push_back(o); sift_up((uint32_t)heap.size() - 1); }
            else sift_up(where[o]);
        }
    };
    picked.push_back(a0); picked.push_back(b0);
    done[a0] = done[b0] = 1;
    coord[a0] = 0; coord[b0] = 1;
    account(a0);
    if (b0 != a0) account(b0);
    uint32_t remaining = 0, lowest = 0;
    for (uint32_t s = 0; s < n; s++) remaining += !done[s];
    std::vector<std::pair<int32_t, uint32_t>> near;  // (position, count) of the picked neighbours of the symbol being placed
    while (remaining) {
        // the unplaced symbol most often adjacent to the placed ones; lowest number on ties, the lowest unplaced when all are 0 (enc.cpp:1868-1891)
        uint32_t best = UINT32_MAX;
        if (!heap.empty()) { best = heap[0]; remove_top(); }
        else { while (done[lowest]) lowest++; best = lowest; }
        // which end: the float sum of count * (distance to the far end - distance to the near end), in line order (enc.cpp:1893-1915)
        near.clear();
        for (uint32_t k = first[best]; k < first[best + 1]; k++)
            if (done[edges[k].other]) near.emplace_back(coord[edges[k].other] - front, edges[k].count);
        std::sort(near.begin(), near.end());
        float side = 0;
        const int size = (int)picked.size();
        for (const auto& pc : near) side += static_cast<float>((size + 1 - 2 * (pc.first + 1)) * (int)pc.second);
        if (side <= 0) { coord[best] = front + size; picked.push_back(best); }
        else { coord[best] = --front; picked.push_front(best); }
        done[best] = 1;
        remaining--;
        account(best);
    }
    for (uint32_t i = 0; i < n; i++) remap[picked[i]] = i;
    return remap;
}

// ------------------------------------------------------------------------------------------------------------------------------

void etc1s_backend::init(const backend_source& src, const backend_params& params, const std::vector<backend_slice_desc>& slices, reoptimize_fn reoptimize) {
    m_frontend = nullptr;
    m_src = src;
    m_reoptimize = std::move(reoptimize);
    m_params = params;
    m_slices = slices;
    m_output = backend_output();
    m_error.clear();
}

// backend.cpp:52-75. The frontend's state is flattened into the arrays of backend_source; the one call back into it re-flattens.
void etc1s_backend::init(etc1s_frontend* fe, const backend_params& params, const std::vector<backend_slice_desc>& slices) {
    init(backend_source(), params, slices, nullptr);
    m_frontend = fe;
    m_frontend_state_changed = false;
    auto flatten = [this](backend_source& src) {
        etc1s_frontend& f = *m_frontend;
        const auto& P = f.endpoint_cluster_params();
        m_fe_endpoints.resize(P.size() * 4);
        for (size_t i = 0; i < P.size(); i++) { m_fe_endpoints[i * 4] = P[i].r; m_fe_endpoints[i * 4 + 1] = P[i].g; m_fe_endpoints[i * 4 + 2] = P[i].b; m_fe_endpoints[i * 4 + 3] = P[i].inten; }
        src.total_blocks = f.get_total_output_blocks();
        src.perceptual = f.get_params().m_perceptual;
        src.source_blocks = f.source_blocks_host();
        src.output_blocks = f.get_output_blocks().data();
        src.block_endpoint_index = f.block_endpoint_clusters().data();
        src.block_selector_index = f.block_selector_cluster_index().data();
        src.total_endpoints = (uint32_t)P.size();
        src.endpoint_color5_inten = m_fe_endpoints.data();
        src.total_selectors = f.get_total_selector_clusters();
        src.selector_blocks = f.optimized_cluster_selectors().data();
    };
    flatten(m_src);
    m_reoptimize = [this, flatten](const std::vector<uint32_t>& new_block_endpoints, std::vector<int>& old_to_new, bool final_codebook,
                                   const std::vector<uint32_t>* block_selector_indices, backend_source& src) {
        m_frontend_state_changed = true;   // the resident blocks / clustering no longer describe what the backend walks
        if (!m_frontend->reoptimize_remapped_endpoints(new_block_endpoints, old_to_new, final_codebook, block_selector_indices)) return false;
        flatten(src);
        return true;
    };
}

void etc1s_backend::create_endpoint_palette() {  // backend.cpp:77-94
    m_output.m_num_endpoints = m_src.total_endpoints;
    m_endpoint_palette.resize(m_src.total_endpoints);
    for (uint32_t i = 0; i < m_src.total_endpoints; i++) {
        const uint8_t* e = m_src.endpoint_color5_inten + (size_t)i * 4;
        m_endpoint_palette[i] = endpoint_entry{e[0], e[1], e[2], e[3]};
    }
    m_palette_colors.resize(m_src.total_endpoints);
    for (uint32_t i = 0; i < m_src.total_endpoints; i++) {
        const endpoint_entry& e = m_endpoint_palette[i];
        block_colors(m_src.perceptual, m_palette_colors[i], color5{e.r, e.g, e.b, e.inten});
    }
}

void etc1s_backend::create_selector_palette() {  // backend.cpp:96-118
    m_output.m_num_selectors = m_src.total_selectors;
    m_selector_palette.resize(m_src.total_selectors);
    m_selector_bytes.resize(m_src.total_selectors);
    for (uint32_t i = 0; i < m_src.total_selectors; i++) {
        m_selector_palette[i] = packed_selectors(m_src.selector_blocks[i]);
        m_selector_bytes[i] = metric::unpack_selectors(m_selector_palette[i]);
    }
}

// backend.cpp:310-330: the slice holding the same mip level of the frame `delta` away
int etc1s_backend::find_video_frame(size_t slice, int delta) const {
    const backend_slice_desc& c = m_slices[slice];
    for (size_t s = 0; s < m_slices.size(); s++) {
        const backend_slice_desc& o = m_slices[s];
        if ((int)o.m_source_file_index == (int)c.m_source_file_index + delta && o.m_mip_index == c.m_mip_index && o.m_num_blocks_x == c.m_num_blocks_x &&
            o.m_num_blocks_y == c.m_num_blocks_y && o.m_alpha == c.m_alpha)
            return (int)s;
    }
    return -1;
}

// What both walks need of a block without knowing anything about the walk: the error of the frontend's block as it stands (the
// reference's cur_err in backend.cpp:507 and :841), its selectors, and -- speculatively -- its error under the endpoints its three causal
// neighbours had in the frontend (what backend.cpp:520-574 evaluates when those neighbours keep their endpoints, which most do). Stateless,
// so it is spread over the host threads; the walks fall back to computing a value themselves where the speculation does not apply.
void etc1s_backend::precompute_block_errors(bool with_neighbours) {
    const bool perceptual = m_src.perceptual;
    const metric::kernels K = metric::pick_kernels();
    m_own_err.assign(m_src.total_blocks, 0);
    m_own_sels.assign(m_src.total_blocks, 0);
    if (with_neighbours) m_neighbour_err.assign((size_t)m_src.total_blocks * 3, UINT64_MAX);
    for (const backend_slice_desc& s : m_slices) {
        const uint32_t nbx = s.m_num_blocks_x, nby = s.m_num_blocks_y, base = s.m_first_block_index;
        // With a resident frontend behind the backend (and its state untouched since compress()) the errors come from the MI355X (k_backend_block_errors: tiles, blocks
        // and clustering are still in HBM); only the selector unpacking stays here. The host loop below is the same arithmetic for the array-driven backend.
        if (m_frontend && with_neighbours && !m_frontend_state_changed && (uint64_t)base + (uint64_t)nbx * nby <= m_src.total_blocks && nbx && nby) {
            std::vector<uint32_t> own((size_t)nbx * nby), nb3((size_t)nbx * nby * 3);
            if (m_frontend->backend_block_errors(base, nbx, nby, true, own.data(), nb3.data())) {
                parallel_rows(nby, nbx, [&, nbx, base](uint32_t y0, uint32_t y1) {
                    for (uint32_t i = y0 * nbx; i < y1 * nbx; i++) {
                        const uint32_t b = base + i;
                        m_own_sels[b] = packed_selectors(m_src.output_blocks[b]);
                        m_own_err[b] = own[i];
                        for (uint32_t p = 0; p < 3; p++) m_neighbour_err[(size_t)b * 3 + p] = nb3[(size_t)i * 3 + p] == UINT32_MAX ? UINT64_MAX : (uint64_t)nb3[(size_t)i * 3 + p];
                    }
                });
                continue;
            }
        }
        parallel_rows(nby, nbx, [&, nbx, base](uint32_t y0, uint32_t y1) {
            for (uint32_t by = y0; by < y1; by++)
                for (uint32_t bx = 0; bx < nbx; bx++) {
                    const uint32_t b = base + bx + by * nbx;
                    block_px px;
                    K.load_pixels(perceptual, px, &m_src.source_blocks[b].m_pixels[0][0]);
                    const bu_etc_block& out = m_src.output_blocks[b];
                    const uint32_t packed = packed_selectors(out);
                    const sel16 sels = metric::unpack_selectors(packed);
                    pal_colors own;
                    block_colors(perceptual, own, header_of(out));
                    m_own_sels[b] = packed;
                    m_own_err[b] = K.block_error(perceptual, px, own, sels);
                    if (!with_neighbours || !m_own_err[b]) continue;
                    const uint32_t mine = m_src.block_endpoint_index[b];
                    uint32_t nb[kNumEndpointPreds];
                    bool any_equal = false;
                    for (uint32_t p = 0; p < kNumEndpointPreds; p++) {
                        const int x = (int)bx + kPredDx[p], y = (int)by + kPredDy[p];
                        nb[p] = (x >= 0 && y >= 0) ? m_src.block_endpoint_index[base + (uint32_t)x + (uint32_t)y * nbx] : UINT32_MAX;
                        any_equal = any_equal || nb[p] == mine;
                    }
                    if (any_equal) continue;   // predicted unless a neighbour is remapped away: left to the walk
                    for (uint32_t p = 0; p < kNumEndpointPreds; p++)
                        if (nb[p] != UINT32_MAX && nb[p] < m_src.total_endpoints) m_neighbour_err[(size_t)b * 3 + p] = K.block_error(perceptual, px, m_palette_colors[nb[p]], sels);
                }
        });
    }
}

// backend.cpp:406-617: every block is predicted from its left, upper or upper-left neighbour when that one uses the same endpoints;
// when none does, a neighbour's endpoints are adopted anyway if the block's error stays within the RDO threshold.
bool etc1s_backend::create_encoder_blocks() {
    const uint32_t total = m_src.total_blocks;
    m_blocks.assign(total, encoder_block{0, 0, 0, 0});
    const float thresh = m_params.m_endpoint_rdo_quality_thresh;
    const bool perceptual = m_src.perceptual;
    const metric::kernels K = metric::pick_kernels();
    struct joined_thread {  // the selector codebook's order only depends on the codebook: sorted on the side while the blocks are walked
        std::thread t;
        void join() { if (t.joinable()) t.join(); }
        ~joined_thread() { join(); }
    } selector_sort;
    if (host_single_threaded()) sort_selector_codebook();
    else selector_sort.t = std::thread([this] { sort_selector_codebook(); });
    std::vector<std::pair<uint64_t, uint64_t>> extents;
    for (const backend_slice_desc& s : m_slices) {
        if ((uint64_t)s.m_first_block_index + (uint64_t)s.m_num_blocks_x * s.m_num_blocks_y > total) return fail("slice exceeds the frontend's blocks");
        extents.emplace_back(s.m_first_block_index, (uint64_t)s.m_first_block_index + (uint64_t)s.m_num_blocks_x * s.m_num_blocks_y);
    }
    std::sort(extents.begin(), extents.end());
    for (size_t i = 1; i < extents.size(); i++)
        if (extents[i].first < extents[i - 1].second) return fail("slices overlap");  // the per-block state is kept once per block, and slices are walked concurrently
    { timer t__; if (thresh > 0.0f) precompute_block_errors(true); sub_time("~ceb/block_errors", t__.seconds()); }
    timer walk_timer;
    struct slice_result { std::vector<uint32_t> unpredicted; uint32_t remapped = 0; const char* error = nullptr; };
    std::vector<slice_result> results(m_slices.size());
    const bool video = m_params.m_video;
    const uint32_t spatial_preds = video ? 2u : 3u;   // in video files predictor 2 means "same as the previous frame", not the upper-left neighbour
    m_cr_target.assign(video ? total : 0, 0);
    // One row of a slice. A block looks at its left, upper and upper-left neighbours AS THE WALK LEFT THEM, so row y can follow row y - 1 one block behind:
    // large slices are walked as a wavefront, the rows dealt round robin to the host threads, each row waiting for the row above through a progress counter.
    auto walk_row = [&](size_t si, uint32_t by, int prev_frame, std::vector<uint32_t>& unpredicted, uint32_t& remapped, const char*& error,
                        const std::atomic<uint32_t>* above, std::atomic<uint32_t>* mine) {
        const backend_slice_desc& s = m_slices[si];
        const uint32_t nbx = s.m_num_blocks_x;
        uint32_t ready = above ? 0u : nbx;   // blocks of the row above known to be finished
        for (uint32_t bx = 0; bx < nbx; bx++) {
            if (bx + 1 > ready && ready < nbx) {   // needs (bx, by - 1) and (bx - 1, by - 1)
                for (uint32_t spins = 0; (ready = above->load(std::memory_order_acquire)) < std::min(nbx, bx + 1); spins++)
                    if (spins > 64) std::this_thread::yield();
            }
            const uint32_t b = s.m_first_block_index + bx + by * nbx;
            encoder_block& m = m_blocks[b];
            m.endpoint_index = m_src.block_endpoint_index[b];
            m.selector_index = m_src.block_selector_index[b];
            m.endpoint_predictor = kNoEndpointPred;
            if (m.endpoint_index >= m_src.total_endpoints || m.selector_index >= m_src.total_selectors) { error = "block index out of range"; if (mine) mine->store(nbx, std::memory_order_release); return; }
            uint32_t neighbour[kNumEndpointPreds];
            bool present[kNumEndpointPreds];
            uint32_t best_pred = UINT32_MAX;
            for (uint32_t p = 0; p < kNumEndpointPreds; p++) {
                const int px = (int)bx + kPredDx[p], py = (int)by + kPredDy[p];
                present[p] = p < spatial_preds && px >= 0 && py >= 0;  // dx, dy <= 0: the far edges cannot be crossed
                if (!present[p]) continue;
                neighbour[p] = m_blocks[s.m_first_block_index + (uint32_t)px + (uint32_t)py * nbx].endpoint_index;
                if (neighbour[p] == m.endpoint_index && best_pred == UINT32_MAX) best_pred = p;
            }
            if (prev_frame >= 0) {   // conditional replenishment wins over the spatial predictors (backend.cpp:457-471)
                const uint32_t pb = m_slices[prev_frame].m_first_block_index + bx + by * nbx;
                if (m_blocks[pb].endpoint_index == m.endpoint_index && m_blocks[pb].selector_index == m.selector_index) { best_pred = 2; m_cr_target[pb] = 1; }
            }
            if (best_pred != UINT32_MAX) {
                m.endpoint_predictor = (uint8_t)best_pred;
            } else if (thresh > 0.0f) {
                const uint64_t cur_err = m_own_err[b];
                if (cur_err) {
                    const uint64_t thresh_err = (uint64_t)(cur_err * std::max(1.0f, thresh));
                    uint64_t best_err = UINT64_MAX;
                    uint32_t best_index = 0;
                    block_px px;
                    sel16 sels;
                    bool have_px = false;
                    for (uint32_t p = 0; p < kNumEndpointPreds; p++) {
                        if (!present[p]) continue;
                        const uint32_t nb_block = s.m_first_block_index + (uint32_t)((int)bx + kPredDx[p]) + (uint32_t)((int)by + kPredDy[p]) * nbx;
                        uint64_t err = m_neighbour_err[(size_t)b * 3 + p];
                        if (err == UINT64_MAX || neighbour[p] != m_src.block_endpoint_index[nb_block]) {   // not speculated, or the neighbour was remapped
                            if (!have_px) { K.load_pixels(perceptual, px, &m_src.source_blocks[b].m_pixels[0][0]); sels = metric::unpack_selectors(m_own_sels[b]); have_px = true; }
                            err = K.block_error(perceptual, px, m_palette_colors[neighbour[p]], sels);
                        }
                        if (err <= thresh_err && err < best_err) { best_err = err; best_pred = p; best_index = neighbour[p]; }  // ascending p: ties keep the lower predictor
                    }
                    if (best_pred != UINT32_MAX) {
                        m.endpoint_index = best_index;
                        m.endpoint_predictor = (uint8_t)best_pred;
                        remapped++;
                    }
                }
            }
            if (m.endpoint_predictor == kNoEndpointPred) unpredicted.push_back(m.endpoint_index);
            if (mine && ((bx + 1) & 31u) == 0) mine->store(bx + 1, std::memory_order_release);
        }
        if (mine) mine->store(nbx, std::memory_order_release);
    };
    auto walk_slice = [&](size_t si) {
        const backend_slice_desc& s = m_slices[si];
        const uint32_t nbx = s.m_num_blocks_x, nby = s.m_num_blocks_y;
        const int prev_frame = video && !s.m_iframe ? find_video_frame(si, -1) : -1;
        std::vector<uint32_t>& all_endpoint_indices = results[si].unpredicted;
        all_endpoint_indices.reserve((size_t)nbx * nby);
        unsigned threads = 8;
        if (const char* e = std::getenv("BU_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) threads = (unsigned)v; }
        threads = std::min<unsigned>(std::min<unsigned>(threads, std::max(1u, std::thread::hardware_concurrency())), nby);
        if (video || threads <= 1 || (uint64_t)nbx * nby < 65536 || nbx < 64) {   // (video frames mark blocks of the previous frame: kept on one thread)
            for (uint32_t by = 0; by < nby; by++) walk_row(si, by, prev_frame, all_endpoint_indices, results[si].remapped, results[si].error, nullptr, nullptr);
            return;
        }
        std::vector<std::atomic<uint32_t>> done(nby);
        for (auto& d : done) d.store(0, std::memory_order_relaxed);
        struct row_out { std::vector<uint32_t> unpredicted; uint32_t remapped = 0; const char* error = nullptr; };
        std::vector<row_out> rows(nby);
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; t++)
            pool.emplace_back([&, t] {
                for (uint32_t by = t; by < nby; by += threads) walk_row(si, by, -1, rows[by].unpredicted, rows[by].remapped, rows[by].error, by ? &done[by - 1] : nullptr, &done[by]);
            });
        for (std::thread& th : pool) th.join();
        for (uint32_t by = 0; by < nby; by++) {
            if (rows[by].error && !results[si].error) results[si].error = rows[by].error;
            results[si].remapped += rows[by].remapped;
            all_endpoint_indices.insert(all_endpoint_indices.end(), rows[by].unpredicted.begin(), rows[by].unpredicted.end());
        }
    };
    if (video) { for (size_t si = 0; si < m_slices.size(); si++) walk_slice(si); }   // a frame reads the finished blocks of the frame before it
    else for_each_slice(m_slices, walk_slice);
    uint32_t remapped = 0;
    std::vector<uint32_t> all_endpoint_indices;  // the unpredicted endpoint indices in coding order: what the palette ordering works from
    all_endpoint_indices.reserve(total);
    for (const slice_result& r : results) {
        if (r.error) return fail(r.error);
        remapped += r.remapped;
        all_endpoint_indices.insert(all_endpoint_indices.end(), r.unpredicted.begin(), r.unpredicted.end());
    }
    sub_time("~ceb/walk", walk_timer.seconds());
    timer sort_timer;
    const bool ok = reoptimize_and_sort_endpoints_codebook(remapped, all_endpoint_indices);
    selector_sort.join();
    sub_time("~ceb/palette_orders", sort_timer.seconds());
    return ok;
}

// backend.cpp:130-244
bool etc1s_backend::reoptimize_and_sort_endpoints_codebook(uint32_t total_remapped, std::vector<uint32_t>& all_endpoint_indices) {
    if (total_remapped && m_params.m_compression_level > 1) {
        // the block -> endpoint assignment changed: let the frontend refit and compact its codebook (backend.cpp:145-191)
        if (!m_reoptimize) return fail("compression levels above 1 need the frontend behind the backend (reoptimize_remapped_endpoints)");
        std::vector<uint32_t> new_block_endpoints(m_src.total_blocks);
        for (uint32_t b = 0; b < m_src.total_blocks; b++) new_block_endpoints[b] = m_blocks[b].endpoint_index;
        std::vector<int> old_to_new;
        if (!m_reoptimize(new_block_endpoints, old_to_new, true, nullptr, m_src)) return fail("reoptimize_remapped_endpoints failed");
        create_endpoint_palette();
        precompute_block_errors(false);  // the frontend rewrote the colours of its output blocks
        for (encoder_block& m : m_blocks) m.endpoint_index = (uint32_t)old_to_new[m.endpoint_index];
        for (uint32_t& i : all_endpoint_indices) i = (uint32_t)old_to_new[i];
    }
    const uint32_t k = m_src.total_endpoints;
    m_endpoint_old_to_new = reorder_palette_by_adjacency(all_endpoint_indices.data(), (uint32_t)all_endpoint_indices.size(), k);
    // old -> new need not be onto: unused new slots point at the first used old entry (backend.cpp:199-243)
    std::vector<uint8_t> old_used(k, 0);
    uint32_t first_old = UINT32_MAX;
    for (const backend_slice_desc& s : m_slices)
        for (uint32_t i = 0, n = s.m_num_blocks_x * s.m_num_blocks_y; i < n; i++) {
            const uint32_t e = m_blocks[s.m_first_block_index + i].endpoint_index;
            old_used[e] = 1;
            first_old = std::min(first_old, e);
        }
    m_new_endpoint_was_used.assign(k, 0);
    m_endpoint_new_to_old.assign(k, first_old);
    for (uint32_t o = 0; o < k; o++)
        if (old_used[o]) { const uint32_t nw = m_endpoint_old_to_new[o]; m_new_endpoint_was_used[nw] = 1; m_endpoint_new_to_old[nw] = o; }
    return true;
}

// backend.cpp:246-309: a greedy nearest-neighbour chain over the selector patterns by Hamming distance of the packed selectors
void etc1s_backend::sort_selector_codebook() {
    const uint32_t k = m_src.total_selectors;
    m_selector_new_to_old.assign(k, 0);
    if (m_params.m_compression_level == 0) {
        for (uint32_t i = 0; i < k; i++) m_selector_new_to_old[i] = i;
    } else if (k) {
        // `remaining` in the reference's order (swap-with-last removal), with the patterns alongside so that the scan reads memory linearly
        std::vector<uint32_t> remaining(k - 1), bits(k - 1);
        for (uint32_t i = 1; i < k; i++) { remaining[i - 1] = i; bits[i - 1] = m_selector_palette[i]; }
        const bool hw_popcnt = __builtin_cpu_supports("popcnt"), avx2 = metric::pick_kernels().isa[0] == 'a';
        uint32_t prev = 0;
        for (uint32_t i = 1; i < k; i++) {
            const uint32_t left = (uint32_t)remaining.size();
            const uint32_t best_j = avx2 ? nearest_pattern_avx2(bits.data(), left, m_selector_palette[prev]) : hw_popcnt ? nearest_pattern_popcnt(bits.data(), left, m_selector_palette[prev]) : nearest_pattern_plain(bits.data(), left, m_selector_palette[prev]);
            prev = remaining[best_j];
            m_selector_new_to_old[i] = prev;
            remaining[best_j] = remaining.back(); bits[best_j] = bits.back();
            remaining.pop_back(); bits.pop_back();
        }
    }
    m_selector_old_to_new.assign(k, 0);
    for (uint32_t i = 0; i < k; i++) m_selector_old_to_new[m_selector_new_to_old[i]] = i;
}

// backend.cpp:619-681: CRC-16 of the slice as plain ETC1 blocks (differential, not flipped), what a transcoder to ETC1 would produce
void etc1s_backend::compute_slice_crcs() {
    static const uint8_t to_raw[4] = {3, 2, 0, 1};  // g_selector_index_to_etc1
    // the two selector bit planes of every palette pattern, once per pattern
    std::vector<uint32_t> planes(m_selector_palette.size());
    for (size_t k = 0; k < planes.size(); k++) {
        const uint32_t sels = m_selector_palette[k];
        uint32_t lo32 = 0;
        for (uint32_t y = 0; y < 4; y++)
            for (uint32_t x = 0; x < 4; x++) {
                const uint32_t raw = to_raw[(sels >> (2 * (y * 4 + x))) & 3], bit = x * 4 + y;
                lo32 |= (raw & 1u) << bit;
                lo32 |= (raw >> 1) << (16 + bit);
            }
        planes[k] = lo32;
    }
    m_output.m_slice_image_crcs.assign(m_slices.size(), 0);
    for_each_slice(m_slices, [&](size_t si) {
        const backend_slice_desc& s = m_slices[si];
        const uint32_t gx = (s.m_width + 3) / 4, gy = (s.m_height + 3) / 4;
        std::vector<uint8_t> img((size_t)gx * gy * 8, 0);
        for (uint32_t by = 0; by < s.m_num_blocks_y && by < gy; by++)
            for (uint32_t bx = 0; bx < s.m_num_blocks_x && bx < gx; bx++) {
                const encoder_block& m = m_blocks[s.m_first_block_index + bx + by * s.m_num_blocks_x];
                const endpoint_entry& e = m_endpoint_palette[m.endpoint_index];
                const uint64_t v = ((uint64_t)e.r << 59) | ((uint64_t)e.g << 51) | ((uint64_t)e.b << 43) | ((uint64_t)e.inten << 37) | ((uint64_t)e.inten << 34) | (1ull << 33) |
                                   planes[m.selector_index];
                const uint64_t be = __builtin_bswap64(v);
                std::memcpy(&img[((size_t)by * gx + bx) * 8], &be, 8);
            }
        m_output.m_slice_image_crcs[si] = crc16_ccitt(img.data(), img.size(), 0);
    });
}

// backend.cpp:687-1485
bool etc1s_backend::encode_image() {
    const uint32_t n_ep = m_src.total_endpoints, n_sel = m_src.total_selectors;
    const bool perceptual = m_src.perceptual;
    const uint32_t level = m_params.m_compression_level;
    const uint32_t kHistFirstSym = n_sel, kHistRleSym = n_sel + kSelectorHistorySize;
    struct slice_stats {
        std::vector<uint32_t> selector_hist, rle_hist, delta_hist, pred_hist;
        uint32_t endpoints_remapped = 0;
        slice_stats(uint32_t n_sel_syms, uint32_t n_ep_syms) : selector_hist(n_sel_syms, 0), rle_hist(kSelectorRleCountTotal, 0), delta_hist(n_ep_syms, 0), pred_hist(kEndpointPredSymbols, 0) {}
    };
    std::vector<slice_stats> stats(m_slices.size(), slice_stats(n_sel + kSelectorHistorySize + 1, n_ep));
    // a slice's symbols stay where the two walks put them -- per block: predictor, endpoint delta, selector, each possibly absent -- and are coded from there in that order
    struct slice_symbols { std::vector<token> pred, delta, sel; };
    std::vector<slice_symbols> slice_tokens(m_slices.size());
    std::vector<uint32_t> block_endpoint_indices(m_src.total_blocks, 0), block_selector_indices(m_src.total_blocks, 0);
    const metric::kernels K = metric::pick_kernels();
    const float selector_thresh = std::max(1.0f, m_params.m_selector_rdo_quality_thresh);
    const float endpoint_thresh = std::max(1.0f, m_params.m_endpoint_rdo_quality_thresh);
    const int max_search = level >= 2 ? 64 : 16;  // backend.cpp:852
    const bool video = m_params.m_video;
    // the endpoint search walks a window of consecutive NEW indices (wrapping once at either end, backend.cpp:865-869): the palette in
    // that order as byte arrays, extended by the largest half-window on both sides so that a window is one contiguous read
    const uint32_t kPad = 64;
    std::vector<pal_colors> sorted_colors(n_ep);
    std::vector<uint8_t> win_r(n_ep + 2 * kPad + 32, 0), win_g(win_r.size(), 0), win_b(win_r.size(), 0), win_i(win_r.size(), 0), win_u(win_r.size(), 0);
    for (uint32_t nw = 0; nw < n_ep; nw++) sorted_colors[nw] = m_palette_colors[m_endpoint_new_to_old[nw]];
    for (uint32_t k = 0; k < n_ep + 2 * kPad; k++) {
        const uint32_t nw = (uint32_t)(((int64_t)k - kPad) % (int64_t)n_ep + n_ep) % n_ep;
        const endpoint_entry& e = m_endpoint_palette[m_endpoint_new_to_old[nw]];
        win_r[k] = e.r; win_g[k] = e.g; win_b[k] = e.b; win_i[k] = e.inten; win_u[k] = m_new_endpoint_was_used[nw];
    }

    // A slice is walked by two loops over its blocks in raster order, each carrying its own state from block to block:
    //   1. endpoints : predictor symbols per 2x2 macroblock, the endpoint search relative to the previous block's index  (state: previous index, runs)
    //   2. selectors : the history-buffer search and the selector symbols (state: history buffer, runs); where the search is needed it first builds the
    //                  block's pixel-to-colour distance table against its FINAL endpoints, which is what it needs loop 1 for
    // Loop 2 only needs loop 1 to be ahead of it, so for slices worth it they run as two threads coupled by a progress counter (four bytes per block cross
    // between them). Until round 3 the tables were a third thread feeding a ring: every table (264 bytes) then crossed from one core's cache to another's,
    // and the selector loop -- 105 ms on its own -- took 160 ms waiting for them; only the blocks that miss the history need a table at all.
    const uint32_t kPipelineMinBlocks = 16384, kPublishEvery = 64;
    unsigned thread_cap = 8;
    if (const char* e = std::getenv("BU_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) thread_cap = (unsigned)v; }
    const bool may_pipeline = thread_cap >= 2 && std::thread::hardware_concurrency() >= 2;

    std::mutex loop_time_lock;
    double loop_times[4] = {0, 0, 0, 0};
    timer walks_timer;
    for_each_slice(m_slices, [&](size_t si) {
        const backend_slice_desc& s = m_slices[si];
        std::vector<uint32_t>&selector_hist = stats[si].selector_hist, &rle_hist = stats[si].rle_hist, &delta_hist = stats[si].delta_hist, &pred_hist = stats[si].pred_hist;
        uint32_t& endpoints_remapped = stats[si].endpoints_remapped;
        const uint32_t nbx = s.m_num_blocks_x, nby = s.m_num_blocks_y, base = s.m_first_block_index, n = nbx * nby;
        if (!n) return;
        // blocks whose endpoints a later block is predicted from must keep them (backend.cpp:740-766)
        std::vector<uint8_t> referenced(n, 0);
        for (uint32_t by = 0; by < nby; by++)
            for (uint32_t bx = 0; bx < nbx; bx++) {
                const uint32_t p = m_blocks[base + bx + by * nbx].endpoint_predictor;
                if (p < (video ? 2u : 3u)) referenced[(bx + kPredDx[p]) + (size_t)(by + kPredDy[p]) * nbx] = 1;
                if (video && m_cr_target[base + bx + by * nbx]) referenced[bx + (size_t)by * nbx] = 1;
            }
        // the symbols of a block, by producer; a run's symbol sits on the block that opens the run (placeholders patched when it closes)
        std::vector<token>& pred_tok = slice_tokens[si].pred;
        std::vector<token>& delta_tok = slice_tokens[si].delta;
        std::vector<token>& sel_tok = slice_tokens[si].sel;
        pred_tok.assign(n, token{0, T_NONE}); delta_tok.assign(n, token{0, T_NONE}); sel_tok.assign(n, token{0, T_NONE});
        std::vector<uint32_t> final_endpoint(n, 0);  // NEW palette index per block once loop 1 has passed it
        const bool pipelined = may_pipeline && n >= kPipelineMinBlocks;
        std::atomic<uint32_t> done1{0};
        auto wait_for = [](const std::atomic<uint32_t>& counter, uint32_t need) {  // until counter >= need
            uint32_t v;
            for (uint32_t spins = 0; (v = counter.load(std::memory_order_acquire)) < need; spins++)
                if (spins > 64) std::this_thread::yield();
            return v;
        };

        auto endpoints_loop = [&]() {
            std::vector<uint32_t> pred_run;   // the blocks (macroblock corners) holding the placeholders of the open run
            int prev_pred_sym = -1;
            uint32_t prev_endpoint = 0;
            auto close_pred_run = [&]() {
                if (pred_run.empty()) return;
                const uint32_t count = (uint32_t)pred_run.size();
                if (count > kEndpointPredMinRepeat) {
                    pred_hist[kEndpointPredRepeatLast]++;
                    pred_tok[pred_run[0]] = token{count, T_PRED_REPEAT};
                } else {
                    pred_hist[prev_pred_sym] += count;
                    for (uint32_t t : pred_run) pred_tok[t] = token{(uint32_t)prev_pred_sym, T_PRED};
                }
                pred_run.clear();
            };
            for (uint32_t by = 0; by < nby; by++)
                for (uint32_t bx = 0; bx < nbx; bx++) {
                    const uint32_t i = bx + by * nbx, b = base + i;
                    encoder_block& m = m_blocks[b];
                    // ---- one endpoint-predictor symbol per 2x2 macroblock, runs of equal symbols collapsed (backend.cpp:776-827)
                    if (!(bx & 1) && !(by & 1)) {
                        uint32_t sym = 0;
                        for (uint32_t y = 0; y < 2; y++)
                            for (uint32_t x = 0; x < 2; x++) {
                                uint32_t pred = kNoEndpointPred;
                                if (bx + x < nbx && by + y < nby) pred = m_blocks[base + (bx + x) + (by + y) * nbx].endpoint_predictor;
                                sym |= pred << (x * 2 + y * 4);
                            }
                        if ((int)sym == prev_pred_sym) {
                            pred_run.push_back(i);
                        } else {
                            close_pred_run();
                            pred_hist[sym]++;
                            pred_tok[i] = token{sym, T_PRED};
                            prev_pred_sym = (int)sym;
                        }
                    }
                    // ---- endpoint index, as a delta to the previous block's in the sorted palette (backend.cpp:829-1009)
                    int new_endpoint = (int)m_endpoint_old_to_new[m.endpoint_index];
                    if (m.endpoint_predictor == kNoEndpointPred) {
                        int delta = new_endpoint - (int)prev_endpoint;
                        if (m_params.m_endpoint_rdo_quality_thresh > 1.0f && std::abs(delta) > 1 && !referenced[i]) {
                            // a palette entry closer to the previous index that keeps the error within the threshold is cheaper to code
                            const color5 cur_c = header_of(m_src.output_blocks[b]);
                            const uint64_t cur_err = m_own_err[b];
                            if (cur_err) {
                                const endpoint_entry cur_e = m_endpoint_palette[m.endpoint_index];
                                const uint64_t thresh_err = (uint64_t)(cur_err * endpoint_thresh);
                                uint64_t best_err = UINT64_MAX;
                                int best_idx = 0;
                                block_px px;
                                K.load_pixels(perceptual, px, &m_src.source_blocks[b].m_pixels[0][0]);
                                const sel16 sels = metric::unpack_selectors(m_own_sels[b]);
                                const int dist = std::min(std::abs(delta) - 1, max_search);
                                int cand[128], n_cand = 0;   // 2 * max_search at most
                                const size_t w0 = (size_t)((int)prev_endpoint - dist + (int)kPad);
                                metric::window_mask wm = K.filter_window(&win_r[w0], &win_g[w0], &win_b[w0], &win_i[w0], &win_u[w0], 2 * dist, cur_e.r, cur_e.g, cur_e.b, cur_c.inten, level <= 1);
                                for (int half = 0; half < 2; half++)
                                    for (uint64_t bits = wm.w[half]; bits; bits &= bits - 1) {
                                        int trial = (int)prev_endpoint - dist + half * 64 + __builtin_ctzll(bits);
                                        if (trial < 0) trial += (int)n_ep; else if (trial >= (int)n_ep) trial -= (int)n_ep;
                                        if (trial != new_endpoint) cand[n_cand++] = trial;
                                    }
                                uint64_t cand_err[128];
                                K.block_errors(perceptual, px, sels, sorted_colors.data(), cand, n_cand, cand_err);
                                for (int k = 0; k < n_cand; k++)
                                    if (cand_err[k] < best_err && cand_err[k] <= thresh_err) { best_err = cand_err[k]; best_idx = cand[k]; }
                                if (best_err != UINT64_MAX) {
                                    m.endpoint_index = m_endpoint_new_to_old[best_idx];
                                    new_endpoint = best_idx;
                                    delta = new_endpoint - (int)prev_endpoint;
                                    endpoints_remapped++;
                                }
                            }
                        }
                        if (delta < 0) delta += (int)n_ep;
                        delta_hist[delta]++;
                        delta_tok[i] = token{(uint32_t)delta, T_ENDPOINT_DELTA};
                    }
                    block_endpoint_indices[b] = m_endpoint_new_to_old[new_endpoint];
                    final_endpoint[i] = (uint32_t)new_endpoint;
                    prev_endpoint = (uint32_t)new_endpoint;
                    if (((i + 1) % kPublishEvery) == 0) done1.store(i + 1, std::memory_order_release);
                }
            close_pred_run();
            done1.store(n, std::memory_order_release);
        };

        auto selectors_loop = [&]() {
            history_buffer history;
            history.reset(m_selector_bytes[m_selector_new_to_old[0]]);
            std::vector<uint32_t> selector_run;
            auto close_selector_run = [&]() {
                if (selector_run.empty()) return;
                const uint32_t count = (uint32_t)selector_run.size();
                if (count >= kSelectorRleThresh) {
                    sel_tok[selector_run[0]] = token{count, T_SELECTOR_RLE};
                    rle_hist[std::min<uint32_t>(count - kSelectorRleThresh, kSelectorRleCountTotal - 1)]++;
                    selector_hist[kHistRleSym]++;
                } else {
                    selector_hist[kHistFirstSym] += count;
                    for (uint32_t t : selector_run) sel_tok[t] = token{kHistFirstSym, T_SELECTOR};
                }
                selector_run.clear();
            };
            // The part of a block's search that does not depend on the history (distance table, own error, limit: metric::search_prep) is issued one block ahead
            // of the search itself, so that the two overlap in the core; `prepared` is the block prep[prepared & 1] holds.
            metric::search_prep prep[2];
            uint32_t prepared = UINT32_MAX;
            auto searched = [&](uint32_t i) { const uint32_t b = base + i; return !(video && (m_blocks[b].endpoint_predictor == 2 || m_cr_target[b])); };
            auto prepare = [&](uint32_t i) {
                const uint32_t b = base + i;
                K.search_prepare(perceptual, &m_src.source_blocks[b].m_pixels[0][0], m_palette_colors[m_endpoint_new_to_old[final_endpoint[i]]], m_selector_bytes[m_blocks[b].selector_index],
                                 selector_thresh, prep[i & 1]);
                prepared = i;
            };
            uint32_t ready = 0;
            for (uint32_t i = 0; i < n; i++) {
                if (std::min(i + 2, n) > ready) ready = wait_for(done1, std::min(i + 2, n));   // the next block's final endpoint is read one block early
                const uint32_t b = base + i;
                encoder_block& m = m_blocks[b];
                // ---- a recently used pattern (history buffer) that is good enough, else the block's own (backend.cpp:1011-1205)
                if (video && m.endpoint_predictor == 2) {   // repeats the previous frame's block: no selector symbol at all (backend.cpp:1011)
                    block_selector_indices[b] = m.selector_index;
                    continue;
                }
                const bool cr_target = video && m_cr_target[b];   // its selectors are what the next frame repeats: not to be traded (backend.cpp:1020, 1036)
                int new_selector = (int)m_selector_old_to_new[m.selector_index];
                int history_index = -1;
                if (cr_target) {
                    history_index = metric::find_first_64(history.v, new_selector);
                } else {
                    // the entry that holds the block's own pattern (levels 0 and 1 look there first); else the block's pixels against the four colours of its final
                    // endpoints: its own error, then the history patterns within thresh * that
                    if (prepared != i) prepare(i);
                    const metric::search_prep& mine = prep[i & 1];
                    if (i + 1 < n && searched(i + 1)) prepare(i + 1);
                    const metric::scan_result best = K.search_history(mine, m_selector_bytes[m.selector_index], history.sel, level <= 1 ? kSelDiffThreshold : 0, history.v, level <= 1 ? new_selector : -1);
                    if (best.index >= 0) { new_selector = history.v[best.index]; history_index = best.index; }
                }
                m.selector_index = m_selector_new_to_old[new_selector];
                if (history_index != 0) close_selector_run();
                if (history_index == 0) {
                    selector_run.push_back(i);
                } else if (history_index > 0) {
                    selector_hist[kHistFirstSym + history_index]++;
                    sel_tok[i] = token{kHistFirstSym + (uint32_t)history_index, T_SELECTOR};
                } else {
                    selector_hist[new_selector]++;
                    sel_tok[i] = token{(uint32_t)new_selector, T_SELECTOR};
                }
                m.selector_history_index = (int8_t)history_index;
                if (history_index < 0) history.add(new_selector, m_selector_bytes[m.selector_index]); else history.use((uint32_t)history_index);
                block_selector_indices[b] = m.selector_index;
            }
            close_selector_run();
        };

        // each loop's own duration (with the pipeline on they overlap: the slowest one is the walk's wall time)
        double d1 = 0, d3 = 0;
        auto timed = [](auto& fn, double& out) { timer t__; fn(); out = t__.seconds(); };
        if (pipelined) {
            std::thread t3([&] { timed(selectors_loop, d3); });
            timed(endpoints_loop, d1);
            t3.join();
        } else {
            timed(endpoints_loop, d1);
            timed(selectors_loop, d3);
        }
        { std::lock_guard<std::mutex> g(loop_time_lock); loop_times[0] += d1; loop_times[2] += d3; }
    });
    sub_time("~ei/endpoints_loop", loop_times[0]); sub_time("~ei/selectors_loop", loop_times[2]);
    sub_time("~ei/walks_wall", walks_timer.seconds());
    timer coding_timer;
    std::vector<uint32_t> selector_hist(n_sel + kSelectorHistorySize + 1, 0), rle_hist(kSelectorRleCountTotal, 0), delta_hist(n_ep, 0), pred_hist(kEndpointPredSymbols, 0);
    uint32_t endpoints_remapped = 0;
    for (const slice_stats& st : stats) {
        for (size_t i = 0; i < selector_hist.size(); i++) selector_hist[i] += st.selector_hist[i];
        for (size_t i = 0; i < rle_hist.size(); i++) rle_hist[i] += st.rle_hist[i];
        for (size_t i = 0; i < delta_hist.size(); i++) delta_hist[i] += st.delta_hist[i];
        for (size_t i = 0; i < pred_hist.size(); i++) pred_hist[i] += st.pred_hist[i];
        endpoints_remapped += st.endpoints_remapped;
    }

    if (endpoints_remapped && level > 1) {  // backend.cpp:1281-1287: refit the palette entries in place (no renumbering)
        if (!m_reoptimize) return fail("compression levels above 1 need the frontend behind the backend (reoptimize_remapped_endpoints)");
        std::vector<int> unused;
        if (!m_reoptimize(block_endpoint_indices, unused, false, &block_selector_indices, m_src)) return fail("reoptimize_remapped_endpoints failed");
        create_endpoint_palette();
    }
    // the slice CRCs only need the final indices: computed on the side while the symbols are coded
    struct joined_thread { std::thread t; ~joined_thread() { if (t.joinable()) t.join(); } } crc_thread;
    if (host_single_threaded()) compute_slice_crcs();
    else crc_thread.t = std::thread([this] { compute_slice_crcs(); });

    // ---- the four models, then the slices (backend.cpp:1298-1472)
    auto model = [](std::vector<uint32_t>& h, huffman_table& t) {
        bool any = false;
        for (uint32_t v : h) if (v) { any = true; break; }
        if (!any) h[0]++;
        return t.init(h, kHuffMaxCodeSize);
    };
    huffman_table pred_model, delta_model, selector_model, rle_model;
    if (!model(pred_hist, pred_model) || !model(delta_hist, delta_model) || !model(selector_hist, selector_model) || !model(rle_hist, rle_model))
        return fail("huffman model construction failed");
    bit_writer w;
    w.restart(4096);
    if (!w.put_table(pred_model) || !w.put_table(delta_model) || !w.put_table(selector_model) || !w.put_table(rle_model)) return fail("huffman table serialisation failed");
    w.put_bits(kSelectorHistorySize, 13);
    w.flush();
    m_output.m_slice_image_tables = w.bytes();
    m_output.m_slice_image_data.assign(m_slices.size(), std::vector<uint8_t>());
    auto code_token = [&](const token& t, bit_writer& w) {
        switch (t.kind) {
        case T_NONE: break;
        case T_PRED: w.put_code(t.value, pred_model); break;
        case T_PRED_REPEAT: w.put_code(kEndpointPredRepeatLast, pred_model); w.put_vlc(t.value - kEndpointPredMinRepeat, kEndpointPredCountVlcBits); break;
        case T_ENDPOINT_DELTA: w.put_code(t.value, delta_model); break;
        case T_SELECTOR: w.put_code(t.value, selector_model); break;
        case T_SELECTOR_RLE: {
            w.put_code(kHistRleSym, selector_model);
            const uint32_t run = t.value - kSelectorRleThresh;
            if (run >= kSelectorRleCountTotal - 1) { w.put_code(kSelectorRleCountTotal - 1, rle_model); w.put_vlc(run, 7); }
            else w.put_code(run, rle_model);
            break;
        }
        }
    };
    auto code_blocks = [&](const slice_symbols& s, size_t first, size_t end, bit_writer& w) {   // bit-stream order: predictor, endpoint delta, selector of every block in turn
        for (size_t i = first; i < end; i++) { code_token(s.pred[i], w); code_token(s.delta[i], w); code_token(s.sel[i], w); }
    };
    for_each_slice(m_slices, [&](size_t si) {
        const slice_symbols& syms = slice_tokens[si];
        const size_t blocks = syms.sel.size();
        // a long slice is coded in pieces on the host threads and the pieces' bits are joined in order (the symbols are independent of each other once the models exist)
        const unsigned pieces = host_single_threaded() ? 1u : (unsigned)std::min<size_t>(8, blocks / 16384);
        bit_writer w;
        if (pieces <= 1) {
            w.restart(blocks * 3 + 64);
            code_blocks(syms, 0, blocks, w);
        } else {
            std::vector<bit_writer> part(pieces);
            std::vector<std::thread> pool;
            const size_t per = (blocks + pieces - 1) / pieces;
            for (unsigned k = 0; k < pieces; k++)
                pool.emplace_back([&, k] {
                    const size_t a = std::min(blocks, k * per), b = std::min(blocks, a + per);
                    bit_writer local;   // on this thread's stack: the writers' accumulators are touched per symbol, and neighbours in one array would share cache lines
                    local.restart((b - a) * 3 + 64);
                    code_blocks(syms, a, b, local);
                    part[k] = std::move(local);
                });
            for (std::thread& th : pool) th.join();
            w.restart(blocks * 3 + 64);
            for (unsigned k = 0; k < pieces; k++) w.append(part[k]);
        }
        w.flush();
        m_output.m_slice_image_data[si] = w.bytes();
    });
    sub_time("~ei/huffman_coding", coding_timer.seconds());
    return true;
}

// backend.cpp:1487-1658: the palette in its final order, each entry as deltas to the previous one; three colour models picked by
// the previous component value, one for the intensity table
bool etc1s_backend::encode_endpoint_palette() {
    const uint32_t k = m_src.total_endpoints;
    std::vector<uint8_t> old_used(k, 0);
    uint32_t first_old = UINT32_MAX;
    for (const backend_slice_desc& s : m_slices)
        for (uint32_t i = 0, n = s.m_num_blocks_x * s.m_num_blocks_y; i < n; i++) {
            const uint32_t e = m_blocks[s.m_first_block_index + i].endpoint_index;
            old_used[e] = 1;
            first_old = std::min(first_old, e);
        }
    std::vector<uint32_t> new_to_old(k, first_old);
    for (uint32_t o = 0; o < k; o++) if (old_used[o]) new_to_old[m_endpoint_old_to_new[o]] = o;
    bool gray = true;
    for (const endpoint_entry& e : m_endpoint_palette) if (e.r != e.g || e.r != e.b) { gray = false; break; }
    const uint32_t comps = gray ? 1u : 3u;
    std::vector<uint32_t> h0(32, 0), h1(32, 0), h2(32, 0), hi(8, 0);
    auto walk = [&](auto&& on_inten, auto&& on_color) {
        int prev[3] = {16, 16, 16}, prev_inten = 0;
        for (uint32_t nw = 0; nw < k; nw++) {
            const endpoint_entry& e = m_endpoint_palette[new_to_old[nw]];
            on_inten((uint32_t)(((int)e.inten - prev_inten) & 7));
            prev_inten = e.inten;
            const int c[3] = {e.r, e.g, e.b};
            for (uint32_t i = 0; i < comps; i++) {
                on_color(prev[i] <= 9 ? 0 : (prev[i] <= 21 ? 1 : 2), (uint32_t)((c[i] - prev[i]) & 31));  // COLOR5_PAL0/1_PREV_HI
                prev[i] = c[i];
            }
        }
    };
    walk([&](uint32_t d) { hi[d]++; }, [&](int which, uint32_t d) { (which == 0 ? h0 : which == 1 ? h1 : h2)[d]++; });
    auto nonempty = [](std::vector<uint32_t>& h) { for (uint32_t v : h) if (v) return; h[0]++; };
    nonempty(h0); nonempty(h1); nonempty(h2);
    huffman_table m0, m1, m2, mi;
    if (!m0.init(h0) || !m1.init(h1) || !m2.init(h2) || !mi.init(hi)) return fail("endpoint palette model construction failed");
    bit_writer w;
    w.restart(8192);
    if (!w.put_table(m0) || !w.put_table(m1) || !w.put_table(m2) || !w.put_table(mi)) return fail("huffman table serialisation failed");
    w.put_bits(gray ? 1 : 0, 1);
    walk([&](uint32_t d) { w.put_code(d, mi); }, [&](int which, uint32_t d) { w.put_code(d, which == 0 ? m0 : which == 1 ? m1 : m2); });
    w.flush();
    m_output.m_endpoint_palette = w.bytes();
    return true;
}

// backend.cpp:1660-1745: each pattern as four bytes XORed with the previous pattern's, Huffman coded; raw bytes when that is not smaller
bool etc1s_backend::encode_selector_palette() {
    const uint32_t k = m_src.total_selectors;
    std::vector<uint32_t> h(256, 0);
    for (uint32_t q = 1; q < k; q++) {
        const uint32_t x = m_selector_palette[m_selector_new_to_old[q]] ^ m_selector_palette[m_selector_new_to_old[q - 1]];
        for (int j = 0; j < 4; j++) h[(x >> (8 * j)) & 255]++;
    }
    if (k < 2) h[0]++;
    huffman_table model;
    if (!model.init(h)) return fail("selector palette model construction failed");
    bit_writer w;
    w.restart((size_t)k * 4 + 64);
    w.put_bits(0, 1); w.put_bits(0, 1); w.put_bits(0, 1);  // global codebook, hybrid codebooks, raw bytes
    if (!w.put_table(model)) return fail("huffman table serialisation failed");
    for (uint32_t q = 0; q < k; q++) {
        const uint32_t cur = m_selector_palette[m_selector_new_to_old[q]];
        if (!q) { for (int j = 0; j < 4; j++) w.put_bits((cur >> (8 * j)) & 255, 8); continue; }
        const uint32_t x = cur ^ m_selector_palette[m_selector_new_to_old[q - 1]];
        for (int j = 0; j < 4; j++) w.put_code((x >> (8 * j)) & 255, model);
    }
    w.flush();
    if (w.bytes().size() >= (size_t)k * 4) {
        w.restart((size_t)k * 4 + 8);
        w.put_bits(0, 1); w.put_bits(0, 1); w.put_bits(1, 1);
        for (uint32_t q = 0; q < k; q++) {
            const uint32_t cur = m_selector_palette[m_selector_new_to_old[q]];
            for (int j = 0; j < 4; j++) w.put_bits((cur >> (8 * j)) & 255, 8);
        }
        w.flush();
    }
    m_output.m_selector_palette = w.bytes();
    return true;
}

uint32_t etc1s_backend::encode() {  // backend.cpp:1747-1776
    m_stage_times.clear();
    m_error.clear();
    if (m_params.m_used_global_codebooks) { fail("global codebooks are not supported"); return 0; }
    if (!m_src.total_blocks || !m_src.total_endpoints || !m_src.total_selectors || !m_src.source_blocks || !m_src.output_blocks ||
        !m_src.block_endpoint_index || !m_src.block_selector_index || !m_src.endpoint_color5_inten || !m_src.selector_blocks) { fail("incomplete backend source"); return 0; }
    m_output = backend_output();
    m_output.m_slice_desc = m_slices;
    m_output.m_tex_format = 0;
    m_output.m_etc1s = true;
    m_output.m_uses_global_codebooks = false;
    m_output.m_srgb = m_src.perceptual;
#define BU_BSTAGE(name, call) do { timer t__; if (!(call)) return 0; m_stage_times.push_back(stage_time{name, t__.seconds()}); } while (0)
    BU_BSTAGE("create_palettes", (create_endpoint_palette(), create_selector_palette(), true));
    BU_BSTAGE("create_encoder_blocks", create_encoder_blocks());
    BU_BSTAGE("encode_image", encode_image());
    BU_BSTAGE("encode_endpoint_palette", encode_endpoint_palette());
    BU_BSTAGE("encode_selector_palette", encode_selector_palette());
#undef BU_BSTAGE
    return m_output.get_output_size_estimate();
}

}  // namespace bu
