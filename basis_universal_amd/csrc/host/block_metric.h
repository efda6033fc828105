// block_metric.h -- the backend's inner loops: error of a 4x4 block against four ETC1S block colours under given selectors, in the
// reference's integer colour metric (color_distance, encoder/basisu_enc.h:1141-1195; linear: sum of squared differences).
//
// The metric is separable once a colour is moved to (l, cr, cb) = (14r+45g+5b, 64r-l, 64b-l): a distance is three squares. Every
// value below fits 32 bits (|dl| <= 16320, |dcr|,|dcb| <= 32640; one perceptual distance < 41e6, sixteen of them < 2^32), so the same
// arithmetic runs in 32-bit SIMD lanes. There are three implementations of each loop -- plain C++, AVX2, AVX-512 -- and a fourth of the
// history scan for CPUs with AVX-512 VBMI, picked once per encode from what the CPU has (BU_BACKEND_ISA=plain|avx2|avx512 caps the
// choice); they are integer-exact and therefore interchangeable, which tests/test_backend_host.py checks by running all of them.
#pragma once
#include <immintrin.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace bu {
namespace metric {

struct alignas(64) block_px { int32_t x[16], y[16], z[16]; };     // the 16 source pixels, pixel p = y*4+x
struct alignas(16) pal_colors { int32_t x[4], y[4], z[4]; };      // the four block colours of one (colour5, intensity table)
struct alignas(64) dist_table { uint32_t d[4][16]; };             // d[k][p]: pixel p against block colour k
struct alignas(16) sel16 { uint8_t s[16]; };                      // 16 selectors, one per byte
// What the history search of one block needs that does not depend on the history: filled by search_prepare (for the NEXT block, while the
// current one is being searched), read by search_history.
struct alignas(64) search_prep {
    dist_table t;            // the block's pixels against the four colours of its final endpoints
    uint8_t bound[64];       // VBMI: min(255, d >> shift), entry [k * 16 + p]: the lower-bound filter's table
    uint8_t exact[4][64];    // VBMI: the four byte planes of d, same layout: exact sums by table look-up
    uint64_t limit;          // ceilf(own error * thresh)
    uint32_t shift;
};

inline void to_metric(bool perceptual, int r, int g, int b, int32_t& x, int32_t& y, int32_t& z) {
    if (perceptual) { const int l = r * 14 + g * 45 + b * 5; x = l; y = r * 64 - l; z = b * 64 - l; }
    else { x = r; y = g; z = b; }
}
inline uint32_t dist1(bool perceptual, int dx, int dy, int dz) {
    if (perceptual) return ((uint32_t)(dx * dx) >> 5) + ((((uint32_t)(dy * dy) >> 5) * 26u) >> 7) + ((((uint32_t)(dz * dz) >> 5) * 3u) >> 7);
    return (uint32_t)(dx * dx + dy * dy + dz * dz);
}
inline sel16 unpack_selectors(uint32_t packed) {  // four selectors per byte -> one per byte
    static const struct lut_t { uint32_t v[256]; lut_t() { for (uint32_t b = 0; b < 256; b++) v[b] = (b & 3) | (((b >> 2) & 3) << 8) | (((b >> 4) & 3) << 16) | ((b >> 6) << 24); } } lut;
    sel16 o;
    const uint32_t w[4] = {lut.v[packed & 255], lut.v[(packed >> 8) & 255], lut.v[(packed >> 16) & 255], lut.v[packed >> 24]};
    std::memcpy(o.s, w, 16);
    return o;
}
inline void load_pixels_plain(bool perceptual, block_px& out, const uint8_t* rgba16) {
    for (int p = 0; p < 16; p++) to_metric(perceptual, rgba16[p * 4], rgba16[p * 4 + 1], rgba16[p * 4 + 2], out.x[p], out.y[p], out.z[p]);
}

// ---- plain C++
inline uint64_t block_error_plain(bool perceptual, const block_px& px, const pal_colors& c, const sel16& sel) {
    uint64_t e = 0;
    for (int p = 0; p < 16; p++) { const int k = sel.s[p]; e += dist1(perceptual, px.x[p] - c.x[k], px.y[p] - c.y[k], px.z[p] - c.z[k]); }
    return e;
}
inline void build_table_plain(bool perceptual, const block_px& px, const pal_colors& c, dist_table& t) {
    for (int k = 0; k < 4; k++)
        for (int p = 0; p < 16; p++) t.d[k][p] = dist1(perceptual, px.x[p] - c.x[k], px.y[p] - c.y[k], px.z[p] - c.z[k]);
}
// Sum over the pixels of d[selector][pixel]. `bound`: the caller only cares about results <= bound; once the first eight pixels alone
// exceed it the partial sum is returned (any value > bound does).
inline uint64_t table_error_plain(const dist_table& t, const sel16& sel, uint64_t bound) {
    uint64_t e = 0;
    for (int p = 0; p < 8; p++) e += t.d[sel.s[p]][p];
    if (e > bound) return e;
    for (int p = 8; p < 16; p++) e += t.d[sel.s[p]][p];
    return e;
}

// The two searches of the backend walk, batched so that each runs inside one ISA-specific function:
//   scan_history : min over the 64 history patterns of the table error, first index on ties, only patterns within `limit` count; with
//                  `sad_limit` > 0 patterns whose selectors differ from `cur` by that much or more (sum of absolute differences) are skipped
//   block_errors : the block's error under each of n candidate colour sets
struct scan_result { uint64_t err; int index; };  // index -1: nothing within the limit
inline int selector_sad(const sel16& a, const sel16& b);
inline scan_result scan_history_plain(const dist_table& t, const sel16& cur, const sel16* hist, int sad_limit, uint64_t limit) {
    scan_result r{UINT64_MAX, -1};
    for (int j = 0; j < 64; j++) {
        if (sad_limit > 0 && selector_sad(cur, hist[j]) >= sad_limit) continue;
        const uint64_t e = table_error_plain(t, hist[j], r.err < limit ? r.err : limit);
        if (e < r.err && e <= limit) { r.err = e; r.index = j; }
    }
    return r;
}
inline void block_errors_plain(bool perceptual, const block_px& px, const sel16& sel, const pal_colors* colors, const int* which, int n, uint64_t* out) {
    for (int i = 0; i < n; i++) out[i] = block_error_plain(perceptual, px, colors[which[i]], sel);
}

// The endpoint search's pre-filter over a window of `count` (<= 128) consecutive palette entries given as byte arrays (colour5, intensity
// table, used flag): bit i of the result says entry i is worth an error evaluation. With `strict` (compression levels 0 and 1) an entry
// must not have a higher intensity table than the block's and must be within 8 in summed colour5 distance (backend.cpp:880-892).
struct window_mask { uint64_t w[2]; };
inline window_mask filter_window_plain(const uint8_t* r, const uint8_t* g, const uint8_t* b, const uint8_t* inten, const uint8_t* used, int count,
                                       int cur_r, int cur_g, int cur_b, int cur_inten, bool strict) {
    window_mask m{{0, 0}};
    for (int i = 0; i < count; i++) {
        if (!used[i]) continue;
        if (strict && (inten[i] > cur_inten || std::abs(cur_r - r[i]) + std::abs(cur_g - g[i]) + std::abs(cur_b - b[i]) > 8)) continue;
        m.w[i >> 6] |= 1ull << (i & 63);
    }
    return m;
}

// ---- AVX2
#define BU_AVX2 __attribute__((target("avx2")))
BU_AVX2 inline window_mask filter_window_avx2(const uint8_t* r, const uint8_t* g, const uint8_t* b, const uint8_t* inten, const uint8_t* used, int count,
                                             int cur_r, int cur_g, int cur_b, int cur_inten, bool strict) {   // reads whole 32-byte chunks: arrays are padded
    window_mask m{{0, 0}};
    const __m256i cr = _mm256_set1_epi8((char)cur_r), cg = _mm256_set1_epi8((char)cur_g), cb = _mm256_set1_epi8((char)cur_b), ci = _mm256_set1_epi8((char)cur_inten);
    const __m256i eight = _mm256_set1_epi8(8), zero = _mm256_setzero_si256();
    for (int c = 0; c * 32 < count; c++) {
        __m256i ok = _mm256_cmpeq_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(used + 32 * c)), zero), zero);  // used != 0
        if (strict) {
            const __m256i R = _mm256_loadu_si256((const __m256i*)(r + 32 * c)), G = _mm256_loadu_si256((const __m256i*)(g + 32 * c)), B = _mm256_loadu_si256((const __m256i*)(b + 32 * c));
            const __m256i dr = _mm256_or_si256(_mm256_subs_epu8(R, cr), _mm256_subs_epu8(cr, R)), dg = _mm256_or_si256(_mm256_subs_epu8(G, cg), _mm256_subs_epu8(cg, G)),
                          db = _mm256_or_si256(_mm256_subs_epu8(B, cb), _mm256_subs_epu8(cb, B));
            const __m256i sum = _mm256_adds_epu8(_mm256_adds_epu8(dr, dg), db);
            ok = _mm256_and_si256(ok, _mm256_cmpeq_epi8(_mm256_max_epu8(sum, eight), eight));                                              // sum <= 8
            ok = _mm256_and_si256(ok, _mm256_cmpeq_epi8(_mm256_max_epu8(_mm256_loadu_si256((const __m256i*)(inten + 32 * c)), ci), ci));  // inten <= cur
        }
        uint64_t bits = (uint32_t)_mm256_movemask_epi8(ok);
        const int left = count - 32 * c;
        if (left < 32) bits &= (1ull << left) - 1;
        m.w[c >> 1] |= bits << (32 * (c & 1));
    }
    return m;
}
BU_AVX2 inline __m256i dist8(bool perceptual, __m256i dx, __m256i dy, __m256i dz) {
    const __m256i xx = _mm256_mullo_epi32(dx, dx), yy = _mm256_mullo_epi32(dy, dy), zz = _mm256_mullo_epi32(dz, dz);
    if (!perceptual) return _mm256_add_epi32(_mm256_add_epi32(xx, yy), zz);
    const __m256i l = _mm256_srli_epi32(xx, 5);
    const __m256i cr = _mm256_srli_epi32(_mm256_mullo_epi32(_mm256_srli_epi32(yy, 5), _mm256_set1_epi32(26)), 7);
    const __m256i cb = _mm256_srli_epi32(_mm256_mullo_epi32(_mm256_srli_epi32(zz, 5), _mm256_set1_epi32(3)), 7);
    return _mm256_add_epi32(_mm256_add_epi32(l, cr), cb);
}
BU_AVX2 inline uint32_t hsum8(__m256i v) {
    __m128i s = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0x4E));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0xB1));
    return (uint32_t)_mm_cvtsi128_si32(s);
}
BU_AVX2 inline uint64_t block_error_avx2(bool perceptual, const block_px& px, const pal_colors& c, const sel16& sel) {
    const __m256i cx = _mm256_castsi128_si256(_mm_load_si128((const __m128i*)c.x)), cy = _mm256_castsi128_si256(_mm_load_si128((const __m128i*)c.y)),
                  cz = _mm256_castsi128_si256(_mm_load_si128((const __m128i*)c.z));
    const __m128i s = _mm_load_si128((const __m128i*)sel.s);
    __m256i acc = _mm256_setzero_si256();
    for (int h = 0; h < 2; h++) {
        const __m256i idx = _mm256_cvtepu8_epi32(h ? _mm_srli_si128(s, 8) : s);
        const __m256i dx = _mm256_sub_epi32(_mm256_load_si256((const __m256i*)(px.x + 8 * h)), _mm256_permutevar8x32_epi32(cx, idx));
        const __m256i dy = _mm256_sub_epi32(_mm256_load_si256((const __m256i*)(px.y + 8 * h)), _mm256_permutevar8x32_epi32(cy, idx));
        const __m256i dz = _mm256_sub_epi32(_mm256_load_si256((const __m256i*)(px.z + 8 * h)), _mm256_permutevar8x32_epi32(cz, idx));
        acc = _mm256_add_epi32(acc, dist8(perceptual, dx, dy, dz));
    }
    return hsum8(acc);
}
BU_AVX2 inline void build_table_avx2(bool perceptual, const block_px& px, const pal_colors& c, dist_table& t) {
    for (int k = 0; k < 4; k++) {
        const __m256i cx = _mm256_set1_epi32(c.x[k]), cy = _mm256_set1_epi32(c.y[k]), cz = _mm256_set1_epi32(c.z[k]);
        for (int h = 0; h < 2; h++) {
            const __m256i dx = _mm256_sub_epi32(_mm256_load_si256((const __m256i*)(px.x + 8 * h)), cx);
            const __m256i dy = _mm256_sub_epi32(_mm256_load_si256((const __m256i*)(px.y + 8 * h)), cy);
            const __m256i dz = _mm256_sub_epi32(_mm256_load_si256((const __m256i*)(px.z + 8 * h)), cz);
            _mm256_store_si256((__m256i*)(t.d[k] + 8 * h), dist8(perceptual, dx, dy, dz));
        }
    }
}
BU_AVX2 inline __m256i table_half_avx2(const dist_table& t, __m128i sel8, int h) {
    const __m256i idx = _mm256_cvtepu8_epi32(sel8);
    __m256i acc = _mm256_and_si256(_mm256_cmpeq_epi32(idx, _mm256_setzero_si256()), _mm256_load_si256((const __m256i*)(t.d[0] + 8 * h)));
    for (int k = 1; k < 4; k++)
        acc = _mm256_add_epi32(acc, _mm256_and_si256(_mm256_cmpeq_epi32(idx, _mm256_set1_epi32(k)), _mm256_load_si256((const __m256i*)(t.d[k] + 8 * h))));
    return acc;
}
BU_AVX2 inline uint64_t table_error_avx2(const dist_table& t, const sel16& sel, uint64_t bound) {
    const __m128i s = _mm_load_si128((const __m128i*)sel.s);
    const __m256i lo = table_half_avx2(t, s, 0);
    if (bound != UINT64_MAX) { const uint64_t e = hsum8(lo); if (e > bound) return e; }
    return hsum8(_mm256_add_epi32(lo, table_half_avx2(t, _mm_srli_si128(s, 8), 1)));
}
BU_AVX2 inline void load_pixels_avx2(bool perceptual, block_px& out, const uint8_t* rgba16) {
    const __m256i m = _mm256_set1_epi32(255);
    for (int h = 0; h < 2; h++) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(rgba16 + 32 * h));
        const __m256i r = _mm256_and_si256(v, m), g = _mm256_and_si256(_mm256_srli_epi32(v, 8), m), b = _mm256_and_si256(_mm256_srli_epi32(v, 16), m);
        if (perceptual) {
            const __m256i l = _mm256_add_epi32(_mm256_add_epi32(_mm256_mullo_epi32(r, _mm256_set1_epi32(14)), _mm256_mullo_epi32(g, _mm256_set1_epi32(45))), _mm256_mullo_epi32(b, _mm256_set1_epi32(5)));
            _mm256_store_si256((__m256i*)(out.x + 8 * h), l);
            _mm256_store_si256((__m256i*)(out.y + 8 * h), _mm256_sub_epi32(_mm256_slli_epi32(r, 6), l));
            _mm256_store_si256((__m256i*)(out.z + 8 * h), _mm256_sub_epi32(_mm256_slli_epi32(b, 6), l));
        } else {
            _mm256_store_si256((__m256i*)(out.x + 8 * h), r); _mm256_store_si256((__m256i*)(out.y + 8 * h), g); _mm256_store_si256((__m256i*)(out.z + 8 * h), b);
        }
    }
}
BU_AVX2 inline scan_result scan_history_avx2(const dist_table& t, const sel16& cur, const sel16* hist, int sad_limit, uint64_t limit) {
    scan_result r{UINT64_MAX, -1};
    uint64_t todo = ~0ull;
    if (sad_limit > 0) {  // the pre-filter for all 64 patterns first, two per SAD instruction
        todo = 0;
        const __m256i c = _mm256_broadcastsi128_si256(_mm_load_si128((const __m128i*)cur.s)), lim = _mm256_set1_epi64x(sad_limit);
        for (int q = 0; q < 32; q++) {
            const __m256i d = _mm256_sad_epu8(c, _mm256_loadu_si256((const __m256i*)(hist + 2 * q)));
            const __m256i sum = _mm256_add_epi64(d, _mm256_bsrli_epi128(d, 8));
            const uint32_t m = (uint32_t)_mm256_movemask_pd(_mm256_castsi256_pd(_mm256_cmpgt_epi64(lim, sum)));  // bits 0 and 2
            todo |= (uint64_t)((m & 1u) | ((m >> 1) & 2u)) << (2 * q);
        }
    }
    while (todo) {  // the survivors four at a time with one shared lane reduction
        int j[4];
        __m256i acc[4];
        int n = 0;
        for (; n < 4 && todo; n++) { j[n] = __builtin_ctzll(todo); todo &= todo - 1; }
        for (int k = 0; k < 4; k++) {
            const __m128i h = _mm_load_si128((const __m128i*)hist[j[k < n ? k : 0]].s);
            acc[k] = _mm256_add_epi32(table_half_avx2(t, h, 0), table_half_avx2(t, _mm_srli_si128(h, 8), 1));
        }
        const __m256i t0 = _mm256_add_epi32(_mm256_unpacklo_epi32(acc[0], acc[1]), _mm256_unpackhi_epi32(acc[0], acc[1]));
        const __m256i t1 = _mm256_add_epi32(_mm256_unpacklo_epi32(acc[2], acc[3]), _mm256_unpackhi_epi32(acc[2], acc[3]));
        const __m256i u = _mm256_add_epi32(_mm256_unpacklo_epi64(t0, t1), _mm256_unpackhi_epi64(t0, t1));
        const __m128i w = _mm_add_epi32(_mm256_castsi256_si128(u), _mm256_extracti128_si256(u, 1));
        alignas(16) uint32_t e[4];
        _mm_store_si128((__m128i*)e, w);
        for (int k = 0; k < n; k++)
            if (e[k] < r.err && e[k] <= limit) { r.err = e[k]; r.index = j[k]; }
    }
    return r;
}
BU_AVX2 inline void block_errors_avx2(bool perceptual, const block_px& px, const sel16& sel, const pal_colors* colors, const int* which, int n, uint64_t* out) {
    const __m128i s = _mm_load_si128((const __m128i*)sel.s);
    const __m256i i0 = _mm256_cvtepu8_epi32(s), i1 = _mm256_cvtepu8_epi32(_mm_srli_si128(s, 8));
    const __m256i x0 = _mm256_load_si256((const __m256i*)px.x), x1 = _mm256_load_si256((const __m256i*)(px.x + 8)), y0 = _mm256_load_si256((const __m256i*)px.y),
                  y1 = _mm256_load_si256((const __m256i*)(px.y + 8)), z0 = _mm256_load_si256((const __m256i*)px.z), z1 = _mm256_load_si256((const __m256i*)(px.z + 8));
    for (int i = 0; i < n; i++) {
        const pal_colors& c = colors[which[i]];
        const __m256i cx = _mm256_castsi128_si256(_mm_load_si128((const __m128i*)c.x)), cy = _mm256_castsi128_si256(_mm_load_si128((const __m128i*)c.y)),
                      cz = _mm256_castsi128_si256(_mm_load_si128((const __m128i*)c.z));
        const __m256i a = dist8(perceptual, _mm256_sub_epi32(x0, _mm256_permutevar8x32_epi32(cx, i0)), _mm256_sub_epi32(y0, _mm256_permutevar8x32_epi32(cy, i0)),
                                _mm256_sub_epi32(z0, _mm256_permutevar8x32_epi32(cz, i0)));
        const __m256i b = dist8(perceptual, _mm256_sub_epi32(x1, _mm256_permutevar8x32_epi32(cx, i1)), _mm256_sub_epi32(y1, _mm256_permutevar8x32_epi32(cy, i1)),
                                _mm256_sub_epi32(z1, _mm256_permutevar8x32_epi32(cz, i1)));
        out[i] = hsum8(_mm256_add_epi32(a, b));
    }
}
#undef BU_AVX2

// ---- AVX-512 (F + BW + VL): all 16 pixels of a block in one register
#define BU_AVX512 __attribute__((target("avx512f,avx512bw,avx512vl,avx2")))
BU_AVX512 inline __m512i dist16(bool perceptual, __m512i dx, __m512i dy, __m512i dz) {
    const __m512i xx = _mm512_mullo_epi32(dx, dx), yy = _mm512_mullo_epi32(dy, dy), zz = _mm512_mullo_epi32(dz, dz);
    if (!perceptual) return _mm512_add_epi32(_mm512_add_epi32(xx, yy), zz);
    const __m512i l = _mm512_srli_epi32(xx, 5);
    const __m512i cr = _mm512_srli_epi32(_mm512_mullo_epi32(_mm512_srli_epi32(yy, 5), _mm512_set1_epi32(26)), 7);
    const __m512i cb = _mm512_srli_epi32(_mm512_mullo_epi32(_mm512_srli_epi32(zz, 5), _mm512_set1_epi32(3)), 7);
    return _mm512_add_epi32(_mm512_add_epi32(l, cr), cb);
}
BU_AVX512 inline uint64_t block_error_avx512(bool perceptual, const block_px& px, const pal_colors& c, const sel16& sel) {
    const __m512i idx = _mm512_cvtepu8_epi32(_mm_load_si128((const __m128i*)sel.s));
    const __m512i cx = _mm512_castsi128_si512(_mm_load_si128((const __m128i*)c.x)), cy = _mm512_castsi128_si512(_mm_load_si128((const __m128i*)c.y)),
                  cz = _mm512_castsi128_si512(_mm_load_si128((const __m128i*)c.z));
    const __m512i dx = _mm512_sub_epi32(_mm512_load_si512((const void*)px.x), _mm512_permutexvar_epi32(idx, cx));
    const __m512i dy = _mm512_sub_epi32(_mm512_load_si512((const void*)px.y), _mm512_permutexvar_epi32(idx, cy));
    const __m512i dz = _mm512_sub_epi32(_mm512_load_si512((const void*)px.z), _mm512_permutexvar_epi32(idx, cz));
    return (uint32_t)_mm512_reduce_add_epi32(dist16(perceptual, dx, dy, dz));
}
BU_AVX512 inline void build_table_avx512(bool perceptual, const block_px& px, const pal_colors& c, dist_table& t) {
    const __m512i x = _mm512_load_si512((const void*)px.x), y = _mm512_load_si512((const void*)px.y), z = _mm512_load_si512((const void*)px.z);
    for (int k = 0; k < 4; k++)
        _mm512_store_si512((void*)t.d[k], dist16(perceptual, _mm512_sub_epi32(x, _mm512_set1_epi32(c.x[k])), _mm512_sub_epi32(y, _mm512_set1_epi32(c.y[k])),
                                                 _mm512_sub_epi32(z, _mm512_set1_epi32(c.z[k]))));
}
BU_AVX512 inline uint64_t table_error_avx512(const dist_table& t, const sel16& sel, uint64_t) {
    const __m512i idx = _mm512_cvtepu8_epi32(_mm_load_si128((const __m128i*)sel.s));
    __m512i acc = _mm512_maskz_mov_epi32(_mm512_cmpeq_epi32_mask(idx, _mm512_setzero_si512()), _mm512_load_si512((const void*)t.d[0]));
    for (int k = 1; k < 4; k++) acc = _mm512_mask_add_epi32(acc, _mm512_cmpeq_epi32_mask(idx, _mm512_set1_epi32(k)), acc, _mm512_load_si512((const void*)t.d[k]));
    return (uint32_t)_mm512_reduce_add_epi32(acc);
}
BU_AVX512 inline void load_pixels_avx512(bool perceptual, block_px& out, const uint8_t* rgba16) {
    const __m512i m = _mm512_set1_epi32(255), v = _mm512_loadu_si512((const void*)rgba16);
    const __m512i r = _mm512_and_si512(v, m), g = _mm512_and_si512(_mm512_srli_epi32(v, 8), m), b = _mm512_and_si512(_mm512_srli_epi32(v, 16), m);
    if (perceptual) {
        const __m512i l = _mm512_add_epi32(_mm512_add_epi32(_mm512_mullo_epi32(r, _mm512_set1_epi32(14)), _mm512_mullo_epi32(g, _mm512_set1_epi32(45))), _mm512_mullo_epi32(b, _mm512_set1_epi32(5)));
        _mm512_store_si512((void*)out.x, l);
        _mm512_store_si512((void*)out.y, _mm512_sub_epi32(_mm512_slli_epi32(r, 6), l));
        _mm512_store_si512((void*)out.z, _mm512_sub_epi32(_mm512_slli_epi32(b, 6), l));
    } else {
        _mm512_store_si512((void*)out.x, r); _mm512_store_si512((void*)out.y, g); _mm512_store_si512((void*)out.z, b);
    }
}
// The exact errors of the patterns whose bit is set in `todo`, four at a time with one shared lane reduction; the minimum within `limit`,
// first index on ties.
BU_AVX512 inline scan_result scan_survivors_avx512(const dist_table& t, const sel16* hist, uint64_t todo, uint64_t limit) {
    const __m512i d0 = _mm512_load_si512((const void*)t.d[0]), d1 = _mm512_load_si512((const void*)t.d[1]), d2 = _mm512_load_si512((const void*)t.d[2]),
                  d3 = _mm512_load_si512((const void*)t.d[3]);
    const __m512i k1 = _mm512_set1_epi32(1), k2 = _mm512_set1_epi32(2), k3 = _mm512_set1_epi32(3);
    uint64_t best = UINT64_MAX;   // (error << 6) | index: the minimum is the smallest error at its first index (an error is below 2^32)
    while (todo) {
        uint32_t j[4];
        bool live[4];
        __m512i acc[4];
        for (int k = 0; k < 4; k++) {   // no branches on the number of patterns left: a spent slot repeats pattern 0 and is not counted
            live[k] = todo != 0;
            j[k] = live[k] ? (uint32_t)__builtin_ctzll(todo) : 0u;
            todo &= todo - (todo != 0);
            const __m512i idx = _mm512_cvtepu8_epi32(_mm_load_si128((const __m128i*)hist[j[k]].s));
            __m512i a = _mm512_maskz_mov_epi32(_mm512_testn_epi32_mask(idx, idx), d0);
            a = _mm512_mask_add_epi32(a, _mm512_cmpeq_epi32_mask(idx, k1), a, d1);
            a = _mm512_mask_add_epi32(a, _mm512_cmpeq_epi32_mask(idx, k2), a, d2);
            acc[k] = _mm512_mask_add_epi32(a, _mm512_cmpeq_epi32_mask(idx, k3), a, d3);
        }
        // lane sums of the four accumulators at once
        const __m512i t0 = _mm512_add_epi32(_mm512_unpacklo_epi32(acc[0], acc[1]), _mm512_unpackhi_epi32(acc[0], acc[1]));
        const __m512i t1 = _mm512_add_epi32(_mm512_unpacklo_epi32(acc[2], acc[3]), _mm512_unpackhi_epi32(acc[2], acc[3]));
        const __m512i u = _mm512_add_epi32(_mm512_unpacklo_epi64(t0, t1), _mm512_unpackhi_epi64(t0, t1));  // per 128-bit lane: partial sums of a, b, c, d
        const __m256i v = _mm256_add_epi32(_mm512_castsi512_si256(u), _mm512_extracti64x4_epi64(u, 1));
        const __m128i w = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
        alignas(16) uint32_t e[4];
        _mm_store_si128((__m128i*)e, w);
        for (int k = 0; k < 4; k++) {
            const uint64_t key = ((uint64_t)e[k] << 6) | j[k];
            best = (live[k] & (e[k] <= limit) & (key < best)) ? key : best;
        }
    }
    return best == UINT64_MAX ? scan_result{UINT64_MAX, -1} : scan_result{best >> 6, (int)(best & 63)};
}
// The pre-filter for all 64 patterns first (four per SAD instruction -> a 64-bit candidate mask), then the survivors four at a time with
// one shared lane reduction.
BU_AVX512 inline scan_result scan_history_avx512(const dist_table& t, const sel16& cur, const sel16* hist, int sad_limit, uint64_t limit) {
    uint64_t todo = ~0ull;
    if (sad_limit > 0) {
        todo = 0;
        const __m512i c = _mm512_broadcast_i32x4(_mm_load_si128((const __m128i*)cur.s)), lim = _mm512_set1_epi64(sad_limit);
        for (int q = 0; q < 16; q++) {
            const __m512i d = _mm512_sad_epu8(c, _mm512_loadu_si512((const void*)(hist + 4 * q)));   // per pattern: two 64-bit halves
            const __m512i sum = _mm512_add_epi64(d, _mm512_bsrli_epi128(d, 8));                      // even 64-bit lanes: the pattern's SAD
            const uint32_t m = _mm512_cmplt_epu64_mask(sum, lim);                                  // bits 0, 2, 4, 6
            todo |= (uint64_t)((m & 1u) | ((m >> 1) & 2u) | ((m >> 2) & 4u) | ((m >> 3) & 8u)) << (4 * q);
        }
    }
    return scan_survivors_avx512(t, hist, todo, limit);
}
// the 16 pixels against one candidate's colours under the block's selectors
BU_AVX512 inline __m512i candidate_distances_avx512(bool perceptual, __m512i x, __m512i y, __m512i z, __m512i idx, const pal_colors& c) {
    const __m512i cx = _mm512_castsi128_si512(_mm_load_si128((const __m128i*)c.x)), cy = _mm512_castsi128_si512(_mm_load_si128((const __m128i*)c.y)),
                  cz = _mm512_castsi128_si512(_mm_load_si128((const __m128i*)c.z));
    return dist16(perceptual, _mm512_sub_epi32(x, _mm512_permutexvar_epi32(idx, cx)), _mm512_sub_epi32(y, _mm512_permutexvar_epi32(idx, cy)), _mm512_sub_epi32(z, _mm512_permutexvar_epi32(idx, cz)));
}
BU_AVX512 inline void block_errors_avx512(bool perceptual, const block_px& px, const sel16& sel, const pal_colors* colors, const int* which, int n, uint64_t* out) {
    const __m512i idx = _mm512_cvtepu8_epi32(_mm_load_si128((const __m128i*)sel.s));
    const __m512i x = _mm512_load_si512((const void*)px.x), y = _mm512_load_si512((const void*)px.y), z = _mm512_load_si512((const void*)px.z);
    int i = 0;
    for (; i + 4 <= n; i += 4) {   // four candidates share one lane reduction (as in scan_survivors_avx512)
        const __m512i a0 = candidate_distances_avx512(perceptual, x, y, z, idx, colors[which[i]]), a1 = candidate_distances_avx512(perceptual, x, y, z, idx, colors[which[i + 1]]),
                      a2 = candidate_distances_avx512(perceptual, x, y, z, idx, colors[which[i + 2]]), a3 = candidate_distances_avx512(perceptual, x, y, z, idx, colors[which[i + 3]]);
        const __m512i t0 = _mm512_add_epi32(_mm512_unpacklo_epi32(a0, a1), _mm512_unpackhi_epi32(a0, a1));
        const __m512i t1 = _mm512_add_epi32(_mm512_unpacklo_epi32(a2, a3), _mm512_unpackhi_epi32(a2, a3));
        const __m512i u = _mm512_add_epi32(_mm512_unpacklo_epi64(t0, t1), _mm512_unpackhi_epi64(t0, t1));
        const __m256i v = _mm256_add_epi32(_mm512_castsi512_si256(u), _mm512_extracti64x4_epi64(u, 1));
        _mm256_storeu_si256((__m256i*)(out + i), _mm256_cvtepu32_epi64(_mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1))));
    }
    for (; i < n; i++) out[i] = (uint32_t)_mm512_reduce_add_epi32(candidate_distances_avx512(perceptual, x, y, z, idx, colors[which[i]]));
}
#undef BU_AVX512

// ---- AVX-512 VBMI: the history scan with a second pre-filter in front of the exact errors. The distance table is cut down to one byte per
// entry, q[k][p] = min(255, d[k][p] >> shift), with the shift chosen so that limit >> shift stays below 1024: the sum of a pattern's q is
// never above its error >> shift, so a pattern whose byte sum exceeds limit >> shift cannot be within the limit. All 64 byte sums cost one
// VPERMB (the 64-entry table IS one register) and one VPSADBW per four patterns. On a 2048^2 photo-like image at level 1: 21.7 of the 64
// patterns pass the selector-SAD test, 3.4 pass both, 2.9 are within the limit (a cap of 2048 lets 6.4 through, 512 3.6: a pixel whose
// distance saturates the byte costs more than the rounding does).
#define BU_VBMI __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi,avx2,bmi2,lzcnt")))
BU_VBMI inline scan_result scan_history_vbmi(const dist_table& t, const sel16& cur, const sel16* hist, int sad_limit, uint64_t limit) {
    const int bits = 64 - (int)_lzcnt_u64(limit), shift = bits > 10 ? bits - 10 : 0;   // limit >> shift < 1024
    const __m128i cnt = _mm_cvtsi32_si128(shift);
    const __m512i cap = _mm512_set1_epi32(255);
    __m512i plane = _mm512_castsi128_si512(_mm512_cvtepi32_epi8(_mm512_min_epu32(_mm512_srl_epi32(_mm512_load_si512((const void*)t.d[0]), cnt), cap)));
    plane = _mm512_inserti32x4(plane, _mm512_cvtepi32_epi8(_mm512_min_epu32(_mm512_srl_epi32(_mm512_load_si512((const void*)t.d[1]), cnt), cap)), 1);
    plane = _mm512_inserti32x4(plane, _mm512_cvtepi32_epi8(_mm512_min_epu32(_mm512_srl_epi32(_mm512_load_si512((const void*)t.d[2]), cnt), cap)), 2);
    plane = _mm512_inserti32x4(plane, _mm512_cvtepi32_epi8(_mm512_min_epu32(_mm512_srl_epi32(_mm512_load_si512((const void*)t.d[3]), cnt), cap)), 3);
    const __m512i pos = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
    const __m512i c = _mm512_broadcast_i32x4(_mm_load_si128((const __m128i*)cur.s)), zero = _mm512_setzero_si512();
    // Per pattern a pair of 64-bit lanes, each (byte sum of eight pixels) | (selector SAD of eight pixels) << 32; two registers of four patterns are
    // folded into one of eight sums, whose 32-bit lanes are compared against (limit >> shift) + 1 and sad_limit in one instruction.
    const __m512i bound = _mm512_set1_epi64((long long)((limit >> shift) + 1) | ((long long)(sad_limit > 0 ? sad_limit : 1 << 20) << 32));
    uint64_t pass[2] = {0, 0};   // 16 bits per eight patterns: bit 2l the byte-sum test, bit 2l+1 the SAD test of pattern (l & 1) * 4 + (l >> 1)
    for (int q = 0; q < 8; q++) {
        const __m512i ha = _mm512_load_si512((const void*)(hist + 8 * q)), hb = _mm512_load_si512((const void*)(hist + 8 * q + 4));
        const __m512i va = _mm512_or_si512(_mm512_sad_epu8(_mm512_permutexvar_epi8(_mm512_or_si512(_mm512_slli_epi16(ha, 4), pos), plane), zero), _mm512_slli_epi64(_mm512_sad_epu8(c, ha), 32));
        const __m512i vb = _mm512_or_si512(_mm512_sad_epu8(_mm512_permutexvar_epi8(_mm512_or_si512(_mm512_slli_epi16(hb, 4), pos), plane), zero), _mm512_slli_epi64(_mm512_sad_epu8(c, hb), 32));
        const uint64_t m = _mm512_cmplt_epu32_mask(_mm512_add_epi64(_mm512_unpacklo_epi64(va, vb), _mm512_unpackhi_epi64(va, vb)), bound);
        pass[q >> 2] |= m << (16 * (q & 3));
    }
    uint64_t todo = 0;
    for (int w = 0; w < 2; w++) {
        const uint64_t ok = _pext_u64(pass[w] & (pass[w] >> 1), 0x5555555555555555ull);   // per byte: a0 b0 a1 b1 a2 b2 a3 b3 (a: first four patterns, b: next four)
        todo |= (_pdep_u64(_pext_u64(ok, 0x55555555ull), 0x0F0F0F0Full) | _pdep_u64(_pext_u64(ok, 0xAAAAAAAAull), 0xF0F0F0F0ull)) << (32 * w);
    }
    return scan_survivors_avx512(t, hist, todo, limit);
}
// The search of one block in two calls: search_prepare_vbmi -- pixels to metric space, distance table, own error, limit, the byte tables -- has no input from the
// history, so the selector walk issues it for block i + 1 BEFORE search_history_vbmi of block i: the walk is one chain of dependent steps per block (find, filter,
// exact sums, update), 260 cycles long when the preparation was at its head, and the core overlaps the two calls on its own once they are independent.
BU_VBMI inline void search_prepare_vbmi(bool perceptual, const uint8_t* rgba16, const pal_colors& colors, const sel16& cur, float thresh, search_prep& out) {
    alignas(64) block_px px;
    load_pixels_avx512(perceptual, px, rgba16);
    build_table_avx512(perceptual, px, colors, out.t);
    const uint64_t own = table_error_avx512(out.t, cur, UINT64_MAX);
    const uint64_t limit = (uint64_t)ceilf(own * thresh);
    const int bits = 64 - (int)_lzcnt_u64(limit), shift = bits > 10 ? bits - 10 : 0;   // limit >> shift < 1024
    out.limit = limit;
    out.shift = (uint32_t)shift;
    // the table's byte planes: plane b = byte b of the 64 distances in table order, two two-source byte permutes (32 distances each) per plane
    const __m512i d0 = _mm512_load_si512((const void*)out.t.d[0]), d1 = _mm512_load_si512((const void*)out.t.d[1]), d2 = _mm512_load_si512((const void*)out.t.d[2]),
                  d3 = _mm512_load_si512((const void*)out.t.d[3]);
    alignas(64) static const struct pick_t {
        uint8_t at[4][64];   // at[b][i]: byte b of dword i of the 128-byte pair (i < 32); the upper half is not used
        pick_t() { for (int b = 0; b < 4; b++) for (int i = 0; i < 64; i++) at[b][i] = (uint8_t)(((i & 31) * 4 + b) & 127); }
    } pick;
    for (int b = 0; b < 4; b++) {
        const __m512i ix = _mm512_load_si512((const void*)pick.at[b]);
        _mm512_store_si512((void*)out.exact[b], _mm512_inserti64x4(_mm512_permutex2var_epi8(d0, ix, d1), _mm512_castsi512_si256(_mm512_permutex2var_epi8(d2, ix, d3)), 1));
    }
    const __m128i cnt = _mm_cvtsi32_si128(shift);
    const __m512i cap = _mm512_set1_epi32(255), ix0 = _mm512_load_si512((const void*)pick.at[0]);
    const __m512i q0 = _mm512_min_epu32(_mm512_srl_epi32(d0, cnt), cap), q1 = _mm512_min_epu32(_mm512_srl_epi32(d1, cnt), cap), q2 = _mm512_min_epu32(_mm512_srl_epi32(d2, cnt), cap),
                  q3 = _mm512_min_epu32(_mm512_srl_epi32(d3, cnt), cap);
    _mm512_store_si512((void*)out.bound, _mm512_inserti64x4(_mm512_permutex2var_epi8(q0, ix0, q1), _mm512_castsi512_si256(_mm512_permutex2var_epi8(q2, ix0, q3)), 1));
}
BU_VBMI inline scan_result search_history_vbmi(const search_prep& pr, const sel16& cur, const sel16* hist, int sad_limit, const int* hist_values, int own_value) {
    if (own_value >= 0) {   // the block's own pattern is in the history: that entry, no search (backend.cpp:1024-1034)
        const __m512i key = _mm512_set1_epi32(own_value);
        const uint64_t m = (uint64_t)_mm512_cmpeq_epi32_mask(_mm512_loadu_si512((const void*)hist_values), key) | ((uint64_t)_mm512_cmpeq_epi32_mask(_mm512_loadu_si512((const void*)(hist_values + 16)), key) << 16) |
                           ((uint64_t)_mm512_cmpeq_epi32_mask(_mm512_loadu_si512((const void*)(hist_values + 32)), key) << 32) | ((uint64_t)_mm512_cmpeq_epi32_mask(_mm512_loadu_si512((const void*)(hist_values + 48)), key) << 48);
        if (m) return scan_result{0, (int)__builtin_ctzll(m)};
    }
    const uint64_t limit = pr.limit;
    const __m512i plane = _mm512_load_si512((const void*)pr.bound);
    const __m512i pos = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15));
    const __m512i c = _mm512_broadcast_i32x4(_mm_load_si128((const __m128i*)cur.s)), zero = _mm512_setzero_si512();
    const __m512i bound = _mm512_set1_epi64((long long)((limit >> pr.shift) + 1) | ((long long)(sad_limit > 0 ? sad_limit : 1 << 20) << 32));
    uint64_t pass[2] = {0, 0};   // as in scan_history_vbmi
    for (int q = 0; q < 8; q++) {
        const __m512i ha = _mm512_load_si512((const void*)(hist + 8 * q)), hb = _mm512_load_si512((const void*)(hist + 8 * q + 4));
        const __m512i va = _mm512_or_si512(_mm512_sad_epu8(_mm512_permutexvar_epi8(_mm512_or_si512(_mm512_slli_epi16(ha, 4), pos), plane), zero), _mm512_slli_epi64(_mm512_sad_epu8(c, ha), 32));
        const __m512i vb = _mm512_or_si512(_mm512_sad_epu8(_mm512_permutexvar_epi8(_mm512_or_si512(_mm512_slli_epi16(hb, 4), pos), plane), zero), _mm512_slli_epi64(_mm512_sad_epu8(c, hb), 32));
        const uint64_t m = _mm512_cmplt_epu32_mask(_mm512_add_epi64(_mm512_unpacklo_epi64(va, vb), _mm512_unpackhi_epi64(va, vb)), bound);
        pass[q >> 2] |= m << (16 * (q & 3));
    }
    uint64_t todo = 0;
    for (int w = 0; w < 2; w++) {
        const uint64_t ok = _pext_u64(pass[w] & (pass[w] >> 1), 0x5555555555555555ull);
        todo |= (_pdep_u64(_pext_u64(ok, 0x55555555ull), 0x0F0F0F0Full) | _pdep_u64(_pext_u64(ok, 0xAAAAAAAAull), 0xF0F0F0F0ull)) << (32 * w);
    }
    if (!todo) return scan_result{UINT64_MAX, -1};
    // The exact errors of what is left, four patterns per round: their 64 selectors are one VPERMB index register, the table's four byte planes give four
    // registers of bytes, VPSADBW sums eight of them at a time, and the sums are put back together with shifts -- a short chain (the masked adds plus lane
    // reduction of scan_survivors_avx512 take three times as long from first load to result, and this loop waits for nothing else).
    const __m512i e0 = _mm512_load_si512((const void*)pr.exact[0]), e1 = _mm512_load_si512((const void*)pr.exact[1]), e2 = _mm512_load_si512((const void*)pr.exact[2]),
                  e3 = _mm512_load_si512((const void*)pr.exact[3]);
    const __m256i lim4 = _mm256_set1_epi64x((long long)limit);
    __m256i best = _mm256_set1_epi64x(-1);   // per lane (error << 6) | index; all ones: nothing
    do {
        uint64_t j[4];
        uint32_t n = 0;
        for (int k = 0; k < 4; k++) {   // no branches on the number of patterns left: a spent slot repeats pattern 0 and is masked out
            const bool live = todo != 0;
            j[k] = live ? (uint64_t)__builtin_ctzll(todo) : 0u;
            n += live;
            todo &= todo - (todo != 0);
        }
        __m512i h = _mm512_castsi128_si512(_mm_load_si128((const __m128i*)hist[j[0]].s));
        h = _mm512_inserti32x4(h, _mm_load_si128((const __m128i*)hist[j[1]].s), 1);
        h = _mm512_inserti32x4(h, _mm_load_si128((const __m128i*)hist[j[2]].s), 2);
        h = _mm512_inserti32x4(h, _mm_load_si128((const __m128i*)hist[j[3]].s), 3);
        const __m512i idx = _mm512_or_si512(_mm512_slli_epi16(h, 4), pos);
        const __m512i s0 = _mm512_sad_epu8(_mm512_permutexvar_epi8(idx, e0), zero), s1 = _mm512_sad_epu8(_mm512_permutexvar_epi8(idx, e1), zero),
                      s2 = _mm512_sad_epu8(_mm512_permutexvar_epi8(idx, e2), zero), s3 = _mm512_sad_epu8(_mm512_permutexvar_epi8(idx, e3), zero);
        const __m512i half = _mm512_add_epi64(_mm512_add_epi64(s0, _mm512_slli_epi64(s1, 8)), _mm512_add_epi64(_mm512_slli_epi64(s2, 16), _mm512_slli_epi64(s3, 24)));
        const __m256i err = _mm512_castsi512_si256(_mm512_maskz_compress_epi64(0x55, _mm512_add_epi64(half, _mm512_bsrli_epi128(half, 8))));   // four errors
        const __m256i key = _mm256_or_si256(_mm256_slli_epi64(err, 6), _mm256_set_epi64x((long long)j[3], (long long)j[2], (long long)j[1], (long long)j[0]));
        const __mmask8 ok = (__mmask8)(_mm256_cmple_epu64_mask(err, lim4) & ((1u << n) - 1u));
        best = _mm256_mask_min_epu64(best, ok, best, key);
    } while (todo);
    const __m128i b2 = _mm_min_epu64(_mm256_castsi256_si128(best), _mm256_extracti128_si256(best, 1));
    const uint64_t b = (uint64_t)_mm_cvtsi128_si64(_mm_min_epu64(b2, _mm_unpackhi_epi64(b2, b2)));
    return b == UINT64_MAX ? scan_result{UINT64_MAX, -1} : scan_result{b >> 6, (int)(b & 63)};
}
#undef BU_VBMI

// sum over the pixels of |selector difference| (SSE2: part of the x86-64 baseline)
inline int selector_sad(const sel16& a, const sel16& b) {
    const __m128i s = _mm_sad_epu8(_mm_load_si128((const __m128i*)a.s), _mm_load_si128((const __m128i*)b.s));
    return _mm_cvtsi128_si32(s) + _mm_cvtsi128_si32(_mm_srli_si128(s, 8));
}
// first j in [0, 64) with v[j] == x, or -1
inline int find_first_64(const int* v, int x) {
    const __m128i key = _mm_set1_epi32(x);
    for (int j = 0; j < 64; j += 4) {
        const int m = _mm_movemask_ps(_mm_castsi128_ps(_mm_cmpeq_epi32(_mm_loadu_si128((const __m128i*)(v + j)), key)));
        if (m) return j + __builtin_ctz((unsigned)m);
    }
    return -1;
}

// search_prepare / search_history: what the selector walk asks per block, in two calls (see the VBMI pair above for why two) -- the history entry that holds the
// block's own pattern number (own_value >= 0: levels 0 and 1 look for it first), else the best history pattern within own error * thresh (ceilf of the float
// product, as the reference computes its limit, backend.cpp:1051).
#define BU_HISTORY_SEARCH(TARGET, SUFFIX) \
    TARGET inline void search_prepare_##SUFFIX(bool perceptual, const uint8_t* rgba16, const pal_colors& colors, const sel16& cur, float thresh, search_prep& out) { \
        block_px px; \
        load_pixels_##SUFFIX(perceptual, px, rgba16); \
        build_table_##SUFFIX(perceptual, px, colors, out.t); \
        out.limit = (uint64_t)ceilf(table_error_##SUFFIX(out.t, cur, UINT64_MAX) * thresh); \
    } \
    TARGET inline scan_result search_history_##SUFFIX(const search_prep& pr, const sel16& cur, const sel16* hist, int sad_limit, const int* hist_values, int own_value) { \
        if (own_value >= 0) { const int at = find_first_64(hist_values, own_value); if (at >= 0) return scan_result{0, at}; } \
        return scan_history_##SUFFIX(pr.t, cur, hist, sad_limit, pr.limit); \
    }
BU_HISTORY_SEARCH(, plain)
BU_HISTORY_SEARCH(__attribute__((target("avx2"))), avx2)
BU_HISTORY_SEARCH(__attribute__((target("avx512f,avx512bw,avx512vl,avx2"))), avx512)
#undef BU_HISTORY_SEARCH

struct kernels {
    void (*search_prepare)(bool, const uint8_t*, const pal_colors&, const sel16&, float, search_prep&);
    scan_result (*search_history)(const search_prep&, const sel16&, const sel16*, int, const int*, int);
    uint64_t (*block_error)(bool, const block_px&, const pal_colors&, const sel16&);
    void (*build_table)(bool, const block_px&, const pal_colors&, dist_table&);
    uint64_t (*table_error)(const dist_table&, const sel16&, uint64_t);
    scan_result (*scan_history)(const dist_table&, const sel16&, const sel16*, int, uint64_t);
    void (*block_errors)(bool, const block_px&, const sel16&, const pal_colors*, const int*, int, uint64_t*);
    void (*load_pixels)(bool, block_px&, const uint8_t*);
    window_mask (*filter_window)(const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, int, int, int, int, int, bool);
    const char* isa;
};
inline kernels pick_kernels() {  // BU_BACKEND_ISA = plain | avx2 | avx512 | vbmi caps the choice (tests run all of them)
    const char* cap = std::getenv("BU_BACKEND_ISA");
    const int level = !cap ? 3 : (!std::strcmp(cap, "plain") ? 0 : (!std::strcmp(cap, "avx2") ? 1 : (!std::strcmp(cap, "avx512") ? 2 : 3)));
    if (level >= 3 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512vbmi") &&
        __builtin_cpu_supports("bmi2"))
        return kernels{search_prepare_vbmi, search_history_vbmi, block_error_avx512, build_table_avx512, table_error_avx512, scan_history_vbmi, block_errors_avx512, load_pixels_avx512, filter_window_avx2, "avx512+vbmi"};
    if (level >= 2 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl"))
        return kernels{search_prepare_avx512, search_history_avx512, block_error_avx512, build_table_avx512, table_error_avx512, scan_history_avx512, block_errors_avx512, load_pixels_avx512, filter_window_avx2, "avx512"};
    if (level >= 1 && __builtin_cpu_supports("avx2")) return kernels{search_prepare_avx2, search_history_avx2, block_error_avx2, build_table_avx2, table_error_avx2, scan_history_avx2, block_errors_avx2, load_pixels_avx2, filter_window_avx2, "avx2"};
    return kernels{search_prepare_plain, search_history_plain, block_error_plain, build_table_plain, table_error_plain, scan_history_plain, block_errors_plain, load_pixels_plain, filter_window_plain, "plain"};
}

}  // namespace metric
}  // namespace bu
