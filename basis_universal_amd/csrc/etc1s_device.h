// etc1s_device.h -- device-side building blocks shared by the ETC1S kernels (gfx950 only).
//
// Everything here is integer-exact with respect to the reference CPU encoder:
//   colour distance      encoder/basisu_enc.h:1141-1195 (perceptual) / :1075-1106 (linear)
//   ETC1S block colours  encoder/basisu_etc.h:584-602 (get_block_colors5), inten tables etc.cpp:304-308
//   etc_block bit layout encoder/basisu_etc.h:91-330
//   hash_hsieh           transcoder/basisu_transcoder.cpp:355-409 (3-byte key case, used by etc.cpp:1072-1089)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bu {

// g_etc1_inten_tables (etc.cpp:304-308). Rows are symmetric: {-b, -a, a, b}.
__device__ __constant__ static const int k_inten_a[8] = {2, 5, 9, 13, 18, 24, 33, 47};
__device__ __constant__ static const int k_inten_b[8] = {8, 17, 29, 42, 60, 80, 106, 183};

__device__ __forceinline__ int inten_delta(int table, int sel) {
    // sel 0..3 -> -b, -a, +a, +b
    const int v = (sel == 0 || sel == 3) ? k_inten_b[table] : k_inten_a[table];
    return (sel < 2) ? -v : v;
}

__device__ __forceinline__ int clamp255(int v) { return min(max(v, 0), 255); }
__device__ __forceinline__ int scale5(int c) { return (c << 3) | (c >> 2); }

// A colour (pixel or block colour) in the basis the distance metric is separable in.
//   perceptual: x = 14r+45g+5b (luma*64), y = 64r - x, z = 64b - x     -> delta_l, delta_cr, delta_cb are plain differences
//   linear:     x = r, y = g, z = b
struct cvec { int x, y, z; };

template <bool PERCEPTUAL>
__device__ __forceinline__ cvec to_cvec(int r, int g, int b) {
    cvec c;
    if (PERCEPTUAL) {
        const int l = r * 14 + g * 45 + b * 5;
        c.x = l; c.y = r * 64 - l; c.z = b * 64 - l;
    } else {
        c.x = r; c.y = g; c.z = b;
    }
    return c;
}

// color_distance(perceptual, a, b, false). |x| <= 16320 and |y|,|z| <= 32640 so the squares fit 31 bits and every
// intermediate is exact in 24-bit-multiplier arithmetic except the *26, which is done with shifts.
template <bool PERCEPTUAL>
__device__ __forceinline__ uint32_t cdist(const cvec& a, const cvec& b) {
    const int dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    if (PERCEPTUAL) {
        const uint32_t l2 = (uint32_t)__mul24(dx, dx) >> 5;
        const uint32_t cr = (uint32_t)__mul24(dy, dy) >> 5;
        const uint32_t cb = (uint32_t)__mul24(dz, dz) >> 5;
        const uint32_t cr26 = (cr << 4) + (cr << 3) + (cr << 1);
        const uint32_t cb3 = (cb << 1) + cb;
        return l2 + (cr26 >> 7) + (cb3 >> 7);
    } else {
        return (uint32_t)(__mul24(dx, dx) + __mul24(dy, dy) + __mul24(dz, dz));
    }
}

// ---- block colours that need no clamping. The four colours of (base, table) are base + d * (1,1,1), d = -b, -a, +a, +b; when
// 0 <= min(base) - b and max(base) + b <= 255 none of them is clamped, and in the basis of cvec a grey offset d moves x by 64 d
// (14 + 45 + 5 = 64) and leaves y and z alone. The two chroma terms of the distance are then the same for all four colours (for
// all eight tables, even), and because floor is monotone
//     min_s [ (dx_s^2 >> 5) + C ] = (min_s dx_s^2 >> 5) + C :
// one chroma term per pixel and candidate instead of four, bit for bit the same minimum. (Perceptual metric only.)
__device__ __forceinline__ bool base_unclamped(int br, int bg, int bb, int table) {
    const int b = k_inten_b[table];
    return min(br, min(bg, bb)) - b >= 0 && max(br, max(bg, bb)) + b <= 255;
}
// perceptual only: the part of the distance the chroma differences contribute
__device__ __forceinline__ uint32_t chroma_term(int dy, int dz) {
    const uint32_t cr = (uint32_t)__mul24(dy, dy) >> 5;
    const uint32_t cb = (uint32_t)__mul24(dz, dz) >> 5;
    const uint32_t cr26 = (cr << 4) + (cr << 3) + (cr << 1);
    const uint32_t cb3 = (cb << 1) + cb;
    return (cr26 >> 7) + (cb3 >> 7);
}
// perceptual only: a LOWER BOUND of the sum over a tile's 16 pixels of chroma_term(y_p - cy, z_p - cz) from the tile's chroma moments -- what a sweep over
// many candidate colours tests before it spends sixteen chroma terms on one (k_refine_sorted). With r_p = y_p - m (m = floor(mean y)), R1 = sum r_p (0..15),
// e = m - cy:  sum (y_p - cy)^2 = sum r_p^2 + 2 e R1 + 16 e^2, so  A = (e^2 >> 6) + sum (r_p^2 >> 10) + ((e R1) >> 9)  <=  sum (y_p - cy)^2 / 1024  (floors only
// take away), B the same for z. A pixel's term is floor(26 floor(dy^2 / 32) / 128) + floor(3 floor(dz^2 / 32) / 128) >= (26 dy^2 + 3 dz^2) / 4096 - 2.204
// (each inner floor loses < 31/32 of its factor, each outer one < 127/128), over 16 pixels >= (26 A + 3 B) / 4 - 35.3. |y|, |z| <= 15045: everything fits 31 bits
// (A, B < 2^24). struct: the tile's side of it (wave-uniform in the caller).
struct chroma_moments { int my, mz, r1y, r1z, r2y, r2z; };   // floor means, sum r_p, sum (r_p^2 >> 10)
__device__ __forceinline__ uint32_t chroma_lower_bound(const chroma_moments& t, int cy, int cz) {
    const int ey = t.my - cy, ez = t.mz - cz;
    const int A = (int)((uint32_t)__mul24(ey, ey) >> 6) + t.r2y + (__mul24(ey, t.r1y) >> 9);
    const int B = (int)((uint32_t)__mul24(ez, ez) >> 6) + t.r2z + (__mul24(ez, t.r1z) >> 9);
    return (uint32_t)max(((__mul24(A, 26) + __mul24(B, 3)) >> 2) - 36, 0);
}
// perceptual only: min over the four unclamped colours of the luma term, dx0 = pixel.x - base.x, a64 / b64 = 64 * the table's two deltas
// The four offsets are +-a, +-b: of each pair the one on dx0's side is the nearer, (|dx0| - a)^2 <= (|dx0| + a)^2, so two squares decide the minimum of four
// (the same integers; |dx0| is shared by a pixel's eight tables). Until round 6 all four were squared: twice the multiplies and minimums of the kernels' innermost term.
__device__ __forceinline__ uint32_t min_luma_term(int dx0, int a64, int b64) {
    const int m = dx0 < 0 ? -dx0 : dx0;
    const int ea = m - a64, eb = m - b64;
    return min((uint32_t)__mul24(ea, ea), (uint32_t)__mul24(eb, eb)) >> 5;
}

// perceptual only: min over the four block colours of the distance to ONE pixel when some of the colours are clamped. An offset d that clamps no channel
// (mn + d >= 0 for the negative ones, mx + d <= 255 for the positive ones; mn / mx = the base colour's smallest / largest channel) still shares the base
// colour's chroma, so it costs one square; only the offsets that do clamp take the full distance. The minimum of the same four integers as min_err4's.
struct mixed_min { uint32_t luma_sq, full; };   // min squared luma difference over the unclamped offsets (not yet >> 5), min distance over the clamped ones
__device__ __forceinline__ uint32_t mixed_min_total(const mixed_min& m, uint32_t chroma) { return min((m.luma_sq >> 5) + chroma, m.full); }
// The four block colours of (scaled base colour, intensity table), clamped per channel (etc.h:584-602).
template <bool PERCEPTUAL>
__device__ __forceinline__ void block_cvecs(cvec out[4], int br, int bg, int bb, int table) {
    const int a = k_inten_a[table], b = k_inten_b[table];
    out[0] = to_cvec<PERCEPTUAL>(clamp255(br - b), clamp255(bg - b), clamp255(bb - b));
    out[1] = to_cvec<PERCEPTUAL>(clamp255(br - a), clamp255(bg - a), clamp255(bb - a));
    out[2] = to_cvec<PERCEPTUAL>(clamp255(br + a), clamp255(bg + a), clamp255(bb + a));
    out[3] = to_cvec<PERCEPTUAL>(clamp255(br + b), clamp255(bg + b), clamp255(bb + b));
}

// bc[s], s = 0..3, as selects on the components: a conditional expression on the whole struct makes clang copy from a selected POINTER,
// which keeps the four colours in scratch memory instead of registers
__device__ __forceinline__ cvec select_cvec(const cvec bc[4], uint32_t s) {
    const int x0 = bc[0].x, x1 = bc[1].x, x2 = bc[2].x, x3 = bc[3].x, y0 = bc[0].y, y1 = bc[1].y, y2 = bc[2].y, y3 = bc[3].y;
    const int z0 = bc[0].z, z1 = bc[1].z, z2 = bc[2].z, z3 = bc[3].z;
    const bool odd = (s & 1u) != 0, up = (s & 2u) != 0;
    cvec c;
    c.x = up ? (odd ? x3 : x2) : (odd ? x1 : x0);
    c.y = up ? (odd ? y3 : y2) : (odd ? y1 : y0);
    c.z = up ? (odd ? z3 : z2) : (odd ? z1 : z0);
    return c;
}

// min over the 4 selectors (error only)
template <bool PERCEPTUAL>
__device__ __forceinline__ uint32_t min_err4(const cvec& p, const cvec bc[4]) {
    const uint32_t e0 = cdist<PERCEPTUAL>(p, bc[0]), e1 = cdist<PERCEPTUAL>(p, bc[1]);
    const uint32_t e2 = cdist<PERCEPTUAL>(p, bc[2]), e3 = cdist<PERCEPTUAL>(p, bc[3]);
    return min(min(e0, e1), min(e2, e3));
}

// first-minimum selector (strict <, ascending s) as in etc.cpp:1188-1219 and etc.h:374-436
template <bool PERCEPTUAL>
__device__ __forceinline__ uint32_t best_sel4(const cvec& p, const cvec bc[4]) {
    uint32_t be = cdist<PERCEPTUAL>(p, bc[0]), bs = 0;
#pragma unroll
    for (uint32_t s = 1; s < 4; s++) {
        const uint32_t e = cdist<PERCEPTUAL>(p, bc[s]);
        if (e < be) { be = e; bs = s; }
    }
    return bs;
}

__device__ __forceinline__ cvec pixel_cvec_lin(uint32_t rgba) {
    cvec c; c.x = rgba & 255; c.y = (rgba >> 8) & 255; c.z = (rgba >> 16) & 255; return c;
}
template <bool PERCEPTUAL>
__device__ __forceinline__ cvec pixel_cvec(uint32_t rgba) {
    return to_cvec<PERCEPTUAL>(rgba & 255, (rgba >> 8) & 255, (rgba >> 16) & 255);
}

// hash_hsieh over the 3 bytes (r,g,b) of an unscaled colour
__device__ __forceinline__ uint32_t hash_hsieh3(uint32_t r, uint32_t g, uint32_t b) {
    uint32_t h = 3u;
    h += r | (g << 8);
    h ^= h << 16;
    h ^= ((uint32_t)(int32_t)(int8_t)b) << 18;
    h += h >> 11;
    h ^= h << 3;  h += h >> 5;
    h ^= h << 4;  h += h >> 17;
    h ^= h << 25; h += h >> 6;
    return h;
}

// etc_block as the big-endian u64 value V; the bytes in memory are bswap64(V) when stored from a little-endian register.
__device__ __forceinline__ uint64_t etc1s_header_bits(uint32_t r5, uint32_t g5, uint32_t b5, uint32_t inten) {
    return ((uint64_t)r5 << 59) | ((uint64_t)g5 << 51) | ((uint64_t)b5 << 43) | ((uint64_t)inten << 37) | ((uint64_t)inten << 34) |
           (1ull << 33) | (1ull << 32); // diff bit, flip bit; delta3 = 0
}
// Contribution of selector `sel` (inten-table index) of pixel (x,y) to the low 32 bits of V (etc.h:232-258):
// raw code {3,2,0,1}[sel]; lsb plane at bit x*4+y, msb plane at bit 16+x*4+y.
__device__ __forceinline__ uint32_t selector_bits(uint32_t x, uint32_t y, uint32_t sel) {
    const uint32_t raw = (0x4Bu >> (sel * 2)) & 3u; // g_selector_index_to_etc1 = {3,2,0,1} packed 2 bits per entry, entry 0 lowest
    const uint32_t bit = x * 4 + y;
    return ((raw & 1u) << bit) | ((raw >> 1) << (16 + bit));
}
__device__ __forceinline__ uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }

__device__ __forceinline__ void unpack_etc1s_header(uint64_t mem_le, uint32_t& r5, uint32_t& g5, uint32_t& b5, uint32_t& inten) {
    const uint64_t v = bswap64(mem_le);
    r5 = (uint32_t)(v >> 59) & 31; g5 = (uint32_t)(v >> 51) & 31; b5 = (uint32_t)(v >> 43) & 31; inten = (uint32_t)(v >> 37) & 7;
}
// selector (inten-table index) of pixel (x,y) from the low 32 bits of V: g_etc1_to_selector_index = {2,3,1,0}
__device__ __forceinline__ uint32_t selector_from_bits(uint32_t lo32, uint32_t x, uint32_t y) {
    const uint32_t bit = x * 4 + y;
    const uint32_t raw = ((lo32 >> bit) & 1u) | (((lo32 >> (16 + bit)) & 1u) << 1);
    return (0x1Eu >> (raw * 2)) & 3u; // {2,3,1,0}: raw0->2, raw1->3, raw2->1, raw3->0 -> bits 00 01 11 10 = 0x1E
}

} // namespace bu
