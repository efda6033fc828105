// uastc_core.h -- UASTC LDR 4x4 block encoder core, single source for the HIP kernels.
//
// What it computes is pinned, bit for bit, by the reference's encode_uastc() (encoder/basisu_uastc_enc.cpp:3126-3645) and the
// pieces it calls: color_cell_compression (encoder/basisu_bc7enc.cpp:1364-1762), the astc_mode* candidate generators
// (uastc_enc.cpp:470-2508), unpack_uastc / transcode_uastc_to_bc7 / the BC7 and BC1 decoders (transcoder/basisu_transcoder.cpp),
// the BC1 / EAC / ETC1 transcode hints (uastc_enc.cpp:2535-3103) and pack_uastc (:110-466). How it is organised is ours:
//   * one generic colour-cell fitter specialised to the only configuration UASTC uses (ASTC endpoint ranks, unit channel
//     weights, non-perceptual metric), one generic candidate builder driven by a per-mode descriptor instead of 19 functions,
//   * candidates live in fixed slots so that independent (block, job) work items can run as separate GPU threads and the
//     reference's first-wins tie rules still hold (slot order = the reference's result order),
//   * the BC7 round trip decodes straight from the transcoded endpoints (bit packing and unpacking cancel),
//   * all tables come from uastc_tables.inc (generated; see tools/gen_uastc_tables.py).
// Float code keeps the reference's evaluation order; build with -ffp-contract=off (SURVEY hazard H4).
//
// The same header compiles with g++ (BU_FN = static inline): tests use that build to diff every stage against the real
// reference on the CPU; the product only ever runs the hipcc build.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

#if defined(__HIPCC__)
#define BU_FN __device__ inline
#define BU_FN_MEMBER __device__ inline
#define BU_FN_HD __host__ __device__ inline
#define BU_FN_BIG __device__ __noinline__
#define BU_TAB static __device__ const
#else
#define BU_FN static inline
#define BU_FN_MEMBER inline
#define BU_FN_HD static inline
#define BU_FN_BIG static
#define BU_TAB static const
#endif

#include "uastc_tables.inc"

namespace bu_uastc {

struct rgba8 { uint8_t c[4]; };

// 24-bit multiplies: every product below has operands far inside +-2^23, and the 24-bit multiplier is full rate where the 32-bit
// one (v_mul_lo_u32) is quarter rate. On the host they are ordinary multiplies.
#if defined(__HIPCC__)
BU_FN int imul24(int a, int b) { return __mul24(a, b); }
BU_FN uint32_t umul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#else
BU_FN int imul24(int a, int b) { return a * b; }
BU_FN uint32_t umul24(uint32_t a, uint32_t b) { return a * b; }
#endif
BU_FN_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
BU_FN float saturatef(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
BU_FN uint32_t astc_levels(uint32_t range) { return (1u + 2u * ku_bise[range * 3 + 1] + 4u * ku_bise[range * 3 + 2]) << ku_bise[range * 3]; }
// astc_interpolate_linear (bc7enc.cpp:177-183) == basist::astc_interpolate(..., srgb=false) (transcoder_uastc.h:77-93)
BU_FN uint32_t astc_lerp(uint32_t l, uint32_t h, uint32_t w) {
    l = (l << 8) | l;
    h = (h << 8) | h;
    return ((umul24(l, 64 - w) + umul24(h, w) + 32) >> 6) >> 8;
}
BU_FN const uint8_t* weight_set(uint32_t bits) { return ku_weights + ((1u << bits) - 2u); }

// ------------------------------------------------------------------------------------------------------------------
// Colour-cell fit: color_cell_compression(mode = 255, ASTC range, weights 1/1/1/1, perceptual off), bc7enc.cpp:1364-1762
//
// Register-resident formulation. A cell is "the texels of a 4x4 block selected by a 16-bit mask": the texels are 16 packed
// RGBA dwords, every loop over them is fully unrolled and predicated on the mask, selectors are kept at their raster position
// (byte-packed, 4 dwords) -- so nothing is gathered, nothing is dynamically indexed and a GPU lane needs no scratch memory.
// Skipping non-members does not change the order in which the members are visited, which is all the reference's running float
// sums depend on. The interpolated colours of a selector are computed (astc_lerp of the weight formula) instead of tabulated.
// ------------------------------------------------------------------------------------------------------------------

#if defined(__HIPCC__)
#define BU_UNROLL _Pragma("unroll")
#else
#define BU_UNROLL
#endif

// Interpolation weight of selector s for `bits` weight bits, computed instead of looked up: ASTC weight unquantisation (replicate
// to 6 bits, +1 above 32) reproduces ku_weights for every set UASTC uses (tests/test_uastc_core_host.py checks it against the table).
BU_FN uint32_t weight_of(uint32_t bits, uint32_t s) {
    uint32_t w = bits == 1 ? s * 63 : (bits == 2 ? s * 21 : (bits == 3 ? s * 9 : (bits == 4 ? (s << 2) | (s >> 2) : (s << 1) | (s >> 4))));
    return w + (w > 32 ? 1u : 0u);
}

struct cell_cfg {
    uint8_t wbits;      // 1..5 weight bits -> 2..32 interpolants
    uint8_t range;      // ASTC endpoint range
    uint8_t alpha;      // fit 4 channels
    uint8_t uber;       // bc7enc_compress_block_params::m_uber_level
    uint8_t ls_passes;  // m_least_squares_passes
    const float* ls_weights;  // the weight set's {w*w, (1-w)*w, (1-w)*(1-w), w} rows (ku_weights_ls + ((1 << wbits) - 2) * 4), or a staged copy of them (LDS)
};

struct sel16 { uint32_t w[4]; };  // 16 selectors, one byte each, texel i in byte i
BU_FN uint32_t sel_get(const sel16& s, int i) { return (s.w[i >> 2] >> (8 * (i & 3))) & 255u; }
BU_FN void sel_set(sel16& s, int i, uint32_t v) { const int sh = 8 * (i & 3); s.w[i >> 2] = (s.w[i >> 2] & ~(255u << sh)) | (v << sh); }

struct cell_fit {
    uint64_t err;
    uint8_t lo[4], hi[4];            // endpoint RANKS (position in ascending unquantised order)
    uint8_t astc_lo[4], astc_hi[4];  // the same endpoints as ASTC endpoint indices
    sel16 sel;                       // selectors at raster positions (members only)
};

BU_FN uint32_t dist_rgb(const uint8_t* a, const uint8_t* b) {
    const int dr = (int)a[0] - (int)b[0], dg = (int)a[1] - (int)b[1], db = (int)a[2] - (int)b[2];
    return (uint32_t)imul24(dr, dr) + (uint32_t)imul24(dg, dg) + (uint32_t)imul24(db, db);
}
BU_FN int px_comp(uint32_t p, int c) { return (int)((p >> (8 * c)) & 255u); }
template <bool GREY> BU_FN int pxc(uint32_t p, int c) { return (GREY && c < 3) ? (int)(p & 255u) : px_comp(p, c); }   // GREY: r = g = b by construction (cell_compress_grey)
BU_FN uint32_t pack_px(const uint8_t* c) { return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24); }

// evaluate_solution (bc7enc.cpp:822-1049), ASTC branch with the non-perceptual selector search.
// Interpolants are not tabulated: astc_lerp(L, H, w) = (64 * L257 + 32 + (H257 - L257) * w) >> 14 with X257 = X * 257, one multiply-add
// per channel, and the weight of a selector comes from the set's (multiplier, shift) pair -- see weight_of.
// FORCED: the selectors are given (m_pForce_selectors, bc7enc.cpp:885-898) and only the error is measured.
template <bool FORCED, bool GREY>
BU_FN uint64_t cell_eval(const uint32_t* px, uint32_t mask, const cell_cfg& cfg, const uint8_t* lo, const uint8_t* hi, cell_fit& best, const sel16* forced) {
    const uint32_t N = 1u << cfg.wbits;
    const uint8_t* SU = ku_sorted_unquant + cfg.range * 256;
    const int nc = cfg.alpha ? 4 : 3;
    int L[4], base[4], slope[4];
    for (int c = 0; c < 4; c++) {
        L[c] = SU[lo[c]];
        const int h = SU[hi[c]];
        base[c] = L[c] * 257 * 64 + 32;
        slope[c] = (h - L[c]) * 257;
    }
    const int dr = slope[0] / 257, dg = slope[1] / 257, db = slope[2] / 257, da = cfg.alpha ? slope[3] / 257 : 0;
#if !defined(__HIPCC__)
    if (GREY && (lo[0] != lo[1] || lo[1] != lo[2] || hi[0] != hi[1] || hi[1] != hi[2])) abort();   // (test builds) a grey cell's proposals are grey
#endif
    const float f = (float)N / ((float)(imul24(dr, dr) + imul24(dg, dg) + imul24(db, db) + imul24(da, da)) + .00000125f);
    // weight(s) = s * wmul + (s >> wshift), +1 above 32: {63, 21, 9} x s for 1..3 bits, bit replication for 4 and 5 bits
    const uint32_t wmul = cfg.wbits == 1 ? 63u : (cfg.wbits == 2 ? 21u : (cfg.wbits == 3 ? 9u : (cfg.wbits == 4 ? 4u : 2u)));
    const uint32_t wshift = cfg.wbits == 4 ? 2u : (cfg.wbits == 5 ? 4u : 31u);
    uint32_t total = 0;  // <= 16 * 4 * 255^2
    sel16 tmp = { { 0, 0, 0, 0 } };
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        if (!((mask >> i) & 1)) continue;
        const uint32_t p = px[i];
        if (FORCED) {
            const uint32_t sf = sel_get(*forced, i);
            uint32_t wf = umul24(sf, wmul) + (sf >> wshift);
            wf += (wf + 31) >> 6;
            BU_UNROLL
            for (int c = 0; c < 4; c++) {
                if (c >= nc) continue;
                const int d = ((base[c] + imul24(slope[c], (int)wf)) >> 14) - pxc<GREY>(p, c);
                total += (uint32_t)imul24(d, d);
            }
            sel_set(tmp, i, sf);
            continue;
        }
        // GREY: the three colour channels carry one value and -- every endpoint proposal being built channel by channel from the same numbers -- one pair of
        // endpoints (the host build checks it), so their three identical terms are one term times three
        int proj = GREY ? 3 * imul24(pxc<GREY>(p, 0) - L[0], dr) : imul24(pxc<GREY>(p, 0) - L[0], dr) + imul24(pxc<GREY>(p, 1) - L[1], dg) + imul24(pxc<GREY>(p, 2) - L[2], db);
        if (cfg.alpha) proj += imul24(pxc<GREY>(p, 3) - L[3], da);
        int s = (int)((float)proj * f + .5f);
        s = clampi(s, 1, (int)N - 1);
        uint32_t w1 = umul24((uint32_t)s, wmul) + ((uint32_t)s >> wshift), w0 = umul24((uint32_t)s - 1, wmul) + (((uint32_t)s - 1) >> wshift);
        w1 += (w1 + 31) >> 6;
        w0 += (w0 + 31) >> 6;
        uint32_t e0 = 0, e1 = 0;
        BU_UNROLL
        for (int c = 0; c < 4; c++) {  // fixed trip count: a run-time bound turns base[] / slope[] into scratch arrays on the GPU
            if (c >= nc || (GREY && (c == 1 || c == 2))) continue;
            const int v = pxc<GREY>(p, c);
            const int d0 = ((base[c] + imul24(slope[c], (int)w0)) >> 14) - v, d1 = ((base[c] + imul24(slope[c], (int)w1)) >> 14) - v;
            const uint32_t times = (GREY && c == 0) ? 3u : 1u;
            e0 += times * (uint32_t)imul24(d0, d0); e1 += times * (uint32_t)imul24(d1, d1);
        }
        if (e0 == e1) {
            if (s == 1) s = 0;  // prefer the non-interpolated endpoint
        } else if (e0 < e1) {
            e1 = e0;
            --s;
        }
        total += e1;
        sel_set(tmp, i, (uint32_t)s);
    }
    if (total < best.err) {
        best.err = total;
        for (int c = 0; c < 4; c++) { best.lo[c] = lo[c]; best.hi[c] = hi[c]; }
        best.sel = tmp;
    }
    return total;
}

// find_optimal_solution (bc7enc.cpp:1103-1282), ASTC branch, mode 255 degeneracy handling (:1051-1101)
template <bool FORCED, bool GREY>
BU_FN uint64_t cell_try(const uint32_t* px, uint32_t mask, const cell_cfg& cfg, const float* xl_in, const float* xh_in, cell_fit& best, const sel16* forced) {
    float xl[4], xh[4];
    for (int c = 0; c < 4; c++) { xl[c] = saturatef(xl_in[c]); xh[c] = saturatef(xh_in[c]); }
    const int top = (int)astc_levels(cfg.range) - 1;
    const uint8_t* NEAR = ku_nearest_rank + cfg.range * 256;
    uint8_t tmin[4], tmax[4];
    for (int c = 0; c < 4; c++) {
        tmin[c] = NEAR[clampi((int)(xl[c] * 255.0f + .5f), 0, 255)];
        tmax[c] = NEAR[clampi((int)(xh[c] * 255.0f + .5f), 0, 255)];
    }
    bool degenerate = false;
    for (int c = 0; c < 3; c++)
        if (tmin[c] == tmax[c] && fabsf(xl[c] - xh[c]) > 0.0f) degenerate = true;
    const uint32_t trials = degenerate ? 4 : 1;
    for (uint32_t t = 0; t < trials; t++) {
        uint8_t a[4], b[4];
        for (int c = 0; c < 4; c++) { a[c] = tmin[c]; b[c] = tmax[c]; }
        if (degenerate) {
            const uint32_t flags = t == 0 ? 1u : (t == 1 ? 0u : t);  // the reference tries 1, 0, 2, 3
            for (int c = 0; c < 3; c++)
                if (a[c] == b[c] && fabsf(xl[c] - xh[c]) > 0.000125f) {
                    if ((flags & 1) && a[c] > 0) a[c]--;
                    if ((flags & 2) && (int)b[c] < top) b[c]++;
                }
        }
        bool differs = best.err == UINT64_MAX;
        for (int c = 0; c < 4; c++) differs = differs || a[c] != best.lo[c] || b[c] != best.hi[c];
        if (differs) cell_eval<FORCED, GREY>(px, mask, cfg, a, b, best, forced);
    }
    return best.err;
}
// the winner's endpoint ranks as ASTC endpoint indices: once per cell, not once per proposal (eight table loads and a wait each time; staging the three tables in
// LDS, on the other hand, measured no gain: the fit is not waiting for them)
BU_FN void cell_astc_indices(const cell_cfg& cfg, cell_fit& best) {
    const uint8_t* SI = ku_sorted_index + cfg.range * 256;
    for (int c = 0; c < 4; c++) { best.astc_lo[c] = SI[best.lo[c]]; best.astc_hi[c] = SI[best.hi[c]]; }
}

// compute_least_squares_endpoints_rgb / _rgba (bc7enc.cpp:394-518) followed by the 1/255 scaling of its callers
template <bool GREY>
BU_FN void cell_least_squares(const uint32_t* px, uint32_t mask, const sel16& sel, const cell_cfg& cfg, float* xl, float* xh) {
    const float* WX = cfg.ls_weights;
    const int nc = cfg.alpha ? 4 : 3;
    double z00 = 0.0, z10 = 0.0, z11 = 0.0;
    double q00[4] = { 0, 0, 0, 0 }, t[4] = { 0, 0, 0, 0 };
    int lo_v[4] = { 255, 255, 255, 255 }, hi_v[4] = { 0, 0, 0, 0 };
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        if (!((mask >> i) & 1)) continue;
        const float* w4 = WX + sel_get(sel, i) * 4;
        z00 += w4[0];
        z10 += w4[1];
        z11 += w4[2];
        const float w = w4[3];
        BU_UNROLL
        for (int c = 0; c < 4; c++) {
            if (c >= nc || (GREY && (c == 1 || c == 2))) continue;   // GREY: green and blue are red's sums, copied below
            const int v = pxc<GREY>(px[i], c);
            q00[c] += w * (float)v;
            t[c] += (double)v;
            lo_v[c] = v < lo_v[c] ? v : lo_v[c];
            hi_v[c] = v > hi_v[c] ? v : hi_v[c];
        }
    }
    if (GREY) { q00[1] = q00[2] = q00[0]; t[1] = t[2] = t[0]; lo_v[1] = lo_v[2] = lo_v[0]; hi_v[1] = hi_v[2] = hi_v[0]; }
    const double z01 = z10;
    double det = z00 * z11 - z01 * z10;
    if (det != 0.0) det = 1.0 / det;
    const double iz00 = z11 * det, iz01 = -z01 * det, iz10 = -z10 * det, iz11 = z00 * det;
    BU_UNROLL
    for (int c = 0; c < 4; c++) {
        if (c >= nc) continue;
        const double q10 = t[c] - q00[c];
        xl[c] = (float)(iz00 * q00[c] + iz01 * q10);
        xh[c] = (float)(iz10 * q00[c] + iz11 * q10);
    }
    if (nc == 3) { xl[3] = 255.0f; xh[3] = 255.0f; }
    BU_UNROLL
    for (int c = 0; c < 4; c++)
        if (c < nc && (xl[c] < 0.0f || xh[c] > 255.0f) && lo_v[c] == hi_v[c]) { xl[c] = (float)lo_v[c]; xh[c] = (float)hi_v[c]; }
    for (int c = 0; c < 4; c++) { xl[c] = xl[c] * (1.0f / 255.0f); xh[c] = xh[c] * (1.0f / 255.0f); }
}

// The optimal single-colour encodings (bc7enc.cpp:605-820); which one applies is a function of (range, weights, alpha).
struct one_colour_kind { const uint8_t* table; uint8_t widx; uint8_t alpha_rank; uint8_t rgba; };
BU_FN bool one_colour_lookup(const cell_cfg& cfg, one_colour_kind& k) {
    const uint32_t N = 1u << cfg.wbits;
    if (cfg.range == 8 && N == 8 && !cfg.alpha) { k.table = ku_opt_4bit_3bit; k.widx = 2; k.alpha_rank = 0; k.rgba = 0; return true; }
    if (cfg.range == 7 && N == 4 && !cfg.alpha) { k.table = ku_opt_r7_2bit; k.widx = 1; k.alpha_rank = 0; k.rgba = 0; return true; }
    if (cfg.range == 8 && N == 4 && cfg.alpha) { k.table = ku_opt_4bit_2bit; k.widx = 1; k.alpha_rank = 0; k.rgba = 1; return true; }
    if (cfg.range == 13 && N == 4 && !cfg.alpha) { k.table = ku_opt_r13_2bit; k.widx = 1; k.alpha_rank = 47; k.rgba = 0; return true; }
    if (cfg.range == 11 && N == 32 && !cfg.alpha) { k.table = ku_opt_r11_5bit; k.widx = 13; k.alpha_rank = 31; k.rgba = 0; return true; }
    return false;
}
template <bool GREY>
BU_FN uint64_t one_colour_fit(const uint32_t* px, uint32_t mask, const cell_cfg& cfg, const one_colour_kind& k, const uint32_t* col, cell_fit& out) {
    const uint8_t* SU = ku_sorted_unquant + cfg.range * 256;
    const uint8_t* SI = ku_sorted_index + cfg.range * 256;
    const uint32_t w = weight_of(cfg.wbits, k.widx);
    int p[4] = { 0, 0, 0, 255 };
    for (int c = 0; c < 4; c++) {
        if (c < 3 || k.rgba) {
            out.lo[c] = k.table[col[c] * 2];
            out.hi[c] = k.table[col[c] * 2 + 1];
        } else {
            out.lo[c] = k.alpha_rank;
            out.hi[c] = k.alpha_rank;
        }
        out.astc_lo[c] = SI[out.lo[c]];
        out.astc_hi[c] = SI[out.hi[c]];
        if (c < 3 || k.rgba) p[c] = (int)astc_lerp(SU[out.lo[c]], SU[out.hi[c]], w);
    }
    uint32_t total = 0;
    out.sel.w[0] = out.sel.w[1] = out.sel.w[2] = out.sel.w[3] = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        if (!((mask >> i) & 1)) continue;
        sel_set(out.sel, i, k.widx);
        BU_UNROLL
        for (int c = 0; c < 4; c++) { if (c == 3 && !k.rgba) continue; const int d = p[c] - pxc<GREY>(px[i], c); total += (uint32_t)imul24(d, d); }
    }
    out.err = total;
    return total;
}

// color_cell_compression (bc7enc.cpp:1364-1762). The endpoint proposals (principal axis, least squares on the current selectors,
// the selector perturbations of the uber levels) are generated by one loop so that cell_try -- and with it the unrolled
// cell_eval -- is instantiated once.
// FORCED (m_pForce_selectors: uastc_rdo's mode-0 endpoint refit, uastc_enc.cpp:4046-4060): no single-colour shortcuts, the
// selectors never change, every proposal is scored on them.
template <bool FORCED, bool GREY = false>
BU_FN uint64_t cell_compress_t(const uint32_t* px, uint32_t mask, const cell_cfg& cfg, cell_fit& best, const sel16* forced) {
    best.err = UINT64_MAX;
    best.sel.w[0] = best.sel.w[1] = best.sel.w[2] = best.sel.w[3] = 0;
    const uint32_t n = (uint32_t)__builtin_popcount(mask);
    one_colour_kind kind;
    const bool has_kind = one_colour_lookup(cfg, kind);

    // sums, "all texels equal", mean
    float sum[4] = { 0, 0, 0, 0 };
    uint32_t first = 0;
    bool have_first = false, same = true;
    const uint32_t cmp_mask = (has_kind && kind.rgba) ? 0xFFFFFFFFu : 0x00FFFFFFu;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        if (!((mask >> i) & 1)) continue;
        for (int c = 0; c < 4; c++) sum[c] = sum[c] + (float)pxc<GREY>(px[i], c);
        if (!have_first) { first = px[i]; have_first = true; }
        else same = same && (((px[i] ^ first) & cmp_mask) == 0);
    }
    if (!FORCED && has_kind && same) {
        const uint32_t col[4] = { first & 255u, (first >> 8) & 255u, (first >> 16) & 255u, first >> 24 };
        return one_colour_fit<GREY>(px, mask, cfg, kind, col, best);
    }
    const float inv_n = 1.0f / (float)n;
    const float inv_n255 = 1.0f / ((float)n * 255.0f);
    float mean_s[4], mean[4];
    for (int c = 0; c < 4; c++) {
        mean_s[c] = sum[c] * inv_n;
        mean[c] = saturatef(sum[c] * inv_n255);
    }

    float axis[4];
    if (cfg.alpha) {
        // incremental 4-D PCA (bc7enc.cpp:1428-1448)
        axis[0] = axis[1] = axis[2] = axis[3] = 0.0f;
        bool started = false;
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            if (!((mask >> i) & 1)) continue;
            float col[4];
            for (int c = 0; c < 4; c++) col[c] = (float)pxc<GREY>(px[i], c) - mean_s[c];
            float v[4];
            for (int c = 0; c < 4; c++) v[c] = started ? axis[c] : col[c];
            started = true;
            float s = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            if (s != 0.0f) {
                s = 1.0f / sqrtf(s);
                v[0] *= s; v[1] *= s; v[2] *= s; v[3] *= s;
            }
            for (int c = 0; c < 4; c++) {
                const float k = col[c];
                axis[c] += (col[0] * k) * v[0] + (col[1] * k) * v[1] + (col[2] * k) * v[2] + (col[3] * k) * v[3];
            }
        }
        float s = axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2] + axis[3] * axis[3];
        if (s != 0.0f) {
            s = 1.0f / sqrtf(s);
            axis[0] *= s; axis[1] *= s; axis[2] *= s; axis[3] *= s;
        }
    } else {
        // covariance + 3 power iterations (bc7enc.cpp:1450-1489)
        float cov[6] = { 0, 0, 0, 0, 0, 0 };
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            if (!((mask >> i) & 1)) continue;
            const float r = (float)pxc<GREY>(px[i], 0) - mean_s[0];
            const float g = (float)pxc<GREY>(px[i], 1) - mean_s[1];
            const float b = (float)pxc<GREY>(px[i], 2) - mean_s[2];
            cov[0] += r * r; cov[1] += r * g; cov[2] += r * b; cov[3] += g * g; cov[4] += g * b; cov[5] += b * b;
        }
        float xr = .9f, xg = 1.0f, xb = .7f;
        for (uint32_t iter = 0; iter < 3; iter++) {
            float r = xr * cov[0] + xg * cov[1] + xb * cov[2];
            float g = xr * cov[1] + xg * cov[3] + xb * cov[4];
            float b = xr * cov[2] + xg * cov[4] + xb * cov[5];
            float m = fabsf(r) > fabsf(g) ? fabsf(r) : fabsf(g);
            m = m > fabsf(b) ? m : fabsf(b);
            if (m > 1e-10f) {
                m = 1.0f / m;
                r *= m; g *= m; b *= m;
            }
            xr = r; xg = g; xb = b;
        }
        float len = xr * xr + xg * xg + xb * xb;
        if (len < 1e-10f) {
            axis[0] = axis[1] = axis[2] = axis[3] = 0.0f;
        } else {
            len = 1.0f / sqrtf(len);
            axis[0] = xr * len; axis[1] = xg * len; axis[2] = xb * len; axis[3] = 0.0f;
        }
    }
    if (axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2] + axis[3] * axis[3] < .5f) {
        axis[0] = axis[1] = axis[2] = 1.0f;
        axis[3] = cfg.alpha ? 1.0f : 0.0f;
        float s = axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2] + axis[3] * axis[3];
        s = 1.0f / sqrtf(s);
        axis[0] *= s; axis[1] *= s; axis[2] *= s; axis[3] *= s;
    }

    float l = 1e+9f, h = -1e+9f;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        if (!((mask >> i) & 1)) continue;
        float q[4];
        for (int c = 0; c < 4; c++) q[c] = (float)pxc<GREY>(px[i], c) - mean_s[c];
        const float d = q[0] * axis[0] + q[1] * axis[1] + q[2] * axis[2] + q[3] * axis[3];
        l = l < d ? l : d;
        h = h > d ? h : d;
    }
    l *= (1.0f / 255.0f);
    h *= (1.0f / 255.0f);
    float xl[4], xh[4];
    for (int c = 0; c < 4; c++) {
        xl[c] = saturatef(mean[c] + axis[c] * l);
        xh[c] = saturatef(mean[c] + axis[c] * h);
    }
    if (xl[0] * 1.0f + xl[1] * 1.0f + xl[2] * 1.0f + xl[3] * 1.0f > xh[0] * 1.0f + xh[1] * 1.0f + xh[2] * 1.0f + xh[3] * 1.0f)
        for (int c = 0; c < 4; c++) { const float t = xl[c]; xl[c] = xh[c]; xh[c] = t; }

    // proposal schedule: 0 = principal axis; 1..ls = least squares on the best selectors so far; then (uber > 0) three perturbations
    // of the selectors as they stood after the least-squares passes (raise the minimum, lower the maximum, both; :1567-1642), then
    // (uber >= 2 and still a poor fit) rescalings of that selector set (:1644-1677)
    const uint32_t top = (1u << cfg.wbits) - 1;
    const uint32_t n_basic = 1 + cfg.ls_passes + (cfg.uber > 0 ? 3u : 0u);
    const int Q = cfg.uber >= 4 ? (int)cfg.uber - 2 : 1;
    const uint32_t n_scaled = cfg.uber >= 2 ? (uint32_t)((Q + 2) * (Q + 2) - 1) : 0;  // (ly, hy) pairs without (0, top)
    sel16 base = { { 0, 0, 0, 0 } };
    uint32_t smin = 256, smax = 0;
    for (uint32_t step = 0; step < n_basic + n_scaled; step++) {
        if (step >= 1) {
            sel16 trial = best.sel;
            if (step == 1u + cfg.ls_passes) {  // freeze the selector set the perturbations start from
                base = best.sel;
                BU_UNROLL
                for (int i = 0; i < 16; i++) {
                    if (!((mask >> i) & 1)) continue;
                    const uint32_t sv = sel_get(base, i);
                    smin = sv < smin ? sv : smin;
                    smax = sv > smax ? sv : smax;
                }
            }
            if (step > cfg.ls_passes && step < n_basic) {
                const uint32_t variant = step - 1 - cfg.ls_passes;
                BU_UNROLL
                for (int i = 0; i < 16; i++) {
                    if (!((mask >> i) & 1)) continue;
                    uint32_t sv = sel_get(base, i);
                    if (variant != 1 && sv == smin && sv < top) sv++;
                    else if (variant != 0 && sv == smax && sv > 0) sv--;
                    sel_set(trial, i, sv);
                }
            } else if (step >= n_basic) {
                if (step == n_basic && !(best.err > ((n * 56) >> 4))) break;
                // enumerate ly in [-Q, 1], hy in [top-1, top+Q], skipping (0, top), in the reference's order
                uint32_t k = step - n_basic;
                const uint32_t skip = (uint32_t)(Q * (Q + 2) + 1);  // index of (0, top) in the full grid
                if (k >= skip) k++;
                const int ly = -Q + (int)(k / (uint32_t)(Q + 2)), hy = (int)top - 1 + (int)(k % (uint32_t)(Q + 2));
                BU_UNROLL
                for (int i = 0; i < 16; i++) {
                    if (!((mask >> i) & 1)) continue;
                    float v = floorf((float)top * ((float)sel_get(base, i) - (float)ly) / ((float)hy - (float)ly) + .5f);
                    v = v < 0.0f ? 0.0f : (v > (float)top ? (float)top : v);
                    sel_set(trial, i, (uint32_t)v);
                }
            }
            cell_least_squares<GREY>(px, mask, trial, cfg, xl, xh);
        }
        if (!cell_try<FORCED, GREY>(px, mask, cfg, xl, xh, best, forced)) { cell_astc_indices(cfg, best); return 0; }
    }

    if (!FORCED && has_kind) {
        // the whole cell as its mean colour (bc7enc.cpp:1679-1755)
        uint32_t col[4];
        for (int c = 0; c < 4; c++) col[c] = (uint32_t)(int)(.5f + mean[c] * 255.0f);
        cell_fit avg;
        cell_astc_indices(cfg, best);
        if (one_colour_fit<GREY>(px, mask, cfg, kind, col, avg) < best.err) best = avg;
        return best.err;
    }
    cell_astc_indices(cfg, best);
    return best.err;
}
// The out-of-line instance takes the texels and the configuration and returns the fit BY VALUE: on the GPU they travel in registers, whereas
// pointer / reference parameters of a real call live in scratch memory (and every load from them waits out a memory round trip).
struct px16 { uint32_t v[16]; };
template <bool GREY>
BU_FN cell_fit cell_compress_rv(px16 px, uint32_t mask, cell_cfg cfg) {
    cell_fit f;
    cell_compress_t<false, GREY>(px.v, mask, cfg, f, nullptr);
    return f;
}
BU_FN uint64_t cell_compress(const uint32_t* px, uint32_t mask, const cell_cfg& cfg, cell_fit& best) {
    px16 p;
    BU_UNROLL
    for (int i = 0; i < 16; i++) p.v[i] = px[i];
    best = cell_compress_rv<false>(p, mask, cfg);
    return best.err;
}
// The same fit for texels whose red, green and blue are one value (the second plane of the dual-plane modes: one channel replicated to grey, uastc_enc.cpp
// e.g. :960-1010). Nothing is computed differently -- the channel accessor just returns the same register for the three colour channels, which lets the compiler
// see that every per-channel expression of the fit is computed three times over and keep one (same operations on the same values: the same results).
BU_FN uint64_t cell_compress_grey(const uint32_t* px, uint32_t mask, const cell_cfg& cfg, cell_fit& best) {
    px16 p;
    BU_UNROLL
    for (int i = 0; i < 16; i++) p.v[i] = px[i];
    best = cell_compress_rv<true>(p, mask, cfg);
    return best.err;
}

// color_cell_compression_est_astc (bc7enc.cpp:1764-1984) with unit channel weights over the texels selected by `mask`: bounding-box
// endpoints, threshold selectors. On the GPU the mask is wave-uniform when every lane ranks the same pattern, so the predicates cost
// nothing. Thresholds are non-decreasing (the interpolants move monotonically from the box's low corner to its high corner), so
// "last threshold not above d" is a count. The reference's early outs (:1880-1882 and its callers' loop conditions) only ever stop
// once the running error exceeds the best total so far, which cannot change which pattern wins, so the full error is computed.
template <int WBITS, int COMPS>
BU_FN uint32_t estimate_masked(const uint32_t* px, uint32_t mask) {
    constexpr int N = 1 << WBITS;
    int lo[4] = { 255, 255, 255, 255 }, hi[4] = { 0, 0, 0, 0 };
    BU_UNROLL
    for (int i = 0; i < 16; i++)
        if ((mask >> i) & 1) {
            BU_UNROLL
            for (int c = 0; c < COMPS; c++) {
                const int v = px_comp(px[i], c);
                lo[c] = v < lo[c] ? v : lo[c];
                hi[c] = v > hi[c] ? v : hi[c];
            }
        }
    int axis[4], col[N][4], thresh[N - 1];
    for (int c = 0; c < COMPS; c++) axis[c] = hi[c] - lo[c];
    int prev_dot = 0;
    BU_UNROLL
    for (int k = 0; k < N; k++) {
        int dot = 0;
        BU_UNROLL
        for (int c = 0; c < COMPS; c++) {
            col[k][c] = k == 0 ? lo[c] : (k == N - 1 ? hi[c] : (int)astc_lerp((uint32_t)lo[c], (uint32_t)hi[c], weight_of(WBITS, (uint32_t)k)));
            dot += imul24(col[k][c], axis[c]);
        }
        if (k) thresh[k - 1] = (prev_dot + dot + 1) >> 1;
        prev_dot = dot;
    }
    uint32_t total = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++)
        if ((mask >> i) & 1) {
            int v[4], d = 0;
            BU_UNROLL
            for (int c = 0; c < COMPS; c++) { v[c] = px_comp(px[i], c); d += imul24(axis[c], v[c]); }
            int pick[4];
            for (int c = 0; c < COMPS; c++) pick[c] = col[0][c];
            BU_UNROLL
            for (int k = 1; k < N; k++) {
                const bool at_least = d >= thresh[k - 1];
                for (int c = 0; c < COMPS; c++) pick[c] = at_least ? col[k][c] : pick[c];
            }
            for (int c = 0; c < COMPS; c++) { const int e = pick[c] - v[c]; total += (uint32_t)imul24(e, e); }
        }
    return total;
}

BU_FN uint32_t estimate_masked_any(uint32_t wbits, uint32_t comps, const uint32_t* px, uint32_t mask) {
    if (wbits == 3) return estimate_masked<3, 3>(px, mask);  // mode 2
    return comps == 4 ? estimate_masked<2, 4>(px, mask) : estimate_masked<2, 3>(px, mask);
}

// ------------------------------------------------------------------------------------------------------------------
// Candidates. One `cand` is one uastc_encode_results (uastc_enc.h:74-81) in a fixed 64-byte slot.
// ------------------------------------------------------------------------------------------------------------------

struct alignas(16) cand {
    uint64_t err;           // m_astc_err
    uint8_t mode, pattern, ccs, valid;
    uint8_t endpoints[18];  // ASTC endpoint indices: per subset, per channel {low, high}
    uint8_t weights[32];    // per texel (x2 for dual plane: plane0, plane1)
    uint8_t pad[2];
};

struct enc_cfg {            // encode_uastc's per-level settings (uastc_enc.cpp:3187-3263)
    uint32_t flags, mode_mask, eac_table_mask;
    uint8_t level, uber, ls_passes, estimate_partition, always_alpha, eac_mul_rad, bc1_hints, la_only_transparent;
};

enum { FLAG_FAVOR_UASTC = 8, FLAG_FAVOR_BC7 = 16, FLAG_ETC1_FASTER = 64, FLAG_ETC1_FASTEST = 128, FLAG_ETC1_NO_FLIP_INDIVIDUAL = 256, FLAG_FAVOR_SIMPLER = 512 };
enum { CLS_SOLID = 1, CLS_ALPHA = 2, CLS_LA = 4 };

BU_FN_HD void make_cfg(uint32_t flags, enc_cfg& c) {
    int level = (int)(flags & 7);
    level = clampi(level, 0, 4);
    c.flags = flags;
    c.level = (uint8_t)level;
    c.mode_mask = 0xFFFFFFFFu; c.uber = 6; c.estimate_partition = 0; c.always_alpha = 1; c.eac_mul_rad = 3; c.eac_table_mask = 0xFFFFFFFFu;
    c.ls_passes = 2; c.bc1_hints = 1; c.la_only_transparent = 0;
    if (level == 0) {
        c.mode_mask = (1u << 0) | (1u << 8) | (1u << 11) | (1u << 12) | (1u << 15);
        c.always_alpha = 0; c.eac_mul_rad = 0; c.eac_table_mask = (1u << 2) | (1u << 8) | (1u << 11) | (1u << 13);
        c.uber = 0; c.ls_passes = 1; c.bc1_hints = 0; c.estimate_partition = 1; c.la_only_transparent = 1;
    } else if (level == 1) {
        c.mode_mask = (1u << 0) | (1u << 4) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 11) | (1u << 12) | (1u << 15) | (1u << 17);
        c.always_alpha = 0; c.eac_mul_rad = 0; c.eac_table_mask = (1u << 2) | (1u << 8) | (1u << 11) | (1u << 13);
        c.uber = 0; c.ls_passes = 1; c.estimate_partition = 1;
    } else if (level == 2) {
        c.mode_mask = (1u << 0) | (1u << 1) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 10) | (1u << 11) | (1u << 12) | (1u << 13) |
                      (1u << 15) | (1u << 16) | (1u << 17);
        c.always_alpha = 0; c.eac_mul_rad = 1;
        c.eac_table_mask = (1u << 0) | (1u << 2) | (1u << 6) | (1u << 7) | (1u << 8) | (1u << 10) | (1u << 11) | (1u << 13);
        c.uber = 1; c.ls_passes = 1; c.estimate_partition = 1;
    } else if (level == 3) {
        c.always_alpha = 0; c.eac_mul_rad = 2; c.uber = 3; c.estimate_partition = 1;
    }
}

// solid / alpha / luminance-alpha classification (uastc_enc.cpp:3135-3152, 3275-3279)
BU_FN uint32_t classify(const rgba8* px, const enc_cfg& cfg) {
    bool solid = true, alpha = false, la = true;
    for (uint32_t i = 0; i < 16; i++) {
        if (px[i].c[3] < 255) alpha = true;
        for (uint32_t c = 0; c < 4; c++) if (px[i].c[c] != px[0].c[c]) solid = false;
        if (px[i].c[0] != px[i].c[1] || px[i].c[0] != px[i].c[2]) la = false;
    }
    if (solid) return CLS_SOLID | (alpha ? CLS_ALPHA : 0) | (la ? CLS_LA : 0);
    if (cfg.la_only_transparent && la && !alpha) la = false;
    return (alpha ? CLS_ALPHA : 0) | (la ? CLS_LA : 0);
}

BU_FN cell_cfg mode_cell_cfg(uint32_t mode, bool alpha, const enc_cfg& e, const float* staged_ls_weights = nullptr) {
    cell_cfg c;
    c.wbits = ku_mode_weight_bits[mode];
    c.range = ku_mode_endpoint_ranges[mode];
    c.alpha = alpha ? 1 : 0;
    c.uber = e.uber;
    c.ls_passes = e.ls_passes;
    c.ls_weights = staged_ls_weights ? staged_ls_weights : ku_weights_ls + ((1u << c.wbits) - 2u) * 4;
    return c;
}

// Order one subset's RGB(A) endpoints so that the low sum comes first (ASTC blue contraction must stay off); returns true
// when they were swapped, in which case the subset's weights have to be mirrored (e.g. uastc_enc.cpp:519-543).
BU_FN bool order_endpoints(uint8_t* ep, uint32_t comps, uint32_t range) {
    const uint8_t* UQ = ku_unquant + range * 256;
    const int s0 = UQ[ep[0]] + UQ[ep[2]] + UQ[ep[4]];
    const int s1 = UQ[ep[1]] + UQ[ep[3]] + UQ[ep[5]];
    if (s1 < s0) {
        for (uint32_t c = 0; c < comps; c++) { const uint8_t t = ep[c * 2]; ep[c * 2] = ep[c * 2 + 1]; ep[c * 2 + 1] = t; }
        return true;
    }
    return false;
}

BU_FN void cand_begin(cand& r, uint32_t mode, uint32_t pattern) {
    r.err = 0; r.mode = (uint8_t)mode; r.pattern = (uint8_t)pattern; r.ccs = 0; r.valid = 1;
    for (uint32_t i = 0; i < 18; i++) r.endpoints[i] = 0;
    for (uint32_t i = 0; i < 32; i++) r.weights[i] = 0;
    r.pad[0] = r.pad[1] = 0;
}

// the block's texels as packed dwords; luminance-alpha modes fit (l, 0, 0, a) so both channels weigh the same (uastc_enc.cpp:1611-1617, 2437-2440)
BU_FN void pack_block_px(const rgba8* px, bool la, uint32_t* out) {
    BU_UNROLL
    for (int i = 0; i < 16; i++) out[i] = la ? ((uint32_t)px[i].c[0] | ((uint32_t)px[i].c[3] << 24)) : pack_px(px[i].c);
}

// luminance/alpha error of a cell fitted on (l,0,0,a) texels, measured the way the LA modes do (uastc_enc.cpp:1708-1728, 2480-2494)
BU_FN uint64_t la_cell_error(const uint32_t* px, uint32_t mask, const cell_fit& f, uint32_t mode) {
    const uint8_t* UQ = ku_unquant + ku_mode_endpoint_ranges[mode] * 256;
    const uint32_t wbits = ku_mode_weight_bits[mode];
    const uint32_t ll = UQ[f.astc_lo[0]], lh = UQ[f.astc_hi[0]], al = UQ[f.astc_lo[3]], ah = UQ[f.astc_hi[3]];
    uint32_t total = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        if (!((mask >> i) & 1)) continue;
        const uint32_t w = weight_of(wbits, sel_get(f.sel, i));
        const int dl = px_comp(px[i], 0) - (int)astc_lerp(ll, lh, w), da = px_comp(px[i], 3) - (int)astc_lerp(al, ah, w);
        total += (uint32_t)(imul24(dl, dl) + imul24(da, da));
    }
    return total;
}

// modes 0, 1, 5, 18 (RGB), 10, 12, 14 (RGBA), 15 (LA): one subset, one plane
BU_FN_BIG void build_single(uint32_t mode, const rgba8* px_in, const enc_cfg& e, cand& r, const float* staged = nullptr) {
    cand_begin(r, mode, 0);
    const uint32_t comps = ku_mode_comps[mode];
    const uint32_t top = (1u << ku_mode_weight_bits[mode]) - 1;
    uint32_t px[16];
    pack_block_px(px_in, comps == 2, px);
    const cell_cfg cc = mode_cell_cfg(mode, comps != 3, e, staged);
    cell_fit f;
    const uint64_t err = cell_compress(px, 0xFFFFu, cc, f);
    if (comps == 2) {
        r.endpoints[0] = f.astc_lo[0]; r.endpoints[1] = f.astc_hi[0]; r.endpoints[2] = f.astc_lo[3]; r.endpoints[3] = f.astc_hi[3];
        BU_UNROLL
        for (int i = 0; i < 16; i++) r.weights[i] = (uint8_t)sel_get(f.sel, i);
        r.err = la_cell_error(px, 0xFFFFu, f, mode);
        return;
    }
    r.err = err;
    for (uint32_t c = 0; c < comps; c++) { r.endpoints[c * 2] = f.astc_lo[c]; r.endpoints[c * 2 + 1] = f.astc_hi[c]; }
    const bool inv = order_endpoints(r.endpoints, comps, cc.range);
    BU_UNROLL
    for (int i = 0; i < 16; i++) r.weights[i] = (uint8_t)(inv ? top - sel_get(f.sel, i) : sel_get(f.sel, i));
}

BU_FN uint32_t bc7_3_to_2(uint32_t p, uint32_t k) {  // bc7_convert_partition_index_3_to_2, transcoder.cpp:14303-14328
    const uint32_t g = k >> 1;
    uint32_t r = g == 0 ? (p <= 1 ? 0u : 1u) : (g == 1 ? (p == 0 ? 0u : 1u) : ((p == 0 || p == 2) ? 0u : 1u));
    return (k & 1) ? 1 - r : r;
}

// The partition a multi-subset mode splits its texels by while fitting: BC7 subsets for modes 2/3/4/9/16, ASTC subsets for 7.
BU_FN uint32_t fit_partition_bits(uint32_t mode, uint32_t pattern) {
    if (mode == 3) return ku_bc7_part3[ku_cp3_bc7[pattern]];
    if (mode == 7) {
        const uint32_t src = ku_bc7_part3[ku_cp7_bc7[pattern]], k = ku_cp7_k[pattern];
        uint32_t bits = 0;
        for (uint32_t i = 0; i < 16; i++) bits |= bc7_3_to_2((src >> (2 * i)) & 3, k) << (2 * i);
        return bits;
    }
    return ku_bc7_part2[ku_cp2_bc7[pattern]];
}
// 2-bit-per-texel partition -> the 16-bit texel masks of its subsets
BU_FN void partition_masks(uint32_t bits, uint32_t* m) {
    const uint32_t lo = bits & 0x55555555u, hi = (bits >> 1) & 0x55555555u;
    uint32_t m1 = 0, m2 = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) { m1 |= ((lo >> (2 * i)) & 1u) << i; m2 |= ((hi >> (2 * i)) & 1u) << i; }
    m[0] = 0xFFFFu & ~(m1 | m2); m[1] = m1; m[2] = m2;
}

// modes 2, 4, 7 (RGB, 2 subsets), 3 (RGB, 3 subsets), 9 (RGBA, 2 subsets), 16 (LA, 2 subsets) for one common pattern
BU_FN_BIG void build_multi(uint32_t mode, uint32_t pattern, const rgba8* px_in, const enc_cfg& e, cand& r, const float* staged = nullptr) {
    cand_begin(r, mode, pattern);
    const uint32_t comps = ku_mode_comps[mode], subsets = ku_mode_subsets[mode];
    const uint32_t top = (1u << ku_mode_weight_bits[mode]) - 1;
    uint32_t px[16];
    pack_block_px(px_in, comps == 2, px);
    const uint32_t part_bits = fit_partition_bits(mode, pattern);
    uint32_t masks[3];
    partition_masks(part_bits, masks);
    const cell_cfg cc = mode_cell_cfg(mode, comps != 3, e, staged);

    // which fitted subset feeds ASTC subset a, and the inverse
    uint32_t src_of_astc[3] = { 0, 1, 2 }, astc_of_src[3] = { 0, 1, 2 };
    if (mode == 3) {
        const uint32_t perm = ku_cp3_perm[pattern];
        for (uint32_t a = 0; a < 3; a++) src_of_astc[a] = ku_astc_to_bc7_perm[perm * 3 + a];
        for (uint32_t a = 0; a < 3; a++) astc_of_src[src_of_astc[a]] = a;
    } else if (mode != 7 && ku_cp2_invert[pattern]) {
        src_of_astc[0] = 1; src_of_astc[1] = 0;
        astc_of_src[0] = 1; astc_of_src[1] = 0;
    }
    uint64_t total = 0;
    for (uint32_t s = 0; s < subsets; s++) {  // one call site: the fit is large once unrolled
        cell_fit f;
        const uint64_t err = cell_compress(px, masks[s], cc, f);
        total += comps == 2 ? la_cell_error(px, masks[s], f, mode) : err;
        const uint32_t a = astc_of_src[s];
        bool inv = false;
        if (comps == 2) {
            uint8_t* ep = r.endpoints + a * 4;
            ep[0] = f.astc_lo[0]; ep[1] = f.astc_hi[0]; ep[2] = f.astc_lo[3]; ep[3] = f.astc_hi[3];
        } else {
            uint8_t* ep = r.endpoints + a * comps * 2;
            for (uint32_t c = 0; c < comps; c++) { ep[c * 2] = f.astc_lo[c]; ep[c * 2 + 1] = f.astc_hi[c]; }
            inv = order_endpoints(ep, comps, cc.range);
        }
        BU_UNROLL
        for (int i = 0; i < 16; i++)
            if ((masks[s] >> i) & 1) r.weights[i] = (uint8_t)(inv ? top - sel_get(f.sel, i) : sel_get(f.sel, i));
    }
    r.err = total;
}

// modes 6 (RGB), 11, 13 (RGBA), 17 (LA): one subset, two weight planes; `rot` is the channel on the second plane
BU_FN_BIG void build_dual(uint32_t mode, uint32_t rot, const rgba8* px, const enc_cfg& e, cand& r, const float* staged = nullptr) {
    cand_begin(r, mode, 0);
    const uint32_t top = (1u << ku_mode_weight_bits[mode]) - 1;
    const cell_cfg cc = mode_cell_cfg(mode, false, e, staged);
    cell_fit fit[2];  // [0] the three remaining channels, [1] the rotated channel replicated to grey
    uint64_t err[2];
    for (uint32_t plane = 0; plane < 2; plane++) {
        uint32_t t[16];
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            uint8_t c[4] = { px[i].c[0], px[i].c[1], px[i].c[2], px[i].c[3] };
            if (plane == 1) {
                const uint8_t v = mode == 17 ? c[3] : c[rot];
                c[0] = c[1] = c[2] = v; c[3] = 255;
            } else if (mode == 17) {
                c[1] = c[0]; c[2] = c[0]; c[3] = 255;
            } else if (mode == 6) {
                c[rot] = 255;
            } else {
                c[rot] = c[3]; c[3] = 255;
            }
            t[i] = pack_px(c);
        }
        err[plane] = (plane == 1 || mode == 17) ? cell_compress_grey(t, 0xFFFFu, cc, fit[plane]) : cell_compress(t, 0xFFFFu, cc, fit[plane]);   // (mode 17: both planes are one channel replicated)
    }
    const cell_fit &fm = fit[0], &fs = fit[1];
    r.err = mode == 17 ? err[0] / 3 + err[1] / 3 : err[0] + err[1] / 3;
    bool inv = false;
    if (mode == 17) {
        r.ccs = 3;
        r.endpoints[0] = fm.astc_lo[0]; r.endpoints[1] = fm.astc_hi[0]; r.endpoints[2] = fs.astc_lo[0]; r.endpoints[3] = fs.astc_hi[0];
    } else {
        r.ccs = (uint8_t)rot;
        for (uint32_t c = 0; c < 3; c++) {
            const cell_fit& f = rot == c ? fs : fm;
            r.endpoints[c * 2] = f.astc_lo[c]; r.endpoints[c * 2 + 1] = f.astc_hi[c];
        }
        if (mode != 6) {
            if (rot == 3) { r.endpoints[6] = fs.astc_lo[0]; r.endpoints[7] = fs.astc_hi[0]; }
            else { r.endpoints[6] = fm.astc_lo[rot]; r.endpoints[7] = fm.astc_hi[rot]; }
        }
        inv = order_endpoints(r.endpoints, mode == 6 ? 3 : 4, cc.range);
    }
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        r.weights[i * 2] = (uint8_t)(inv ? top - sel_get(fm.sel, i) : sel_get(fm.sel, i));
        r.weights[i * 2 + 1] = (uint8_t)(inv ? top - sel_get(fs.sel, i) : sel_get(fs.sel, i));
    }
}

// estimate_partition2 / estimate_partition2_list and the inlined variants of modes 3 and 7 (uastc_enc.cpp:638-671, 828-860, 1362-1405,
// 1542-1594): rank the common patterns by the cheap bounding-box estimate. Writes the `want` best patterns (ascending error,
// earlier pattern first among equals).
BU_FN void estimate_patterns(uint32_t mode, const rgba8* px_in, uint32_t want, uint32_t* out) {
    const uint32_t comps = ku_mode_comps[mode] == 3 ? 3 : 4, subsets = ku_mode_subsets[mode], wbits = ku_mode_weight_bits[mode];
    const uint32_t total = mode == 3 ? 11 : (mode == 7 ? 19 : 30);
    uint32_t px[16];
    pack_block_px(px_in, ku_mode_comps[mode] == 2, px);
    uint64_t best_err[8];
    for (uint32_t i = 0; i < 8; i++) { best_err[i] = UINT64_MAX; if (i < want) out[i] = 0; }
    for (uint32_t pat = 0; pat < total; pat++) {
        const uint32_t bits = mode == 3 ? ku_bc7_part3[ku_cp3_bc7[pat]] : (mode == 7 ? ku_pat7[pat] : ku_bc7_part2[ku_cp2_bc7[pat]]);
        uint32_t m[3];
        partition_masks(bits, m);
        uint64_t err = (uint64_t)estimate_masked_any(wbits, comps, px, m[0]) + estimate_masked_any(wbits, comps, px, m[1]);
        if (subsets == 3) err += estimate_masked_any(wbits, comps, px, m[2]);
        for (uint32_t i = 0; i < want; i++)
            if (err < best_err[i]) {
                for (uint32_t j = want - 1; j > i; --j) { out[j] = out[j - 1]; best_err[j] = best_err[j - 1]; }
                out[i] = pat;
                best_err[i] = err;
                break;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Slots: the fixed position of every (mode, variant) in the reference's result order (uastc_enc.cpp:3295-3362).
// ------------------------------------------------------------------------------------------------------------------

enum { MAX_SLOTS = 176 };

BU_FN_HD uint32_t mode_variants(uint32_t mode, const enc_cfg& e) {
    if (!((e.mode_mask >> mode) & 1) || mode == 8) return 0;
    switch (mode) {
    case 2: case 4: return e.estimate_partition ? 1 : 30;
    case 3: return e.estimate_partition ? 1 : 11;
    case 7: return e.estimate_partition ? 1 : 19;
    case 9: case 16: return e.estimate_partition ? 4 : 30;
    case 6: return 3;
    case 11: case 13: return 4;
    default: return 1;
    }
}
// call order of the mode generators; the first three only for luminance-alpha blocks, 0..18 only for opaque blocks, the rest
// only when alpha modes are tried
BU_FN_HD uint32_t mode_order(uint32_t i) {
    const uint8_t order[18] = { 15, 16, 17, 0, 1, 2, 3, 4, 5, 6, 7, 18, 9, 10, 11, 12, 13, 14 };
    return order[i];
}
BU_FN bool mode_applies(uint32_t mode, uint32_t cls, const enc_cfg& e) {
    const bool alpha = (cls & CLS_ALPHA) != 0, la = (cls & CLS_LA) != 0;
    if (mode >= 15 && mode <= 17) return la;
    if (mode <= 7 || mode == 18) return !alpha;
    return alpha || e.always_alpha;
}
BU_FN_HD uint32_t slot_base(uint32_t mode, const enc_cfg& e) {
    uint32_t base = 0;
    for (uint32_t i = 0; i < 18; i++) {
        const uint32_t m = mode_order(i);
        if (m == mode) return base;
        base += mode_variants(m, e);
    }
    return base;
}
BU_FN_HD uint32_t total_slots(const enc_cfg& e) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < 18; i++) n += mode_variants(mode_order(i), e);
    return n;
}

// Candidates `first_variant .. first_variant + n_variants` of one mode for one block -> out[0 .. n_variants).
// The unit of GPU work is (block, mode, variant range).
BU_FN void run_mode(uint32_t mode, const rgba8* px, const enc_cfg& e, cand* out, uint32_t first_variant, uint32_t n_variants, const float* staged = nullptr) {
    const uint32_t subsets = ku_mode_subsets[mode], planes = ku_mode_planes[mode];
    if (planes == 2) {
        for (uint32_t v = 0; v < n_variants; v++) build_dual(mode, first_variant + v, px, e, out[v], staged);
    } else if (subsets == 1) {
        build_single(mode, px, e, out[0], staged);
    } else if (e.estimate_partition) {
        uint32_t pats[8];
        const uint32_t want = (mode == 9 || mode == 16) ? 4 : 1;
        estimate_patterns(mode, px, want, pats);
        for (uint32_t v = 0; v < n_variants; v++) build_multi(mode, pats[first_variant + v], px, e, out[v], staged);
    } else {
        for (uint32_t v = 0; v < n_variants; v++) build_multi(mode, first_variant + v, px, e, out[v], staged);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Decoding a candidate: the UASTC/ASTC view (unpack_uastc, transcoder.cpp:15743-15879) and the BC7 view
// (transcode_uastc_to_bc7 :16034-16526 followed by encode_bc7_block + unpack_bc7, which cancel except for the decode itself)
// ------------------------------------------------------------------------------------------------------------------

BU_FN uint32_t astc_pattern_bits(uint32_t mode, uint32_t pattern) {
    const uint32_t subsets = ku_mode_subsets[mode];
    if (subsets < 2) return 0;
    return subsets == 3 ? ku_pat3[pattern] : (mode == 7 ? ku_pat7[pattern] : ku_pat2[pattern]);
}

BU_FN void decode_uastc(const cand& r, rgba8* out) {
    const uint32_t mode = r.mode, comps = ku_mode_comps[mode], planes = ku_mode_planes[mode];
    const uint8_t* UQ = ku_unquant + ku_mode_endpoint_ranges[mode] * 256;
    const uint8_t* W = weight_set(ku_mode_weight_bits[mode]);
    const uint32_t pat = astc_pattern_bits(mode, r.pattern);
    for (uint32_t i = 0; i < 16; i++) {
        const uint8_t* ep = r.endpoints + ((pat >> (2 * i)) & 3) * comps * 2;
        for (uint32_t c = 0; c < 4; c++) {
            const uint32_t w = W[(planes == 2 && c == r.ccs) ? r.weights[i * 2 + 1] : r.weights[i * planes]];
            if (comps == 2) {
                const uint32_t k = c == 3 ? 2 : 0;
                out[i].c[c] = (uint8_t)astc_lerp(UQ[ep[k]], UQ[ep[k + 1]], w);
            } else if (c < comps) {
                out[i].c[c] = (uint8_t)astc_lerp(UQ[ep[c * 2]], UQ[ep[c * 2 + 1]], w);
            } else {
                out[i].c[c] = 255;
            }
        }
    }
}

// determine_unique_pbits / determine_shared_pbits (transcoder.cpp:15897-16013): quantise float endpoints to comp_bits + p-bit
BU_FN void bc7_pbit_quantise(bool shared, uint32_t total_comps, uint32_t comp_bits, const float* xl, const float* xh, uint8_t* lo, uint8_t* hi, uint32_t* pbits) {
    const uint32_t total_bits = comp_bits + 1;
    const int iscalep = (1 << total_bits) - 1;
    const float scalep = (float)iscalep;
    float best0 = 1e+9f, best1 = 1e+9f;
    for (int p = 0; p < 2; p++) {
        uint8_t xmin[4], xmax[4], slo[4], shi[4];
        for (uint32_t c = 0; c < 4; c++) {
            xmin[c] = (uint8_t)clampi(((int)((xl[c] * scalep - (float)p) / 2.0f + .5f)) * 2 + p, p, iscalep - 1 + p);
            xmax[c] = (uint8_t)clampi(((int)((xh[c] * scalep - (float)p) / 2.0f + .5f)) * 2 + p, p, iscalep - 1 + p);
            slo[c] = (uint8_t)(xmin[c] << (8 - total_bits));
            slo[c] = (uint8_t)(slo[c] | (slo[c] >> total_bits));
            shi[c] = (uint8_t)(xmax[c] << (8 - total_bits));
            shi[c] = (uint8_t)(shi[c] | (shi[c] >> total_bits));
        }
        if (shared) {
            float err = 0;
            for (uint32_t i = 0; i < total_comps; i++) {
                const float a = ((float)slo[i] / 255.0f) - xl[i], b = ((float)shi[i] / 255.0f) - xh[i];
                err += a * a + b * b;
            }
            if (err < best0) {
                best0 = err;
                pbits[0] = (uint32_t)p; pbits[1] = (uint32_t)p;
                for (uint32_t c = 0; c < 4; c++) { lo[c] = xmin[c] >> 1; hi[c] = xmax[c] >> 1; }
            }
        } else {
            float err0 = 0, err1 = 0;
            for (uint32_t i = 0; i < total_comps; i++) {
                const float a = (float)slo[i] - xl[i] * 255.0f, b = (float)shi[i] - xh[i] * 255.0f;
                err0 += a * a;
                err1 += b * b;
            }
            if (err0 < best0) { best0 = err0; pbits[0] = (uint32_t)p; for (uint32_t c = 0; c < 4; c++) lo[c] = xmin[c] >> 1; }
            if (err1 < best1) { best1 = err1; pbits[1] = (uint32_t)p; for (uint32_t c = 0; c < 4; c++) hi[c] = xmax[c] >> 1; }
        }
    }
}

BU_FN uint32_t bc7_dequant_p(uint32_t v, uint32_t pbit, uint32_t bits) {  // bc7u::bc7_dequant with p-bit (transcoder.cpp:29770)
    const uint32_t total = bits + 1;
    v = (v << 1) | pbit;
    v <<= (8 - total);
    return v | (v >> total);
}
BU_FN uint32_t bc7_dequant(uint32_t v, uint32_t bits) { v <<= (8 - bits); return v | (v >> bits); }
BU_FN uint32_t bc7_lerp(uint32_t l, uint32_t h, uint32_t w) { return (umul24(l, 64 - w) + umul24(h, w) + 32) >> 6; }

BU_FN_BIG void decode_bc7(const cand& r, rgba8* out) {
    const uint32_t mode = r.mode, comps = ku_mode_comps[mode], range = ku_mode_endpoint_ranges[mode];
    const uint8_t* UQ = ku_unquant + range * 256;
    const uint8_t* ep = r.endpoints;
    switch (mode) {
    case 0: case 5: case 10: case 12: case 14: case 15: case 18: {  // -> BC7 mode 6
        float xl[4], xh[4];
        if (comps == 2) {
            xl[0] = xl[1] = xl[2] = (float)UQ[ep[0]] / 255.0f; xh[0] = xh[1] = xh[2] = (float)UQ[ep[1]] / 255.0f;
            xl[3] = (float)UQ[ep[2]] / 255.0f; xh[3] = (float)UQ[ep[3]] / 255.0f;
        } else {
            for (uint32_t c = 0; c < 4; c++) {
                xl[c] = c < comps ? (float)UQ[ep[c * 2]] / 255.0f : 1.0f;
                xh[c] = c < comps ? (float)UQ[ep[c * 2 + 1]] / 255.0f : 1.0f;
            }
        }
        uint8_t lo[4] = { 0, 0, 0, 0 }, hi[4] = { 0, 0, 0, 0 };
        uint32_t pb[2] = { 0, 0 };
        bc7_pbit_quantise(false, comps == 2 ? 4 : comps, 7, xl, xh, lo, hi, pb);
        if (comps == 3) { lo[3] = 127; hi[3] = 127; }
        const uint8_t five_to_four[32] = { 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 6, 7, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15 };
        const uint8_t three_to_four[8] = { 0, 2, 4, 6, 9, 11, 13, 15 };
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t w = r.weights[i];
            const uint32_t s = mode == 18 ? five_to_four[w] : (mode == 14 ? w * 5 : ((mode == 5 || mode == 12) ? three_to_four[w] : w));
            for (uint32_t c = 0; c < 4; c++) out[i].c[c] = (uint8_t)bc7_lerp((lo[c] << 1) | pb[0], (hi[c] << 1) | pb[1], ku_bc7_weights4[s]);
        }
        break;
    }
    case 1: case 4: case 2: case 9: case 16: {  // -> BC7 mode 3 (1, 4), mode 1 (2), mode 7 (9, 16): two subsets with p-bits
        const uint32_t ncomp = (mode == 9 || mode == 16) ? 4 : 3;
        const uint32_t bits = mode == 2 ? 6 : (ncomp == 4 ? 5 : 7);
        const uint32_t part = mode == 1 ? 0 : ku_bc7_part2[ku_cp2_bc7[r.pattern]];
        const bool invert = mode != 1 && ku_cp2_invert[r.pattern];
        uint8_t lo[2][4], hi[2][4];
        uint32_t pb[2][2];
        for (uint32_t s = 0; s < 2; s++) {
            float xl[4], xh[4];
            const uint8_t* e = ep + (mode == 1 ? 0 : s * comps * 2);
            if (comps == 2) {
                xl[0] = xl[1] = xl[2] = (float)UQ[e[0]] / 255.0f; xh[0] = xh[1] = xh[2] = (float)UQ[e[1]] / 255.0f;
                xl[3] = (float)UQ[e[2]] / 255.0f; xh[3] = (float)UQ[e[3]] / 255.0f;
            } else {
                for (uint32_t c = 0; c < 4; c++) {
                    // mode 1 stores 8-bit endpoints directly, mode 2 4-bit ones replicated to 8 bits
                    xl[c] = c < comps ? (float)UQ[e[c * 2]] / 255.0f : 1.0f;
                    xh[c] = c < comps ? (float)UQ[e[c * 2 + 1]] / 255.0f : 1.0f;
                }
            }
            const uint32_t d = invert ? 1 - s : s;
            for (uint32_t c = 0; c < 4; c++) { lo[d][c] = 0; hi[d][c] = 0; }
            pb[d][0] = pb[d][1] = 0;
            bc7_pbit_quantise(mode == 2, ncomp, bits, xl, xh, lo[d], hi[d], pb[d]);
        }
        const uint8_t* W = weight_set(mode == 2 ? 3 : 2);
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t s = (part >> (2 * i)) & 3;
            for (uint32_t c = 0; c < 4; c++)
                out[i].c[c] = c < ncomp ? (uint8_t)bc7_lerp(bc7_dequant_p(lo[s][c], pb[s][0], bits), bc7_dequant_p(hi[s][c], pb[s][mode == 2 ? 0 : 1], bits), W[r.weights[i]]) : 255;
        }
        break;
    }
    case 3: case 7: {  // -> BC7 mode 2: three subsets, 5-bit endpoints, no p-bits
        uint8_t lo[3][3], hi[3][3];
        uint32_t part;
        if (mode == 3) {
            part = ku_bc7_part3[ku_cp3_bc7[r.pattern]];
            const uint32_t perm = ku_cp3_perm[r.pattern];
            for (uint32_t s = 0; s < 3; s++) {
                const uint32_t d = ku_astc_to_bc7_perm[perm * 3 + s];
                for (uint32_t c = 0; c < 3; c++) {
                    lo[d][c] = (uint8_t)((UQ[ep[c * 2 + s * 6]] * 31 + 127) / 255);
                    hi[d][c] = (uint8_t)((UQ[ep[c * 2 + 1 + s * 6]] * 31 + 127) / 255);
                }
            }
        } else {
            part = ku_bc7_part3[ku_cp7_bc7[r.pattern]];
            for (uint32_t d = 0; d < 3; d++) {
                const uint32_t s = bc7_3_to_2(d, ku_cp7_k[r.pattern]);
                for (uint32_t c = 0; c < 3; c++) {
                    lo[d][c] = (uint8_t)((UQ[ep[c * 2 + s * 6]] * 31 + 127) / 255);
                    hi[d][c] = (uint8_t)((UQ[ep[c * 2 + 1 + s * 6]] * 31 + 127) / 255);
                }
            }
        }
        const uint8_t* W = weight_set(2);
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t s = (part >> (2 * i)) & 3;
            for (uint32_t c = 0; c < 3; c++) out[i].c[c] = (uint8_t)bc7_lerp(bc7_dequant(lo[s][c], 5), bc7_dequant(hi[s][c], 5), W[r.weights[i]]);
            out[i].c[3] = 255;
        }
        break;
    }
    default: {  // 6, 11, 13, 17 -> BC7 mode 5: 7-bit colour + 8-bit alpha, separate index planes, channel rotation
        uint8_t lo[4], hi[4];
        const uint32_t rot = (r.ccs + 1u) & 3u;
        if (comps == 2) {
            lo[0] = lo[1] = lo[2] = (uint8_t)((UQ[ep[0]] * 127 + 127) / 255);
            hi[0] = hi[1] = hi[2] = (uint8_t)((UQ[ep[1]] * 127 + 127) / 255);
            lo[3] = UQ[ep[2]]; hi[3] = UQ[ep[3]];
        } else {
            for (uint32_t ac = 0; ac < 4; ac++) {
                const uint32_t bc = ac == r.ccs ? 3 : (ac == 3 ? r.ccs : ac);
                uint32_t l = 255, h = 255;
                if (ac < comps) { l = UQ[ep[ac * 2]]; h = UQ[ep[ac * 2 + 1]]; }
                if (bc < 3) { l = (l * 127 + 127) / 255; h = (h * 127 + 127) / 255; }
                lo[bc] = (uint8_t)l; hi[bc] = (uint8_t)h;
            }
        }
        const uint8_t* W = weight_set(2);
        for (uint32_t i = 0; i < 16; i++) {
            uint32_t cs = r.weights[i * 2], as = r.weights[i * 2 + 1];
            if (mode == 13) { cs = cs ? 3 : 0; as = as ? 3 : 0; }
            uint8_t v[4];
            for (uint32_t c = 0; c < 3; c++) v[c] = (uint8_t)bc7_lerp(bc7_dequant(lo[c], 7), bc7_dequant(hi[c], 7), W[cs]);
            v[3] = (uint8_t)bc7_lerp(lo[3], hi[3], W[as]);
            if (rot >= 1) { const uint8_t t = v[3]; v[3] = v[rot - 1]; v[rot - 1] = t; }
            for (uint32_t c = 0; c < 4; c++) out[i].c[c] = v[c];
        }
        break;
    }
    }
}

struct block_err { uint64_t rgb, rgba, la; };
BU_FN block_err block_error(const rgba8* a, const rgba8* b) {  // compute_block_error, uastc_enc.cpp:2510-2533
    uint64_t e[4] = { 0, 0, 0, 0 };
    for (uint32_t i = 0; i < 16; i++)
        for (uint32_t c = 0; c < 4; c++) { const int d = (int)a[i].c[c] - (int)b[i].c[c]; e[c] += (uint32_t)imul24(d, d); }
    block_err r;
    r.la = e[0] + e[3]; r.rgb = e[0] + e[1] + e[2]; r.rgba = r.rgb + e[3];
    return r;
}

// ---- The two decodes of a candidate and their errors against the source texels in one pass, without the decoded images: decode_uastc / decode_bc7 + block_error
// fused per texel, every loop unrolled, every table of the candidate (endpoints per subset, p-bits) held as scalars and picked by compare-and-select. The general
// decoders above write byte arrays that are indexed by run-time values -- on the GPU those live in scratch memory, and the scoring kernel spent its time waiting for
// them (1.9 ms for ~600 arithmetic instructions per candidate). Same integers, same order of operations per texel and channel.
struct chan_err { uint32_t e[4]; };   // per channel: sum over the texels of the squared difference (<= 16 * 255^2)
BU_FN void chan_err_add(chan_err& a, uint32_t px, uint32_t r, uint32_t g, uint32_t b, uint32_t al) {
    const int d0 = px_comp(px, 0) - (int)r, d1 = px_comp(px, 1) - (int)g, d2 = px_comp(px, 2) - (int)b, d3 = px_comp(px, 3) - (int)al;
    a.e[0] += (uint32_t)imul24(d0, d0); a.e[1] += (uint32_t)imul24(d1, d1); a.e[2] += (uint32_t)imul24(d2, d2); a.e[3] += (uint32_t)imul24(d3, d3);
}
BU_FN block_err chan_err_totals(const chan_err& a) {
    block_err r;
    r.la = (uint64_t)a.e[0] + a.e[3]; r.rgb = (uint64_t)a.e[0] + a.e[1] + a.e[2]; r.rgba = r.rgb + a.e[3];
    return r;
}
BU_FN uint32_t pick3(uint32_t s, uint32_t v0, uint32_t v1, uint32_t v2) { return s == 0 ? v0 : (s == 1 ? v1 : v2); }

// decode_uastc + block_error
template <int COMPS, int PLANES, int SUBSETS>
BU_FN void uastc_errors_t(const cand& r, const uint32_t* px, chan_err& out) {
    const uint32_t mode = r.mode, wbits = ku_mode_weight_bits[mode];
    const uint8_t* UQ = ku_unquant + ku_mode_endpoint_ranges[mode] * 256;
    const uint32_t pat = SUBSETS > 1 ? astc_pattern_bits(mode, r.pattern) : 0u;
    uint32_t L[3][4], H[3][4];
    BU_UNROLL
    for (int s = 0; s < 3; s++)
        BU_UNROLL
        for (int c = 0; c < 4; c++) {
            L[s][c] = 255; H[s][c] = 255;
            if (s < SUBSETS) {
                if (COMPS == 2) { const int k = c == 3 ? 2 : 0; L[s][c] = UQ[r.endpoints[s * 4 + k]]; H[s][c] = UQ[r.endpoints[s * 4 + k + 1]]; }
                else if (c < COMPS) { L[s][c] = UQ[r.endpoints[s * COMPS * 2 + c * 2]]; H[s][c] = UQ[r.endpoints[s * COMPS * 2 + c * 2 + 1]]; }
            }
        }
    out.e[0] = out.e[1] = out.e[2] = out.e[3] = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        const uint32_t s = SUBSETS > 1 ? (pat >> (2 * i)) & 3u : 0u;
        const uint32_t w0 = weight_of(wbits, r.weights[i * PLANES]), w1 = PLANES == 2 ? weight_of(wbits, r.weights[i * PLANES + (PLANES - 1)]) : w0;
        uint32_t v[4];
        BU_UNROLL
        for (int c = 0; c < 4; c++) {
            if (COMPS != 2 && c >= COMPS) { v[c] = 255; continue; }
            const uint32_t l = SUBSETS == 1 ? L[0][c] : pick3(s, L[0][c], L[1][c], L[2][c]), h = SUBSETS == 1 ? H[0][c] : pick3(s, H[0][c], H[1][c], H[2][c]);
            const uint32_t w = (PLANES == 2 && (uint32_t)c == r.ccs) ? w1 : w0;
            v[c] = astc_lerp(l, h, w);
        }
        chan_err_add(out, px[i], v[0], v[1], v[2], v[3]);
    }
}
BU_FN void uastc_errors(const cand& r, const uint32_t* px, chan_err& out) {
    const uint32_t mode = r.mode, comps = ku_mode_comps[mode], planes = ku_mode_planes[mode], subsets = ku_mode_subsets[mode];
    if (planes == 2) {
        if (comps == 3) uastc_errors_t<3, 2, 1>(r, px, out); else if (comps == 4) uastc_errors_t<4, 2, 1>(r, px, out); else uastc_errors_t<2, 2, 1>(r, px, out);
    } else if (comps == 3) {
        if (subsets == 1) uastc_errors_t<3, 1, 1>(r, px, out); else if (subsets == 2) uastc_errors_t<3, 1, 2>(r, px, out); else uastc_errors_t<3, 1, 3>(r, px, out);
    } else if (comps == 4) {
        if (subsets == 1) uastc_errors_t<4, 1, 1>(r, px, out); else uastc_errors_t<4, 1, 2>(r, px, out);
    } else {
        if (subsets == 1) uastc_errors_t<2, 1, 1>(r, px, out); else uastc_errors_t<2, 1, 2>(r, px, out);
    }
}

// decode_bc7 + block_error
BU_FN void bc7_unquant_floats(const cand& r, uint32_t first, uint32_t comps, const uint8_t* UQ, float* xl, float* xh) {   // the float endpoints of the subset whose indices start at `first`
    if (comps == 2) {
        xl[0] = xl[1] = xl[2] = (float)UQ[r.endpoints[first]] / 255.0f; xh[0] = xh[1] = xh[2] = (float)UQ[r.endpoints[first + 1]] / 255.0f;
        xl[3] = (float)UQ[r.endpoints[first + 2]] / 255.0f; xh[3] = (float)UQ[r.endpoints[first + 3]] / 255.0f;
    } else {
        BU_UNROLL
        for (int c = 0; c < 4; c++) {
            xl[c] = (uint32_t)c < comps ? (float)UQ[r.endpoints[first + c * 2]] / 255.0f : 1.0f;
            xh[c] = (uint32_t)c < comps ? (float)UQ[r.endpoints[first + c * 2 + 1]] / 255.0f : 1.0f;
        }
    }
}
BU_FN void bc7_errors(const cand& r, const uint32_t* px, chan_err& out) {
    const uint32_t mode = r.mode, comps = ku_mode_comps[mode];
    const uint8_t* UQ = ku_unquant + ku_mode_endpoint_ranges[mode] * 256;
    out.e[0] = out.e[1] = out.e[2] = out.e[3] = 0;
    switch (mode) {
    case 0: case 5: case 10: case 12: case 14: case 15: case 18: {  // -> BC7 mode 6
        float xl[4], xh[4];
        bc7_unquant_floats(r, 0, comps, UQ, xl, xh);
        uint8_t lo[4] = { 0, 0, 0, 0 }, hi[4] = { 0, 0, 0, 0 };
        uint32_t pb[2] = { 0, 0 };
        bc7_pbit_quantise(false, comps == 2 ? 4 : comps, 7, xl, xh, lo, hi, pb);
        if (comps == 3) { lo[3] = 127; hi[3] = 127; }
        uint32_t l[4], h[4];
        for (int c = 0; c < 4; c++) { l[c] = ((uint32_t)lo[c] << 1) | pb[0]; h[c] = ((uint32_t)hi[c] << 1) | pb[1]; }
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            const uint32_t w = r.weights[i];
            // five_to_four / three_to_four of decode_bc7 and the BC7 4-bit weights, as literals (a nibble / a byte per entry)
            const uint32_t five = (uint32_t)((w < 16 ? 0x7666554433221100ull : 0xFFEEDDCCBBAA9998ull) >> (4 * (w & 15))) & 15u;
            const uint32_t three = (0xFDB96420u >> (4 * (w & 7))) & 15u;
            const uint32_t sx = mode == 18 ? five : (mode == 14 ? w * 5 : ((mode == 5 || mode == 12) ? three : w));
            const uint32_t wt = (uint32_t)((sx < 8 ? 0x1E1A15110D090400ull : 0x403C37332F2B2622ull) >> (8 * (sx & 7))) & 255u;
            chan_err_add(out, px[i], bc7_lerp(l[0], h[0], wt), bc7_lerp(l[1], h[1], wt), bc7_lerp(l[2], h[2], wt), bc7_lerp(l[3], h[3], wt));
        }
        break;
    }
    case 1: case 4: case 2: case 9: case 16: {  // -> BC7 mode 3 (1, 4), mode 1 (2), mode 7 (9, 16): two subsets with p-bits
        const uint32_t ncomp = (mode == 9 || mode == 16) ? 4 : 3;
        const uint32_t bits = mode == 2 ? 6 : (ncomp == 4 ? 5 : 7);
        const uint32_t part = mode == 1 ? 0 : ku_bc7_part2[ku_cp2_bc7[r.pattern]];
        const bool invert = mode != 1 && ku_cp2_invert[r.pattern];
        uint32_t l[2][4], h[2][4];   // dequantised, by BC7 subset
        BU_UNROLL
        for (int s = 0; s < 2; s++) {
            float xl[4], xh[4];
            bc7_unquant_floats(r, mode == 1 ? 0u : (uint32_t)s * comps * 2, comps, UQ, xl, xh);
            uint8_t lo[4] = { 0, 0, 0, 0 }, hi[4] = { 0, 0, 0, 0 };
            uint32_t pb[2] = { 0, 0 };
            bc7_pbit_quantise(mode == 2, ncomp, bits, xl, xh, lo, hi, pb);
            uint32_t dl[4], dh[4];
            for (int c = 0; c < 4; c++) { dl[c] = bc7_dequant_p(lo[c], pb[0], bits); dh[c] = bc7_dequant_p(hi[c], pb[mode == 2 ? 0 : 1], bits); }
            // ASTC subset s is BC7 subset 1 - s when the pattern is inverted
            for (int c = 0; c < 4; c++) {
                if (s == 0) { l[0][c] = dl[c]; h[0][c] = dh[c]; l[1][c] = dl[c]; h[1][c] = dh[c]; }   // (both, so that every entry is defined; overwritten below)
                else if (invert) { l[0][c] = dl[c]; h[0][c] = dh[c]; }
                else { l[1][c] = dl[c]; h[1][c] = dh[c]; }
            }
            if (s == 0 && invert) { /* subset 0's values belong at index 1: they are there already (copied to both) */ }
        }
        const uint32_t wb = mode == 2 ? 3u : 2u;
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            const uint32_t s = (part >> (2 * i)) & 3u, wt = weight_of(wb, r.weights[i]);
            uint32_t v[4];
            for (int c = 0; c < 4; c++) v[c] = (uint32_t)c < ncomp ? bc7_lerp(s ? l[1][c] : l[0][c], s ? h[1][c] : h[0][c], wt) : 255u;
            chan_err_add(out, px[i], v[0], v[1], v[2], v[3]);
        }
        break;
    }
    case 3: case 7: {  // -> BC7 mode 2: three subsets, 5-bit endpoints, no p-bits
        uint32_t l[3][3], h[3][3];
        uint32_t part;
        if (mode == 3) {
            part = ku_bc7_part3[ku_cp3_bc7[r.pattern]];
            const uint32_t perm = ku_cp3_perm[r.pattern];
            BU_UNROLL
            for (int d = 0; d < 3; d++) {
                // the ASTC subset that lands in BC7 subset d
                uint32_t sa = 0;
                for (uint32_t s = 0; s < 3; s++) if (ku_astc_to_bc7_perm[perm * 3 + s] == (uint32_t)d) sa = s;
                BU_UNROLL
                for (int c = 0; c < 3; c++) {
                    const uint32_t lv = pick3(sa, r.endpoints[c * 2], r.endpoints[c * 2 + 6], r.endpoints[c * 2 + 12]), hv = pick3(sa, r.endpoints[c * 2 + 1], r.endpoints[c * 2 + 7], r.endpoints[c * 2 + 13]);
                    l[d][c] = bc7_dequant((UQ[lv] * 31 + 127) / 255, 5); h[d][c] = bc7_dequant((UQ[hv] * 31 + 127) / 255, 5);
                }
            }
        } else {
            part = ku_bc7_part3[ku_cp7_bc7[r.pattern]];
            BU_UNROLL
            for (int d = 0; d < 3; d++) {
                const uint32_t sa = bc7_3_to_2((uint32_t)d, ku_cp7_k[r.pattern]);
                BU_UNROLL
                for (int c = 0; c < 3; c++) {
                    const uint32_t lv = sa ? r.endpoints[c * 2 + 6] : r.endpoints[c * 2], hv = sa ? r.endpoints[c * 2 + 7] : r.endpoints[c * 2 + 1];
                    l[d][c] = bc7_dequant((UQ[lv] * 31 + 127) / 255, 5); h[d][c] = bc7_dequant((UQ[hv] * 31 + 127) / 255, 5);
                }
            }
        }
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            const uint32_t s = (part >> (2 * i)) & 3u, wt = weight_of(2, r.weights[i]);
            chan_err_add(out, px[i], bc7_lerp(pick3(s, l[0][0], l[1][0], l[2][0]), pick3(s, h[0][0], h[1][0], h[2][0]), wt),
                         bc7_lerp(pick3(s, l[0][1], l[1][1], l[2][1]), pick3(s, h[0][1], h[1][1], h[2][1]), wt),
                         bc7_lerp(pick3(s, l[0][2], l[1][2], l[2][2]), pick3(s, h[0][2], h[1][2], h[2][2]), wt), 255u);
        }
        break;
    }
    default: {  // 6, 11, 13, 17 -> BC7 mode 5: 7-bit colour + 8-bit alpha, separate index planes, channel rotation
        uint32_t lo[4] = { 0, 0, 0, 0 }, hi[4] = { 0, 0, 0, 0 };   // by BC7 channel: 0..2 colour (7 bits), 3 the rotated-out channel (8 bits)
        const uint32_t rot = (r.ccs + 1u) & 3u;
        if (comps == 2) {
            lo[0] = lo[1] = lo[2] = (UQ[r.endpoints[0]] * 127u + 127u) / 255u;
            hi[0] = hi[1] = hi[2] = (UQ[r.endpoints[1]] * 127u + 127u) / 255u;
            lo[3] = UQ[r.endpoints[2]]; hi[3] = UQ[r.endpoints[3]];
        } else {
            BU_UNROLL
            for (int ac = 0; ac < 4; ac++) {
                const uint32_t bc = (uint32_t)ac == r.ccs ? 3u : (ac == 3 ? r.ccs : (uint32_t)ac);
                uint32_t lv = 255, hv = 255;
                if ((uint32_t)ac < comps) { lv = UQ[r.endpoints[ac * 2]]; hv = UQ[r.endpoints[ac * 2 + 1]]; }
                if (bc < 3) { lv = (lv * 127 + 127) / 255; hv = (hv * 127 + 127) / 255; }
                BU_UNROLL
                for (int k = 0; k < 4; k++) if (bc == (uint32_t)k) { lo[k] = lv; hi[k] = hv; }
            }
        }
        uint32_t cl[3], ch[3];
        for (int c = 0; c < 3; c++) { cl[c] = bc7_dequant(lo[c], 7); ch[c] = bc7_dequant(hi[c], 7); }
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            uint32_t cs = r.weights[i * 2], as = r.weights[i * 2 + 1];
            if (mode == 13) { cs = cs ? 3 : 0; as = as ? 3 : 0; }
            const uint32_t wc = weight_of(2, cs), wa = weight_of(2, as);
            const uint32_t v0 = bc7_lerp(cl[0], ch[0], wc), v1 = bc7_lerp(cl[1], ch[1], wc), v2 = bc7_lerp(cl[2], ch[2], wc), v3 = bc7_lerp(lo[3], hi[3], wa);
            // undo the rotation: the alpha plane's value goes to channel rot - 1, that channel's to alpha
            chan_err_add(out, px[i], rot == 1 ? v3 : v0, rot == 2 ? v3 : v1, rot == 3 ? v3 : v2, rot == 0 ? v3 : (rot == 1 ? v0 : (rot == 2 ? v1 : v2)));
        }
        break;
    }
    }
}

// Per-candidate scores used by the final choice (uastc_enc.cpp:3403-3488)
struct cand_score { uint64_t overall; float uastc_rms; };
BU_FN cand_score score_candidate(const cand& r, const rgba8* px, uint32_t cls, const enc_cfg& e) {
    uint32_t packed[16];
    BU_UNROLL
    for (int i = 0; i < 16; i++) packed[i] = pack_px(px[i].c);
    chan_err cu, cb;
    uastc_errors(r, packed, cu);
    bc7_errors(r, packed, cb);
    const block_err eu = chan_err_totals(cu), eb = chan_err_totals(cb);
    const bool favor_uastc = (e.flags & FLAG_FAVOR_UASTC) != 0, favor_bc7 = !favor_uastc && (e.flags & FLAG_FAVOR_BC7) != 0;
    const uint32_t bc7_w = favor_bc7 ? 100 : (favor_uastc ? 0 : 50), uastc_w = favor_bc7 ? 0 : 100;
    const uint64_t u = (cls & CLS_LA) ? eu.la : ((cls & CLS_ALPHA) ? eu.rgba : eu.rgb);
    const uint64_t b = (cls & CLS_LA) ? eb.la : ((cls & CLS_ALPHA) ? eb.rgba : eb.rgb);
    cand_score s;
    s.overall = (b * bc7_w) / 100 + (u * uastc_w) / 100;
    s.uastc_rms = sqrtf((float)u);
    return s;
}

BU_FN float mode_bias(uint32_t mode) { return (mode == 0 || mode == 10) ? .8f : 1.0f; }  // get_uastc_mode_weight, :3110-3124

// Choose among the valid slots, in slot order (uastc_enc.cpp:3391-3548). `V` supplies valid(i), overall(i), rms(i), mode(i).
template <class V>
BU_FN uint32_t choose_candidate(const V& v, uint32_t n_slots, const enc_cfg& e) {
    uint32_t count = 0, first = 0;
    for (uint32_t i = 0; i < n_slots; i++)
        if (v.valid(i)) { if (!count) first = i; count++; }
    if (count <= 1) return first;
    double best_rms = 1e+20f;
    for (uint32_t i = 0; i < n_slots; i++) {
        if (!v.valid(i)) continue;
        if (!v.overall(i)) return i;
        const float rms = v.rms(i);
        if ((double)rms < best_rms) best_rms = rms;
    }
    const bool favor_uastc = (e.flags & FLAG_FAVOR_UASTC) != 0, favor_bc7 = !favor_uastc && (e.flags & FLAG_FAVOR_BC7) != 0;
    const bool window = !(best_rms == 0.0f || favor_bc7);
    uint64_t best = UINT64_MAX;
    uint32_t best_i = first;
    for (uint32_t i = 0; i < n_slots; i++) {
        if (!v.valid(i)) continue;
        if (window && !((double)v.rms(i) / best_rms <= (double)1.3f)) continue;
        const float wgt = (e.flags & FLAG_FAVOR_SIMPLER) ? mode_bias(v.mode(i)) : 1.0f;
        const uint64_t w = (uint64_t)((float)v.overall(i) * wgt);
        if (w < best) {
            best = w;
            best_i = i;
            if (!best) break;
        }
    }
    return best_i;
}

// ------------------------------------------------------------------------------------------------------------------
// Transcode hints: BC1 (uastc_enc.cpp:2535-2629), EAC A8 (:3012-3103), ETC1 (:2714-3010)
// ------------------------------------------------------------------------------------------------------------------

struct bc1_blk { uint16_t c0, c1; uint32_t sel; };  // selectors: 2 bits per texel, texel 0 in the low bits

// Everything below works on texels as packed dwords held in registers (r | g << 8 | b << 16 | a << 24), selector sets as 2 bits per texel in one dword, and blocks by
// value: on the GPU an image that reaches a function through a pointer, or a byte array indexed by a run-time value, lives in scratch memory, and these functions used
// to spend their time waiting for it (round 3; the same change as uastc_errors / bc7_errors).

// bcu::unpack_bc1 (transcoder/basisu_dds_transcoder.inl:23-80) + the RGB error against `px` in one pass
BU_FN uint32_t bc1_error(const bc1_blk& b, const uint32_t* px) {
    int col[4][3];
    const uint32_t c[2] = { b.c0, b.c1 };
    for (int k = 0; k < 2; k++) {
        const uint32_t r = (c[k] >> 11) & 31, g = (c[k] >> 5) & 63, bl = c[k] & 31;
        col[k][0] = (int)((r << 3) | (r >> 2)); col[k][1] = (int)((g << 2) | (g >> 4)); col[k][2] = (int)((bl << 3) | (bl >> 2));
    }
    const bool four = b.c0 > b.c1;
    for (int ch = 0; ch < 3; ch++) {
        col[2][ch] = four ? (col[0][ch] * 2 + col[1][ch]) / 3 : (col[0][ch] + col[1][ch]) / 2;
        col[3][ch] = four ? (col[1][ch] * 2 + col[0][ch]) / 3 : 0;
    }
    uint32_t total = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        const uint32_t sl = (b.sel >> (2 * i)) & 3u;
        const bool odd = (sl & 1u) != 0, upper = (sl & 2u) != 0;
        BU_UNROLL
        for (int ch = 0; ch < 3; ch++) {
            const int v = upper ? (odd ? col[3][ch] : col[2][ch]) : (odd ? col[1][ch] : col[0][ch]);
            const int d = px_comp(px[i], ch) - v;
            total += (uint32_t)imul24(d, d);
        }
    }
    return total;
}

// bc1_find_sels (transcoder.cpp:17857-17885): linear selectors 0..3 along low->high for 5:6:5 endpoints; 2 bits per texel
BU_FN uint32_t bc1_pick_selectors(const uint32_t* px, const int* l, const int* h) {
    int br[4], bg[4], bb[4];
    br[0] = (l[0] << 3) | (l[0] >> 2); bg[0] = (l[1] << 2) | (l[1] >> 4); bb[0] = (l[2] << 3) | (l[2] >> 2);
    br[3] = (h[0] << 3) | (h[0] >> 2); bg[3] = (h[1] << 2) | (h[1] >> 4); bb[3] = (h[2] << 3) | (h[2] >> 2);
    br[1] = (br[0] * 2 + br[3]) / 3; bg[1] = (bg[0] * 2 + bg[3]) / 3; bb[1] = (bb[0] * 2 + bb[3]) / 3;
    br[2] = (br[3] * 2 + br[0]) / 3; bg[2] = (bg[3] * 2 + bg[0]) / 3; bb[2] = (bb[3] * 2 + bb[0]) / 3;
    int ar = br[3] - br[0], ag = bg[3] - bg[0], ab = bb[3] - bb[0];
    int dots[4];
    for (int i = 0; i < 4; i++) dots[i] = imul24(br[i], ar) + imul24(bg[i], ag) + imul24(bb[i], ab);
    const int t0 = dots[0] + dots[1], t1 = dots[1] + dots[2], t2 = dots[2] + dots[3];
    ar *= 2; ag *= 2; ab *= 2;
    uint32_t sels = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) {
        const int d = imul24(px_comp(px[i], 0), ar) + imul24(px_comp(px[i], 1), ag) + imul24(px_comp(px[i], 2), ab);
        sels |= (uint32_t)(3 - ((d <= t0) + (d < t1) + (d < t2))) << (2 * i);
    }
    return sels;
}

BU_FN bc1_blk bc1_solid(uint32_t r, uint32_t g, uint32_t b) {  // encode_bc1_solid_block, transcoder.cpp:17999-18042
    uint32_t mask = 0xAA;
    uint32_t max16 = ((uint32_t)ku_bc1_match5[r * 2] << 11) | ((uint32_t)ku_bc1_match6[g * 2] << 5) | ku_bc1_match5[b * 2];
    uint32_t min16 = ((uint32_t)ku_bc1_match5[r * 2 + 1] << 11) | ((uint32_t)ku_bc1_match6[g * 2 + 1] << 5) | ku_bc1_match5[b * 2 + 1];
    if (min16 == max16) {
        mask = 0;
        if (min16 > 0) min16--;
        else { max16 = 1; min16 = 0; mask = 0x55; }
    }
    if (max16 < min16) { const uint32_t t = max16; max16 = min16; min16 = t; mask ^= 0x55; }
    bc1_blk out;
    out.c0 = (uint16_t)max16; out.c1 = (uint16_t)min16;
    out.sel = mask * 0x01010101u;
    return out;
}

// basist::encode_bc1 (transcoder.cpp:18047-18283) with flags 0 (use_given false) or cEncodeBC1UseSelectors (the selectors in `given`, 2 bits per texel)
BU_FN bc1_blk bc1_encode(const uint32_t* px, bool use_given, uint32_t given) {
    int avg[3] = { -1, 0, 0 };
    int l[3] = { 0, 0, 0 }, h[3] = { 0, 0, 0 };
    uint32_t sels = given;
    int tot[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 }, mn[3] = { 255, 255, 255 };
    BU_UNROLL
    for (int i = 0; i < 16; i++)
        BU_UNROLL
        for (int c = 0; c < 3; c++) {
            const int v = px_comp(px[i], c);
            tot[c] += v; mx[c] = v > mx[c] ? v : mx[c]; mn[c] = v < mn[c] ? v : mn[c];
        }
    if (!use_given) {
        if (mx[0] == mn[0] && mx[1] == mn[1] && mx[2] == mn[2]) return bc1_solid((uint32_t)mn[0], (uint32_t)mn[1], (uint32_t)mn[2]);
        for (int c = 0; c < 3; c++) avg[c] = (tot[c] + 8) >> 4;
        int icov[6] = { 0, 0, 0, 0, 0, 0 };
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            const int r = px_comp(px[i], 0) - avg[0], g = px_comp(px[i], 1) - avg[1], b = px_comp(px[i], 2) - avg[2];
            icov[0] += imul24(r, r); icov[1] += imul24(r, g); icov[2] += imul24(r, b); icov[3] += imul24(g, g); icov[4] += imul24(g, b); icov[5] += imul24(b, b);
        }
        float cov[6];
        for (int i = 0; i < 6; i++) cov[i] = (float)icov[i] * (1.0f / 255.0f);
        float xr = (float)(mx[0] - mn[0]), xg = (float)(mx[1] - mn[1]), xb = (float)(mx[2] - mn[2]);
        for (int it = 0; it < 4; it++) {
            const float r = xr * cov[0] + xg * cov[1] + xb * cov[2];
            const float g = xr * cov[1] + xg * cov[3] + xb * cov[4];
            const float b = xr * cov[2] + xg * cov[4] + xb * cov[5];
            xr = r; xg = g; xb = b;
        }
        float k = fabsf(xr) > fabsf(xg) ? fabsf(xr) : fabsf(xg);
        k = k > fabsf(xb) ? k : fabsf(xb);
        int sa[3] = { 306, 601, 117 };
        if (k >= 2) {
            const float m = 1024.0f / k;
            sa[0] = (int)(xr * m); sa[1] = (int)(xg * m); sa[2] = (int)(xb * m);
        }
        int low_dot = INT32_MAX, high_dot = INT32_MIN;
        uint32_t low_px = 0, high_px = 0;   // the first texel with the least / the greatest projection, itself rather than its index
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            const int dot = imul24(px_comp(px[i], 0), sa[0]) + imul24(px_comp(px[i], 1), sa[1]) + imul24(px_comp(px[i], 2), sa[2]);
            if (dot < low_dot) { low_dot = dot; low_px = px[i]; }
            if (dot > high_dot) { high_dot = dot; high_px = px[i]; }
        }
        for (int c = 0; c < 3; c++) {
            const uint32_t mul = c == 1 ? 63 : 31;
            uint32_t v = (uint32_t)px_comp(low_px, c) * mul + 128;
            l[c] = (int)((v + (v >> 8)) >> 8);
            v = (uint32_t)px_comp(high_px, c) * mul + 128;
            h[c] = (int)((v + (v >> 8)) >> 8);
        }
        sels = bc1_pick_selectors(px, l, h);
    }
    {
        // one least-squares pass (compute_least_squares_endpoints_rgb, transcoder.cpp:17922-17997)
        uint32_t q00[3] = { 0, 0, 0 }, wacc = 0;
        BU_UNROLL
        for (int i = 0; i < 16; i++) {
            const uint32_t sel = (sels >> (2 * i)) & 3u;
            wacc += sel == 0 ? 0x000009u : (sel == 1 ? 0x010204u : (sel == 2 ? 0x040201u : 0x090000u));
            BU_UNROLL
            for (int c = 0; c < 3; c++) q00[c] += sel * (uint32_t)px_comp(px[i], c);
        }
        const float z00 = (float)((wacc >> 16) & 0xFF), z10 = (float)((wacc >> 8) & 0xFF), z11 = (float)(wacc & 0xFF), z01 = z10;
        float det = z00 * z11 - z01 * z10;
        if (fabsf(det) < 1e-8f) {
            if (avg[0] < 0) for (int c = 0; c < 3; c++) avg[c] = (tot[c] + 8) >> 4;
            l[0] = ku_bc1_match5[avg[0] * 2]; l[1] = ku_bc1_match6[avg[1] * 2]; l[2] = ku_bc1_match5[avg[2] * 2];
            h[0] = ku_bc1_match5[avg[0] * 2 + 1]; h[1] = ku_bc1_match6[avg[1] * 2 + 1]; h[2] = ku_bc1_match5[avg[2] * 2 + 1];
        } else {
            det = 3.0f / det;
            const float iz00 = z11 * det, iz01 = -z01 * det, iz10 = -z10 * det, iz11 = z00 * det;
            for (int c = 0; c < 3; c++) {
                const float fq00 = (float)q00[c], ft = (float)tot[c];
                const float fq10 = ft * 3.0f - fq00;
                float xl = iz00 * fq00 + iz01 * fq10, xh = iz10 * fq00 + iz11 * fq10;
                if ((xl < 0.0f || xh > 255.0f) && mn[c] == mx[c]) { xl = (float)mn[c]; xh = (float)mx[c]; }
                const float scale = c == 1 ? (63.0f / 255.0f) : (31.0f / 255.0f);
                const int top = c == 1 ? 63 : 31;
                l[c] = clampi((int)(xl * scale + .5f), 0, top);
                h[c] = clampi((int)(xh * scale + .5f), 0, top);
            }
        }
        sels = bc1_pick_selectors(px, l, h);
    }
    uint32_t lc16 = ((uint32_t)l[0] << 11) | ((uint32_t)l[1] << 5) | (uint32_t)l[2];
    uint32_t hc16 = ((uint32_t)h[0] << 11) | ((uint32_t)h[1] << 5) | (uint32_t)h[2];
    bc1_blk out;
    if (lc16 == hc16) {
        uint32_t mask = 0;
        if (hc16 > 0) hc16--;
        else { hc16 = 0; lc16 = 1; mask = 0x55; }
        out.c0 = (uint16_t)lc16; out.c1 = (uint16_t)hc16; out.sel = mask * 0x01010101u;
    } else {
        uint32_t invert = 0;
        if (lc16 < hc16) { const uint32_t t = lc16; lc16 = hc16; hc16 = t; invert = 0x55555555u; }
        // linear selector 0..3 -> BC1 code {0, 2, 3, 1}, all sixteen at once: code = (s >> 1) | ((s ^ (s >> 1)) & 1) << 1
        const uint32_t hi = (sels >> 1) & 0x55555555u, lo = sels & 0x55555555u;
        out.c0 = (uint16_t)lc16; out.c1 = (uint16_t)hc16; out.sel = (hi | ((hi ^ lo) << 1)) ^ invert;
    }
    return out;
}

// The candidate as pack_uastc stores it: every subset/plane anchor weight has its top bit clear (weights mirrored and the
// endpoints of that subset/plane swapped otherwise) -- uastc_enc.cpp:262-338. The BC1 hint transcodes see this form.
BU_FN void normalise_anchors(cand& r) {
    const uint32_t mode = r.mode, subsets = ku_mode_subsets[mode], planes = ku_mode_planes[mode], comps = ku_mode_comps[mode];
    const uint32_t wbits = ku_mode_weight_bits[mode], top = (1u << wbits) - 1;
    const uint32_t pat = astc_pattern_bits(mode, r.pattern);
    const uint8_t* anchors = subsets == 3 ? ku_anchor3 + r.pattern * 3 : (mode == 7 ? ku_anchor7 + r.pattern * 3 : ku_anchor2 + r.pattern * 3);
    for (uint32_t p = 0; p < planes; p++)
        for (uint32_t s = 0; s < subsets; s++) {
            const uint32_t anchor = subsets >= 2 ? anchors[s] : 0;
            if (!(r.weights[anchor * planes + p] & (1u << (wbits - 1)))) continue;
            for (uint32_t i = 0; i < 16; i++)
                if (((pat >> (2 * i)) & 3) == s) r.weights[i * planes + p] = (uint8_t)(top - r.weights[i * planes + p]);
            for (uint32_t c = 0; c < comps; c++) {
                if (planes == 2) {
                    const uint32_t comp_plane = comps == 2 ? c : (c == r.ccs ? 1u : 0u);
                    if (comp_plane != p) continue;
                }
                uint8_t* e = r.endpoints + (planes == 2 ? 0 : s * comps * 2) + c * 2;
                const uint8_t t = e[0]; e[0] = e[1]; e[1] = t;
            }
        }
}

BU_FN uint32_t bc1_weight_translate(uint32_t wbits, uint32_t w) {  // s_uastc{1..5}_to_bc1, transcoder.cpp:17729-17735 (2 bits per entry in a literal)
    return wbits == 5 ? (uint32_t)(0x555fffffaaaaa000ull >> (2 * w)) & 3u : (wbits == 4 ? (0x57ffaa80u >> (2 * w)) & 3u : (wbits == 3 ? (0x5fa0u >> (2 * w)) & 3u : (wbits == 2 ? (0x78u >> (2 * w)) & 3u : w)));
}

BU_FN uint32_t pack565_scaled(uint32_t r, uint32_t g, uint32_t b) {  // dxt1_block::pack_color(scaled, bias 127), uastc_enc.cpp:69-80
    r = (r * 31 + 127) / 255; g = (g * 63 + 127) / 255; b = (b * 31 + 127) / 255;
    return (r << 11) | (g << 5) | b;
}

// compute_bc1_hints (uastc_enc.cpp:2535-2629); `norm` is the anchor-normalised winner, `decoded` its UASTC decode, `px` the source texels (both packed dwords)
BU_FN void bc1_hints(const cand& norm, const uint32_t* px, const uint32_t* decoded, bool& hint0, bool& hint1) {
    hint0 = hint1 = false;
    const uint32_t mode = norm.mode;
    const bool has0 = ku_mode_has_bc1_hint0[mode] != 0, has1 = ku_mode_has_bc1_hint1[mode] != 0;
    if (!has0 && !has1) return;
    const uint32_t wbits = ku_mode_weight_bits[mode], planes = ku_mode_planes[mode], comps = ku_mode_comps[mode];
    // the first plane's weights translated to BC1's four levels, 2 bits per texel
    uint32_t tw = 0;
    BU_UNROLL
    for (int i = 0; i < 16; i++) tw |= bc1_weight_translate(wbits, planes == 2 ? norm.weights[i * 2] : norm.weights[i]) << (2 * i);
    const uint32_t et = bc1_error(bc1_encode(decoded, false, 0), px);
    uint32_t e0 = 0, e1 = 0;
    if (has1) {  // transcode_uastc_to_bc1_hint1, transcoder.cpp:18700-18728: BC1 code -> linear selector {0, 3, 1, 2}, all sixteen at once: (low bit, high ^ low)
        const uint32_t hi = (tw >> 1) & 0x55555555u, lo = tw & 0x55555555u;
        e1 = bc1_error(bc1_encode(decoded, true, (lo << 1) | (hi ^ lo)), px);
    }
    if (has0) {  // transcode_uastc_to_bc1_hint0, :18602-18697
        const uint8_t* UQ = ku_unquant + ku_mode_endpoint_ranges[mode] * 256;
        uint32_t lc, hc;
        if (comps == 2) { lc = pack565_scaled(UQ[norm.endpoints[0]], UQ[norm.endpoints[0]], UQ[norm.endpoints[0]]); hc = pack565_scaled(UQ[norm.endpoints[1]], UQ[norm.endpoints[1]], UQ[norm.endpoints[1]]); }
        else { lc = pack565_scaled(UQ[norm.endpoints[0]], UQ[norm.endpoints[2]], UQ[norm.endpoints[4]]); hc = pack565_scaled(UQ[norm.endpoints[1]], UQ[norm.endpoints[3]], UQ[norm.endpoints[5]]); }
        bc1_blk b;
        if (lc == hc) {
            uint32_t mask = 0;
            if (hc > 0) hc--;
            else { hc = 0; lc = 1; mask = 0x55; }
            b.c0 = (uint16_t)lc; b.c1 = (uint16_t)hc; b.sel = mask * 0x01010101u;
        } else {
            uint32_t sels = tw;
            if (lc < hc) { const uint32_t t = lc; lc = hc; hc = t; sels ^= 0x55555555u; }
            b.c0 = (uint16_t)lc; b.c1 = (uint16_t)hc; b.sel = sels;
        }
        e0 = bc1_error(b, px);
    }
    const float t_err = sqrtf((float)et), t0 = sqrtf((float)e0), t1 = sqrtf((float)e1);
    if (has0 && t0 <= t_err * 1.075f) hint0 = true;
    if (has1 && t1 <= t_err * 1.075f) hint1 = true;
}

// uastc_pack_eac_a8 (uastc_enc.cpp:3019-3103) with base_search_rad 0: only the table and multiplier are kept
BU_FN void eac_a8_hint(const rgba8* decoded, uint32_t mul_rad, uint32_t table_mask, uint32_t& out_table, uint32_t& out_mul) {
    uint32_t amin = 255, amax = 0;
    for (uint32_t i = 0; i < 16; i++) { const uint32_t a = decoded[i].c[3]; amin = a < amin ? a : amin; amax = a > amax ? a : amax; }
    out_table = 13; out_mul = 1;
    if (amin == amax) return;
    out_table = 0; out_mul = 0;
    const uint32_t arange = amax - amin;
    uint64_t best = UINT64_MAX;
    for (uint32_t table = 0; table < 16; table++) {
        if (!((table_mask >> table) & 1)) continue;
        const signed char* T = ku_eac_tables + table * 8;
        const float range = (float)(T[7] - T[3]);
        const float tpos = (float)(0 - T[3]) / range;
        const int center = (int)roundf((float)amin + ((float)amax - (float)amin) * tpos);
        const int base = clampi(center, 0, 255);
        const int mul = (int)roundf((float)arange / range);
        const int mlo = clampi(mul - (int)mul_rad, 1, 15), mhi = clampi(mul + (int)mul_rad, 1, 15);
        for (int m = mlo; m <= mhi; m++) {
            uint64_t total = 0;
            for (uint32_t i = 0; i < 16; i++) {
                const int a = decoded[i].c[3];
                uint32_t best_s = 0xFFFFFFFFu;
                for (uint32_t s = 0; s < 8; s++) {
                    const int v = clampi(m * T[s] + base, 0, 255);
                    const uint32_t err = (uint32_t)(a > v ? a - v : v - a);
                    best_s = err < best_s ? err : best_s;
                }
                total += best_s * best_s;
                if (total >= best) break;
            }
            if (total < best) {
                best = total;
                out_mul = (uint32_t)m; out_table = table;
                if (!best) return;
            }
        }
    }
}

struct etc1_hint { uint8_t flip, diff, inten0, inten1, bias; };
BU_TAB unsigned char ku_etc1_bias_order[32] = { 13, 0, 22, 29, 27, 12, 26, 9, 30, 31, 8, 10, 25, 2, 23, 5, 15, 7, 3, 11, 6, 17, 28, 18, 1, 19, 20, 21, 24, 4, 14, 16 };

BU_FN void ycbcr(const uint8_t* c, int* o) {  // rgb_to_y_cb_cr, uastc_enc.cpp:2638-2644
    const int y = c[0] * 54 + c[1] * 183 + c[2] * 19;
    o[0] = y; o[1] = ((int)c[2] << 8) - y; o[2] = ((int)c[0] << 8) - y;
}
BU_FN uint64_t ycbcr_diff(const int* a, const int* b) {  // color_diff, :2646-2652
    const int64_t dy = a[0] - b[0], dcb = a[1] - b[1], dcr = a[2] - b[2];
    return (uint64_t)(dy * dy * 4 + dcr * dcr + dcb * dcb);
}

BU_FN uint32_t etc1_bias_apply(uint32_t v_in, uint32_t c, uint32_t bias, uint32_t limit, uint32_t sub) {  // apply_etc1_bias, transcoder.cpp:16547-16612
    int delta;
    switch (bias) {
    case 2: delta = sub ? 0 : (c == 0 ? -1 : 0); break;
    case 5: delta = sub ? 0 : (c == 1 ? -1 : 0); break;
    case 6: delta = sub ? 0 : (c == 2 ? -1 : 0); break;
    case 7: delta = sub ? 0 : (c == 0 ? 1 : 0); break;
    case 11: delta = sub ? 0 : (c == 1 ? 1 : 0); break;
    case 15: delta = sub ? 0 : (c == 2 ? 1 : 0); break;
    case 18: delta = sub ? (c == 0 ? -1 : 0) : 0; break;
    case 19: delta = sub ? (c == 1 ? -1 : 0) : 0; break;
    case 20: delta = sub ? (c == 2 ? -1 : 0) : 0; break;
    case 21: delta = sub ? (c == 0 ? 1 : 0) : 0; break;
    case 24: delta = sub ? (c == 1 ? 1 : 0) : 0; break;
    case 8: delta = sub ? (c == 2 ? 1 : 0) : 0; break;
    case 10: delta = -2; break;
    case 27: delta = sub ? 0 : -1; break;
    case 28: delta = sub ? -1 : 1; break;
    case 29: delta = sub ? 1 : 0; break;
    case 30: delta = sub ? -1 : 0; break;
    case 31: delta = sub ? 0 : 1; break;
    default: { const uint32_t divs[3] = { 1, 3, 9 }; delta = (int)((bias / divs[c]) % 3) - 1; break; }
    }
    int v = (int)v_in;
    if (v == 0) v += delta == -2 ? 3 : delta + 1;
    else if (v == (int)limit) v += delta - 1;
    else {
        v += delta;
        if (v < 0 || v > (int)limit) v = (v - delta) - delta;
    }
    return (uint32_t)v;
}

BU_FN bool etc1_estimate_flipped(const rgba8* p) {  // pack_etc1_estimate_flipped, uastc_enc.cpp:2654-2712
    int sums[3][2][2];
    for (uint32_t c = 0; c < 3; c++)
        for (uint32_t qx = 0; qx < 2; qx++)
            for (uint32_t qy = 0; qy < 2; qy++) {
                int t = 0;
                for (uint32_t y = 0; y < 2; y++) for (uint32_t x = 0; x < 2; x++) t += p[(qx * 2 + x) + (qy * 2 + y) * 4].c[c];
                sums[c][qx][qy] = t;
            }
    int upper[3], lower[3], left[3], right[3];
    for (uint32_t c = 0; c < 3; c++) {
        upper[c] = (sums[c][0][0] + sums[c][1][0] + 4) / 8;
        lower[c] = (sums[c][0][1] + sums[c][1][1] + 4) / 8;
        left[c] = (sums[c][0][0] + sums[c][0][1] + 4) / 8;
        right[c] = (sums[c][1][0] + sums[c][1][1] + 4) / 8;
    }
    int ul = 0, lr = 0;
    for (uint32_t i = 0; i < 4; i++)
        for (uint32_t j = 0; j < 2; j++) {
            const struct { uint32_t x, y; const int* avg; } q[4] = { { i, j, upper }, { i, 2 + j, lower }, { j, i, left }, { 2 + j, i, right } };
            for (uint32_t k = 0; k < 4; k++) {
                const uint8_t* c = p[q[k].x + q[k].y * 4].c;
                const int* a = q[k].avg;
                const int gd = (((int)c[0] - a[0]) + ((int)c[1] - a[1]) + ((int)c[2] - a[2]) + 1) / 3;
                int d = 0;
                for (uint32_t ch = 0; ch < 3; ch++) { const int e = (int)c[ch] - clampi(a[ch] + gd, 0, 255); d += e * e; }
                if (k < 2) ul += d; else lr += d;
            }
        }
    return ul < lr;
}

// ---- compute_etc1_hints (uastc_enc.cpp:2714-3010) for non-solid blocks.
// The search is 2 flips x 2 colour modes x up to 32 biases, each trial fitting an intensity table per sub-block against the DECODED
// texels and then scoring the trial against the SOURCE texels, all in a Y/Cb/Cr-like integer space. Everything below is laid out
// so that a GPU lane keeps the 2 x 16 x 3 texel values in registers: flip and sub-block are template parameters (so every texel
// index is a compile-time constant) and the distances are written for the 32 x 32 + 64-bit multiply-add (etc1_fit_subblock).
// The reference's row-wise early outs (:2885, :2905, :2946, :2964) only stop once a running error has reached the best one, so
// they cannot change a result and are dropped; the one early out that does (non-flipped: stop trying intensity tables at the
// first one that is not better, :2906-2907) is kept.

struct ycc { int y, cb, cr; };
BU_FN ycc to_ycc(int r, int g, int b) { const int y = imul24(r, 54) + imul24(g, 183) + imul24(b, 19); ycc o = { y, (b << 8) - y, (r << 8) - y }; return o; }
struct texels_ycc { ycc t[16]; };
// The SOURCE texels in the same space, one register each: y | b << 16 | r << 24 (cb = 256 b - y, cr = 256 r - y). They are only read once per texel and fit, so
// they are kept packed -- but kept: reading them through the caller's pointer is a scratch-memory load and a full wait per texel inside the innermost loop.
struct texels_packed { uint32_t t[16]; };
BU_FN uint32_t pack_ycc_source(const rgba8& p) { return (uint32_t)(imul24(p.c[0], 54) + imul24(p.c[1], 183) + imul24(p.c[2], 19)) | ((uint32_t)p.c[2] << 16) | ((uint32_t)p.c[0] << 24); }
BU_FN ycc unpack_ycc_source(uint32_t v) { const int y = (int)(v & 0xFFFFu); ycc o = { y, (int)((v >> 8) & 0xFF00u) - y, (int)((v >> 16) & 0xFF00u) - y }; return o; }
// raster index of texel j (0..7) of sub-block SUB; in the flipped layout j runs row by row (g_etc1_pixel_coords, etc.cpp:314-337)
template <int FLIP, int SUB> constexpr int etc1_texel(int j) { return FLIP ? (SUB * 8 + j) : ((j & 3) * 4 + SUB * 2 + (j >> 2)); }

struct etc1_subblock_stats { int mn[3], mx[3]; uint32_t sum[3]; };
template <int FLIP, int SUB>
BU_FN void etc1_stats(const rgba8* decoded, etc1_subblock_stats& s) {
    for (int c = 0; c < 3; c++) { s.mn[c] = 255; s.mx[c] = 0; s.sum[c] = 0; }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; j++)
        for (int c = 0; c < 3; c++) {
            const int v = decoded[etc1_texel<FLIP, SUB>(j)].c[c];
            s.sum[c] += (uint32_t)v; s.mn[c] = v < s.mn[c] ? v : s.mn[c]; s.mx[c] = v > s.mx[c] ? v : s.mx[c];
        }
}

// ---- The two functions the search spends its time in, in 64-bit integer form. For a block colour c and a texel t
//     D(c, t) = 4 (yc - yt)^2 + (cbc - cbt)^2 + (crc - crt)^2 = N(c) + N(t) - 2 (4 yc yt + cbc cbt + crc crt),        N(v) = 4 y^2 + cb^2 + cr^2,
// so with the colour's -32 y, -8 cb, -8 cr and 4 N(c) prepared once per table, 4 (D - N(t)) is three v_mad_i64_i32 per texel and colour (4 issue
// cycles each on gfx950, profiles/valu_calibration.json) where the double-precision form is three subtractions, a multiply, two v_fma_f64 and a
// compare + two selects for the minimum. The colour's index rides in the two low bits, so one minimum over the four keys gives the reference's FIRST minimum and
// which colour it was; N(t) is the same for every colour and table and is added once per sub-block (its sum over the texels). The keys are kept as the
// bit patterns of doubles in [2^52, 2^53) (offset 2^52 + 2^42 folded into the accumulator's start value): they order like the integers, so
// the minimum is one v_min_f64, and subtracting the offset as doubles returns the exact integer. Every total is an integer below 2^53: the
// doubles that come out are the ones the reference computes, bit for bit.
// the eight intensity tables are -b, -a, a, b with these a (small) and b (large); a byte per table in a 64-bit literal, so a table's modifiers are two scalar shifts
// and not a load from a constant array that the loop over the tables would wait for
BU_FN int etc1_inten_small(uint32_t table) { return (int)((0x2F2118120D090502ull >> (8 * table)) & 255u); }   // 2, 5, 9, 13, 18, 24, 33, 47
BU_FN int etc1_inten_large(uint32_t table) { return (int)((0xB76A503C2A1D1108ull >> (8 * table)) & 255u); }   // 8, 17, 29, 42, 60, 80, 106, 183
struct etc1_colour { int a, b, c; long long n; };   // -32 y, -8 cb, -8 cr; 4 N(colour) + colour index + key offset
BU_TAB long long ku_etc1_key_offset = 0x4330000000000000ll + (1ll << 42);   // bits of the double 2^52 + 2^42; 4 (D - N(t)) > -2^39 keeps every key above 2^52
BU_FN double etc1_key_value(long long key) {   // 4 (D - N(t)) of a key, colour index dropped
    return __builtin_bit_cast(double, key & ~3ll) - __builtin_bit_cast(double, ku_etc1_key_offset);
}
BU_FN long long etc1_key_min(long long x, long long y) {
#if defined(__HIPCC__)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(double, x)), "v"(__builtin_bit_cast(double, y)));
    return __builtin_bit_cast(long long, r);
#else
    return x < y ? x : y;
#endif
}
BU_FN void etc1_colour_keys(const int* base, uint32_t table, etc1_colour* col) {
    const int small = etc1_inten_small(table), large = etc1_inten_large(table);
    for (uint32_t k = 0; k < 4; k++) {
        const int d = k == 0 ? -large : (k == 1 ? -small : (k == 2 ? small : large));
        const ycc c = to_ycc(clampi(base[0] + d, 0, 255), clampi(base[1] + d, 0, 255), clampi(base[2] + d, 0, 255));
        col[k].a = imul24(c.y, -32); col[k].b = imul24(c.cb, -8); col[k].c = imul24(c.cr, -8);
        const int y4 = 4 * c.y, cb2 = 2 * c.cb, cr2 = 2 * c.cr;
        col[k].n = (long long)y4 * y4 + ((long long)cb2 * cb2 + ((long long)cr2 * cr2 + (ku_etc1_key_offset + (long long)k)));
    }
}
// x * y + z. Written out for the GPU: left to itself the compiler starts the chain of three from zero and adds the colour's constant with a
// fourth instruction (v_lshl_add_u64) per key.
BU_FN long long etc1_mad(int x, int y, long long z) {
#if defined(__HIPCC__)
    long long r;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z) : "vcc");
    return r;
#else
    return (long long)x * y + z;
#endif
}
BU_FN long long etc1_key(const etc1_colour& c, const ycc& t) { return etc1_mad(c.a, t.y, etc1_mad(c.b, t.cb, etc1_mad(c.c, t.cr, c.n))); }
// ---- The same sums when NO colour of the table clamps, i.e. when the four block colours are base + d (1, 1, 1) exactly. (1, 1, 1) is pure luma in this space
// (54 + 183 + 19 = 256 and both chroma rows sum to zero), so with dy = y(base) - y(t)
//     D(base + d, t) = 4 (dy + 256 d)^2 + dcb^2 + dcr^2 = D(base, t) + 2048 d (dy + 128 d),
// and what depends on the table is a 32-bit integer. With the table's modifiers -b, -a, a, b (a < b) the minimum over the four colours is
// min(a (128 a - |dy|), b (128 b - |dy|)), and the FIRST colour in table order that reaches it (the reference's strict "<") is -b or -a for dy >= 0 (-b on a tie),
// +a or +b for dy < 0 (+a on a tie). The texel sums of D(base, t) are closed forms in the sub-block's moments. Exact integers throughout, so the totals are
// the same doubles as the general form's. This form is only worth taking when EVERY lane of the wave can take it (a wave runs both sides of a branch
// its lanes disagree on): the encoder therefore hands the finish kernel its blocks grouped by how much head room their colours have (uastc_kernels.hip,
// etc1_order_key), and the test below is wave-wide. The host build decides per block; both forms give the same numbers, so it does not matter which ran.
#if defined(__HIPCC__)
#define BU_WAVE_ALL(cond) (__builtin_amdgcn_ballot_w64(!(cond)) == 0ull)
#else
#define BU_WAVE_ALL(cond) (cond)
#endif
struct etc1_moments { int sy, scb, scr; double sn; };   // sums over a sub-block's 8 texels of y, cb, cr and of N(texel)
BU_FN void etc1_moments_add(etc1_moments& m, const ycc& c) {
    m.sy += c.y; m.scb += c.cb; m.scr += c.cr;
    const double y2 = (double)(2 * c.y), cb = (double)c.cb, cr = (double)c.cr;
    m.sn += __builtin_fma(y2, y2, __builtin_fma(cb, cb, cr * cr));
}
BU_FN double etc1_moment_distance(const etc1_moments& m, const ycc& c) {   // sum over the sub-block's texels t of D(c, t)
    const double y2 = (double)(2 * c.y), cb = (double)c.cb, cr = (double)c.cr;
    const double n = __builtin_fma(y2, y2, __builtin_fma(cb, cb, cr * cr));
    const double dot = __builtin_fma(2.0 * y2, (double)m.sy, __builtin_fma(cb, (double)m.scb, cr * (double)m.scr));
    return __builtin_fma(8.0, n, m.sn) - 2.0 * dot;
}
BU_FN uint32_t etc1_unclamped_tables(const int* base) {   // number of leading tables whose colours do not clamp for this base colour
    int lo = base[0] < base[1] ? base[0] : base[1]; lo = base[2] < lo ? base[2] : lo;
    int hi = base[0] > base[1] ? base[0] : base[1]; hi = base[2] > hi ? base[2] : hi;
    const int room = lo < 255 - hi ? lo : 255 - hi;
    uint32_t u = 0;
    for (uint32_t t = 0; t < 8; t++) u += etc1_inten_large(t) <= room ? 1u : 0u;
    return u;
}

// Best intensity table of one sub-block for base colour `base` (uastc_enc.cpp:2858-2918: first minimum over the tables below `limit` of the texel sums of the
// distance to the nearest block colour; the non-flipped search stops at the first table that is not better), then the sub-block's error against
// the SOURCE texels when every texel takes the colour nearest to its DECODED value (:2925-2973).
template <int FLIP, int SUB>
BU_FN void etc1_fit_subblock(const texels_ycc& dec, const texels_packed& src, const etc1_moments& mdec, const etc1_moments& msrc, const int* base, uint32_t limit,
                             uint32_t& table_out, double& err_out) {
    double best = 1e300;
    uint32_t best_table = 0;
    const uint32_t unclamped = etc1_unclamped_tables(base);
    const ycc bc = to_ycc(base[0], base[1], base[2]);
    etc1_colour col[4];
    int ady[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    double base_total = 0.0;   // the texel sum of D(base, t)
    if (!BU_WAVE_ALL(unclamped == 0)) {   // some lane of the wave may get to use the short form
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < 8; j++) { const int d = bc.y - dec.t[etc1_texel<FLIP, SUB>(j)].y; ady[j] = d < 0 ? -d : d; }
        base_total = etc1_moment_distance(mdec, bc);
    }
    for (uint32_t table = 0; table < limit; table++) {
        double total;
        if (BU_WAVE_ALL(table < unclamped)) {
            const int a = etc1_inten_small(table), b = etc1_inten_large(table);
            const int a2 = 128 * a * a, b2 = 128 * b * b;
            int g = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < 8; j++) {
                const int ga = a2 - imul24(a, ady[j]), gb = b2 - imul24(b, ady[j]);
                g += ga < gb ? ga : gb;
            }
            total = __builtin_fma(2048.0, (double)g, base_total);
        } else {
            etc1_colour_keys(base, table, col);
            double total4 = 0.0;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < 8; j++) {
                const ycc& t = dec.t[etc1_texel<FLIP, SUB>(j)];
                total4 += etc1_key_value(etc1_key_min(etc1_key_min(etc1_key(col[0], t), etc1_key(col[1], t)), etc1_key_min(etc1_key(col[2], t), etc1_key(col[3], t))));
            }
            total = __builtin_fma(total4, 0.25, mdec.sn);
        }
        if (!FLIP && total >= best) break;
        if (total < best) { best = total; best_table = table; }
    }
    table_out = best_table;
    if (BU_WAVE_ALL(best_table < unclamped)) {
        const int a = etc1_inten_small(best_table), b = etc1_inten_large(best_table);
        const int a2 = 128 * a * a, b2 = 128 * b * b;
        int s = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int j = 0; j < 8; j++) {
            const int ti = etc1_texel<FLIP, SUB>(j);
            const int dy = bc.y - dec.t[ti].y;
            const int ga = a2 - imul24(a, ady[j]), gb = b2 - imul24(b, ady[j]);
            const bool nonneg = dy >= 0;
            const bool large = nonneg ? gb <= ga : gb < ga;
            const int mag = large ? b : a, d = nonneg ? -mag : mag;
            const int dys = bc.y - (int)(src.t[ti] & 0xFFFFu);
            s += imul24(d, dys + 128 * d);
        }
        err_out = __builtin_fma(2048.0, (double)s, etc1_moment_distance(msrc, bc));
        return;
    }
    etc1_colour_keys(base, best_table, col);
    double err4 = 0.0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; j++) {
        const int ti = etc1_texel<FLIP, SUB>(j);
        const ycc& t = dec.t[ti];
        const long long m = etc1_key_min(etc1_key_min(etc1_key(col[0], t), etc1_key(col[1], t)), etc1_key_min(etc1_key(col[2], t), etc1_key(col[3], t)));
        // the chosen colour is selected field by field (a dynamically indexed local array would live in scratch memory on the GPU, and
        // conditional struct copies become branches)
        const bool odd = ((int)m & 1) != 0, upper = ((int)m & 2) != 0;
        etc1_colour ch;
        ch.a = upper ? (odd ? col[3].a : col[2].a) : (odd ? col[1].a : col[0].a);
        ch.b = upper ? (odd ? col[3].b : col[2].b) : (odd ? col[1].b : col[0].b);
        ch.c = upper ? (odd ? col[3].c : col[2].c) : (odd ? col[1].c : col[0].c);
        ch.n = upper ? (odd ? col[3].n : col[2].n) : (odd ? col[1].n : col[0].n);
        err4 += etc1_key_value(etc1_key(ch, unpack_ycc_source(src.t[ti])));
    }
    err_out = __builtin_fma(err4, 0.25, msrc.sn);
}

struct etc1_search { double best_err; etc1_hint best; };

// The ETC1 bias list repeats itself per sub-block: apply_etc1_bias (transcoder.cpp:16547-16612) moves sub-block `s` of bias b by a delta vector
// that many biases share (most touch one sub-block only). ku_bias_slot[order][s][i] numbers the distinct vectors in list order (order 0 =
// the sorted list, 1 = 0..31); bit i of ku_bias_first[order][s] is set where a vector occurs for the first time. (generated: see the note in
// etc1_trials)
BU_TAB unsigned char ku_bias_slot[2][2][32] = {
    { { 0, 1, 2, 0, 1, 3, 4, 5, 0, 4, 0, 6, 7, 3, 8, 9, 2, 10, 11, 12, 13, 14, 4, 0, 15, 0, 0, 0, 0, 13, 10, 12 }, { 0, 1, 2, 3, 0, 4, 3, 5, 1, 0, 2, 6, 7, 0, 8, 0, 0, 0, 9, 0, 0, 10, 1, 4, 11, 12, 13, 14, 15, 13, 14, 15 } },
    { { 0, 1, 2, 3, 4, 5, 4, 6, 7, 8, 9, 10, 2, 7, 6, 11, 10, 12, 7, 7, 7, 7, 11, 13, 7, 14, 15, 0, 15, 7, 7, 15 }, { 0, 1, 2, 3, 4, 2, 2, 2, 5, 6, 7, 2, 8, 2, 9, 2, 10, 11, 8, 12, 4, 9, 5, 13, 10, 14, 15, 2, 0, 15, 0, 2 } },
};
BU_TAB unsigned int ku_bias_first[2][2] = { { 0x013ED8E7u, 0x1F2458AFu }, { 0x06828FBFu, 0x068B571Fu } };
// The same lists as 64-bit literals (a byte / a nibble per entry): the search reads them once per bias, and a load from a constant array there is a memory wait in
// front of every trial (the index is wave-uniform, so the literal form is a few scalar shifts).
BU_FN uint32_t etc1_bias_in_order(uint32_t i) {
    const uint64_t w = i < 8 ? 0x091a0c1b1d16000dull : (i < 16 ? 0x051702190a081f1eull : (i < 24 ? 0x121c11060b03070full : 0x100e041815141301ull));
    return (uint32_t)(w >> (8 * (i & 7))) & 255u;
}
BU_FN uint32_t etc1_bias_slot(uint32_t order, uint32_t sub, uint32_t i) {
    const uint64_t lo = order ? (sub ? 0x2928276522243210ull : 0xb672a98764543210ull) : (sub ? 0x0807620153403210ull : 0x9837604054310210ull);
    const uint64_t hi = order ? (sub ? 0x20f02fead594c8baull : 0xf77f0fe7db7777caull) : (sub ? 0xfedfedcb41a00900ull : 0xcad0000f04edcba2ull);
    return (uint32_t)((i < 16 ? lo : hi) >> (4 * (i & 15))) & 15u;
}
// Per-block scratch for the repeats: 32 entries (16 per sub-block) of {error, table}; entry i of this block lives at [i * stride] (the GPU
// keeps it in LDS, one column per lane). Optional: without it every trial is evaluated from scratch, with the same result.
// The struct travels by value (a pointer to it would point into scratch memory on the GPU: one more dependent load per use). The pointers stay generic: declaring
// them as LDS pointers (address_space(3), ds_read / ds_write instead of flat accesses) measured 10 % SLOWER for the finish kernel (tools/uastc_time.py, A/B on one box).
struct hint_cache { double* err; unsigned char* table; unsigned int stride; };   // err == nullptr: no cache
BU_FN hint_cache no_hint_cache() { hint_cache c = { nullptr, nullptr, 0 }; return c; }

template <int FLIP>
BU_FN void etc1_trials(uint32_t mode, const rgba8* decoded, const texels_ycc& dec, const texels_packed& src, const enc_cfg& e, uint32_t last_individ,
                       uint32_t last_bias, bool sorted_table, etc1_search& out, hint_cache cache) {
    const bool has_bias = ku_mode_has_etc1_bias[mode] != 0;
    const uint32_t order = sorted_table ? 0u : 1u;
    etc1_subblock_stats st[2];
    etc1_stats<FLIP, 0>(decoded, st[0]);
    etc1_stats<FLIP, 1>(decoded, st[1]);
    etc1_moments mdec[2], msrc[2];
    for (int sub = 0; sub < 2; sub++) { mdec[sub].sy = mdec[sub].scb = mdec[sub].scr = 0; mdec[sub].sn = 0.0; msrc[sub] = mdec[sub]; }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int j = 0; j < 8; j++) {
        etc1_moments_add(mdec[0], dec.t[etc1_texel<FLIP, 0>(j)]);
        etc1_moments_add(mdec[1], dec.t[etc1_texel<FLIP, 1>(j)]);
        etc1_moments_add(msrc[0], unpack_ycc_source(src.t[etc1_texel<FLIP, 0>(j)]));
        etc1_moments_add(msrc[1], unpack_ycc_source(src.t[etc1_texel<FLIP, 1>(j)]));
    }
    for (uint32_t individ = 0; individ < last_individ; individ++) {
        const uint32_t mul = individ ? 15 : 31;
        uint32_t unbiased[2][3];
        for (uint32_t sub = 0; sub < 2; sub++)
            for (uint32_t c = 0; c < 3; c++) unbiased[sub][c] = (st[sub].sum[c] * mul + 1020) / (8 * 255);
        for (uint32_t bi = 0; bi < last_bias; bi++) {
            // 0 should come first, but 13 is the (0,0,0) bias (uastc_enc.cpp:2732-2733)
            const uint32_t bias = sorted_table ? etc1_bias_in_order(bi) : bi;
            int base[2][3];
            for (uint32_t c = 0; c < 3; c++) {
                const uint32_t c0 = has_bias ? etc1_bias_apply(unbiased[0][c], c, bias, mul, 0) : unbiased[0][c];
                const uint32_t c1 = has_bias ? etc1_bias_apply(unbiased[1][c], c, bias, mul, 1) : unbiased[1][c];
                if (individ) {
                    base[0][c] = (int)((c0 << 4) | c0);
                    base[1][c] = (int)((c1 << 4) | c1);
                } else {
                    const uint32_t c1d = (uint32_t)((int)c0 + clampi((int)c1 - (int)c0, -4, 3));  // set_block_color5_clamp, etc.h:674-690
                    base[0][c] = (int)((c0 << 3) | (c0 >> 2));
                    base[1][c] = (int)((c1d << 3) | (c1d >> 2));
                }
            }
            uint32_t limit[2];
            for (uint32_t sub = 0; sub < 2; sub++) {
                int range = 0;
                for (uint32_t c = 0; c < 3; c++) {
                    const int pos = st[sub].mx[c] - base[sub][c], neg = base[sub][c] - st[sub].mn[c];
                    const int ap = pos < 0 ? -pos : pos, an = neg < 0 ? -neg : neg;
                    range = ap > range ? ap : range;
                    range = an > range ? an : range;
                }
                limit[sub] = e.level == 4 ? 8 : (range > 51 ? 8 : (range >= 7 ? 4 : 2));
            }
            // Sub-block 0's colour depends on its own delta vector only; sub-block 1's too in individual mode (in differential mode it is
            // coded relative to sub-block 0). A vector seen before in this (flip, mode) pass gives the same table and error: reuse them.
            uint32_t t0, t1;
            double e0, e1;
            const bool reuse = cache.err != nullptr && has_bias;
            const uint32_t s0 = etc1_bias_slot(order, 0, bi), s1 = 16u + etc1_bias_slot(order, 1, bi);
            if (reuse && !(((order ? 0x06828FBFu : 0x013ED8E7u) >> bi) & 1u)) {
                t0 = cache.table[s0 * cache.stride]; e0 = cache.err[s0 * cache.stride];
            } else {
                etc1_fit_subblock<FLIP, 0>(dec, src, mdec[0], msrc[0], base[0], limit[0], t0, e0);
                if (reuse) { cache.table[s0 * cache.stride] = (unsigned char)t0; cache.err[s0 * cache.stride] = e0; }
            }
            if (reuse && individ && !(((order ? 0x068B571Fu : 0x1F2458AFu) >> bi) & 1u)) {
                t1 = cache.table[s1 * cache.stride]; e1 = cache.err[s1 * cache.stride];
            } else {
                etc1_fit_subblock<FLIP, 1>(dec, src, mdec[1], msrc[1], base[1], limit[1], t1, e1);
                if (reuse && individ) { cache.table[s1 * cache.stride] = (unsigned char)t1; cache.err[s1 * cache.stride] = e1; }
            }
            const double err = e0 + e1;
            if (err < out.best_err) {
                out.best_err = err;
                out.best.flip = (uint8_t)FLIP; out.best.diff = (uint8_t)(individ == 0); out.best.inten0 = (uint8_t)t0; out.best.inten1 = (uint8_t)t1; out.best.bias = (uint8_t)bias;
            }
        }
    }
}

BU_FN_BIG void etc1_hints(uint32_t mode, const rgba8* px, const rgba8* decoded, const enc_cfg& e, etc1_hint& best, hint_cache cache = no_hint_cache()) {
    const bool faster = (e.flags & FLAG_ETC1_FASTER) != 0, fastest = (e.flags & FLAG_ETC1_FASTEST) != 0;
    const bool has_bias = ku_mode_has_etc1_bias[mode] != 0;
    uint32_t last_bias = 1;
    bool sorted_table = false;
    const bool flip_estimate = e.level <= 1 || faster || fastest;
    if (has_bias) {
        sorted_table = e.level <= 3;
        switch (e.level) {
        case 0: last_bias = fastest ? 1 : (faster ? 1 : 2); break;
        case 1: last_bias = fastest ? 1 : (faster ? 3 : 5); break;
        case 2: last_bias = fastest ? 1 : (faster ? 10 : 20); break;
        case 3: last_bias = fastest ? 1 : (faster ? 16 : 32); break;
        default: last_bias = 32; break;
        }
    }
    texels_ycc dec;
    texels_packed src;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int i = 0; i < 16; i++) { dec.t[i] = to_ycc(decoded[i].c[0], decoded[i].c[1], decoded[i].c[2]); src.t[i] = pack_ycc_source(px[i]); }
    uint32_t first_flip = 0, last_flip = 2, last_individ = 2;
    if (e.flags & FLAG_ETC1_NO_FLIP_INDIVIDUAL) { last_flip = 1; last_individ = 1; }
    else if (flip_estimate) { if (etc1_estimate_flipped(decoded)) first_flip = 1; last_flip = first_flip + 1; }
    etc1_search s;
    s.best_err = 1e300;
    s.best.flip = s.best.diff = s.best.inten0 = s.best.inten1 = s.best.bias = 0;
    if (first_flip == 0) etc1_trials<0>(mode, decoded, dec, src, e, last_individ, last_bias, sorted_table, s, cache);
    if (last_flip == 2 || first_flip == 1) etc1_trials<1>(mode, decoded, dec, src, e, last_individ, last_bias, sorted_table, s, cache);
    best = s.best;
}

// pack_etc1_block_solid_color (etc.cpp:181-257): diff, intensity table, selector and the packed base colour for a solid block
struct etc1_solid { uint8_t diff, inten, selector, r, g, b; };
BU_FN void etc1_solid_fit(const uint8_t* col, etc1_solid& out) {
    const uint32_t next[4] = { 1, 2, 0, 1 };
    uint32_t best_err = 0xFFFFFFFFu, best_i = 0, best_x = 0, best_c1 = 0, best_c2 = 0;
    bool done = false;
    for (uint32_t i = 0; i < 3 && !done; i++) {
        const uint32_t c1 = col[next[i]], c2 = col[next[i + 1]];
        for (int delta = -1; delta <= 1 && !done; delta++) {
            const int v = clampi((int)col[i] + delta, 0, 255);
            const unsigned short* t = ku_etc1_solid_cfg + ku_etc1_solid_cfg_ofs[v];
            do {
                const uint32_t x = *t++;
                const unsigned short* inv = ku_etc1_inverse + (x & 0xFF) * 256;
                const uint32_t p1 = inv[c1], p2 = inv[c2];
                const int d0 = v - (int)col[i];
                const uint32_t err = (uint32_t)(d0 * d0) + (p1 >> 8) * (p1 >> 8) + (p2 >> 8) * (p2 >> 8);
                if (err < best_err) {
                    best_err = err; best_x = x; best_c1 = p1 & 0xFF; best_c2 = p2 & 0xFF; best_i = i;
                    if (!best_err) { done = true; break; }
                }
            } while (*t != 0xFFFF);
        }
    }
    uint8_t rgb[3];
    rgb[best_i] = (uint8_t)((best_x >> 8) & 255);
    rgb[next[best_i]] = (uint8_t)best_c1;
    rgb[next[best_i + 1]] = (uint8_t)best_c2;
    out.diff = (uint8_t)(best_x & 1); out.inten = (uint8_t)((best_x >> 1) & 7); out.selector = (uint8_t)((best_x >> 4) & 3);
    out.r = rgb[0]; out.g = rgb[1]; out.b = rgb[2];
}

// ------------------------------------------------------------------------------------------------------------------
// pack_uastc (uastc_enc.cpp:110-466)
// ------------------------------------------------------------------------------------------------------------------

struct bit_writer {
    uint8_t* buf;
    uint32_t ofs;
};
BU_FN void put_bits(bit_writer& w, uint64_t code, uint32_t n) {
    while (n) {
        const uint32_t in_byte = w.ofs & 7;
        const uint32_t k = n < 8 - in_byte ? n : 8 - in_byte;
        w.buf[w.ofs >> 3] = (uint8_t)(w.buf[w.ofs >> 3] | (uint8_t)(code << in_byte));
        code >>= k;
        n -= k;
        w.ofs += k;
    }
}

BU_FN void pack_solid(const uint8_t* rgba, uint8_t* out16) {
    uint8_t buf[32];
    for (uint32_t i = 0; i < 32; i++) buf[i] = 0;
    bit_writer w = { buf, 0 };
    put_bits(w, ku_mode_code[8], ku_mode_code_len[8]);
    for (uint32_t c = 0; c < 4; c++) put_bits(w, rgba[c], 8);
    etc1_solid s;
    etc1_solid_fit(rgba, s);
    put_bits(w, s.diff, 1); put_bits(w, s.inten, 3); put_bits(w, s.selector, 2);
    put_bits(w, s.r, 5); put_bits(w, s.g, 5); put_bits(w, s.b, 5);
    for (uint32_t i = 0; i < 16; i++) out16[i] = buf[i];
}

// `norm` must already be anchor-normalised
BU_FN void pack_block(const cand& norm, const etc1_hint& etc1, uint32_t eac_table, uint32_t eac_mul, bool hint0, bool hint1, uint8_t* out16) {
    const uint32_t mode = norm.mode, subsets = ku_mode_subsets[mode], planes = ku_mode_planes[mode], comps = ku_mode_comps[mode];
    const uint32_t wbits = ku_mode_weight_bits[mode], range = ku_mode_endpoint_ranges[mode];
    uint8_t buf[32];
    for (uint32_t i = 0; i < 32; i++) buf[i] = 0;
    bit_writer w = { buf, 0 };
    put_bits(w, ku_mode_code[mode], ku_mode_code_len[mode]);
    if (ku_mode_has_bc1_hint0[mode]) put_bits(w, hint0 ? 1 : 0, 1);
    if (ku_mode_has_bc1_hint1[mode]) put_bits(w, hint1 ? 1 : 0, 1);
    put_bits(w, etc1.flip, 1); put_bits(w, etc1.diff, 1); put_bits(w, etc1.inten0, 3); put_bits(w, etc1.inten1, 3);
    if (ku_mode_has_etc1_bias[mode]) put_bits(w, etc1.bias, 5);
    if (ku_mode_has_alpha[mode]) put_bits(w, eac_table | (eac_mul << 4), 8);
    if (subsets == 3) put_bits(w, norm.pattern, 4);
    else if (subsets == 2) put_bits(w, norm.pattern, 5);
    if (planes == 2 && mode != 17) put_bits(w, norm.ccs, 2);

    const uint32_t total_values = comps * 2 * subsets;
    const uint32_t ep_bits = ku_bise[range * 3], ep_trits = ku_bise[range * 3 + 1], ep_quints = ku_bise[range * 3 + 2];
    uint32_t tq_accum = 0, tq_mul = 1;
    for (uint32_t i = 0; i < total_values; i++) {
        const uint32_t tq = norm.endpoints[i] >> ep_bits;
        if (ep_trits) {
            tq_accum += tq * tq_mul; tq_mul *= 3;
            if (tq_mul == 243) { put_bits(w, tq_accum, 8); tq_accum = 0; tq_mul = 1; }
        } else if (ep_quints) {
            tq_accum += tq * tq_mul; tq_mul *= 5;
            if (tq_mul == 125) { put_bits(w, tq_accum, 7); tq_accum = 0; tq_mul = 1; }
        }
    }
    if (tq_mul > 1) {
        uint32_t nb;
        if (ep_trits) nb = tq_mul == 3 ? 2 : (tq_mul == 9 ? 4 : (tq_mul == 27 ? 5 : 7));
        else nb = tq_mul == 5 ? 3 : 5;
        put_bits(w, tq_accum, nb);
    }
    for (uint32_t i = 0; i < total_values; i++) put_bits(w, norm.endpoints[i] & ((1u << ep_bits) - 1), ep_bits);

    const uint8_t* anchors = subsets == 3 ? ku_anchor3 + norm.pattern * 3 : (mode == 7 ? ku_anchor7 + norm.pattern * 3 : ku_anchor2 + norm.pattern * 3);
    const uint32_t plane_shift = planes == 2 ? 1 : 0;
    for (uint32_t i = 0; i < 16 * planes; i++) {
        uint32_t nb = wbits;
        for (uint32_t s = 0; s < subsets; s++)
            if ((subsets >= 2 ? anchors[s] : 0u) == (i >> plane_shift)) { nb--; break; }
        put_bits(w, norm.weights[i], nb);
    }
    for (uint32_t i = 0; i < 16; i++) out16[i] = buf[i];
}

// ------------------------------------------------------------------------------------------------------------------
// Whole-block drivers
// ------------------------------------------------------------------------------------------------------------------

// hints + packing of the chosen candidate (uastc_enc.cpp:3550-3644)
BU_FN_BIG void finish_block(const rgba8* px, const enc_cfg& e, const cand& chosen, uint8_t* out16, hint_cache cache = no_hint_cache()) {
    cand best = chosen;
    rgba8 decoded[16];
    decode_uastc(best, decoded);
    normalise_anchors(best);
    bool h0 = false, h1 = false;
    if (e.bc1_hints) {
        uint32_t spx[16], dpx[16];   // source and decoded texels as packed dwords, in registers for the whole hint computation
        BU_UNROLL
        for (int i = 0; i < 16; i++) { spx[i] = pack_px(px[i].c); dpx[i] = pack_px(decoded[i].c); }
        bc1_hints(best, spx, dpx, h0, h1);
    }
    uint32_t eac_table = 0, eac_mul = 0;
    if (ku_mode_has_alpha[best.mode]) eac_a8_hint(decoded, e.eac_mul_rad, e.eac_table_mask, eac_table, eac_mul);
    etc1_hint eh;
    etc1_hints(best.mode, px, decoded, e, eh, cache);
    pack_block(best, eh, eac_table, eac_mul, h0, h1, out16);
}

struct array_view {  // choose_candidate accessor over plain arrays
    const cand* slots; const cand_score* score;
    BU_FN_MEMBER bool valid(uint32_t i) const { return slots[i].valid != 0; }
    BU_FN_MEMBER uint64_t overall(uint32_t i) const { return score[i].overall; }
    BU_FN_MEMBER float rms(uint32_t i) const { return score[i].uastc_rms; }
    BU_FN_MEMBER uint32_t mode(uint32_t i) const { return slots[i].mode; }
};

// encode_uastc (uastc_enc.cpp:3126) for one block, everything in one call (host tests; the GPU splits it into jobs)
BU_FN_BIG void encode_block(const uint8_t* rgba64, uint32_t flags, uint8_t* out16, cand* scratch /* MAX_SLOTS */) {
    const rgba8* px = (const rgba8*)rgba64;
    enc_cfg e;
    make_cfg(flags, e);
    const uint32_t cls = classify(px, e);
    if (cls & CLS_SOLID) { pack_solid(rgba64, out16); return; }
    const uint32_t n_slots = total_slots(e);
    for (uint32_t i = 0; i < n_slots; i++) scratch[i].valid = 0;
    for (uint32_t i = 0; i < 18; i++) {
        const uint32_t m = mode_order(i), nv = mode_variants(m, e);
        if (nv && mode_applies(m, cls, e)) run_mode(m, px, e, scratch + slot_base(m, e), 0, nv);
    }
    cand_score score[MAX_SLOTS];
    for (uint32_t i = 0; i < n_slots; i++) {
        score[i].overall = 0; score[i].uastc_rms = 0;
        if (scratch[i].valid) score[i] = score_candidate(scratch[i], px, cls, e);
    }
    array_view v = { scratch, score };
    double cache_err[32];
    unsigned char cache_table[32];
    const hint_cache hc = { cache_err, cache_table, 1 };
    finish_block(px, e, scratch[choose_candidate(v, n_slots, e)], out16, hc);
}

}  // namespace bu_uastc
