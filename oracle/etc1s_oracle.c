/* oracle/etc1s_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference's ETC1S frontend hot path, written from the behaviour of
 * /root/reference (citations are encoder/... relative to the reference root). It is the checker the HIP path is diffed
 * against on the GPU box, where /root/reference does not exist. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (basis_universal_amd/) never does.
 *
 * Parity is PINNED: tests/test_oracle_vs_reference.py checks every function here bit-for-bit against the real reference
 * (oracle/_ref/libref_harness.so, built from the reference sources) on Kodak images and seeded synthetic inputs, and
 * against the committed fixtures under tests/golden/ that were produced by that same reference build.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off (no FMA contraction: the reference's x86-64 baseline build has none, and the
 * few float expressions below are rounding-sensitive -- SURVEY hazard H4).
 */
#include "etc1s_oracle.h"
#include "etc1s_tables.h"
#include <stdlib.h>
#include <string.h>

/* ---- constants (encoder/basisu_etc.cpp:304-311, 339-361) */
static const int k_inten[8][4] = {
    {-8, -2, 2, 8},     {-17, -5, 5, 17},   {-29, -9, 9, 29},    {-42, -13, 13, 42},
    {-60, -18, 18, 60}, {-80, -24, 24, 80}, {-106, -33, 33, 106}, {-183, -47, 47, 183}};
/* selector index (into k_inten rows) -> raw ETC1 2-bit code */
static const uint8_t k_sel_to_raw[4] = {3, 2, 0, 1};
static const uint8_t k_raw_to_sel[4] = {2, 3, 1, 0};

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int scale5(int c) { return (c << 3) | (c >> 2); }

/* ---- enc.h:1141-1195, 1075-1106 */
uint32_t orc_color_distance(int perceptual, const uint8_t* a, const uint8_t* b) {
    int dr = (int)a[0] - (int)b[0], dg = (int)a[1] - (int)b[1], db = (int)a[2] - (int)b[2];
    if (!perceptual) return (uint32_t)(dr * dr + dg * dg + db * db);
    int dl = dr * 14 + dg * 45 + db * 5;
    int dcr = dr * 64 - dl;
    int dcb = db * 64 - dl;
    return ((uint32_t)(dl * dl) >> 5) + ((((uint32_t)(dcr * dcr) >> 5) * 26u) >> 7) + ((((uint32_t)(dcb * dcb) >> 5) * 3u) >> 7);
}

/* ---- transcoder/basisu_transcoder.cpp:355-409 specialised to len == 3 (zero 4-byte words, 3 tail bytes) */
uint32_t orc_hash_hsieh3(uint8_t r, uint8_t g, uint8_t b) {
    uint32_t h = 3;
    h += (uint32_t)r | ((uint32_t)g << 8);
    h ^= h << 16;
    h ^= ((uint32_t)(int32_t)(int8_t)b) << 18;
    h += h >> 11;
    h ^= h << 3;  h += h >> 5;
    h ^= h << 4;  h += h >> 17;
    h ^= h << 25; h += h >> 6;
    return h;
}

/* ---- etc_block bit layout (etc.h:91-330): the 8 bytes are a big-endian u64 */
static uint64_t pack_etc1s(int r5, int g5, int b5, int inten, const uint8_t sel[16] /* [y*4+x], inten-table indices */) {
    uint64_t v = ((uint64_t)r5 << 59) | ((uint64_t)g5 << 51) | ((uint64_t)b5 << 43) | ((uint64_t)inten << 37) | ((uint64_t)inten << 34) |
                 (1ull << 33) /* diff */ | (1ull << 32) /* flip */;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            uint32_t raw = k_sel_to_raw[sel[y * 4 + x]];
            int bit = x * 4 + y;
            v |= (uint64_t)(raw & 1) << bit;
            v |= (uint64_t)(raw >> 1) << (16 + bit);
        }
    return v;
}
static void store_be64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (56 - 8 * i)); }
static uint64_t load_be64(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v; }
static void unpack_etc1s(const uint8_t* blk, int* r5, int* g5, int* b5, int* inten, uint8_t sel[16]) {
    uint64_t v = load_be64(blk);
    *r5 = (int)((v >> 59) & 31); *g5 = (int)((v >> 51) & 31); *b5 = (int)((v >> 43) & 31);
    *inten = (int)((v >> 37) & 7);
    if (sel)
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int bit = x * 4 + y;
                uint32_t raw = (uint32_t)((v >> bit) & 1) | ((uint32_t)((v >> (16 + bit)) & 1) << 1);
                sel[y * 4 + x] = k_raw_to_sel[raw];
            }
}
static void block_colors5(uint8_t out[4][3], int r5, int g5, int b5, int inten) {
    int r = scale5(r5), g = scale5(g5), b = scale5(b5);
    for (int s = 0; s < 4; s++) {
        int yd = k_inten[inten][s];
        out[s][0] = (uint8_t)clampi(r + yd, 0, 255);
        out[s][1] = (uint8_t)clampi(g + yd, 0, 255);
        out[s][2] = (uint8_t)clampi(b + yd, 0, 255);
    }
}

/* ---- etc1_optimizer (etc.cpp:948-1278) */
typedef struct {
    const uint8_t* px; uint32_t n; int quality, perceptual;
    float avg[3]; int spread;
    uint8_t bloom[128];
    /* best solution */
    int valid; uint64_t err; int r, g, b, inten;
    uint8_t *best_sel, *trial_sel, *tmp_sel;
    /* fast path */
    uint16_t* luma;
} opt_t;

/* check_for_redundant_solution (etc.cpp:1072-1089): 1024-bit Bloom filter, k=2. Returns 1 if definitely new. */
static int bloom_test_and_set(opt_t* o, int r, int g, int b) {
    uint32_t kh = orc_hash_hsieh3((uint8_t)r, (uint8_t)g, (uint8_t)b);
    uint32_t h0 = kh & 1023, h1 = (kh >> 10) & 1023;
    if ((o->bloom[h0 >> 3] & (1 << (h0 & 7))) && (o->bloom[h1 >> 3] & (1 << (h1 & 7)))) return 0;
    o->bloom[h0 >> 3] |= (uint8_t)(1 << (h0 & 7));
    o->bloom[h1 >> 3] |= (uint8_t)(1 << (h1 & 7));
    return 1;
}

/* evaluate_solution_slow (etc.cpp:1104-1278) and evaluate_solution_fast (etc.cpp:1280-1506). Returns 1 if best improved. */
static int evaluate(opt_t* o, int r5, int g5, int b5) {
    if (!bloom_test_and_set(o, r5, g5, b5)) return 0;
    const int br = scale5(r5), bg = scale5(g5), bb = scale5(b5);
    uint64_t trial_err; int trial_inten = 0, trial_valid = 0;
    if (o->quality >= ORC_QUALITY_MEDIUM) {
        trial_err = (uint64_t)INT64_MAX;
        for (int t = 0; t < 8; t++) {
            if (o->quality <= ORC_QUALITY_MEDIUM && !((orc_inten_enable_by_spread[o->spread] >> t) & 1)) continue;
            uint8_t bc[4][3];
            for (int s = 0; s < 4; s++) {
                int yd = k_inten[t][s];
                bc[s][0] = (uint8_t)clampi(br + yd, 0, 255); bc[s][1] = (uint8_t)clampi(bg + yd, 0, 255); bc[s][2] = (uint8_t)clampi(bb + yd, 0, 255);
            }
            uint64_t total = 0;
            for (uint32_t i = 0; i < o->n; i++) {
                const uint8_t* p = o->px + 4 * i;
                uint32_t be = orc_color_distance(o->perceptual, p, bc[0]); uint32_t bs = 0;
                for (uint32_t s = 1; s < 4; s++) { uint32_t e = orc_color_distance(o->perceptual, p, bc[s]); if (e < be) { be = e; bs = s; } }
                o->tmp_sel[i] = (uint8_t)bs;
                total += be;
            }
            if (total < trial_err) { trial_err = total; trial_inten = t; trial_valid = 1; uint8_t* sw = o->trial_sel; o->trial_sel = o->tmp_sel; o->tmp_sel = sw; }
        }
    } else {
        /* fast: linear metric forced (etc.cpp:1313), selectors by luma midpoints, tables scanned 7..0 */
        trial_err = UINT64_MAX;
        uint32_t lmin = 0xFFFFFFFFu, lmax = 0;
        for (uint32_t i = 0; i < o->n; i++) { if (o->luma[i] < lmin) lmin = o->luma[i]; if (o->luma[i] > lmax) lmax = o->luma[i]; }
        for (int t = 7; t >= 0; --t) {
            uint8_t bc[4][3]; uint32_t bi[4];
            for (int s = 0; s < 4; s++) {
                int yd = k_inten[t][s];
                bc[s][0] = (uint8_t)clampi(br + yd, 0, 255); bc[s][1] = (uint8_t)clampi(bg + yd, 0, 255); bc[s][2] = (uint8_t)clampi(bb + yd, 0, 255);
                bi[s] = (uint32_t)bc[s][0] + bc[s][1] + bc[s][2];
            }
            const uint32_t mid[3] = {bi[0] + bi[1], bi[1] + bi[2], bi[2] + bi[3]};
            uint64_t total = 0;
            if (lmax * 2 < mid[0]) {
                if (bi[0] > lmax) { uint32_t me = bi[0] - lmax; if ((uint64_t)me >= trial_err) continue; }
                for (uint32_t i = 0; i < o->n; i++) { o->tmp_sel[i] = 0; total += orc_color_distance(0, bc[0], o->px + 4 * i); }
            } else if (lmin * 2 >= mid[2]) {
                if (lmin > bi[3]) { uint32_t me = lmin - bi[3]; if ((uint64_t)me >= trial_err) continue; }
                for (uint32_t i = 0; i < o->n; i++) { o->tmp_sel[i] = 3; total += orc_color_distance(0, bc[3], o->px + 4 * i); }
            } else {
                /* the reference walks pixels in sorted-luma order; the selector of a pixel only depends on its own luma */
                for (uint32_t i = 0; i < o->n; i++) {
                    uint32_t y2 = (uint32_t)o->luma[i] * 2, s = 0;
                    while (s < 3 && y2 >= mid[s]) s++;
                    o->tmp_sel[i] = (uint8_t)s;
                    total += orc_color_distance(0, bc[s], o->px + 4 * i);
                }
            }
            if (total < trial_err) {
                trial_err = total; trial_inten = t; trial_valid = 1; uint8_t* sw = o->trial_sel; o->trial_sel = o->tmp_sel; o->tmp_sel = sw;
                if (!total) break;
            }
        }
    }
    if (trial_err < o->err) {
        o->err = trial_err; o->r = r5; o->g = g5; o->b = b5; o->inten = trial_inten; o->valid = trial_valid;
        uint8_t* sw = o->best_sel; o->best_sel = o->trial_sel; o->trial_sel = sw;
        return 1;
    }
    return 0;
}

int orc_etc1_optimize(const uint8_t* rgba, uint32_t n, int quality, int perceptual,
                      uint8_t out_color5[3], uint32_t* out_inten, uint64_t* out_err, uint8_t* out_selectors) {
    opt_t o; memset(&o, 0, sizeof(o));
    o.px = rgba; o.n = n; o.quality = quality; o.perceptual = perceptual;
    uint8_t* selbuf = (uint8_t*)malloc((size_t)n * 3 + 3);
    o.best_sel = selbuf; o.trial_sel = selbuf + n + 1; o.tmp_sel = selbuf + 2 * (size_t)n + 2;
    o.luma = (quality == ORC_QUALITY_FAST) ? (uint16_t*)malloc(sizeof(uint16_t) * (n ? n : 1)) : NULL;
    /* init (etc.cpp:998-1070): float running sums in pixel order, then a true division by n */
    float sr = 0.0f, sg = 0.0f, sb = 0.0f;
    int mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t* p = rgba + 4 * i;
        for (int c = 0; c < 3; c++) { if (p[c] < mn[c]) mn[c] = p[c]; if (p[c] > mx[c]) mx[c] = p[c]; }
        sr += (float)p[0]; sg += (float)p[1]; sb += (float)p[2];
        if (o.luma) o.luma[i] = (uint16_t)(p[0] + p[1] + p[2]);
    }
    o.avg[0] = sr / (float)n; o.avg[1] = sg / (float)n; o.avg[2] = sb / (float)n;
    int spread = mx[0] - mn[0]; if (mx[1] - mn[1] > spread) spread = mx[1] - mn[1]; if (mx[2] - mn[2] > spread) spread = mx[2] - mn[2];
    o.spread = spread;
    const int limit = 31;
    int br = clampi((int)(uint32_t)(o.avg[0] * limit / 255.0f + .5f), 0, limit);
    int bg = clampi((int)(uint32_t)(o.avg[1] * limit / 255.0f + .5f), 0, limit);
    int bb = clampi((int)(uint32_t)(o.avg[2] * limit / 255.0f + .5f), 0, limit);
    o.valid = 0; o.err = UINT64_MAX;
    /* compute -> compute_internal_cluster_fit (etc.cpp:776-792, 948-996) */
    uint32_t perms = quality == ORC_QUALITY_FAST ? 4 : quality == ORC_QUALITY_MEDIUM ? 16 : quality == ORC_QUALITY_SLOW ? 64 : 165;
    evaluate(&o, br, bg, bb);
    if (o.err != 0 && o.valid) {
        for (uint32_t i = 0; i < perms; i++) {
            const int base[3] = {scale5(o.r), scale5(o.g), scale5(o.b)};
            int dsum[3] = {0, 0, 0};
            for (int q = 0; q < 4; q++) {
                int cnt = (int)((orc_cluster_fit_order[i] >> (8 * q)) & 255), yd = k_inten[o.inten][q];
                for (int c = 0; c < 3; c++) dsum[c] += cnt * (clampi(base[c] + yd, 0, 255) - base[c]);
            }
            if (!dsum[0] && !dsum[1] && !dsum[2]) continue;
            int t[3];
            for (int c = 0; c < 3; c++) {
                float avg_delta = (float)dsum[c] / 8;
                t[c] = clampi((int)(int32_t)((o.avg[c] - avg_delta) * limit / 255.0f + .5f), 0, limit);
            }
            evaluate(&o, t[0], t[1], t[2]);
            if (o.err == 0) break;
        }
    }
    int ok = o.valid;
    if (ok) {
        out_color5[0] = (uint8_t)o.r; out_color5[1] = (uint8_t)o.g; out_color5[2] = (uint8_t)o.b;
        *out_inten = (uint32_t)o.inten; *out_err = o.err;
        if (out_selectors) memcpy(out_selectors, o.best_sel, n);
    }
    free(selbuf); free(o.luma);
    return ok;
}

static int level_to_block_quality(int level) { return level == 0 ? ORC_QUALITY_FAST : level == 1 ? ORC_QUALITY_MEDIUM : level == 6 ? ORC_QUALITY_UBER : ORC_QUALITY_SLOW; }
/* frontend.cpp:1530-1533: levels 0,1 medium; 6 uber; others slow (the basis_etc1_pack_params default) */
static int level_to_cluster_quality(int level) { return level <= 1 ? ORC_QUALITY_MEDIUM : level == 6 ? ORC_QUALITY_UBER : ORC_QUALITY_SLOW; }

void orc_encode_etc1s_blocks(const uint8_t* pixel_blocks, uint32_t n_blocks, int comp_level, int perceptual, uint8_t* out_blocks) {
    for (uint32_t i = 0; i < n_blocks; i++) {
        uint8_t c[3], sel[16]; uint32_t inten; uint64_t err;
        if (!orc_etc1_optimize(pixel_blocks + 64ull * i, 16, level_to_block_quality(comp_level), perceptual, c, &inten, &err, sel)) abort();
        store_be64(out_blocks + 8ull * i, pack_etc1s(c[0], c[1], c[2], (int)inten, sel));
    }
}

void orc_determine_selectors(const uint8_t* pixel_blocks, uint32_t n_blocks, const uint8_t* color5_inten, int perceptual, uint8_t* out_blocks) {
    for (uint32_t i = 0; i < n_blocks; i++) {
        const uint8_t* ci = color5_inten + 4ull * i;
        uint8_t bc[4][3], sel[16];
        block_colors5(bc, ci[0], ci[1], ci[2], ci[3]);
        for (int p = 0; p < 16; p++) {
            const uint8_t* px = pixel_blocks + 64ull * i + 4 * p;
            uint32_t be = orc_color_distance(perceptual, bc[0], px), bs = 0;
            for (uint32_t s = 1; s < 4; s++) { uint32_t e = orc_color_distance(perceptual, bc[s], px); if (e < be) { be = e; bs = s; } }
            sel[p] = (uint8_t)bs;
        }
        store_be64(out_blocks + 8ull * i, pack_etc1s(ci[0], ci[1], ci[2], ci[3], sel));
    }
}

/* g_etc1_pixel_indices[flip=1][subblock][i] (etc.cpp:352-361): subblock 0 = rows 0-1, subblock 1 = rows 2-3, row-major */
static void gather_cluster_pixels(const uint8_t* pixel_blocks, const uint32_t* idx, uint32_t count, uint8_t* dst) {
    for (uint32_t k = 0; k < count; k++) {
        uint32_t block = idx[k] >> 1, sub = idx[k] & 1;
        memcpy(dst + 32ull * k, pixel_blocks + 64ull * block + 32 * sub, 32);
    }
}

void orc_generate_endpoint_codebook(const uint8_t* pixel_blocks, uint32_t n_clusters, const uint32_t* offsets, const uint32_t* indices,
                                    int comp_level, int perceptual, uint32_t step, uint8_t* params, uint64_t* err, uint8_t* valid) {
    for (uint32_t ci = 0; ci < n_clusters; ci++) {
        const uint32_t cnt = offsets[ci + 1] - offsets[ci], npx = cnt * 8;
        uint8_t* px = (uint8_t*)malloc(32ull * (cnt ? cnt : 1));
        gather_cluster_pixels(pixel_blocks, indices + offsets[ci], cnt, px);
        uint8_t c[3]; uint32_t inten; uint64_t e;
        if (!orc_etc1_optimize(px, npx, level_to_cluster_quality(comp_level), perceptual, c, &inten, &e, NULL)) abort();
        int use_new = 0;
        if (!step || !valid[ci]) use_new = 1;
        else {
            uint8_t bc[4][3];
            block_colors5(bc, params[4 * ci], params[4 * ci + 1], params[4 * ci + 2], params[4 * ci + 3]);
            uint64_t prev = 0;
            for (uint32_t i = 0; i < npx; i++) {
                uint32_t be = 0xFFFFFFFFu;
                for (int s = 0; s < 4; s++) { uint32_t d = orc_color_distance(perceptual, px + 4 * i, bc[s]); if (d < be) be = d; }
                prev += be;
            }
            if (prev > e) use_new = 1;
        }
        if (use_new) { params[4 * ci] = c[0]; params[4 * ci + 1] = c[1]; params[4 * ci + 2] = c[2]; params[4 * ci + 3] = (uint8_t)inten; err[ci] = e; valid[ci] = 1; }
        free(px);
    }
}

void orc_refine_endpoint_clusterization(const uint8_t* pixel_blocks, uint32_t n_blocks, const uint32_t* block_cluster,
                                        const uint8_t* cluster_params, uint32_t n_clusters,
                                        uint32_t n_parents, const uint32_t* cand_offsets, const uint32_t* cand_indices, const uint8_t* block_parent,
                                        int perceptual, uint32_t* out_best_cluster) {
    for (uint32_t bi = 0; bi < n_blocks; bi++) {
        const uint32_t cur = block_cluster[bi];
        const int cur_inten = cluster_params[4 * cur + 3];
        uint32_t first = 0, total = n_clusters;
        if (n_parents) { first = cand_offsets[block_parent[bi]]; total = cand_offsets[block_parent[bi] + 1] - first; }
        uint64_t best_err = (uint64_t)INT64_MAX; uint32_t best = 0;
        for (uint32_t k = 0; k < total; k++) {
            const uint32_t ci = n_parents ? cand_indices[first + k] : k;
            const uint8_t* cp = cluster_params + 4 * ci;
            uint64_t tot = 0;
            if (cp[3] > cur_inten) tot = (uint64_t)INT64_MAX;
            else {
                uint8_t bc[4][3];
                block_colors5(bc, cp[0], cp[1], cp[2], cp[3]);
                for (int p = 0; p < 16; p++) {
                    const uint8_t* px = pixel_blocks + 64ull * bi + 4 * p;
                    uint32_t be = 0xFFFFFFFFu;
                    for (int s = 0; s < 4; s++) { uint32_t d = orc_color_distance(perceptual, px, bc[s]); if (d < be) be = d; }
                    tot += be;
                }
            }
            if (tot < best_err || (ci == cur && tot == best_err)) { best_err = tot; best = ci; if (!best_err) break; }
        }
        out_best_cluster[bi] = best;
    }
}

void orc_create_optimized_selector_codebook(const uint8_t* pixel_blocks, const uint8_t* encoded_blocks, uint32_t n_clusters,
                                            const uint32_t* offsets, const uint32_t* block_indices, int perceptual, uint8_t* inout_selector_blocks) {
    for (uint32_t ci = 0; ci < n_clusters; ci++) {
        const uint32_t cnt = offsets[ci + 1] - offsets[ci];
        if (!cnt) continue;
        uint64_t tot[16][4]; memset(tot, 0, sizeof(tot));
        for (uint32_t k = 0; k < cnt; k++) {
            const uint32_t bi = block_indices[offsets[ci] + k];
            int r5, g5, b5, inten; uint8_t bc[4][3];
            unpack_etc1s(encoded_blocks + 8ull * bi, &r5, &g5, &b5, &inten, NULL);
            block_colors5(bc, r5, g5, b5, inten);
            for (int p = 0; p < 16; p++)
                for (int s = 0; s < 4; s++) tot[p][s] += orc_color_distance(perceptual, bc[s], pixel_blocks + 64ull * bi + 4 * p);
        }
        uint8_t sel[16];
        for (int p = 0; p < 16; p++) { int bs = 0; for (int s = 1; s < 4; s++) if (tot[p][s] < tot[p][bs]) bs = s; sel[p] = (uint8_t)bs; }
        /* m_optimized_cluster_selectors[ci].set_selector(...) touches only the selector bytes of the (resized, zero-initialised
           or previously-written) entry; colour bytes of these entries are never meaningful */
        uint64_t v = load_be64(inout_selector_blocks + 8ull * ci) & ~0xFFFFFFFFull;
        v |= pack_etc1s(0, 0, 0, 0, sel) & 0xFFFFFFFFull;
        store_be64(inout_selector_blocks + 8ull * ci, v);
    }
}

void orc_find_optimal_selector_clusters(const uint8_t* pixel_blocks, uint8_t* encoded_blocks, uint32_t n_blocks,
                                        const uint8_t* selector_blocks, uint32_t n_selectors,
                                        uint32_t n_parents, const uint32_t* cand_offsets, const uint32_t* cand_indices, const uint8_t* block_parent,
                                        int perceptual, uint32_t chunk, uint32_t* out_idx) {
    uint8_t* unpacked = (uint8_t*)malloc(16ull * (n_selectors ? n_selectors : 1));
    for (uint32_t i = 0; i < n_selectors; i++) { int a, b, c, d; unpack_etc1s(selector_blocks + 8ull * i, &a, &b, &c, &d, unpacked + 16ull * i); }
    uint32_t prev_best = 0;
    for (uint32_t bi = 0; bi < n_blocks; bi++) {
        const uint32_t first_in_chunk = (bi / chunk) * chunk;
        if (bi == first_in_chunk) prev_best = 0;
        uint8_t* blk = encoded_blocks + 8ull * bi;
        uint32_t best = 0;
        if (bi > first_in_chunk && !memcmp(pixel_blocks + 64ull * bi, pixel_blocks + 64ull * (bi - 1), 64)) {
            best = prev_best; /* frontend.cpp:2557-2564 */
        } else {
            int r5, g5, b5, inten; uint8_t bc[4][3]; uint32_t te[4][16];
            unpack_etc1s(blk, &r5, &g5, &b5, &inten, NULL);
            block_colors5(bc, r5, g5, b5, inten);
            for (int s = 0; s < 4; s++) for (int p = 0; p < 16; p++) te[s][p] = orc_color_distance(perceptual, pixel_blocks + 64ull * bi + 4 * p, bc[s]);
            uint32_t first = 0, total = n_selectors;
            if (n_parents) { first = cand_offsets[block_parent[bi]]; total = cand_offsets[block_parent[bi] + 1] - first; }
            uint64_t best_err = (uint64_t)INT64_MAX;
            for (uint32_t k = 0; k < total; k++) {
                const uint32_t ci = n_parents ? cand_indices[first + k] : k;
                const uint8_t* sl = unpacked + 16ull * ci;
                uint64_t e = 0; for (int p = 0; p < 16; p++) e += te[sl[p]][p];
                if (e < best_err) { best_err = e; best = ci; }
            }
            prev_best = best;
        }
        uint64_t v = (load_be64(blk) & ~0xFFFFFFFFull) | (load_be64(selector_blocks + 8ull * best) & 0xFFFFFFFFull);
        store_be64(blk, v);
        out_idx[bi] = best;
    }
    free(unpacked);
}

void orc_endpoint_training_vectors(const uint8_t* etc1s_blocks, uint32_t n_blocks, float* out6) {
    for (uint32_t i = 0; i < n_blocks; i++) {
        int r5, g5, b5, inten; uint8_t bc[4][3];
        unpack_etc1s(etc1s_blocks + 8ull * i, &r5, &g5, &b5, &inten, NULL);
        block_colors5(bc, r5, g5, b5, inten);
        for (int c = 0; c < 3; c++) { out6[6ull * i + c] = bc[0][c] * (1.0f / 255.0f); out6[6ull * i + 3 + c] = bc[3][c] * (1.0f / 255.0f); }
    }
}

void orc_selector_training_vectors(const uint8_t* encoded_blocks, uint32_t n_blocks, int perceptual, float* out16, uint64_t* out_weight) {
    for (uint32_t i = 0; i < n_blocks; i++) {
        int r5, g5, b5, inten; uint8_t bc[4][3], sel[16];
        unpack_etc1s(encoded_blocks + 8ull * i, &r5, &g5, &b5, &inten, sel);
        block_colors5(bc, r5, g5, b5, inten);
        for (int p = 0; p < 16; p++) out16[16ull * i + p] = (float)sel[p];
        uint32_t w = orc_color_distance(perceptual, bc[0], bc[3]) / 300u;
        out_weight[i] = w < 1 ? 1 : (w > 4096 ? 4096 : w);
    }
}
