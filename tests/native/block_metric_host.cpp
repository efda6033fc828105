// block_metric_host.cpp -- TEST-ONLY host build of csrc/host/block_metric.h: every ISA variant of the backend's history scan against the
// plain C++ one on tables, histories and limits the backend tests do not reach (distances up to the metric's maximum, limits from 0 to
// beyond every error, SAD test on and off). Compiled by tests/helpers.py with g++ -O2.
#include <cstdint>
#include <cstring>

#include "../../basis_universal_amd/csrc/host/block_metric.h"

using namespace bu::metric;

static uint64_t rng_next(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

extern "C" {

// bit i of the result: variant i exists on this CPU (0 plain, 1 avx2, 2 avx512, 3 vbmi)
int bm_variants() {
    int m = 1;
    if (__builtin_cpu_supports("avx2")) m |= 2;
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl")) m |= 4;
    if ((m & 4) && __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("bmi2")) m |= 8;
    return m;
}

// `cases` random scans; magnitude: distances are drawn below 2^magnitude (26 covers one perceptual distance, < 41e6). Returns the number of
// (case, variant) pairs that differ from the plain scan, 0 if all agree.
int bm_scan_check(uint64_t seed, int cases, int magnitude, int variants) {
    int bad = 0;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    alignas(64) sel16 hist[64];
    for (int c = 0; c < cases; c++) {
        dist_table t;
        const int mag = 1 + (int)(rng_next(s) % (uint64_t)magnitude);
        for (int k = 0; k < 4; k++)
            for (int p = 0; p < 16; p++) {
                uint64_t v = rng_next(s) & ((1ull << mag) - 1);
                if (rng_next(s) % 5 == 0) v = v >> (rng_next(s) % 20);          // small entries next to large ones
                if (v > 41000000ull) v = 41000000ull;
                t.d[k][p] = (uint32_t)v;
            }
        sel16 cur;
        for (int p = 0; p < 16; p++) cur.s[p] = (uint8_t)(rng_next(s) & 3);
        const int style = (int)(rng_next(s) % 3);                               // 0 unrelated patterns, 1 a few pixels off the current one, 2 copies
        for (int j = 0; j < 64; j++)
            for (int p = 0; p < 16; p++) {
                const uint8_t r = (uint8_t)(rng_next(s) & 3);
                hist[j].s[p] = style == 0 ? r : (style == 1 ? ((rng_next(s) % 6 == 0) ? r : cur.s[p]) : cur.s[p]);
            }
        const uint64_t own = table_error_plain(t, cur, UINT64_MAX);
        uint64_t limit;
        switch (rng_next(s) % 6) {
            case 0: limit = 0; break;
            case 1: limit = own; break;
            case 2: limit = own + own / 4; break;
            case 3: limit = rng_next(s) % (own + 2); break;
            case 4: limit = (1ull << 33) + 5; break;
            default: limit = own * 3 + 7; break;
        }
        const int sad_limit = (rng_next(s) & 1) ? 0 : 1 + (int)(rng_next(s) % 12);
        const scan_result want = scan_history_plain(t, cur, hist, sad_limit, limit);
        scan_result got[3];
        int n = 0;
        if (variants & 2) got[n++] = scan_history_avx2(t, cur, hist, sad_limit, limit);
        if (variants & 4) got[n++] = scan_history_avx512(t, cur, hist, sad_limit, limit);
        if (variants & 8) got[n++] = scan_history_vbmi(t, cur, hist, sad_limit, limit);
        for (int i = 0; i < n; i++)
            if (got[i].index != want.index || (want.index >= 0 && got[i].err != want.err)) bad++;
    }
    return bad;
}

// the per-block search (search_prepare: pixels -> table -> own error -> limit; search_history: own pattern look-up, scan) of every variant against the plain one, on random pixels and colours
int bm_search_check(uint64_t seed, int cases, int variants) {
    int bad = 0;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 7;
    alignas(64) sel16 hist[64];
    for (int c = 0; c < cases; c++) {
        const bool perceptual = rng_next(s) & 1;
        alignas(64) uint8_t rgba[64];
        const int spread = 1 + (int)(rng_next(s) % 255);
        const int base[3] = {(int)(rng_next(s) & 255), (int)(rng_next(s) & 255), (int)(rng_next(s) & 255)};
        for (int p = 0; p < 16; p++) {
            for (int ch = 0; ch < 3; ch++) { int v = base[ch] + (int)(rng_next(s) % (uint64_t)spread) - spread / 2; rgba[p * 4 + ch] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
            rgba[p * 4 + 3] = 255;
        }
        pal_colors col;
        for (int k = 0; k < 4; k++) {
            const int d = (k - 2) * (int)(rng_next(s) % 60);
            int rgb[3];
            for (int ch = 0; ch < 3; ch++) { int v = base[ch] + d; rgb[ch] = v < 0 ? 0 : v > 255 ? 255 : v; }
            to_metric(perceptual, rgb[0], rgb[1], rgb[2], col.x[k], col.y[k], col.z[k]);
        }
        sel16 cur;
        for (int p = 0; p < 16; p++) cur.s[p] = (uint8_t)(rng_next(s) & 3);
        for (int j = 0; j < 64; j++)
            for (int p = 0; p < 16; p++) hist[j].s[p] = (rng_next(s) % 5 == 0) ? (uint8_t)(rng_next(s) & 3) : cur.s[p];
        const float thresh = 1.0f + (float)(rng_next(s) % 300) / 100.0f;
        const int sad_limit = (rng_next(s) & 1) ? 0 : 11;
        int values[64];
        for (int j = 0; j < 64; j++) values[j] = (int)(rng_next(s) % 40);          // small range: repeats, so that "first index" matters
        const int own = (rng_next(s) % 3 == 0) ? -1 : (int)(rng_next(s) % 60);       // -1: no look-up; 40..59: never present
        search_prep pr;
        search_prepare_plain(perceptual, rgba, col, cur, thresh, pr);
        const scan_result want = search_history_plain(pr, cur, hist, sad_limit, values, own);
        scan_result got[3];
        int n = 0;
        if (variants & 2) { search_prep q; search_prepare_avx2(perceptual, rgba, col, cur, thresh, q); got[n++] = search_history_avx2(q, cur, hist, sad_limit, values, own); }
        if (variants & 4) { search_prep q; search_prepare_avx512(perceptual, rgba, col, cur, thresh, q); got[n++] = search_history_avx512(q, cur, hist, sad_limit, values, own); }
        if (variants & 8) { search_prep q; search_prepare_vbmi(perceptual, rgba, col, cur, thresh, q); got[n++] = search_history_vbmi(q, cur, hist, sad_limit, values, own); }
        for (int i = 0; i < n; i++)
            if (got[i].index != want.index || (want.index >= 0 && got[i].err != want.err)) bad++;
    }
    return bad;
}

}  // extern "C"
