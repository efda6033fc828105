// UASTC LDR 4x4 rate-distortion post-pass: the per-block pieces of uastc_rdo_blocks
// (encoder/basisu_uastc_enc.cpp:3824-4089) on top of uastc_core.h -- unpacking a packed block (transcoder/basisu_transcoder.cpp:
// 15274-15735), the selector bit field of each mode, the LZ match cost estimate, one trial ("this block with that block's selector
// bits"), and the write-back with the mode-0 endpoint refit. Single source like the core: hipcc compiles it for the GPU strips kernel,
// g++ compiles it into the test-only host library that is diffed against the reference's uastc_rdo.
#pragma once
#include "uastc_core.h"

namespace bu_uastc {

struct rdo_params {              // uastc_rdo_params, uastc_enc.h:94-134
    float lambda;
    float max_allowed_rms_increase_ratio;
    float skip_block_rms_thresh;
    float max_smooth_block_std_dev;
    float smooth_block_max_error_scale;
    uint32_t lz_dict_size;
    uint32_t lz_literal_cost;
    uint32_t endpoint_refinement;
};

// g_uastc_mode_selector_bits (uastc_enc.cpp:3728-3735): first bit and length of the weight field of every mode
BU_TAB unsigned char ku_sel_first[19] = { 65, 69, 73, 89, 89, 68, 66, 89, 0, 97, 65, 66, 81, 94, 92, 62, 98, 61, 49 };
BU_TAB unsigned char ku_sel_len[19] = { 63, 31, 46, 29, 30, 47, 62, 30, 0, 30, 63, 62, 47, 30, 31, 63, 30, 62, 79 };

// read_bits (basisu_enc.h:130-151): `n` <= 64 bits at bit `ofs` of a 16-byte block, bits past the block read as zero
BU_FN uint64_t block_bits(const uint8_t* b, uint32_t ofs, uint32_t n) {
    uint64_t v = 0;
    for (uint32_t got = 0; got < n;) {
        const uint32_t in_byte = ofs & 7, k = (n - got) < (8 - in_byte) ? (n - got) : (8 - in_byte);
        const uint32_t byte = (ofs >> 3) < 16 ? b[ofs >> 3] : 0u;
        v |= (uint64_t)((byte >> in_byte) & ((1u << k) - 1)) << got;
        got += k;
        ofs += k;
    }
    return v;
}

BU_FN uint32_t hint_bits(uint32_t mode) {
    return (ku_mode_has_bc1_hint0[mode] ? 1u : 0u) + (ku_mode_has_bc1_hint1[mode] ? 1u : 0u) + 8u + (ku_mode_has_etc1_bias[mode] ? 5u : 0u) +
           (ku_mode_has_alpha[mode] ? 8u : 0u);
}

// the weights of a block from its weight bit field (lo = field bits 0..63, hi = the rest): every anchor weight is stored without its
// top bit (transcoder.cpp:15573-15690; the reference's special cases all read the same layout)
BU_FN void parse_weights(cand& r, uint64_t lo, uint64_t hi) {
    const uint32_t mode = r.mode, subsets = ku_mode_subsets[mode], planes = ku_mode_planes[mode], wbits = ku_mode_weight_bits[mode];
    const uint8_t* anchors = subsets == 3 ? ku_anchor3 + r.pattern * 3 : (mode == 7 ? ku_anchor7 + r.pattern * 3 : ku_anchor2 + r.pattern * 3);
    const uint32_t plane_shift = planes == 2 ? 1 : 0;
    for (uint32_t i = 0; i < 16 * planes; i++) {
        uint32_t nb = wbits;
        for (uint32_t s = 0; s < subsets; s++)
            if ((subsets >= 2 ? anchors[s] : 0u) == (i >> plane_shift)) { nb--; break; }
        r.weights[i] = (uint8_t)(lo & ((1u << nb) - 1));
        lo = nb ? ((lo >> nb) | (hi << (64 - nb))) : lo;  // nb == 0: the anchors of 1-bit weights (mode 13)
        hi >>= nb;
    }
}

// unpack_uastc(blk, unpacked, blue_contract_check = false, read_hints = *): everything but the hints. Returns false where the reference
// does (invalid mode code, pattern index out of range). A solid block comes back as mode 8 with its colour in endpoints[0..3].
BU_FN bool unpack_block(const uint8_t* blk, cand& r) {
    uint32_t mode = 19;
    for (uint32_t m = 0; m < 19; m++)
        if ((blk[0] & ((1u << ku_mode_code_len[m]) - 1)) == ku_mode_code[m]) { mode = m; break; }
    if (mode >= 19) return false;
    cand_begin(r, mode, 0);
    uint32_t ofs = ku_mode_code_len[mode];
    if (mode == 8) {
        for (uint32_t c = 0; c < 4; c++) r.endpoints[c] = (uint8_t)block_bits(blk, ofs + 8 * c, 8);
        return true;
    }
    ofs += hint_bits(mode);
    const uint32_t subsets = ku_mode_subsets[mode], planes = ku_mode_planes[mode], comps = ku_mode_comps[mode];
    if (subsets == 3) {
        r.pattern = (uint8_t)block_bits(blk, ofs, 4); ofs += 4;
        if (r.pattern >= 11) return false;                      // TOTAL_ASTC_BC7_COMMON_PARTITIONS3
    } else if (subsets == 2) {
        r.pattern = (uint8_t)block_bits(blk, ofs, 5); ofs += 5;
        if (r.pattern >= (mode == 7 ? 19u : 30u)) return false;  // TOTAL_BC7_3_ASTC2_COMMON_PARTITIONS / ..._PARTITIONS2
    }
    if (planes == 2) {
        if (mode == 17) r.ccs = 3;
        else { r.ccs = (uint8_t)block_bits(blk, ofs, 2); ofs += 2; }
    }
    const uint32_t range = ku_mode_endpoint_ranges[mode];
    const uint32_t total_values = comps * 2 * subsets;
    const uint32_t ep_bits = ku_bise[range * 3], ep_trits = ku_bise[range * 3 + 1], ep_quints = ku_bise[range * 3 + 2];
    uint32_t groups = 0, per_group = 0, radix = 0;
    if (ep_trits) { groups = (total_values + 4) / 5; per_group = 5; radix = 3; }
    else if (ep_quints) { groups = (total_values + 2) / 3; per_group = 3; radix = 5; }
    uint32_t packed[8];
    for (uint32_t g = 0; g < groups; g++) {
        uint32_t nb = ep_trits ? 8 : 7;
        if (g == groups - 1) {
            const uint32_t left = total_values - (groups - 1) * per_group;
            if (ep_trits) nb = left == 1 ? 2 : (left == 2 ? 4 : (left == 3 ? 5 : (left == 4 ? 7 : 8)));
            else nb = left == 1 ? 3 : (left == 2 ? 5 : 7);
        }
        packed[g] = (uint32_t)block_bits(blk, ofs, nb);
        ofs += nb;
    }
    uint32_t accum = 0, left_in_group = 0, next_group = 0;
    for (uint32_t i = 0; i < total_values; i++) {
        uint32_t v = (uint32_t)block_bits(blk, ofs, ep_bits);
        ofs += ep_bits;
        if (groups) {
            if (!left_in_group) { accum = packed[next_group++]; left_in_group = per_group; }
            v |= (accum % radix) << ep_bits;
            accum /= radix;
            left_in_group--;
        }
        r.endpoints[i] = (uint8_t)v;
    }
    const uint32_t len = ku_sel_len[mode];
    parse_weights(r, block_bits(blk, ofs, len < 64 ? len : 64), len > 64 ? block_bits(blk, ofs + 64, len - 64) : 0);
    return true;
}

// compute_match_cost_estimate (uastc_enc.cpp:3773-3790). The two tdefl tables it indexes are "extra bits of the DEFLATE distance
// code": floor(log2(d)) - 1 below 512 (0 under 4), floor(log2(d >> 8)) + 7 from there on.
BU_FN uint32_t floor_log2(uint32_t v) { return v ? 31u - (uint32_t)__builtin_clz(v) : 0u; }
BU_FN uint32_t match_cost(uint32_t dist) {
    uint32_t cost = 7 + 5;
    if (dist < 512) cost += dist < 4 ? 0 : floor_log2(dist) - 1;
    else {
        const uint32_t idx = (dist < 32767 ? dist : 32767) >> 8;
        cost += idx < 2 ? 0 : floor_log2(idx) + 7;
        if (dist >= 32768) cost += floor_log2(dist) - 14;  // one more bit per halving it takes to get below 32768
    }
    return cost;
}

// sum of the block's UASTC error and the error of its BC7 transcode, halved (uastc_enc.cpp:3872-3893, 3971-3991)
BU_FN uint64_t rdo_block_error(const cand& r, const rgba8* px) {
    uint32_t packed[16];
    BU_UNROLL
    for (int i = 0; i < 16; i++) packed[i] = pack_px(px[i].c);
    chan_err cu, cb;   // the fused decode + error of the scoring pass (uastc_core.h): no decoded images in scratch memory
    uastc_errors(r, packed, cu);
    bc7_errors(r, packed, cb);
    return (chan_err_totals(cu).rgba + chan_err_totals(cb).rgba) / 2;
}

// What uastc_rdo_blocks derives from a block before looking at its neighbours (:3837-3915): the error it has now, the smooth-block scale,
// its selector field, whether it is left alone.
struct rdo_block_info {
    float ms_err, rms_err, scale;
    uint32_t mode;      // 8: solid, nothing to do
    uint32_t skip;      // too much error already
};
BU_FN bool rdo_prepare(const uint8_t* blk, const rgba8* px, const rdo_params& p, cand& unpacked, rdo_block_info& o) {
    if (!unpack_block(blk, unpacked)) return false;
    o.mode = unpacked.mode;
    o.skip = 0; o.ms_err = o.rms_err = 0.0f; o.scale = 1.0f;
    if (o.mode == 8) return true;
    float max_dev = 0.0f;
    for (int c = 0; c < 4; c++) {  // tracked_stat::get_std_dev, basisu_math.h:3554
        int64_t total = 0, total2 = 0;
        for (int i = 0; i < 16; i++) { total += px[i].c[c]; total2 += (int)px[i].c[c] * (int)px[i].c[c]; }
        const float dev = sqrtf((float)(16 * total2 - total * total)) / 16.0f;
        max_dev = c == 0 ? dev : (max_dev > dev ? max_dev : dev);
    }
    float yl = max_dev / p.max_smooth_block_std_dev;
    yl = yl < 0.0f ? 0.0f : (yl > 1.0f ? 1.0f : yl);
    yl = yl * yl;
    o.scale = p.smooth_block_max_error_scale + (1.0f - p.smooth_block_max_error_scale) * yl;
    o.ms_err = (float)rdo_block_error(unpacked, px) * (1.0f / 64.0f);
    o.rms_err = sqrtf(o.ms_err);
    o.skip = o.rms_err >= p.skip_block_rms_thresh ? 1u : 0u;
    return true;
}

// One trial (:3945-4008): `base` with the weight field (lo, hi). Returns false when the error grew past the allowed ratio.
BU_FN bool rdo_trial(const cand& base, uint64_t lo, uint64_t hi, const rgba8* px, const rdo_block_info& info, const rdo_params& p, float& ms_err) {
    cand t = base;
    parse_weights(t, lo, hi);
    ms_err = (float)rdo_block_error(t, px) * (1.0f / 64.0f);
    return !(sqrtf(ms_err) > info.rms_err * p.max_allowed_rms_increase_ratio);
}

// ---- Trials by table. A trial keeps the block's endpoints and replaces its weights, and both decoders are per texel: the UASTC colour of
// texel i is astc_lerp(low_i, high_i, W[w]) and its BC7 transcode is bc7_lerp(low'_i, high'_i, W'[w]) with endpoints that do not depend on
// the weights (decode_uastc / decode_bc7: the anchor-bit endpoint swaps of the real transcode cancel against the mirrored weights). So
// for one block the error of "weight k takes value v" is a table E[k][v], and a trial's UASTC+BC7 error is the sum of 16 (32 for dual
// plane) table entries -- same integers as decoding the trial twice. Dual-plane modes split by channel: the ccs channel reads plane 1.

// BC7 interpolation weight (0..64) decode_bc7 applies for UASTC weight value w of `mode` (transcoder.cpp: the s_uastc*_to_bc7 conversions)
BU_FN uint32_t bc7_weight_for(uint32_t mode, uint32_t w) {
    switch (mode) {
    case 0: case 10: case 15: return ku_bc7_weights4[w];
    case 18: { const uint8_t five_to_four[32] = { 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 6, 7, 8, 9, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15 };
               return ku_bc7_weights4[five_to_four[w]]; }
    case 14: return ku_bc7_weights4[w * 5];
    case 5: case 12: { const uint8_t three_to_four[8] = { 0, 2, 4, 6, 9, 11, 13, 15 }; return ku_bc7_weights4[three_to_four[w]]; }
    case 2: return weight_set(3)[w];
    case 13: return w ? 64u : 0u;
    default: return weight_set(2)[w];  // 1, 4, 9, 16; 3, 7; 6, 11, 17
    }
}

// per texel: the colours both decoders produce at weight 0 and at the top weight, RGBA packed
struct texel_ends { uint32_t ul[16], uh[16], bl[16], bh[16]; };
BU_FN void rdo_texel_ends(const cand& c, texel_ends& o) {
    cand z = c;
    rgba8 dec[16];
    for (int i = 0; i < 32; i++) z.weights[i] = 0;
    decode_uastc(z, dec);
    for (int i = 0; i < 16; i++) o.ul[i] = pack_px(dec[i].c);
    decode_bc7(z, dec);
    for (int i = 0; i < 16; i++) o.bl[i] = pack_px(dec[i].c);
    const uint8_t top = (uint8_t)((1u << ku_mode_weight_bits[c.mode]) - 1);
    for (int i = 0; i < 32; i++) z.weights[i] = top;
    decode_uastc(z, dec);
    for (int i = 0; i < 16; i++) o.uh[i] = pack_px(dec[i].c);
    decode_bc7(z, dec);
    for (int i = 0; i < 16; i++) o.bh[i] = pack_px(dec[i].c);
}

constexpr uint32_t RDO_TABLE_WORDS = 512;
// E[k][v]: weight slot k (texel k / planes, plane k % planes) at value v
BU_FN uint32_t rdo_weight_error(uint32_t mode, uint32_t ccs, uint32_t k, uint32_t v, uint32_t ul, uint32_t uh, uint32_t bl, uint32_t bh, uint32_t px) {
    const uint32_t planes = ku_mode_planes[mode], plane = planes == 2 ? (k & 1u) : 0u;
    const uint32_t wu = weight_set(ku_mode_weight_bits[mode])[v], wb = bc7_weight_for(mode, v);
    uint32_t e = 0;
    for (uint32_t c = 0; c < 4; c++) {
        if (planes == 2 && ((c == ccs) != (plane == 1))) continue;
        const int t = px_comp(px, (int)c);
        const int du = (int)astc_lerp((uint32_t)px_comp(ul, (int)c), (uint32_t)px_comp(uh, (int)c), wu) - t;
        const int db = (int)bc7_lerp((uint32_t)px_comp(bl, (int)c), (uint32_t)px_comp(bh, (int)c), wb) - t;
        e += (uint32_t)(imul24(du, du) + imul24(db, db));
    }
    return e;
}

// which weight slots are stored with one bit less (the anchors; pack_block / parse_weights)
BU_FN uint32_t rdo_anchor_mask(uint32_t mode, uint32_t pattern) {
    const uint32_t subsets = ku_mode_subsets[mode], planes = ku_mode_planes[mode];
    if (planes == 2) return 3u;
    if (subsets == 1) return 1u;
    const uint8_t* anchors = subsets == 3 ? ku_anchor3 + pattern * 3 : (mode == 7 ? ku_anchor7 + pattern * 3 : ku_anchor2 + pattern * 3);
    uint32_t m = 0;
    for (uint32_t s = 0; s < subsets; s++) m |= 1u << anchors[s];
    return m;
}

// a trial's (uastc_err + bc7_err) from the table, table[(k << weight_bits) + v] (at most 512 entries: 16 x 32 for mode 18, 32 x 4 dual plane)
// Written as 16-slot groups with the indices extracted first, so the table reads of a group are independent (one wait per group on the GPU).
template <class TABLE>
BU_FN uint32_t rdo_trial_sum_n(const TABLE& table, uint32_t n_slots, uint32_t wbits, uint32_t anchor_mask, uint64_t lo, uint64_t hi) {
    uint32_t total = 0;
    for (uint32_t g = 0; g < n_slots; g += 16) {
        uint32_t idx[16];
        BU_UNROLL
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t nb = wbits - ((anchor_mask >> (g + k)) & 1u);
            idx[k] = ((g + k) << wbits) + ((uint32_t)lo & ((1u << nb) - 1));
            lo = nb ? ((lo >> nb) | (hi << (64 - nb))) : lo;  // nb == 0: the anchors of 1-bit weights (mode 13)
            hi >>= nb;
        }
        BU_UNROLL
        for (uint32_t k = 0; k < 16; k++) total += table[idx[k]];
    }
    return total;
}
template <class TABLE>
BU_FN uint32_t rdo_trial_sum(const TABLE& table, uint32_t mode, uint32_t anchor_mask, uint64_t lo, uint64_t hi) {
    return rdo_trial_sum_n(table, 16 * ku_mode_planes[mode], ku_mode_weight_bits[mode], anchor_mask, lo, hi);
}

// The accepted trial written back (:4012-4073): mode 0 gets its endpoints refitted to the new selectors when that lowers the UASTC
// error. The hints are left zero -- nothing later in the strip reads them (every mode's selector field starts past bit 48, the hints end
// before bit 30) -- and are recomputed for all modified blocks afterwards (uastc_recompute_hints, :3647-3726 == finish_block).
BU_FN void rdo_write_back(const cand& base, uint64_t lo, uint64_t hi, const rgba8* px, const rdo_params& p, uint8_t* out16, bool& refined) {
    cand c = base;
    parse_weights(c, lo, hi);
    refined = false;
    if (p.endpoint_refinement && c.mode == 0) {
        rgba8 dec[16];
        decode_uastc(c, dec);
        const uint64_t before = block_error(px, dec).rgba;
        uint32_t packed[16];
        pack_block_px(px, false, packed);
        sel16 forced = { { 0, 0, 0, 0 } };
        for (int i = 0; i < 16; i++) sel_set(forced, i, c.weights[i]);
        cell_cfg cc;
        cc.wbits = 4; cc.range = 19; cc.alpha = 0; cc.uber = 0; cc.ls_passes = 1; cc.ls_weights = ku_weights_ls + 14 * 4;
        cell_fit f;
        cell_compress_t<true>(packed, 0xFFFFu, cc, f, &forced);
        cand fitted = c;
        for (uint32_t ch = 0; ch < 3; ch++) { fitted.endpoints[ch * 2] = f.astc_lo[ch]; fitted.endpoints[ch * 2 + 1] = f.astc_hi[ch]; }
        for (int i = 0; i < 16; i++) fitted.weights[i] = (uint8_t)sel_get(f.sel, i);
        decode_uastc(fitted, dec);
        if (block_error(px, dec).rgba < before) { c = fitted; refined = true; }
    }
    const etc1_hint none = { 0, 0, 0, 0, 0 };
    pack_block(c, none, 0, 0, false, false, out16);
}

// set_block_bits (uastc_enc.cpp:3737-3754) for a whole selector field: the trial block is the current block with this field replaced
BU_FN void put_field(uint8_t* b, uint32_t ofs, uint32_t n, uint64_t lo, uint64_t hi) {
    for (uint32_t done = 0; done < n;) {
        const uint32_t in_byte = ofs & 7, k = (n - done) < (8 - in_byte) ? (n - done) : (8 - in_byte);
        const uint32_t m = ((1u << k) - 1) << in_byte;
        b[ofs >> 3] = (uint8_t)((b[ofs >> 3] & ~m) | (((uint32_t)lo << in_byte) & m));
        lo = (lo >> k) | (hi << (64 - k));
        hi >>= k;
        done += k;
        ofs += k;
    }
}

// Deferred form of the write-back. Only a mode-0 block's endpoint bits (up to bit 64) change in the refit, and only the selector fields of
// modes 15, 17 and 18 start below bit 65 -- so as long as no block of those modes looks at a modified mode-0 block, the strip walk can
// store the raw trial bits (put_field) and leave the refit to a parallel pass; a block of a "sensitive" mode first settles the pending
// refits in its window. The refit of a block depends on nothing but that block, so when it runs does not matter.
BU_FN bool rdo_mode_reads_endpoint_bits(uint32_t mode) { return mode == 15 || mode == 17 || mode == 18; }
BU_FN bool rdo_refit_block(const rgba8* px, const rdo_params& p, uint8_t* blk16, bool& refined) {
    cand c;
    refined = false;
    if (!unpack_block(blk16, c)) return false;
    if (c.mode != 0 || !p.endpoint_refinement) return true;
    const uint64_t lo = block_bits(blk16, ku_sel_first[0], 63);
    uint8_t out[16];
    rdo_write_back(c, lo, 0, px, p, out, refined);
    if (refined)  // an unrefined block keeps its bytes (pack_block would zero the stale hints; harmless, but keep the bits stable)
        for (int i = 0; i < 16; i++) blk16[i] = out[i];
    return true;
}

// uastc_recompute_hints (:3647-3726) of a block whose weights (and maybe endpoints) changed
BU_FN bool rdo_rehint(const rgba8* px, const enc_cfg& e, uint8_t* blk16, hint_cache cache = no_hint_cache()) {
    cand c;
    if (!unpack_block(blk16, c)) return false;
    if (c.mode == 8) return true;
    finish_block(px, e, c, blk16, cache);
    return true;
}

}  // namespace bu_uastc
