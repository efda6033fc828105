// tests/native/uastc_host.cpp -- TEST INFRASTRUCTURE ONLY: the UASTC device core (basis_universal_amd/csrc/uastc_core.h) compiled
// for the host with g++, so that every stage can be diffed against the real reference (oracle/_ref) on the CPU, where there is
// no GPU. The product never builds or loads this file; libbasisu_hip.so compiles the same header with hipcc.
#include <cstring>
#include "../../basis_universal_amd/csrc/uastc_core.h"

using namespace bu_uastc;
#define HC_API extern "C" __attribute__((visibility("default")))

// `mask` selects the texels of the 16 given ones that form the cell (the reference sees them gathered, in the same order)
HC_API uint64_t hc_cell_compress(const uint8_t* px16, uint32_t mask, uint32_t wbits, uint32_t range, int alpha, uint32_t uber, uint32_t ls_passes, uint8_t* out24) {
    cell_cfg cfg;
    cfg.wbits = (uint8_t)wbits; cfg.range = (uint8_t)range; cfg.alpha = (uint8_t)alpha; cfg.uber = (uint8_t)uber; cfg.ls_passes = (uint8_t)ls_passes;
    uint32_t px[16];
    for (int i = 0; i < 16; i++) px[i] = pack_px(px16 + i * 4);
    cell_fit f;
    memset(&f, 0, sizeof(f));
    const uint64_t e = cell_compress(px, mask, cfg, f);
    memcpy(out24, f.astc_lo, 4); memcpy(out24 + 4, f.astc_hi, 4);
    for (int i = 0; i < 16; i++) out24[8 + i] = (uint8_t)sel_get(f.sel, i);
    return e;
}
HC_API uint64_t hc_cell_estimate(uint32_t wbits, uint32_t comps, const uint8_t* px16, uint32_t mask) {
    uint32_t px[16];
    for (int i = 0; i < 16; i++) px[i] = pack_px(px16 + i * 4);
    return estimate_masked_any(wbits, comps, px, mask);
}
HC_API uint32_t hc_weight_of(uint32_t bits, uint32_t s) { return weight_of(bits, s); }
HC_API uint32_t hc_weight_table(uint32_t bits, uint32_t s) { return weight_set(bits)[s]; }

HC_API void hc_encode_uastc(const uint8_t* blocks, uint32_t n, uint32_t flags, uint8_t* out) {
    static cand scratch[MAX_SLOTS];
    for (uint32_t i = 0; i < n; i++) encode_block(blocks + (size_t)i * 64, flags, out + (size_t)i * 16, scratch);
}
