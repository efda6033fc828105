// tests/native/uastc_host.cpp -- TEST INFRASTRUCTURE ONLY: the UASTC device core (basis_universal_amd/csrc/uastc_core.h) compiled
// for the host with g++, so that every stage can be diffed against the real reference (oracle/_ref) on the CPU, where there is
// no GPU. The product never builds or loads this file; libbasisu_hip.so compiles the same header with hipcc.
#include <cstring>
#include "../../basis_universal_amd/csrc/uastc_core.h"

using namespace bu_uastc;
#define HC_API extern "C" __attribute__((visibility("default")))

HC_API uint64_t hc_cell_compress(const uint8_t* px, uint32_t n, uint32_t wbits, uint32_t range, int alpha, uint32_t uber, uint32_t ls_passes,
                                 const uint8_t* force_sel, uint8_t* out24) {
    cell_cfg cfg;
    cfg.wbits = (uint8_t)wbits; cfg.range = (uint8_t)range; cfg.alpha = (uint8_t)alpha; cfg.uber = (uint8_t)uber; cfg.ls_passes = (uint8_t)ls_passes;
    cfg.force_sel = force_sel;
    cell_fit f;
    memset(&f, 0, sizeof(f));
    const uint64_t e = cell_compress((const rgba8*)px, n, cfg, f);
    memcpy(out24, f.astc_lo, 4); memcpy(out24 + 4, f.astc_hi, 4); memcpy(out24 + 8, f.sel, 16);
    return e;
}
HC_API uint64_t hc_cell_estimate(uint32_t wbits, uint32_t comps, const uint8_t* px, uint32_t n, uint64_t best) {
    return cell_estimate(wbits, comps, (const rgba8*)px, n, best);
}

HC_API void hc_encode_uastc(const uint8_t* blocks, uint32_t n, uint32_t flags, uint8_t* out) {
    static cand scratch[MAX_SLOTS];
    for (uint32_t i = 0; i < n; i++) encode_block(blocks + (size_t)i * 64, flags, out + (size_t)i * 16, scratch);
}
