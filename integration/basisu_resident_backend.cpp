// integration/basisu_resident_backend.cpp -- the translation unit a maintainer of the reference compiles INSTEAD OF
// encoder/basisu_backend.cpp together with basisu_resident_frontend.cpp ("resident path"): basisu_backend's
// constructor / init / encode (what basis_compressor::process_backend calls, comp.cpp:3526-3557) on this repository's backend
// (include/basisu_hip_backend.h: same bytes as basisu_backend::encode, tests/test_backend_host.py), which reads the resident
// frontend directly and runs its call-back (reoptimize_remapped_endpoints at compression levels above 1) on the device.
// basisu_backend_output is filled field by field, so basisu_file / create_ktx2_file of the reference serialise it unchanged.
// This file is OURS and contains no reference code; it only includes the reference's headers.
#include "encoder/basisu_backend.h"

#include <vector>

#include "basisu_resident.h"

namespace basisu {

basisu_backend::basisu_backend() { clear(); }

void basisu_backend::clear() {
    m_pFront_end = NULL;
    m_params.clear();
    m_output.clear();
}

void basisu_backend::init(basisu_frontend* pFront_end, basisu_backend_params& params, const basisu_backend_slice_desc_vec& slice_descs) {
    m_pFront_end = pFront_end;
    m_params = params;
    m_slices = slice_descs;
}

uint32_t basisu_backend::encode() {
    bu_frontend* f = bu_resident_handle(m_pFront_end);
    if (!f || m_params.m_used_global_codebooks || !m_params.m_etc1s) { error_printf("basisu_backend (resident): needs the resident ETC1S frontend, no global codebooks\n"); return 0; }
    const bool video = m_pFront_end->get_params().m_tex_type == basist::cBASISTexTypeVideoFrames;
    std::vector<bu_backend_slice_desc> sl(m_slices.size());
    for (size_t i = 0; i < m_slices.size(); i++) {
        const basisu_backend_slice_desc& s = m_slices[i];
        bu_backend_slice_desc& d = sl[i];
        d.first_block_index = s.m_first_block_index;
        d.orig_width = s.m_orig_width; d.orig_height = s.m_orig_height;
        d.width = s.m_width; d.height = s.m_height;
        d.num_blocks_x = s.m_num_blocks_x; d.num_blocks_y = s.m_num_blocks_y;
        d.source_file_index = s.m_source_file_index; d.mip_index = s.m_mip_index;
        d.alpha = s.m_alpha; d.iframe = s.m_iframe; d.reserved[0] = d.reserved[1] = 0;
    }
    bu_backend_params bp;
    bp.endpoint_rdo_quality_thresh = m_params.m_endpoint_rdo_quality_thresh;
    bp.selector_rdo_quality_thresh = m_params.m_selector_rdo_quality_thresh;
    bp.compression_level = m_params.m_compression_level;
    bp.video = video ? 1u : 0u;
    // Whatever happens below, the resident frontend's device state is let go when this function returns -- after the backend, which reads it: `fg` is declared
    // first, so `g` (the backend) is destroyed first. A long-lived host that carries on after a failed encode does not keep the frontend's buffers registered.
    struct frontend_guard { const basisu_frontend* fe; ~frontend_guard() { bu_resident_release(fe); } } fg{m_pFront_end};
    bu_backend* be = bu_backend_create();
    if (!be) return 0;
    struct guard { bu_backend* b; ~guard() { bu_backend_destroy(b); } } g{be};
    if (!bu_backend_init(be, f, &bp, sl.data(), (uint32_t)sl.size())) { error_printf("basisu_backend (resident): init failed: %s\n", bu_backend_error(be)); return 0; }
    const uint32_t total = bu_backend_encode(be);
    if (!total) { error_printf("basisu_backend (resident): encode failed: %s\n", bu_backend_error(be)); return 0; }

    auto fetch = [&](const char* name, uint32_t slice, uint8_vec& out) {
        const uint64_t need = bu_backend_get(be, name, slice, nullptr, 0);
        if (need == ~0ull) return false;
        out.resize((size_t)need);
        return need == 0 || bu_backend_get(be, name, slice, out.data(), need) == need;
    };
    m_output.m_slice_desc = m_slices;
    m_output.m_etc1s = m_params.m_etc1s;
    m_output.m_uses_global_codebooks = false;
    m_output.m_srgb = m_pFront_end->get_params().m_perceptual;
    uint32_t v = 0;
    if (bu_backend_get(be, "num_endpoints", 0, &v, 4) != 4) return 0;
    m_output.m_num_endpoints = v;
    if (bu_backend_get(be, "num_selectors", 0, &v, 4) != 4) return 0;
    m_output.m_num_selectors = v;
    if (!fetch("endpoint_palette", 0, m_output.m_endpoint_palette) || !fetch("selector_palette", 0, m_output.m_selector_palette) ||
        !fetch("slice_image_tables", 0, m_output.m_slice_image_tables))
        return 0;
    m_output.m_slice_image_data.resize(m_slices.size());
    for (uint32_t i = 0; i < m_slices.size(); i++)
        if (!fetch("slice_image_data", i, m_output.m_slice_image_data[i])) return 0;
    uint8_vec crcs;
    if (!fetch("slice_image_crcs", 0, crcs) || crcs.size() != m_slices.size() * 2) return 0;
    m_output.m_slice_image_crcs.resize(m_slices.size());
    if (!crcs.empty()) memcpy(m_output.m_slice_image_crcs.data(), crcs.data(), crcs.size());
    return total;   // the guards: backend first, then the frontend's device state (nothing after encode() needs it: the getters read the object's host members)
}

} // namespace basisu
