// integration/basisu_resident_uastc.cpp -- the UASTC LDR 4x4 hot path of basis_compressor on the MI355X, under the reference's own driver:
// basis_compressor::encode_slices_to_uastc_4x4_ldr (encoder/basisu_comp.cpp:1973-2093: the job loop over encode_uastc at :2020-2033 and the uastc_rdo
// call at :2066-2082) re-implemented on the C ABI of include/basisu_hip.h -- one bu_hip_encode_uastc_blocks (+ bu_hip_uastc_rdo) per slice instead
// of one encode_uastc call per block on the job pool. Everything else of basis_compressor (image reading, mip generation, source-block extraction,
// the .basis / KTX2 writers, statistics) stays the reference's untouched object code.
//
// The reference has no seam here, and the function is a member of a class defined in basisu_comp.cpp together with everything else, so
// it cannot be replaced by leaving a translation unit out the way integration/basisu_resident_{frontend,backend}.cpp replace theirs. A maintainer would
// change the function body in place; to show the same thing WITHOUT editing the reference, oracle/Makefile links this file's definition over the
// reference's by weakening that one symbol in its copy of basisu_comp.o (`objcopy --weaken-symbol`): _ref/basisu_hip_uastc. Same output file as the stock
// tool for -uastc [-uastc_level L] [-uastc_rdo_l X] (tests/test_gpu_reference_seam.py). There is no CPU path behind this file.
// OURS; includes the reference's headers, contains none of its code.
#include "encoder/basisu_comp.h"

#include <mutex>

#include "basisu_hip.h"

namespace basisu {

// the accelerator context the reference hands around is the shim's wrapper (integration/basisu_hip_shim.cpp)
struct opencl_context { bu_hip_context* h; };

namespace {
// used when the caller did not ask for the accelerator seam (no -opencl): this build always runs the UASTC search on the GPU
bu_hip_context* own_context() {
    static std::mutex lock;
    static bu_hip_context* ctx = nullptr;
    std::lock_guard<std::mutex> g(lock);
    if (!ctx) { bu_hip_init(0); ctx = bu_hip_create_context(); }
    return ctx;
}
}  // namespace

basis_compressor::error_code basis_compressor::encode_slices_to_uastc_4x4_ldr() {
    debug_printf("basis_compressor::encode_slices_to_uastc_4x4_ldr (MI355X)\n");
    static_assert(sizeof(pixel_block) == sizeof(bu_pixel_block) && sizeof(basist::uastc_block) == sizeof(bu_uastc_block), "layout");

    m_uastc_slice_textures.resize(m_slice_descs.size());
    for (uint32_t slice_index = 0; slice_index < m_slice_descs.size(); slice_index++)
        m_uastc_slice_textures[slice_index].init(texture_format::cUASTC4x4, m_slice_descs[slice_index].m_orig_width, m_slice_descs[slice_index].m_orig_height);

    m_uastc_backend_output.m_tex_format = basist::basis_tex_format::cUASTC_LDR_4x4;
    m_uastc_backend_output.m_etc1s = false;
    m_uastc_backend_output.m_slice_desc = m_slice_descs;
    m_uastc_backend_output.m_slice_image_data.resize(m_slice_descs.size());
    m_uastc_backend_output.m_slice_image_crcs.resize(m_slice_descs.size());

    bu_hip_context* ctx = m_pOpenCL_context ? m_pOpenCL_context->h : own_context();
    if (!ctx) { error_printf("basis_compressor (MI355X UASTC): no HIP context\n"); return cECFailedEncodeUASTC; }

    for (uint32_t slice_index = 0; slice_index < m_slice_descs.size(); slice_index++) {
        gpu_image& tex = m_uastc_slice_textures[slice_index];
        const basisu_backend_slice_desc& slice_desc = m_slice_descs[slice_index];
        const uint32_t total_blocks = tex.get_total_blocks();
        if ((uint64_t)slice_desc.m_first_block_index + total_blocks > m_source_blocks.size()) return cECFailedEncodeUASTC;

        uint32_t uastc_flags = m_params.m_pack_uastc_ldr_4x4_flags;   // comp.cpp:2016-2018
        if ((m_params.m_rdo_uastc_ldr_4x4) && (m_params.m_rdo_uastc_ldr_4x4_favor_simpler_modes_in_rdo_mode)) uastc_flags |= cPackUASTCFavorSimplerModes;

        // the slice's tiles are what extract_source_blocks left (comp.cpp:3207-3268: extract_block_clamped in block-raster order, the order gpu_image stores blocks in)
        const bu_pixel_block* tiles = reinterpret_cast<const bu_pixel_block*>(&m_source_blocks[slice_desc.m_first_block_index]);
        bu_uastc_block* blocks = reinterpret_cast<bu_uastc_block*>(tex.get_ptr());
        if (!bu_hip_set_pixel_blocks(ctx, total_blocks, tiles) || !bu_hip_encode_uastc_blocks(ctx, blocks, uastc_flags)) {
            error_printf("basis_compressor (MI355X UASTC): encode failed: %s\n", bu_hip_last_error(ctx));
            return cECFailedEncodeUASTC;
        }

        if (m_params.m_rdo_uastc_ldr_4x4) {   // comp.cpp:2066-2082
            bu_uastc_rdo_params rdo;
            bu_hip_uastc_rdo_default_params(&rdo);
            rdo.m_lambda = m_params.m_rdo_uastc_ldr_4x4_quality_scalar;
            rdo.m_max_allowed_rms_increase_ratio = m_params.m_rdo_uastc_ldr_4x4_max_allowed_rms_increase_ratio;
            rdo.m_skip_block_rms_thresh = m_params.m_rdo_uastc_ldr_4x4_skip_block_rms_thresh;
            rdo.m_lz_dict_size = m_params.m_rdo_uastc_ldr_4x4_dict_size;
            rdo.m_smooth_block_max_error_scale = m_params.m_rdo_uastc_ldr_4x4_max_smooth_block_error_scale;
            rdo.m_max_smooth_block_std_dev = m_params.m_rdo_uastc_ldr_4x4_smooth_block_max_std_dev;
            const uint32_t total_jobs = (m_params.m_rdo_uastc_ldr_4x4_multithreading && m_params.m_pJob_pool)
                                            ? basisu::minimum<uint32_t>(4, (uint32_t)m_params.m_pJob_pool->get_total_threads()) : 0;
            uint32_t stats[4] = {0, 0, 0, 0};
            if (!bu_hip_uastc_rdo(ctx, blocks, &rdo, m_params.m_pack_uastc_ldr_4x4_flags, total_jobs, stats)) {
                error_printf("basis_compressor (MI355X UASTC): RDO failed: %s\n", bu_hip_last_error(ctx));
                return cECFailedUASTCRDOPostProcess;
            }
        }

        m_uastc_backend_output.m_slice_image_data[slice_index].resize(tex.get_size_in_bytes());
        memcpy(&m_uastc_backend_output.m_slice_image_data[slice_index][0], tex.get_ptr(), tex.get_size_in_bytes());
        m_uastc_backend_output.m_slice_image_crcs[slice_index] = basist::crc16(tex.get_ptr(), tex.get_size_in_bytes(), 0);
    }
    return cECSuccess;
}

}  // namespace basisu
