// integration/process_bench.cpp -- times the reference's own driver, basis_compressor::init() + process() (encoder/basisu_comp.cpp:429-679,
// 828-998), on a raw RGBA raster with the command line tool's ETC1S defaults (comp level 1, sRGB metrics; basisu_tool.cpp:1220-1227), so
// that PNG decoding and file writing stay outside the measurement (figure ii). The SAME source is linked three ways by
// oracle/Makefile -- nothing in here knows which:
//   _ref/process_bench            stock reference objects (CPU)
//   _ref/process_bench_hip        + integration/basisu_hip_shim.cpp: the reference frontend calling the kernels through its accelerator seam
//   _ref/process_bench_resident   + integration/basisu_resident_{frontend,backend}.cpp: the whole ETC1S path resident on the GPU
// OURS; it only calls the reference's public API.
//
// usage: process_bench <raw rgba file> <width> <height> <quality 1..255> <comp level 0..6> <threads> <use_opencl 0|1> <repeats> [out.basis | -] [parallel images]
// prints one JSON line: seconds of init / process per repeat, the output size and an FNV-1a hash of the .basis bytes.
// With `parallel images` = N > 0 the image is compressed N x `repeats` times through the reference's own throughput driver, basis_parallel_compress
// (comp.cpp:5466-5559: a pool of `threads` workers, one basis_compressor -- and with use_opencl one accelerator context -- per image in flight): the JSON line
// then carries the wall seconds of each call, and whether every image's bytes equal the others'.
//
// Allocator policy of THIS application (all three link variants alike): a compressor allocates and frees ~300 MB of multi-megabyte arrays per image; with glibc's
// defaults each of them is a fresh mapping whose pages fault in one by one and is unmapped again on free. main() keeps large blocks on the heap instead
// (mallopt; BU_BENCH_KEEP_MALLOC_DEFAULTS=1 leaves glibc alone). Until round 4 the frontend LIBRARY did this to whatever process loaded it -- which only the
// resident variant did; a library has no business there (it recycles its own blocks privately now: csrc/host/block_pool.cpp), an application does.
#include <malloc.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "encoder/basisu_comp.h"

using namespace basisu;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 9) { std::fprintf(stderr, "usage: %s raw w h quality level threads use_opencl repeats [out.basis]\n", argv[0]); return 2; }
    const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]);
    const int quality = std::atoi(argv[4]), level = std::atoi(argv[5]);
    const uint32_t threads = (uint32_t)std::atoi(argv[6]);
    const bool use_opencl = std::atoi(argv[7]) != 0;
    const int repeats = std::atoi(argv[8]);
    if (!std::getenv("BU_BENCH_KEEP_MALLOC_DEFAULTS")) {
        mallopt(M_MMAP_THRESHOLD, 1 << 30);
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 64 << 20);
    }
    basisu_encoder_init(use_opencl, false);
    image img(w, h);
    {
        FILE* f = std::fopen(argv[1], "rb");
        if (!f || std::fread(img.get_ptr(), 4, (size_t)w * h, f) != (size_t)w * h) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
        std::fclose(f);
    }
    const int parallel = argc > 10 ? std::atoi(argv[10]) : 0;
    if (parallel > 0) {
        auto make = [&] {
            basis_compressor_params p;
            p.m_source_images.push_back(img);
            p.m_use_opencl = use_opencl;
            p.m_perceptual = true;
            p.m_ktx2_and_basis_srgb_transfer_function = true;
            p.m_quality_level = quality;
            p.m_etc1s_compression_level = level;
            p.m_write_output_basis_or_ktx2_files = false;
            p.m_compute_stats = false;
            p.m_status_output = false;
            p.m_debug = false;
            return p;
        };
        basisu::vector<basis_compressor_params> pv;
        for (int i = 0; i < parallel; i++) pv.push_back(make());
        std::vector<double> t_call;
        uint64_t hash0 = 0; size_t bytes0 = 0; bool all_same = true, ok = true;
        for (int r = 0; r < repeats; r++) {
            basisu::vector<parallel_results> res;
            const double t0 = now();
            ok &= basis_parallel_compress(threads < 1 ? 1 : threads, pv, res);
            t_call.push_back(now() - t0);
            for (size_t i = 0; i < res.size(); i++) {
                const uint8_vec& out = res[i].m_basis_file;
                uint64_t h = 1469598103934665603ull;
                for (size_t k = 0; k < out.size(); k++) { h ^= out[k]; h *= 1099511628211ull; }
                if (r == 0 && i == 0) { hash0 = h; bytes0 = out.size(); }
                all_same &= h == hash0 && out.size() == bytes0;
            }
        }
        if (!ok) { std::fprintf(stderr, "basis_parallel_compress failed\n"); return 1; }
        std::printf("{\"width\": %u, \"height\": %u, \"quality\": %d, \"level\": %d, \"threads\": %u, \"use_opencl\": %d, \"parallel_images\": %d, \"bytes\": %zu, \"fnv1a64\": \"%016llx\", "
                    "\"all_images_identical\": %s, \"call_s\": [", w, h, quality, level, threads, use_opencl ? 1 : 0, parallel, bytes0, (unsigned long long)hash0, all_same ? "true" : "false");
        for (size_t i = 0; i < t_call.size(); i++) std::printf("%s%.6f", i ? ", " : "", t_call[i]);
        std::printf("]}\n");
        basisu_encoder_deinit();
        return 0;
    }
    job_pool jpool(threads < 1 ? 1 : threads);   // total pool size including the calling thread (basisu_tool.cpp:2345-2346)
    std::vector<double> t_init, t_process;
    uint64_t hash = 0; size_t bytes = 0;
    for (int r = 0; r < repeats; r++) {
        basis_compressor_params p;
        p.m_source_images.push_back(img);
        p.m_pJob_pool = &jpool;
        p.m_multithreading = threads > 1;
        p.m_use_opencl = use_opencl;
        p.m_perceptual = true;
        p.m_ktx2_and_basis_srgb_transfer_function = true;
        p.m_quality_level = quality;
        p.m_etc1s_compression_level = level;
        p.m_write_output_basis_or_ktx2_files = false;
        p.m_compute_stats = false;
        p.m_status_output = false;
        p.m_debug = false;
        basis_compressor c;
        const double t0 = now();
        if (!c.init(p)) { std::fprintf(stderr, "basis_compressor::init failed\n"); return 1; }
        const double t1 = now();
        const basis_compressor::error_code ec = c.process();
        const double t2 = now();
        if (ec != basis_compressor::cECSuccess) { std::fprintf(stderr, "basis_compressor::process failed: %d\n", (int)ec); return 1; }
        t_init.push_back(t1 - t0); t_process.push_back(t2 - t1);
        const uint8_vec& out = c.get_output_basis_file();
        bytes = out.size();
        hash = 1469598103934665603ull;
        for (size_t i = 0; i < out.size(); i++) { hash ^= out[i]; hash *= 1099511628211ull; }
        if (argc > 9 && r == 0 && std::strcmp(argv[9], "-") != 0) { FILE* f = std::fopen(argv[9], "wb"); if (f) { std::fwrite(out.data(), 1, out.size(), f); std::fclose(f); } }
    }
    std::printf("{\"width\": %u, \"height\": %u, \"quality\": %d, \"level\": %d, \"threads\": %u, \"use_opencl\": %d, \"bytes\": %zu, \"fnv1a64\": \"%016llx\", \"init_s\": [", w, h, quality,
                level, threads, use_opencl ? 1 : 0, bytes, (unsigned long long)hash);
    for (size_t i = 0; i < t_init.size(); i++) std::printf("%s%.6f", i ? ", " : "", t_init[i]);
    std::printf("], \"process_s\": [");
    for (size_t i = 0; i < t_process.size(); i++) std::printf("%s%.6f", i ? ", " : "", t_process[i]);
    std::printf("]}\n");
    basisu_encoder_deinit();
    return 0;
}
