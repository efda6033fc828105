// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or called from the product.
//
// A thin C ABI (for ctypes) over the REAL reference encoder, compiled from the reference sources where they lie
// (-I/root/reference, see oracle/Makefile). It lets tests/ drive the reference's own hot-path functions directly and
// snapshot every intermediate of basisu_frontend so that the HIP path can be diffed stage by stage:
//   * etc1_optimizer (encoder/basisu_etc.cpp:776-1278) on arbitrary pixel lists
//   * basisu_frontend stage methods (encoder/basisu_frontend.cpp:733-2715) called one at a time in the same order
//     as basisu_frontend::compress() (frontend.cpp:159-316), single-threaded job pool (SURVEY hazard H1)
//   * tree_vector_quant / generate_hierarchical_codebook_threaded (encoder/basisu_enc.h:1546-2354)
//   * encode_uastc (encoder/basisu_uastc_enc.cpp:3126)
// All code in this file is ours; the reference is only #included and linked.
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <functional>
#include <thread>
#include <mutex>
#include <atomic>
#include <condition_variable>
#include <random>
#include <memory>
#include <limits>
#include <sstream>
#include <iostream>

// The frontend keeps all stage state private; the harness needs to read it. Access specifiers do not change layout
// under the Itanium ABI, and the reference objects themselves are compiled untouched.
#define private public
#define protected public
#include "encoder/basisu_frontend.h"
#include "encoder/basisu_uastc_enc.h"
#include "encoder/basisu_comp.h"
#include "encoder/basisu_bc7enc.h"
#include "encoder/basisu_gpu_texture.h"
#undef private
#undef protected

using namespace basisu;

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {
struct frontend_handle {
	basisu_frontend fe;
	job_pool jp;
	std::vector<pixel_block> blocks;
	basisu_frontend::params p;
	std::unique_ptr<basisu_backend> be;  // see ref_backend_run
	explicit frontend_handle(uint32_t threads = 1) : jp(threads < 1 ? 1 : threads) {}
};

template <typename T>
uint64_t emit(const std::vector<T>& v, void* buf, uint64_t cap) {
	const uint64_t need = (uint64_t)v.size() * sizeof(T);
	if (buf && cap >= need && need) memcpy(buf, v.data(), need);
	return need;
}

// CSR serialisation of a vector<uint_vec>: [n, off_0..off_n, idx...] as u32
std::vector<uint32_t> csr(const basisu::vector<uint_vec>& lists) {
	std::vector<uint32_t> out;
	out.push_back((uint32_t)lists.size());
	uint32_t ofs = 0;
	for (uint32_t i = 0; i < lists.size(); i++) { out.push_back(ofs); ofs += (uint32_t)lists[i].size(); }
	out.push_back(ofs);
	for (uint32_t i = 0; i < lists.size(); i++)
		for (uint32_t j = 0; j < lists[i].size(); j++) out.push_back(lists[i][j]);
	return out;
}

basis_etc_quality level_to_block_quality(int level) {
	// frontend.cpp:783-788 (level 0 fast, 1 medium, 6 uber, else the basis_etc1_pack_params default = slow)
	if (level == 0) return cETCQualityFast;
	if (level == 1) return cETCQualityMedium;
	if (level == (int)BASISU_MAX_ETC1S_COMPRESSION_LEVEL) return cETCQualityUber;
	return cETCQualitySlow;
}
} // namespace

REF_API int ref_init(void) {
	return basisu_encoder_init(false, false) ? 1 : 0;
}

// Runs the reference etc1_optimizer over n RGBA pixels exactly as init_etc1_images / generate_endpoint_codebook do.
// quality: 0 fast, 1 medium, 2 slow, 3 uber (basis_etc_quality). out_selectors may be NULL.
REF_API int ref_etc1_optimize(const uint8_t* rgba, uint32_t n, int quality, int perceptual,
	uint8_t* out_color5, uint32_t* out_inten, uint64_t* out_err, uint8_t* out_selectors) {
	etc1_optimizer opt;
	etc1_optimizer::params prm;
	etc1_optimizer::results res;
	prm.m_quality = (basis_etc_quality)quality;
	prm.m_num_src_pixels = n;
	prm.m_pSrc_pixels = reinterpret_cast<const color_rgba*>(rgba);
	prm.m_use_color4 = false;
	prm.m_perceptual = perceptual != 0;
	std::vector<uint8_t> sels(n ? n : 1);
	res.m_pSelectors = sels.data();
	res.m_n = n;
	opt.init(prm, res);
	if (!opt.compute()) return 0;
	out_color5[0] = res.m_block_color_unscaled.r;
	out_color5[1] = res.m_block_color_unscaled.g;
	out_color5[2] = res.m_block_color_unscaled.b;
	*out_inten = res.m_block_inten_table;
	*out_err = res.m_error;
	if (out_selectors) memcpy(out_selectors, sels.data(), n);
	return 1;
}

// init_etc1_images' CPU branch (frontend.cpp:765-818) over a bare array of pixel blocks.
REF_API void ref_encode_etc1s_blocks(const uint8_t* pixel_blocks, uint32_t n_blocks, int comp_level, int perceptual, uint8_t* out_blocks) {
	const pixel_block* src = reinterpret_cast<const pixel_block*>(pixel_blocks);
	etc_block* dst = reinterpret_cast<etc_block*>(out_blocks);
	for (uint32_t i = 0; i < n_blocks; i++) {
		etc1_optimizer opt;
		etc1_optimizer::params prm;
		etc1_optimizer::results res;
		prm.m_quality = level_to_block_quality(comp_level);
		prm.m_num_src_pixels = 16;
		prm.m_pSrc_pixels = src[i].get_ptr();
		prm.m_perceptual = perceptual != 0;
		uint8_t sels[16];
		res.m_pSelectors = sels;
		res.m_n = 16;
		opt.init(prm, res);
		if (!opt.compute()) abort();
		etc_block& blk = dst[i];
		memset(&blk, 0, sizeof(blk));
		blk.set_block_color5_etc1s(res.m_block_color_unscaled);
		blk.set_inten_tables_etc1s(res.m_block_inten_table);
		blk.set_flip_bit(true);
		for (uint32_t y = 0; y < 4; y++)
			for (uint32_t x = 0; x < 4; x++)
				blk.set_selector(x, y, sels[x + y * 4]);
	}
}

// etc_block::determine_selectors (etc.h:374-436) for blocks with given colour5+inten, flip=1, delta3=0.
REF_API void ref_determine_selectors(const uint8_t* pixel_blocks, uint32_t n_blocks, const uint8_t* color5_inten /*4B each*/, int perceptual, uint8_t* out_blocks) {
	const pixel_block* src = reinterpret_cast<const pixel_block*>(pixel_blocks);
	etc_block* dst = reinterpret_cast<etc_block*>(out_blocks);
	for (uint32_t i = 0; i < n_blocks; i++) {
		etc_block& blk = dst[i];
		memset(&blk, 0, sizeof(blk));
		color_rgba c(color5_inten[i * 4 + 0], color5_inten[i * 4 + 1], color5_inten[i * 4 + 2], 255);
		blk.set_block_color5(c, c);
		blk.set_flip_bit(true);
		blk.set_inten_table(0, color5_inten[i * 4 + 3]);
		blk.set_inten_table(1, color5_inten[i * 4 + 3]);
		blk.determine_selectors(src[i].get_ptr(), perceptual != 0);
	}
}

REF_API uint32_t ref_color_distance(int perceptual, const uint8_t* a, const uint8_t* b) {
	return color_distance(perceptual != 0, color_rgba(a[0], a[1], a[2], 255), color_rgba(b[0], b[1], b[2], 255), false);
}

// ---------------------------------------------------------------- frontend, stage by stage

// threads: total size of the job pool including the caller (1 = the pinned, single-threaded configuration; more = what the tool does by
// default, basisu_tool.cpp:2331-2348 -- NOT bit-identical to 1 thread above 262,144 distinct selector vectors, SURVEY hazard H1)
REF_API void* ref_frontend_create_mt(const uint8_t* pixel_blocks, uint32_t n_blocks, uint32_t max_endpoint_clusters,
	uint32_t max_selector_clusters, int comp_level, int perceptual, uint32_t threads);
REF_API void* ref_frontend_create(const uint8_t* pixel_blocks, uint32_t n_blocks, uint32_t max_endpoint_clusters,
	uint32_t max_selector_clusters, int comp_level, int perceptual) {
	return ref_frontend_create_mt(pixel_blocks, n_blocks, max_endpoint_clusters, max_selector_clusters, comp_level, perceptual, 1);
}
REF_API void* ref_frontend_create_mt(const uint8_t* pixel_blocks, uint32_t n_blocks, uint32_t max_endpoint_clusters,
	uint32_t max_selector_clusters, int comp_level, int perceptual, uint32_t threads) {
	frontend_handle* h = new frontend_handle(threads);
	h->blocks.resize(n_blocks);
	memcpy(h->blocks.data(), pixel_blocks, (size_t)n_blocks * sizeof(pixel_block));
	basisu_frontend::params& p = h->p;
	p.m_num_source_blocks = n_blocks;
	p.m_pSource_blocks = h->blocks.data();
	p.m_max_endpoint_clusters = max_endpoint_clusters;
	p.m_max_selector_clusters = max_selector_clusters;
	p.m_perceptual = perceptual != 0;
	p.m_compression_level = comp_level;
	p.m_tex_type = basist::cBASISTexType2D;
	p.m_multithreaded = threads > 1;
	p.m_validate = false;
	p.m_pJob_pool = &h->jp;
	p.m_pGlobal_codebooks = nullptr;
	p.m_pOpenCL_context = nullptr;
	if (!h->fe.init(p)) { delete h; return nullptr; }
	h->fe.m_total_blocks = n_blocks;
	h->fe.m_total_pixels = n_blocks * 16;
	return h;
}

REF_API void ref_frontend_destroy(void* hv) { delete static_cast<frontend_handle*>(hv); }

// Calls one private stage method by name. Returns the method's return value (or 1), -1 if unknown.
REF_API int64_t ref_frontend_call(void* hv, const char* name, uint32_t arg) {
	basisu_frontend& fe = static_cast<frontend_handle*>(hv)->fe;
	const std::string n(name);
	if (n == "compress") return fe.compress() ? 1 : 0;
	if (n == "init_etc1_images") { fe.init_etc1_images(); return 1; }
	if (n == "init_endpoint_training_vectors") { fe.init_endpoint_training_vectors(); return 1; }
	if (n == "generate_endpoint_clusters") { fe.generate_endpoint_clusters(); return 1; }
	if (n == "introduce_new_endpoint_clusters") { fe.introduce_new_endpoint_clusters(); return 1; }
	if (n == "generate_endpoint_codebook") { fe.generate_endpoint_codebook(arg); return 1; }
	if (n == "refine_endpoint_clusterization") return fe.refine_endpoint_clusterization();
	if (n == "eliminate_redundant_or_empty_endpoint_clusters") { fe.eliminate_redundant_or_empty_endpoint_clusters(); return 1; }
	if (n == "generate_block_endpoint_clusters") { fe.generate_block_endpoint_clusters(); return 1; }
	if (n == "create_initial_packed_texture") { fe.create_initial_packed_texture(); return 1; }
	if (n == "generate_selector_clusters") { fe.generate_selector_clusters(); return 1; }
	if (n == "compute_selector_clusters_within_each_parent_cluster") { fe.compute_selector_clusters_within_each_parent_cluster(); return 1; }
	if (n == "create_optimized_selector_codebook") { fe.create_optimized_selector_codebook(arg); return 1; }
	if (n == "find_optimal_selector_clusters_for_each_block") { fe.find_optimal_selector_clusters_for_each_block(); return 1; }
	if (n == "introduce_special_selector_clusters") { fe.introduce_special_selector_clusters(); return 1; }
	if (n == "refine_block_endpoints_given_selectors") return fe.refine_block_endpoints_given_selectors();
	if (n == "optimize_selector_codebook") { fe.optimize_selector_codebook(); return 1; }
	if (n == "finalize") { fe.finalize(); return 1; }
	if (n == "use_hierarchical_endpoint_codebooks") return fe.m_use_hierarchical_endpoint_codebooks;
	if (n == "use_hierarchical_selector_codebooks") return fe.m_use_hierarchical_selector_codebooks;
	if (n == "endpoint_refinement") return fe.m_endpoint_refinement;
	if (n == "num_endpoint_codebook_iterations") return fe.m_num_endpoint_codebook_iterations;
	if (n == "num_selector_codebook_iterations") return fe.m_num_selector_codebook_iterations;
	return -1;
}

// Serialises one piece of frontend state. Returns the number of bytes required; copies only if cap is large enough.
REF_API uint64_t ref_frontend_get(void* hv, const char* name, void* buf, uint64_t cap) {
	basisu_frontend& fe = static_cast<frontend_handle*>(hv)->fe;
	const std::string n(name);
	auto raw = [&](const void* p, uint64_t bytes) -> uint64_t {
		if (buf && cap >= bytes && bytes) memcpy(buf, p, bytes);
		return bytes;
	};
	if (n == "etc1_blocks") return raw(fe.m_etc1_blocks_etc1s.data(), fe.m_etc1_blocks_etc1s.size() * 8ull);
	if (n == "encoded_blocks") return raw(fe.m_encoded_blocks.data(), fe.m_encoded_blocks.size() * 8ull);
	if (n == "orig_encoded_blocks") return raw(fe.m_orig_encoded_blocks.data(), fe.m_orig_encoded_blocks.size() * 8ull);
	if (n == "optimized_cluster_selectors") return raw(fe.m_optimized_cluster_selectors.data(), fe.m_optimized_cluster_selectors.size() * 8ull);
	if (n == "block_selector_cluster_index") return raw(fe.m_block_selector_cluster_index.data(), fe.m_block_selector_cluster_index.size() * 4ull);
	if (n == "block_parent_endpoint_cluster") return raw(fe.m_block_parent_endpoint_cluster.data(), fe.m_block_parent_endpoint_cluster.size());
	if (n == "block_parent_selector_cluster") return raw(fe.m_block_parent_selector_cluster.data(), fe.m_block_parent_selector_cluster.size());
	if (n == "endpoint_clusters") return emit(csr(fe.m_endpoint_clusters), buf, cap);
	if (n == "endpoint_parent_clusters") return emit(csr(fe.m_endpoint_parent_clusters), buf, cap);
	if (n == "endpoint_clusters_within_each_parent_cluster") return emit(csr(fe.m_endpoint_clusters_within_each_parent_cluster), buf, cap);
	if (n == "selector_cluster_block_indices") return emit(csr(fe.m_selector_cluster_block_indices), buf, cap);
	if (n == "selector_parent_cluster_block_indices") return emit(csr(fe.m_selector_parent_cluster_block_indices), buf, cap);
	if (n == "selector_clusters_within_each_parent_cluster") return emit(csr(fe.m_selector_clusters_within_each_parent_cluster), buf, cap);
	if (n == "block_endpoint_clusters_indices") {
		std::vector<uint32_t> v(fe.m_block_endpoint_clusters_indices.size());
		for (size_t i = 0; i < v.size(); i++) v[i] = fe.m_block_endpoint_clusters_indices[(uint32_t)i][0];
		return emit(v, buf, cap);
	}
	if (n == "endpoint_cluster_etc_params") {
		// per cluster: r, g, b, inten (4 bytes), then a u8 valid flag padded to 8 bytes total, then u64 color_error
		std::vector<uint8_t> v(fe.m_endpoint_cluster_etc_params.size() * 16);
		for (size_t i = 0; i < fe.m_endpoint_cluster_etc_params.size(); i++) {
			const auto& e = fe.m_endpoint_cluster_etc_params[(uint32_t)i];
			v[i * 16 + 0] = e.m_color_unscaled[0].r;
			v[i * 16 + 1] = e.m_color_unscaled[0].g;
			v[i * 16 + 2] = e.m_color_unscaled[0].b;
			v[i * 16 + 3] = (uint8_t)e.m_inten_table[0];
			v[i * 16 + 4] = e.m_valid ? 1 : 0;
			memcpy(&v[i * 16 + 8], &e.m_color_error[0], 8);
		}
		return emit(v, buf, cap);
	}
	if (n == "endpoint_training_vecs") {
		// per training vec: 6 floats + u64 weight = 32 bytes
		const auto& tv = fe.m_endpoint_clusterizer.get_training_vecs();
		std::vector<uint8_t> v(tv.size() * 32);
		for (size_t i = 0; i < tv.size(); i++) {
			memcpy(&v[i * 32], tv[(uint32_t)i].first.get_ptr(), 24);
			memcpy(&v[i * 32 + 24], &tv[(uint32_t)i].second, 8);
		}
		return emit(v, buf, cap);
	}
	return ~0ull;
}

// ---------------------------------------------------------------- TSVQ

// generate_hierarchical_codebook_threaded (enc.h:2218) single-threaded over n weighted vectors of dimension dim (6 or 16).
// Outputs CSR blobs ([n, offsets(n+1), indices]) for the codebook and the parent codebook into caller buffers.
REF_API int ref_tsvq(uint32_t dim, const float* vecs, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size,
	uint32_t max_parent_codebook_size, int even_odd_pairs_equal,
	uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words) {
	basisu::vector<uint_vec> codebook, parent;
	job_pool jp(1);
	bool ok = false;
	if (dim == 6) {
		tree_vector_quant<vec<6, float>> q;
		for (uint32_t i = 0; i < n; i++) { vec<6, float> v; for (uint32_t k = 0; k < 6; k++) v[k] = vecs[i * 6 + k]; q.add_training_vec(v, weights[i]); }
		ok = generate_hierarchical_codebook_threaded(q, max_codebook_size, max_parent_codebook_size, codebook, parent, 0, &jp, even_odd_pairs_equal != 0);
	} else if (dim == 16) {
		tree_vector_quant<vec16F> q;
		for (uint32_t i = 0; i < n; i++) { vec16F v; for (uint32_t k = 0; k < 16; k++) v[k] = vecs[i * 16 + k]; q.add_training_vec(v, weights[i]); }
		ok = generate_hierarchical_codebook_threaded(q, max_codebook_size, max_parent_codebook_size, codebook, parent, 0, &jp, even_odd_pairs_equal != 0);
	}
	if (!ok) return 0;
	std::vector<uint32_t> a = csr(codebook), b = csr(parent);
	if (a.size() > cap_codebook_words || b.size() > cap_parent_words) return -1;
	memcpy(out_codebook, a.data(), a.size() * 4);
	memcpy(out_parent, b.data(), b.size() * 4);
	return 1;
}

// The same with the job pool and thread count the multi-threaded tool passes (frontend.cpp:2195-2204). `internal` != 0 calls
// generate_hierarchical_codebook_threaded_internal (enc.h:2086-2215) directly on the rows as given (they must be distinct; limit_clusterizers
// as the outer function derives it, enc.h:2305-2307), so that the T-way partition can be exercised below the outer function's 262,144 gate (enc.h:2316).
REF_API int ref_tsvq_mt(uint32_t dim, const float* vecs, const uint64_t* weights, uint32_t n, uint32_t max_codebook_size,
	uint32_t max_parent_codebook_size, uint32_t max_threads, int internal,
	uint32_t* out_codebook, uint64_t cap_codebook_words, uint32_t* out_parent, uint64_t cap_parent_words) {
	basisu::vector<uint_vec> codebook, parent;
	job_pool jp(max_threads ? max_threads : 1);
	bool ok = false;
	const bool limit = n > max_codebook_size;
	if (dim == 6) {
		tree_vector_quant<vec<6, float>> q;
		for (uint32_t i = 0; i < n; i++) { vec<6, float> v; for (uint32_t k = 0; k < 6; k++) v[k] = vecs[i * 6 + k]; q.add_training_vec(v, weights[i]); }
		ok = internal ? generate_hierarchical_codebook_threaded_internal(q, max_codebook_size, max_parent_codebook_size, codebook, parent, max_threads, limit, &jp)
		              : generate_hierarchical_codebook_threaded(q, max_codebook_size, max_parent_codebook_size, codebook, parent, max_threads, &jp, false);
	} else if (dim == 16) {
		tree_vector_quant<vec16F> q;
		for (uint32_t i = 0; i < n; i++) { vec16F v; for (uint32_t k = 0; k < 16; k++) v[k] = vecs[i * 16 + k]; q.add_training_vec(v, weights[i]); }
		ok = internal ? generate_hierarchical_codebook_threaded_internal(q, max_codebook_size, max_parent_codebook_size, codebook, parent, max_threads, limit, &jp)
		              : generate_hierarchical_codebook_threaded(q, max_codebook_size, max_parent_codebook_size, codebook, parent, max_threads, &jp, false);
	}
	if (!ok) return 0;
	std::vector<uint32_t> a = csr(codebook), b = csr(parent);
	if (a.size() > cap_codebook_words || b.size() > cap_parent_words) return -1;
	memcpy(out_codebook, a.data(), a.size() * 4);
	memcpy(out_parent, b.data(), b.size() * 4);
	return 1;
}

// ---------------------------------------------------------------- ETC1S backend, first stage
// basisu_backend::create_encoder_blocks (encoder/basisu_backend.cpp:406-617) on a finished frontend: one 2D slice of num_blocks_x x
// num_blocks_y blocks. Outputs per block (raster order): the endpoint index after the endpoint-prediction RDO remap (:441-586, BEFORE the
// codebook re-sort: at levels > 1 this call applies reoptimize_remapped_endpoints to the frontend, exactly as the reference does), the
// endpoint predictor, the selector index; and the two codebook remap tables the stage ends with.
REF_API int ref_backend_create_encoder_blocks(void* hv, uint32_t num_blocks_x, uint32_t num_blocks_y, float endpoint_rdo_thresh, float selector_rdo_thresh,
	int32_t* out_endpoint_index, uint32_t* out_predictor, int32_t* out_selector_index, uint32_t* out_endpoint_old_to_new, uint32_t* out_selector_new_to_old, double* seconds) {
	frontend_handle* h = static_cast<frontend_handle*>(hv);
	basisu_backend be;
	basisu_backend_params bp;
	bp.m_etc1s = true;
	bp.m_compression_level = h->p.m_compression_level;
	bp.m_endpoint_rdo_quality_thresh = endpoint_rdo_thresh;
	bp.m_selector_rdo_quality_thresh = selector_rdo_thresh;
	bp.m_used_global_codebooks = false;
	basisu_backend_slice_desc_vec slices(1);
	slices[0].m_first_block_index = 0;
	slices[0].m_orig_width = slices[0].m_width = num_blocks_x * 4;
	slices[0].m_orig_height = slices[0].m_height = num_blocks_y * 4;
	slices[0].m_num_blocks_x = num_blocks_x;
	slices[0].m_num_blocks_y = num_blocks_y;
	slices[0].m_num_macroblocks_x = (num_blocks_x + 1) / 2;
	slices[0].m_num_macroblocks_y = (num_blocks_y + 1) / 2;
	slices[0].m_iframe = true;
	if ((uint64_t)num_blocks_x * num_blocks_y != h->fe.m_total_blocks) return 0;
	be.init(&h->fe, bp, slices);
	be.create_endpoint_palette();
	be.create_selector_palette();
	interval_timer tm;
	tm.start();
	be.create_encoder_blocks();
	if (seconds) *seconds = tm.get_elapsed_secs();
	for (uint32_t y = 0; y < num_blocks_y; y++)
		for (uint32_t x = 0; x < num_blocks_x; x++) {
			const encoder_block& m = be.m_slice_encoder_blocks[0](x, y);
			const size_t i = (size_t)y * num_blocks_x + x;
			out_endpoint_index[i] = m.m_endpoint_index;
			out_predictor[i] = m.m_endpoint_predictor;
			out_selector_index[i] = m.m_selector_index;
		}
	if (out_endpoint_old_to_new) for (uint32_t i = 0; i < be.m_endpoint_remap_table_old_to_new.size(); i++) out_endpoint_old_to_new[i] = be.m_endpoint_remap_table_old_to_new[i];
	if (out_selector_new_to_old) for (uint32_t i = 0; i < be.m_selector_remap_table_new_to_old.size(); i++) out_selector_new_to_old[i] = be.m_selector_remap_table_new_to_old[i];
	return 1;
}

// basisu_backend::encode() (backend.cpp:1747) as a whole on a finished frontend, for timing the stages downstream of the hot path.
REF_API uint32_t ref_backend_encode(void* hv, uint32_t num_blocks_x, uint32_t num_blocks_y, float endpoint_rdo_thresh, float selector_rdo_thresh, double* seconds) {
	frontend_handle* h = static_cast<frontend_handle*>(hv);
	basisu_backend be;
	basisu_backend_params bp;
	bp.m_etc1s = true;
	bp.m_compression_level = h->p.m_compression_level;
	bp.m_endpoint_rdo_quality_thresh = endpoint_rdo_thresh;
	bp.m_selector_rdo_quality_thresh = selector_rdo_thresh;
	basisu_backend_slice_desc_vec slices(1);
	slices[0].m_orig_width = slices[0].m_width = num_blocks_x * 4;
	slices[0].m_orig_height = slices[0].m_height = num_blocks_y * 4;
	slices[0].m_num_blocks_x = num_blocks_x;
	slices[0].m_num_blocks_y = num_blocks_y;
	slices[0].m_num_macroblocks_x = (num_blocks_x + 1) / 2;
	slices[0].m_num_macroblocks_y = (num_blocks_y + 1) / 2;
	slices[0].m_iframe = true;
	be.init(&h->fe, bp, slices);
	interval_timer tm;
	tm.start();
	const uint32_t bytes = be.encode();
	if (seconds) *seconds = tm.get_elapsed_secs();
	return bytes;
}

// basisu_backend::encode() on a finished frontend with any number of slices (slices = n x {first_block_index, num_blocks_x, num_blocks_y});
// the backend stays alive in the handle so that ref_backend_get can serialise its output (basisu_backend_output, backend.h:218-276) and
// its per-block state. At compression levels above 1 this MODIFIES the frontend (reoptimize_remapped_endpoints), as the reference does.
// The texture type the frontend (and through it the backend) works under; call before the stages run. 3 = cBASISTexTypeVideoFrames.
REF_API void ref_frontend_set_tex_type(void* hv, uint32_t tex_type) {
	frontend_handle* h = static_cast<frontend_handle*>(hv);
	h->p.m_tex_type = (basist::basis_texture_type)tex_type;
	h->fe.m_params.m_tex_type = (basist::basis_texture_type)tex_type;
}

REF_API uint32_t ref_backend_run(void* hv, const uint32_t* slices3, uint32_t n_slices, float endpoint_rdo_thresh, float selector_rdo_thresh, double* seconds) {
	frontend_handle* h = static_cast<frontend_handle*>(hv);
	h->be.reset(new basisu_backend());
	basisu_backend_params bp;
	bp.m_etc1s = true;
	bp.m_compression_level = h->p.m_compression_level;
	bp.m_endpoint_rdo_quality_thresh = endpoint_rdo_thresh;
	bp.m_selector_rdo_quality_thresh = selector_rdo_thresh;
	basisu_backend_slice_desc_vec slices(n_slices);
	for (uint32_t i = 0; i < n_slices; i++) {
		const uint32_t nbx = slices3[i * 3 + 1], nby = slices3[i * 3 + 2];
		slices[i].m_first_block_index = slices3[i * 3];
		slices[i].m_orig_width = slices[i].m_width = nbx * 4;
		slices[i].m_orig_height = slices[i].m_height = nby * 4;
		slices[i].m_num_blocks_x = nbx;
		slices[i].m_num_blocks_y = nby;
		slices[i].m_num_macroblocks_x = (nbx + 1) / 2;
		slices[i].m_num_macroblocks_y = (nby + 1) / 2;
		slices[i].m_source_file_index = i;
		// as basis_compressor sets it: only video has i-frames, and for ETC1S only the first frame is one (comp.cpp:3016-3030)
		slices[i].m_iframe = (h->fe.m_params.m_tex_type == basist::cBASISTexTypeVideoFrames) && (i == 0);
	}
	h->be->init(&h->fe, bp, slices);
	interval_timer tm;
	tm.start();
	const uint32_t bytes = h->be->encode();
	if (seconds) *seconds = tm.get_elapsed_secs();
	return bytes;
}

REF_API uint64_t ref_backend_get(void* hv, const char* name, uint32_t slice, void* buf, uint64_t cap) {
	frontend_handle* h = static_cast<frontend_handle*>(hv);
	if (!h->be) return ~0ull;
	const basisu_backend& be = *h->be;
	const basisu_backend_output& o = be.get_output();
	const std::string n(name);
	auto bytes = [&](const uint8_vec& v) -> uint64_t { if (buf && cap >= v.size() && v.size()) memcpy(buf, v.data(), v.size()); return v.size(); };
	if (n == "endpoint_palette") return bytes(o.m_endpoint_palette);
	if (n == "selector_palette") return bytes(o.m_selector_palette);
	if (n == "slice_image_tables") return bytes(o.m_slice_image_tables);
	if (n == "slice_image_data") return slice < o.m_slice_image_data.size() ? bytes(o.m_slice_image_data[slice]) : ~0ull;
	if (n == "slice_image_crcs") { std::vector<uint16_t> v(o.m_slice_image_crcs.begin(), o.m_slice_image_crcs.end()); return emit(v, buf, cap); }
	if (n == "num_endpoints") return emit(std::vector<uint32_t>{o.m_num_endpoints}, buf, cap);
	if (n == "num_selectors") return emit(std::vector<uint32_t>{o.m_num_selectors}, buf, cap);
	if (n == "encoder_blocks") {  // u32 x 4 per block in block order: endpoint index, predictor, selector index, history index + 1
		std::vector<uint32_t> v((size_t)h->fe.m_total_blocks * 4, 0);
		for (uint32_t s = 0; s < be.m_slices.size(); s++)
			for (uint32_t y = 0; y < be.m_slices[s].m_num_blocks_y; y++)
				for (uint32_t x = 0; x < be.m_slices[s].m_num_blocks_x; x++) {
					const encoder_block& m = be.m_slice_encoder_blocks[s](x, y);
					const size_t i = (size_t)be.m_slices[s].m_first_block_index + x + (size_t)y * be.m_slices[s].m_num_blocks_x;
					v[i * 4] = m.m_endpoint_index; v[i * 4 + 1] = m.m_endpoint_predictor; v[i * 4 + 2] = m.m_selector_index; v[i * 4 + 3] = (uint32_t)(m.m_selector_history_buf_index + 1);
				}
		return emit(v, buf, cap);
	}
	if (n == "endpoint_remap_old_to_new") { std::vector<uint32_t> v(be.m_endpoint_remap_table_old_to_new.begin(), be.m_endpoint_remap_table_old_to_new.end()); return emit(v, buf, cap); }
	if (n == "selector_remap_new_to_old") { std::vector<uint32_t> v(be.m_selector_remap_table_new_to_old.begin(), be.m_selector_remap_table_new_to_old.end()); return emit(v, buf, cap); }
	return ~0ull;
}

// Overwrites the finished state of the handle's frontend with arbitrary codebooks and assignments (everything the backend reads through the
// getters of frontend.h:119-156), so that the reference backend can be run on states no image would produce (backend fuzzing). The encoded
// blocks are rebuilt from the codebooks: colour5 + table of the block's endpoint cluster, selectors of its selector cluster.
REF_API int ref_frontend_set_state(void* hv, uint32_t n_endpoints, const uint8_t* color5_inten, uint32_t n_selectors, const uint8_t* selectors16,
	const uint32_t* block_endpoint, const uint32_t* block_selector) {
	frontend_handle* h = static_cast<frontend_handle*>(hv);
	basisu_frontend& fe = h->fe;
	const uint32_t n = fe.m_total_blocks;
	fe.m_endpoint_clusters.clear(); fe.m_endpoint_clusters.resize(n_endpoints);
	fe.m_endpoint_cluster_etc_params.clear(); fe.m_endpoint_cluster_etc_params.resize(n_endpoints);
	for (uint32_t i = 0; i < n_endpoints; i++) {
		auto& e = fe.m_endpoint_cluster_etc_params[i];
		e.m_color_unscaled[0] = color_rgba(color5_inten[i * 4], color5_inten[i * 4 + 1], color5_inten[i * 4 + 2], 255);
		e.m_inten_table[0] = color5_inten[i * 4 + 3];
		e.m_color_used[0] = true;
		e.m_valid = true;
	}
	fe.m_selector_cluster_block_indices.clear(); fe.m_selector_cluster_block_indices.resize(n_selectors);
	fe.m_optimized_cluster_selectors.clear(); fe.m_optimized_cluster_selectors.resize(n_selectors);
	for (uint32_t i = 0; i < n_selectors; i++) {
		etc_block& b = fe.m_optimized_cluster_selectors[i];
		memset(&b, 0, sizeof(b));
		for (uint32_t y = 0; y < 4; y++) for (uint32_t x = 0; x < 4; x++) b.set_selector(x, y, selectors16[i * 16 + y * 4 + x] & 3);
	}
	fe.m_block_endpoint_clusters_indices.resize(n);
	fe.m_block_selector_cluster_index.resize(n);
	fe.m_encoded_blocks.resize(n);
	for (uint32_t b = 0; b < n; b++) {
		if (block_endpoint[b] >= n_endpoints || block_selector[b] >= n_selectors) return 0;
		fe.m_block_endpoint_clusters_indices[b][0] = fe.m_block_endpoint_clusters_indices[b][1] = block_endpoint[b];
		fe.m_block_selector_cluster_index[b] = block_selector[b];
		fe.m_endpoint_clusters[block_endpoint[b]].push_back(b * 2); fe.m_endpoint_clusters[block_endpoint[b]].push_back(b * 2 + 1);
		fe.m_selector_cluster_block_indices[block_selector[b]].push_back(b);
		etc_block& blk = fe.m_encoded_blocks[b];
		memset(&blk, 0, sizeof(blk));
		blk.set_diff_bit(true);
		blk.set_flip_bit(true);
		blk.set_block_color5_etc1s(fe.m_endpoint_cluster_etc_params[block_endpoint[b]].m_color_unscaled[0]);
		blk.set_inten_tables_etc1s(fe.m_endpoint_cluster_etc_params[block_endpoint[b]].m_inten_table[0]);
		blk.set_raw_selector_bits(fe.m_optimized_cluster_selectors[block_selector[b]].get_raw_selector_bits());
	}
	return 1;
}

// basisu_frontend::reoptimize_remapped_endpoints (frontend.cpp:2996) on the handle's frontend: what the reference backend calls at levels > 1.
REF_API int ref_frontend_reoptimize(void* hv, const uint32_t* new_block_endpoints, uint32_t n, int32_t* old_to_new, int final_codebook, const uint32_t* block_selector_indices) {
	basisu_frontend& fe = static_cast<frontend_handle*>(hv)->fe;
	uint_vec nb(n);
	for (uint32_t i = 0; i < n; i++) nb[i] = new_block_endpoints[i];
	int_vec o2n;
	uint_vec sel;
	if (block_selector_indices) { sel.resize(n); for (uint32_t i = 0; i < n; i++) sel[i] = block_selector_indices[i]; }
	fe.reoptimize_remapped_endpoints(nb, o2n, final_codebook != 0, block_selector_indices ? &sel : nullptr);
	for (uint32_t i = 0; i < o2n.size(); i++) old_to_new[i] = o2n[i];
	return 1;
}

// basisu_file::init on the kept backend's output (encoder/basisu_basis_file.cpp:290): the .basis container, with optional key-values
// given as n x (key C string, value bytes, value size).
REF_API uint64_t ref_basis_file(void* hv, uint32_t tex_type, uint32_t userdata0, uint32_t userdata1, int y_flipped, uint32_t us_per_frame,
	const char** keys, const uint8_t** values, const uint32_t* value_sizes, uint32_t n_kv, uint8_t* buf, uint64_t cap) {
	frontend_handle* h = static_cast<frontend_handle*>(hv);
	if (!h->be) return 0;
	basist::key_value_vec kvs;
	for (uint32_t i = 0; i < n_kv; i++) {
		basist::key_value kv;
		kv.m_key.append((const uint8_t*)keys[i], strlen(keys[i]) + 1);
		if (value_sizes[i]) kv.m_value.append(values[i], value_sizes[i]);
		kvs.push_back(kv);
	}
	basisu_file f;
	if (!f.init(h->be->get_output(), (basist::basis_texture_type)tex_type, userdata0, userdata1, y_flipped != 0, us_per_frame, kvs)) return 0;
	const uint8_vec& d = f.get_compressed_data();
	if (buf && cap >= d.size() && d.size()) memcpy(buf, d.data(), d.size());
	return d.size();
}

// image_resample (encoder/basisu_enc.cpp:1022), the function behind basis_compressor::generate_mipmaps: RGBA8 in, RGBA8 out (tightly packed).
REF_API int ref_image_resample(const uint8_t* src_rgba, uint32_t src_w, uint32_t src_h, uint8_t* dst_rgba, uint32_t dst_w, uint32_t dst_h, int srgb, const char* filter,
	float filter_scale, int wrapping, uint32_t first_comp, uint32_t num_comps) {
	image src(src_rgba, src_w, src_h, 4), dst(dst_w, dst_h);
	if (!image_resample(src, dst, srgb != 0, filter, filter_scale, wrapping != 0, first_comp, num_comps)) return 0;
	for (uint32_t y = 0; y < dst_h; y++) memcpy(dst_rgba + (size_t)y * dst_w * 4, &dst(0, y), (size_t)dst_w * 4);
	return 1;
}

// The backend's coding tools on their own (known-answer style tests with synthetic inputs).
REF_API uint64_t ref_huffman_table_bytes(const uint32_t* freq, uint32_t n, uint32_t max_code_size, uint8_t* out_sizes, uint16_t* out_codes, uint8_t* out_bytes, uint64_t cap) {
	histogram h(n);
	for (uint32_t i = 0; i < n; i++) h[i] = freq[i];
	huffman_encoding_table t;
	if (!t.init(h, max_code_size)) return ~0ull;
	for (uint32_t i = 0; i < n; i++) { out_sizes[i] = t.get_code_sizes()[i]; out_codes[i] = t.get_codes()[i]; }
	bitwise_coder c;
	c.init(1024);
	c.emit_huffman_table(t);
	c.put_vlc(n, 4);
	c.flush();
	const uint8_vec& b = c.get_bytes();
	if (out_bytes && cap >= b.size() && b.size()) memcpy(out_bytes, b.data(), b.size());
	return b.size();
}
REF_API uint32_t ref_crc16(const uint8_t* data, uint64_t size, uint32_t crc) { return basist::crc16(data, (size_t)size, (uint16_t)crc); }
REF_API void ref_palette_reorder(const uint32_t* indices, uint32_t num_indices, uint32_t num_syms, uint32_t* out_old_to_new) {
	palette_index_reorderer r;
	r.init(num_indices, indices, num_syms, nullptr, nullptr, 0);
	for (uint32_t i = 0; i < num_syms; i++) out_old_to_new[i] = r.get_remap_table()[i];
}

// ---------------------------------------------------------------- UASTC

REF_API void ref_encode_uastc(const uint8_t* pixel_blocks, uint32_t n_blocks, uint32_t flags, uint8_t* out_blocks16) {
	for (uint32_t i = 0; i < n_blocks; i++) {
		basist::uastc_block blk;
		encode_uastc(pixel_blocks + (size_t)i * 64, blk, flags);
		memcpy(out_blocks16 + (size_t)i * 16, &blk, 16);
	}
}


// uastc_rdo (uastc_enc.h:139, uastc_enc.cpp:4095-4163) in place over packed blocks. params: lambda, max_allowed_rms_increase_ratio,
// skip_block_rms_thresh, max_smooth_block_std_dev, smooth_block_max_error_scale; uparams: lz_dict_size, lz_literal_cost,
// endpoint_refinement. total_jobs > 1 splits into strips exactly like the reference's multithreaded path (strips are independent, so the
// thread schedule cannot change the result).
REF_API int ref_uastc_rdo(uint8_t* blocks16, const uint8_t* pixel_blocks, uint32_t n_blocks, const float* params, const uint32_t* uparams,
                          uint32_t flags, uint32_t total_jobs) {
	uastc_rdo_params p;
	p.m_lambda = params[0];
	p.m_max_allowed_rms_increase_ratio = params[1];
	p.m_skip_block_rms_thresh = params[2];
	p.m_max_smooth_block_std_dev = params[3];
	p.m_smooth_block_max_error_scale = params[4];
	p.m_lz_dict_size = uparams[0];
	p.m_lz_literal_cost = uparams[1];
	p.m_endpoint_refinement = uparams[2] != 0;
	if (total_jobs > 1) {
		job_pool jp(total_jobs);
		return uastc_rdo(n_blocks, (basist::uastc_block*)blocks16, (const color_rgba*)pixel_blocks, p, flags, &jp, total_jobs) ? 1 : 0;
	}
	return uastc_rdo(n_blocks, (basist::uastc_block*)blocks16, (const color_rgba*)pixel_blocks, p, flags, nullptr, 0) ? 1 : 0;
}

// ---------------------------------------------------------------- UASTC: tables and staged hooks

// Raw bytes of the reference's extern data / run-time generated tables (after ref_init), for tools/gen_uastc_tables.py
// and for table parity tests. Returns the size in bytes (~0 = unknown name); copies when cap suffices.
REF_API uint64_t ref_table(const char* name, void* buf, uint64_t cap) {
	using namespace basist;
	struct ent { const char* n; const void* p; uint64_t sz; };
	static const ent tabs[] = {
#define T(x) { #x, (const void*)&x, sizeof(x) }
		T(g_uastc_mode_weight_bits), T(g_uastc_mode_weight_ranges), T(g_uastc_mode_endpoint_ranges), T(g_uastc_mode_subsets),
		T(g_uastc_mode_planes), T(g_uastc_mode_comps), T(g_uastc_mode_has_etc1_bias), T(g_uastc_mode_has_bc1_hint0),
		T(g_uastc_mode_has_bc1_hint1), T(g_uastc_mode_has_alpha), T(g_uastc_mode_is_la), T(g_uastc_mode_huff_codes),
		T(g_astc_bc7_patterns2), T(g_astc_bc7_patterns3), T(g_bc7_3_astc2_patterns2),
		T(g_astc_bc7_pattern2_anchors), T(g_astc_bc7_pattern3_anchors), T(g_bc7_3_astc2_patterns2_anchors),
		T(g_astc_to_bc7_partition_index_perm_tables), T(g_bc7_to_astc_partition_index_perm_tables),
		T(g_astc_bise_range_table), T(g_astc_unquant), T(g_astc_sorted_order_unquant),
		T(g_bc7_weights1), T(g_bc7_weights2), T(g_bc7_weights3), T(g_bc7_weights4), T(g_astc_weights4), T(g_astc_weights5),
		T(g_bc7_weights1x), T(g_bc7_weights2x), T(g_bc7_weights3x), T(g_bc7_weights4x), T(g_astc_weights4x), T(g_astc_weights5x),
		T(g_bc7_partition2), T(g_bc7_partition3), T(g_bc7_table_anchor_index_second_subset),
		T(g_bc7_table_anchor_index_third_subset_1), T(g_bc7_table_anchor_index_third_subset_2),
		T(g_etc2_eac_tables), T(g_etc1_inten_tables),
#undef T
	};
	// struct arrays flattened to plain integers so the consumer does not depend on padding
	std::vector<uint32_t> flat;
	if (!strcmp(name, "common_partitions2")) for (auto& d : g_astc_bc7_common_partitions2) { flat.push_back(d.m_bc7); flat.push_back(d.m_astc); flat.push_back(d.m_invert); }
	else if (!strcmp(name, "common_partitions3")) for (auto& d : g_astc_bc7_common_partitions3) { flat.push_back(d.m_bc7); flat.push_back(d.m_astc); flat.push_back(d.m_astc_to_bc7_perm); }
	else if (!strcmp(name, "bc73_astc2_partitions")) for (auto& d : g_bc7_3_astc2_common_partitions) { flat.push_back(d.m_bc73); flat.push_back(d.m_astc2); flat.push_back(d.k); }
	if (!flat.empty()) { const uint64_t need = flat.size() * 4; if (buf && cap >= need) memcpy(buf, flat.data(), need); return need; }
	for (const ent& e : tabs)
		if (!strcmp(e.n, name)) { if (buf && cap >= e.sz) memcpy(buf, e.p, e.sz); return e.sz; }
	return ~0ull;
}

// color_cell_compression (encoder/basisu_bc7enc.cpp:1364) exactly as basisu_uastc_enc.cpp calls it: mode 255, ASTC endpoint range,
// unit channel weights, non-perceptual. out: [0..3] low astc endpoint, [4..7] high, [8..23] selectors. Returns the error.
REF_API uint64_t ref_color_cell_compression(const uint8_t* rgba, uint32_t n, uint32_t weight_bits, uint32_t endpoint_range, int has_alpha,
	uint32_t uber_level, uint32_t ls_passes, const uint8_t* force_selectors, uint8_t* out24) {
	static const uint32_t* wt[6] = { nullptr, basist::g_bc7_weights1, basist::g_bc7_weights2, basist::g_bc7_weights3, basist::g_astc_weights4, basist::g_astc_weights5 };
	static const float* wx[6] = { nullptr, g_bc7_weights1x, g_bc7_weights2x, g_bc7_weights3x, g_astc_weights4x, g_astc_weights5x };
	color_cell_compressor_params p; memset(&p, 0, sizeof(p));
	p.m_num_pixels = n; p.m_pPixels = (const basist::color_quad_u8*)rgba;
	p.m_num_selector_weights = 1u << weight_bits; p.m_pSelector_weights = wt[weight_bits]; p.m_pSelector_weightsx = (const bc7enc_vec4F*)wx[weight_bits];
	p.m_astc_endpoint_range = endpoint_range; p.m_weights[0] = p.m_weights[1] = p.m_weights[2] = p.m_weights[3] = 1;
	p.m_has_alpha = has_alpha; p.m_pForce_selectors = force_selectors;
	bc7enc_compress_block_params cp; memset(&cp, 0, sizeof(cp));
	cp.m_max_partitions_mode1 = 64; cp.m_least_squares_passes = ls_passes; cp.m_weights[0] = cp.m_weights[1] = cp.m_weights[2] = cp.m_weights[3] = 1; cp.m_uber_level = uber_level;
	color_cell_compressor_results r; memset(&r, 0, sizeof(r));
	uint8_t sel[16] = {0}, tmp[16] = {0};
	r.m_pSelectors = sel; r.m_pSelectors_temp = tmp;
	const uint64_t err = color_cell_compression(255, &p, &r, &cp);
	memcpy(out24, r.m_astc_low_endpoint.m_c, 4); memcpy(out24 + 4, r.m_astc_high_endpoint.m_c, 4); memcpy(out24 + 8, sel, 16);
	return err;
}

REF_API uint64_t ref_ccell_est(uint32_t weight_bits, uint32_t comps, const uint8_t* rgba, uint32_t n, uint64_t best_so_far) {
	static const uint32_t* wt[6] = { nullptr, basist::g_bc7_weights1, basist::g_bc7_weights2, basist::g_bc7_weights3, basist::g_astc_weights4, basist::g_astc_weights5 };
	const uint32_t w[4] = { 1, 1, 1, 1 };
	return color_cell_compression_est_astc(1u << weight_bits, comps, wt[weight_bits], n, (const basist::color_quad_u8*)rgba, best_so_far, w);
}

// 16-byte UASTC block -> 16 RGBA pixels via the reference transcoder (basisu_transcoder.cpp:15886)
REF_API int ref_unpack_uastc(const uint8_t* blk16, uint8_t* out_rgba64) {
	basist::uastc_block b; memcpy(&b, blk16, 16);
	return basist::unpack_uastc(b, (basist::color32*)out_rgba64, false) ? 1 : 0;
}
// 16-byte UASTC block -> BC7 (transcode_uastc_to_bc7, :16537) -> pixels (unpack_block cBC7)
REF_API int ref_uastc_to_bc7_pixels(const uint8_t* blk16, uint8_t* out_bc7_16, uint8_t* out_rgba64) {
	basist::uastc_block b; memcpy(&b, blk16, 16);
	if (!basist::transcode_uastc_to_bc7(b, out_bc7_16)) return 0;
	return unpack_block(texture_format::cBC7, out_bc7_16, (color_rgba*)out_rgba64, false) ? 1 : 0;
}
// basist::encode_bc1 (:18047) + unpack
REF_API void ref_encode_bc1(const uint8_t* rgba64, uint32_t flags, uint8_t* out8, uint8_t* out_rgba64) {
	basist::encode_bc1(out8, rgba64, flags);
	if (out_rgba64) unpack_block(texture_format::cBC1, out8, (color_rgba*)out_rgba64, false);
}
REF_API void ref_pack_etc1_solid(const uint8_t* rgb, uint8_t* out8) {
	etc_block b; memset(&b, 0, sizeof(b));
	pack_etc1_block_solid_color(b, rgb);
	memcpy(out8, &b, 8);
}

// ---------------------------------------------------------------- whole-encoder anchors (quality -> cluster counts)

// Mirrors nothing: just runs the real basis_compressor on a raw RGBA image and reports the frontend parameters it chose
// (comp.cpp:3325-3420) plus the .basis bytes, so tests can pin the quality->codebook-size mapping of the host mirror.
REF_API int ref_compress_etc1s(const uint8_t* rgba, uint32_t w, uint32_t h, int quality, int comp_level, int perceptual,
	uint32_t* out_max_endpoint_clusters, uint32_t* out_max_selector_clusters, uint8_t* out_file, uint64_t cap, uint64_t* out_size) {
	job_pool jp(1);
	basis_compressor_params params;
	params.m_source_images.resize(1);
	params.m_source_images[0].init(rgba, w, h, 4);
	params.m_quality_level = quality;
	params.m_etc1s_compression_level = comp_level;
	params.m_perceptual = perceptual != 0;
	params.m_multithreading = false;
	params.m_pJob_pool = &jp;
	params.m_status_output = false;
	params.m_compute_stats = false;
	params.m_mip_gen = false;
	params.m_check_for_alpha = true;
	params.m_uastc = false;
	basis_compressor comp;
	if (!comp.init(params)) return 0;
	if (comp.process() != basis_compressor::cECSuccess) return 0;
	const auto& fp = comp.m_frontend.get_params();
	*out_max_endpoint_clusters = fp.m_max_endpoint_clusters;
	*out_max_selector_clusters = fp.m_max_selector_clusters;
	const uint8_vec& f = comp.get_output_basis_file();
	*out_size = f.size();
	if (out_file && cap >= f.size()) memcpy(out_file, f.data(), f.size());
	return 1;
}

// basis_compressor_params::set_format_mode_and_quality_effort (comp.cpp:76-205) for the two LDR formats of this package: what the unified
// -quality / -effort pair turns into. out = { ETC1S quality level, ETC1S compression level, UASTC pack flags, UASTC RDO flag }, *lambda = the RDO scalar.
REF_API int ref_quality_effort(int uastc, int quality, int effort, int32_t* out, float* lambda) {
	basis_compressor_params params;
	if (!params.set_format_mode_and_quality_effort(uastc ? basist::basis_tex_format::cUASTC_LDR_4x4 : basist::basis_tex_format::cETC1S, quality, effort, true)) return 0;
	out[0] = params.m_quality_level;
	out[1] = params.m_etc1s_compression_level;
	out[2] = (int32_t)(uint32_t)params.m_pack_uastc_ldr_4x4_flags;
	out[3] = params.m_rdo_uastc_ldr_4x4 ? 1 : 0;
	*lambda = params.m_rdo_uastc_ldr_4x4_quality_scalar;
	return 1;
}
