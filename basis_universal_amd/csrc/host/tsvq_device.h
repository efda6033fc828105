// tsvq_device.h -- host driver of the device TSVQ (row a8): the tree, the variance priority queue and the order in which nodes
// are split stay exactly as in tree_vector_quant<>::generate (encoder/basisu_enc.h:1616-1660); the splits themselves
// (split_node, enc.h:1737-1800) run on the GPU through bu_hip_tsvq_split, many nodes per launch.
//
// A node's split depends only on the node, so the driver may compute splits AHEAD of the reference's schedule: it replays the
// priority queue until it reaches a node whose split is not known yet, then asks the device to split every such candidate in
// the queue at once (bounded by the number of splits still needed), and resumes the replay. Speculative results that the
// replay never reaches are simply dropped, so the final tree is the reference's tree.
//
// The reference's multi-threaded configuration (generate_hierarchical_codebook_threaded_internal, enc.h:2086-2215; the tool's default) builds
// a T-leaf tree and then T independent trees, one per leaf: here those T trees share every device round (run_trees), see build().
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <type_traits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/basisu_hip.h"
#include "../../../include/basisu_hip_frontend.h"
#include "tsvq.h"

namespace bu {

// Host threads per parallel region of the bookkeeping between device stages: 8 unless BU_HOST_THREADS says otherwise (1..32).
inline unsigned host_threads() {
    static const unsigned t = [] {
        const unsigned hw = std::thread::hardware_concurrency();
        unsigned want = 8;
        if (const char* e = std::getenv("BU_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 32) want = (unsigned)v; }
        return hw ? (want < hw ? want : hw) : 1u;
    }();
    return t;
}


// The original training-vector indices behind every distinct vector: either one std::vector per distinct vector, or CSR
// (offsets[u] .. offsets[u+1] into one index array), which is what a sort-based de-duplication yields for free.
struct vec_groups {
    const std::vector<std::vector<uint32_t>>* g;
    size_t size(uint32_t u) const { return (*g)[u].size(); }
    void copy(uint32_t u, uint32_t* dst) const { std::memcpy(dst, (*g)[u].data(), (*g)[u].size() * sizeof(uint32_t)); }
};
struct csr_groups {
    const uint32_t* offsets; const uint32_t* items;
    size_t size(uint32_t u) const { return offsets[u + 1] - offsets[u]; }
    void copy(uint32_t u, uint32_t* dst) const { std::memcpy(dst, items + offsets[u], size(u) * sizeof(uint32_t)); }
};
// groups of BLOCKS whose members are the blocks' two sub-block training vectors (ids 2b, 2b+1): the endpoint side
struct csr_block_pair_groups {
    const uint32_t* offsets; const uint32_t* blocks;
    size_t size(uint32_t u) const { return 2 * (size_t)(offsets[u + 1] - offsets[u]); }
    void copy(uint32_t u, uint32_t* dst) const {
        for (uint32_t j = offsets[u]; j < offsets[u + 1]; j++) { *dst++ = blocks[j] * 2; *dst++ = blocks[j] * 2 + 1; }
    }
};

class device_tsvq {
    // a node's children live over the node's span of the NEXT member buffer, cyclic (include/basisu_hip.h: BU_TSVQ_BUFFERS)
    static uint32_t child_buf(uint32_t buf) { return (buf + 1u) % BU_TSVQ_BUFFERS; }
public:
    struct stats { uint32_t rounds = 0, splits_computed = 0, splits_used = 0; double t_create = 0, t_device = 0, t_replay = 0, t_expand = 0; };

    // rows: n distinct vectors of `dim` floats, ascending; groups[u]: original training-vector indices of unique vector u.
    template <class Groups>
    static bool hierarchical_codebook(bu_hip_context* ctx, uint32_t dim, const std::vector<float>& rows, const std::vector<uint64_t>& weights,
                                      const Groups& groups, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                                      std::vector<std::vector<uint32_t>>& codebook, std::vector<std::vector<uint32_t>>& parent_codebook, stats* st = nullptr,
                                      std::vector<uint32_t>* parent_of_unique = nullptr, uint32_t* parent_count = nullptr,
                                      std::vector<uint32_t>* leaf_of_unique = nullptr, uint32_t* leaf_count = nullptr, const bu_comm* comm = nullptr,
                                      uint32_t max_threads = 0, uint32_t min_unique_for_threads = kThreadedCodebookMinUnique) {
        bu_tsvq_root root;
        const auto t0 = std::chrono::steady_clock::now();
        bu_tsvq* q = weights.empty() ? nullptr : bu_hip_tsvq_create(ctx, dim, rows.data(), weights.data(), (uint32_t)weights.size(), &root);
        if (st) st->t_create = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return q && build(ctx, q, root, (uint32_t)weights.size(), groups, max_codebook_size, max_parent_codebook_size, codebook, parent_codebook, st, parent_of_unique, parent_count,
                          leaf_of_unique, leaf_count, nullptr, nullptr, comm,
                          codebook_partitions((uint32_t)weights.size(), max_codebook_size, max_threads, min_unique_for_threads));
    }

    // Selector vectors: keys[u] packs the 16 selector values of distinct vector u (value 0 in the top two bits), ascending.
    template <class Groups>
    static bool hierarchical_codebook_packed16(bu_hip_context* ctx, const std::vector<uint32_t>& keys, const std::vector<uint64_t>& weights,
                                               const Groups& groups, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                                               std::vector<std::vector<uint32_t>>& codebook, std::vector<std::vector<uint32_t>>& parent_codebook, stats* st = nullptr,
                                               uint32_t max_threads = 0, uint32_t min_unique_for_threads = kThreadedCodebookMinUnique) {
        bu_tsvq_root root;
        const auto t0 = std::chrono::steady_clock::now();
        bu_tsvq* q = weights.empty() ? nullptr : bu_hip_tsvq_create_packed16(ctx, keys.data(), weights.data(), (uint32_t)weights.size(), &root);
        if (st) st->t_create = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return q && build(ctx, q, root, (uint32_t)weights.size(), groups, max_codebook_size, max_parent_codebook_size, codebook, parent_codebook, st, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr, codebook_partitions((uint32_t)weights.size(), max_codebook_size, max_threads, min_unique_for_threads));
    }

    // The same with the distinct vectors already resident (outputs of bu_hip_k_unique_selector_vectors): nothing is uploaded.
    template <class Groups>
    static bool hierarchical_codebook_packed16_device(bu_hip_context* ctx, const uint32_t* d_keys, const uint64_t* d_weights, uint32_t n_unique, const Groups& groups,
                                                      uint32_t max_codebook_size, uint32_t max_parent_codebook_size, std::vector<std::vector<uint32_t>>& codebook,
                                                      std::vector<std::vector<uint32_t>>& parent_codebook, stats* st = nullptr,
                                                      std::vector<uint32_t>* parent_of_unique = nullptr, uint32_t* parent_count = nullptr,
                                                      std::vector<uint32_t>* leaf_of_unique = nullptr, uint32_t* leaf_count = nullptr,
                                                      uint32_t* d_leaf_of_unique = nullptr, uint32_t* d_parent_of_unique = nullptr, const bu_comm* comm = nullptr,
                                                      uint32_t max_threads = 0, uint32_t min_unique_for_threads = kThreadedCodebookMinUnique) {
        bu_tsvq_root root;
        const auto t0 = std::chrono::steady_clock::now();
        bu_tsvq* q = n_unique ? bu_hip_tsvq_create_packed16_device(ctx, d_keys, d_weights, n_unique, &root) : nullptr;
        if (st) st->t_create = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return q && build(ctx, q, root, n_unique, groups, max_codebook_size, max_parent_codebook_size, codebook, parent_codebook, st, parent_of_unique, parent_count,
                          leaf_of_unique, leaf_count, d_leaf_of_unique, d_parent_of_unique, comm,
                          codebook_partitions(n_unique, max_codebook_size, max_threads, min_unique_for_threads));
    }

    // The endpoint side with the distinct vectors resident as the de-duplication left them (48-bit keys + group offsets, bu_hip_k_unique_endpoint_vectors): the rows are
    // made on the device, and the result stays there -- leaf, parent and first list position of every distinct vector, the block count of every leaf.
    static bool hierarchical_codebook_endpoint_device(bu_hip_context* ctx, const uint64_t* d_unique_keys, const uint32_t* d_group_offsets, uint32_t n_unique,
                                                      uint32_t max_codebook_size, uint32_t max_parent_codebook_size, uint32_t* leaf_count, uint32_t* parent_count,
                                                      uint32_t* d_leaf_of_unique, uint32_t* d_parent_of_unique, uint32_t* d_first_pos, uint32_t* d_leaf_sizes, stats* st = nullptr,
                                                      const bu_comm* comm = nullptr, uint32_t max_threads = 0, uint32_t min_unique_for_threads = kThreadedCodebookMinUnique) {
        bu_tsvq_root root;
        const auto t0 = std::chrono::steady_clock::now();
        bu_tsvq* q = n_unique ? bu_hip_tsvq_create_endpoint_device(ctx, d_unique_keys, d_group_offsets, n_unique, &root) : nullptr;
        if (st) st->t_create = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<std::vector<uint32_t>> unused_a, unused_b;
        const csr_groups no_groups{nullptr, nullptr};
        finish_extra fx{d_group_offsets, d_first_pos, d_leaf_sizes};
        return q && build(ctx, q, root, n_unique, no_groups, max_codebook_size, max_parent_codebook_size, unused_a, unused_b, st, nullptr, parent_count, nullptr, leaf_count,
                          d_leaf_of_unique, d_parent_of_unique, comm, codebook_partitions(n_unique, max_codebook_size, max_threads, min_unique_for_threads), &fx);
    }

    // parent lists from a parent-of-unique map (what build() would have produced): vectors ascending inside a parent, members group by group
    template <class Groups>
    static void expand_parents(const std::vector<uint32_t>& parent_of_unique, uint32_t parents, const Groups& groups, std::vector<std::vector<uint32_t>>& lists) {
        const uint32_t n = (uint32_t)parent_of_unique.size();
        std::vector<uint32_t> ofs(parents + 1, 0), sorted(n);
        for (uint32_t u = 0; u < n; u++) ofs[parent_of_unique[u] + 1]++;
        for (uint32_t c = 0; c < parents; c++) ofs[c + 1] += ofs[c];
        { std::vector<uint32_t> pos(ofs.begin(), ofs.end() - 1); for (uint32_t u = 0; u < n; u++) sorted[pos[parent_of_unique[u]]++] = u; }
        lists.clear(); lists.resize(parents);
        for (uint32_t c = 0; c < parents; c++) {
            size_t total = 0;
            for (uint32_t i = ofs[c]; i < ofs[c + 1]; i++) total += groups.size(sorted[i]);
            lists[c].resize(total);
            uint32_t* dst = lists[c].data();
            for (uint32_t i = ofs[c]; i < ofs[c + 1]; i++) { groups.copy(sorted[i], dst); dst += groups.size(sorted[i]); }
        }
    }

private:
    struct finish_extra { const uint32_t* d_group_offsets; uint32_t* d_first_pos; uint32_t* d_leaf_sizes; };
    // Debug aid (BU_TSVQ_VERIFY=1): a split is a pure function of its node, so re-running every node of a batch on its own must
    // reproduce the batched result bit for bit, including the children's member lists. Reports the first differences to stderr.
    static void verify_batch(bu_hip_context* ctx, bu_tsvq* q, const std::vector<bu_tsvq_node>& batch, const bu_tsvq_split* batched, uint32_t round) {
        std::vector<uint32_t> lists0, lists1;
        for (size_t i = 0; i < batch.size(); i++) {
            const bu_tsvq_node& b = batch[i];
            lists0.resize(b.count); lists1.resize(b.count);
            if (!bu_hip_tsvq_read_members(ctx, q, child_buf(b.buf), b.start, b.count, lists0.data())) return;
            for (int rep = 0; rep < 2; rep++) {
                bu_tsvq_split s; std::memset(&s, 0, sizeof(s));
                if (!bu_hip_tsvq_split(ctx, q, &b, 1, &s)) return;
                if (!bu_hip_tsvq_read_members(ctx, q, child_buf(b.buf), b.start, b.count, lists1.data())) return;
                bu_tsvq_split ref = batched[i]; ref.pad = 0; s.pad = 0;
                const bool same_out = s.ok == ref.ok && (!s.ok || std::memcmp(&s, &ref, sizeof(s)) == 0);
                const bool same_list = !s.ok || lists0 == lists1;
                if (!same_out || !same_list) {
                    size_t first = 0; while (first < b.count && lists0[first] == lists1[first]) first++;
                    std::fprintf(stderr, "[tsvq verify] round %u batch %zu/%zu node{buf %u start %u count %u} rep %d: out %s lists %s | batched ok %u l %u r %u lvar %.9g rvar %.9g | serial ok %u l %u r %u lvar %.9g rvar %.9g | first list diff at %zu\n",
                                 round, i, batch.size(), b.buf, b.start, b.count, rep, same_out ? "same" : "DIFF", same_list ? "same" : "DIFF", batched[i].ok, batched[i].l_count,
                                 batched[i].r_count, batched[i].l_var, batched[i].r_var, s.ok, s.l_count, s.r_count, s.l_var, s.r_var, first);
                }
            }
        }
    }

    // One tree of the build: the reference's node table + variance queue (enc.h:1616-1660), with the splits it has been handed so far.
    struct node {
        float var; uint64_t weight; float origin[16];
        int32_t left = -1, right = -1; int codebook_index = -1;
        uint32_t buf, start, count;
        int32_t cached = -1; // index into the split cache
    };
    struct tree {
        std::vector<node> nodes;
        variance_heap heap;
        // the queued nodes whose split has not been asked for yet, (variance, node) in creation order: what pending_nodes() offers the next round. (A scan of the whole
        // queue per round found the same set; with 16,128 leaves that was ~1.5 ms of an 8192^2 step.)
        std::vector<std::pair<float, uint32_t>> uncached;
        uint32_t leaves = 1, next_codebook_index = 0, max_leaves = 0;
        void reset(const bu_tsvq_root& root, uint32_t buf, uint32_t start, uint32_t count, uint32_t max_leaves_) {
            nodes.clear(); nodes.reserve((size_t)std::min<uint64_t>(max_leaves_, count) * 2 + 1);
            node r; r.var = root.var; r.weight = root.weight; std::memcpy(r.origin, root.origin, sizeof(r.origin));
            r.buf = buf; r.start = start; r.count = count;
            nodes.push_back(r);
            heap.reset(0, r.var);
            uncached.clear();
            if (count > 1) uncached.emplace_back(r.var, 0u);
            leaves = 1; next_codebook_index = 0; max_leaves = max_leaves_;
        }
        // generate()'s loop (enc.h:1636-1655) while the splits it needs are known; true = it stopped at a node whose split is not known yet
        // (kid[2 c + side]: the cache index of the split of that child of cache entry c, where a deep round has computed it already; -1 otherwise)
        bool replay(const std::vector<bu_tsvq_split>& cache, const std::vector<int32_t>& kid, uint32_t& splits_used) {
            while (heap.size() && leaves < max_leaves) {
                const uint32_t ni = heap.top_index();
                if (nodes[ni].count > 1 && nodes[ni].cached < 0) return true;
                heap.pop();
                if (nodes[ni].count <= 1) continue;
                const size_t ci = (size_t)nodes[ni].cached;
                const bu_tsvq_split& s = cache[ci];
                if (!s.ok) continue; // prep_split / refine_split returned false: the node stays a leaf
                splits_used++;
                const uint32_t li = (uint32_t)nodes.size(), ri = li + 1;
                nodes[ni].left = (int32_t)li; nodes[ni].right = (int32_t)ri;
                nodes[ni].codebook_index = (int)next_codebook_index++;
                node l, r;
                l.var = s.l_var; l.weight = s.l_weight; std::memcpy(l.origin, s.l_centroid, sizeof(l.origin));
                l.buf = child_buf(nodes[ni].buf); l.start = nodes[ni].start; l.count = s.l_count;
                r.var = s.r_var; r.weight = s.r_weight; std::memcpy(r.origin, s.r_centroid, sizeof(r.origin));
                r.buf = child_buf(nodes[ni].buf); r.start = nodes[ni].start + s.l_count; r.count = s.r_count;
                // enc.h:1766-1792: a child with var <= 0 but differing members gets a tiny variance; rows are distinct, so any
                // child with more than one member qualifies
                if (l.var <= 0.0f && l.count > 1) l.var = 1e-4f;
                if (r.var <= 0.0f && r.count > 1) r.var = 1e-4f;
                l.cached = kid[2 * ci]; r.cached = kid[2 * ci + 1];
                nodes.push_back(l); nodes.push_back(r);
                if (l.var > 0.0f && l.count > 1) { heap.push(li, l.var); if (l.cached < 0) uncached.emplace_back(l.var, li); }
                if (r.var > 0.0f && r.count > 1) { heap.push(ri, r.var); if (r.cached < 0) uncached.emplace_back(r.var, ri); }
                leaves++;
            }
            return false;
        }
        // every queued node whose split is unknown, largest variance first, at most as many as there are splits left to do.
        // The queue's top is among them (it is the maximum) -- otherwise the replay could not advance.
        void pending_nodes(std::vector<std::pair<float, uint32_t>>& pending) {
            // (a node with an unknown split is never popped -- the replay stops at it -- so everything created and not yet handed to a round is still queued)
            size_t keep = 0;
            for (const auto& u : uncached) if (nodes[u.second].cached < 0) uncached[keep++] = u;
            uncached.resize(keep);
            pending = uncached;
            const size_t want = std::min<size_t>(pending.size(), (size_t)(max_leaves - leaves));
            if (want < pending.size()) {
                std::nth_element(pending.begin(), pending.begin() + want, pending.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
                pending.resize(want);
            }
        }
        // A bound for the speculation of a deep round: with R splits left to do, a node whose variance is below the R-th largest variance in the queue will not be
        // popped before the leaf budget is spent (popping is by descending variance; new nodes only push it further down), so its descendants are not worth a
        // launch slot. 0 while the queue holds no more than R nodes. (Only a bound on wasted work: whatever the replay does need and does not find, the next round computes.)
        float speculation_floor(std::vector<float>& scratch) const {
            const size_t R = (size_t)(max_leaves - leaves);
            if (!R || heap.size() <= R) return 0.0f;
            scratch.resize(heap.size());
            for (uint32_t i = 0; i < heap.size(); i++) scratch[i] = heap.entry_priority(i + 1);
            std::nth_element(scratch.begin(), scratch.begin() + (R - 1), scratch.end(), [](float a, float b) { return a > b; });
            return scratch[R - 1] > 0.0f ? scratch[R - 1] : 0.0f;
        }
        // retrieve(max_clusters) (enc.h:1598-1628): cut the tree after its first max_clusters-1 splits, depth first, left before right;
        // cut(ni) is called for every cut node in that order
        template <class F> void cuts(uint32_t max_clusters, F&& cut) const {
            std::vector<uint32_t> stack;
            uint32_t ni = 0;
            for (;;) {
                const node& cur = nodes[ni];
                if (cur.left < 0 || (2 + cur.codebook_index) > (int)max_clusters) {
                    cut(ni);
                    if (stack.empty()) break;
                    ni = stack.back(); stack.pop_back();
                    continue;
                }
                stack.push_back((uint32_t)cur.right);
                ni = (uint32_t)cur.left;
            }
        }
    };

    // Splits every tree of `trees` to its leaf budget: each device round takes the pending nodes of ALL trees (they are independent), so
    // the T sub-trees of the partitioned build fill a round T times as well as one tree does.
    // Deep rounds (bu_hip_tsvq_split_deep, bu_hip_tuning::tsvq_deep_levels): with the batch, the device also splits the descendants of its one-workgroup nodes
    // `deep_levels` generations down (bounded per tree by speculation_floor()), so the replay below runs through several generations before it needs the device again.
    static bool run_trees(bu_hip_context* ctx, bu_tsvq* q, std::vector<tree*>& trees, std::vector<bu_tsvq_split>& cache, std::vector<int32_t>& kid, stats& local, const bu_comm* comm,
                          bool dbg_serial, bool dbg_verify, uint32_t deep_levels) {
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        std::vector<bu_tsvq_node> batch;
        std::vector<std::pair<uint32_t, uint32_t>> batch_nodes;   // (tree, node)
        std::vector<std::pair<float, uint32_t>> pending;
        std::vector<float> floor_scratch;
        std::vector<bu_tsvq_split> deep;
        std::vector<int32_t> gen_idx[2];
        const bool sharing = comm && comm->world > 1;
        if (sharing || dbg_serial || dbg_verify) deep_levels = 0;   // (verify re-runs a node and compares the children's lists: they must not have been split further)
        for (;;) {
            batch.clear(); batch_nodes.clear();
            uint64_t splits_left = 0;   // over the trees of this round
            for (uint32_t t = 0; t < trees.size(); t++) {
                tree& tr = *trees[t];
                if (!tr.replay(cache, kid, local.splits_used)) continue;
                splits_left += tr.max_leaves - tr.leaves;
                tr.pending_nodes(pending);
                uint32_t floor_bits = 0;
                if (deep_levels) { const float f = tr.speculation_floor(floor_scratch); std::memcpy(&floor_bits, &f, 4); }
                for (const auto& p : pending) {
                    const node& nd = tr.nodes[p.second];
                    bu_tsvq_node b; std::memset(&b, 0, sizeof(b));
                    b.pad = floor_bits;
                    b.buf = nd.buf; b.start = nd.start; b.count = nd.count; b.weight = nd.weight; std::memcpy(b.origin, nd.origin, sizeof(b.origin));
                    batch.push_back(b); batch_nodes.emplace_back(t, p.second);
                }
            }
            if (batch.empty()) return true;
            const size_t base = cache.size();
            cache.resize(base + batch.size());
            kid.resize(2 * cache.size(), -1);
            if (sharing && batch.size() > 1) {
                // Multi-GPU: the nodes of a round are independent, so every rank splits a share of them (largest first onto the least loaded
                // rank: the same assignment on every rank) and the child member lists + result records are merged by ONE exact sum all-reduce of
                // a staging buffer in which everybody else's entries are zero. Afterwards every rank holds every result, bit for bit.
                const auto td = now();
                std::vector<uint32_t> order(batch.size());
                for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return batch[a].count > batch[b].count; });
                std::vector<uint64_t> load(comm->world, 0);
                std::vector<uint8_t> mine(batch.size(), 0);
                std::vector<uint32_t> my_idx;
                for (uint32_t i : order) {
                    uint32_t r = 0;
                    for (uint32_t k = 1; k < comm->world; k++) if (load[k] < load[r]) r = k;
                    load[r] += batch[i].count;
                    if (r == comm->rank) { mine[i] = 1; my_idx.push_back(i); }
                }
                std::sort(my_idx.begin(), my_idx.end());
                std::vector<bu_tsvq_node> my_nodes;
                for (uint32_t i : my_idx) my_nodes.push_back(batch[i]);
                std::vector<bu_tsvq_split> my_out(my_nodes.size());
                // A rank whose own share fails must still enter the collective -- the others are on their way into it -- and everybody must learn of the failure: one more
                // (zero-member) entry behind the batch is everybody's, and its record carries an error word that the sum turns into "how many ranks failed".
                const bool local_ok = my_nodes.empty() || bu_hip_tsvq_split(ctx, q, my_nodes.data(), (uint32_t)my_nodes.size(), my_out.data()) != 0;
                const size_t nb = batch.size();
                std::vector<bu_tsvq_split> rec(nb + 1);
                std::memset(rec.data(), 0, rec.size() * sizeof(bu_tsvq_split));
                if (local_ok) for (size_t j = 0; j < my_idx.size(); j++) { rec[my_idx[j]] = my_out[j]; rec[my_idx[j]].pad = 0; }
                rec[nb].pad = local_ok ? 0u : 1u;
                std::vector<bu_tsvq_node> xnodes(batch);
                { bu_tsvq_node z; std::memset(&z, 0, sizeof(z)); xnodes.push_back(z); }
                mine.push_back(1);
                void* d_staging = nullptr; uint64_t n_u64 = 0;
                if (!bu_hip_tsvq_exchange_pack(ctx, q, xnodes.data(), mine.data(), rec.data(), (uint32_t)xnodes.size(), &d_staging, &n_u64)) return false;   // (the device itself is gone: nothing to enter the collective with)
                if (!comm->all_reduce_u64(comm->user, d_staging, n_u64)) return false;
                if (!bu_hip_tsvq_exchange_unpack(ctx, q, xnodes.data(), mine.data(), rec.data(), (uint32_t)xnodes.size())) return false;
                if (rec[nb].pad != 0) return false;   // some rank's share failed: every rank fails this round together
                std::memcpy(cache.data() + base, rec.data(), nb * sizeof(bu_tsvq_split));
                local.t_device += secs(td, now());
            } else if (dbg_serial) { // debug: one node per launch
                for (size_t i = 0; i < batch.size(); i++)
                    if (!bu_hip_tsvq_split(ctx, q, &batch[i], 1, cache.data() + base + i)) return false;
            } else if (deep_levels) {
                const auto td = now();
                const size_t n = batch.size();
                // As deep as the leaf budget can still use: every generation doubles the nodes, and in the rounds where the device is full (hundreds of nodes) a split
                // nobody pops is paid for in device time -- two generations for every node of the batch measured SLOWER than none (4096^2 q128: 17.9 against 17.1 ms
                // per step). While the batch with all its descendants fits the splits that are left, nothing attempted is likely to be wasted.
                uint32_t levels = deep_levels;
                while (levels && (uint64_t)n * ((2u << levels) - 1u) > splits_left) levels--;
                deep.resize(n * ((2u << levels) - 2u) + 1);
                if (!bu_hip_tsvq_split_deep(ctx, q, batch.data(), (uint32_t)n, cache.data() + base, levels, deep.data())) return false;
                local.t_device += secs(td, now());
                // the descendants that were attempted join the cache behind the batch; kid[] links every entry to its children's entries
                gen_idx[0].resize(n);
                for (size_t i = 0; i < n; i++) gen_idx[0][i] = (int32_t)(base + i);
                for (uint32_t g = 1; g <= levels; g++) {
                    const std::vector<int32_t>& up = gen_idx[(g - 1) & 1];
                    std::vector<int32_t>& cur = gen_idx[g & 1];
                    const size_t w = (size_t)1 << g;
                    const bu_tsvq_split* src = deep.data() + n * (w - 2);
                    cur.assign(n * w, -1);
                    for (size_t k = 0; k < n * w; k++) {
                        if (up[k >> 1] < 0 || src[k].ok == 3) continue;
                        cur[k] = (int32_t)cache.size();
                        kid[2 * (size_t)up[k >> 1] + (k & 1)] = cur[k];
                        cache.push_back(src[k]);
                        kid.push_back(-1); kid.push_back(-1);
                        local.splits_computed++;
                    }
                }
            } else {
                const auto td = now();
                if (!bu_hip_tsvq_split(ctx, q, batch.data(), (uint32_t)batch.size(), cache.data() + base)) return false;
                local.t_device += secs(td, now());
            }
            if (dbg_verify) verify_batch(ctx, q, batch, cache.data() + base, local.rounds);
            for (size_t i = 0; i < batch.size(); i++) trees[batch_nodes[i].first]->nodes[batch_nodes[i].second].cached = (int32_t)(base + i);
            local.rounds++; local.splits_computed += (uint32_t)batch.size();
        }
    }

    // The end of the one-tree-per-rank build: every rank contributes the trees it built -- every member buffer over each tree's span, and the tree's node table --
    // to one staging buffer in which everybody else's entries are zero; ONE exact u64 sum all-reduce (bu_comm) makes it whole everywhere. The buffer is the one
    // the per-round exchange uses (bu_hip_tsvq_exchange_pack / _unpack): per tree one pseudo-node per member buffer over its span, {buf b - 1} for the list in
    // buffer b ("a node's children live in the next buffer"); the serialised node tables ride in the record slots of zero-member pseudo-nodes
    // behind them (a table's region is a whole number of records, so a record has one owner).
    // local_ok = false: this rank's own trees failed -- it still takes part (the others are waiting in the collective) and everybody returns false.
    static bool exchange_trees(bu_hip_context* ctx, bu_tsvq* q, const bu_comm* comm, std::vector<tree>& subs, const std::vector<bu_tsvq_node>& spans, const std::vector<uint32_t>& owner,
                               bool local_ok = true) {
        static_assert(std::is_trivially_copyable<node>::value, "node tables are exchanged as bytes");
        const size_t rec = sizeof(bu_tsvq_split), T = subs.size();
        struct table_header { uint32_t n_nodes, leaves, next_codebook_index, pad; };
        std::vector<size_t> rec_first(T + 1);
        size_t n_rec = 0;
        for (size_t t = 0; t < T; t++) {
            rec_first[t] = n_rec;
            const size_t cap_nodes = 2 * (size_t)std::min<uint64_t>(subs[t].max_leaves, spans[t].count) + 1;   // a tree of L leaves has 2 L - 1 nodes
            n_rec += (sizeof(table_header) + cap_nodes * sizeof(node) + rec - 1) / rec;
        }
        rec_first[T] = n_rec;
        constexpr size_t B = BU_TSVQ_BUFFERS;
        std::vector<bu_tsvq_node> nodes(B * T + n_rec + 1);   // + the error word's entry (everybody's)
        std::memset(nodes.data(), 0, nodes.size() * sizeof(bu_tsvq_node));
        std::vector<uint8_t> mine(nodes.size(), 0);
        std::vector<bu_tsvq_split> recs(nodes.size());
        std::memset(recs.data(), 0, recs.size() * sizeof(bu_tsvq_split));
        char* blob = reinterpret_cast<char*>(recs.data() + B * T);
        for (size_t t = 0; t < T; t++) {
            const bool my = owner[t] == comm->rank;
            for (size_t b = 0; b < B; b++) {   // pseudo-node b: "children" = the tree's span of buffer (b + 1) % B
                nodes[B * t + b].buf = (uint32_t)b; nodes[B * t + b].start = spans[t].start; nodes[B * t + b].count = spans[t].count;
                mine[B * t + b] = my ? 1 : 0;
            }
            for (size_t r = rec_first[t]; r < rec_first[t + 1]; r++) mine[B * T + r] = my ? 1 : 0;
            if (!my || !local_ok) continue;
            const tree& tr = subs[t];
            if (sizeof(table_header) + tr.nodes.size() * sizeof(node) > (rec_first[t + 1] - rec_first[t]) * rec) return false;   // cannot happen: 2 L - 1 nodes
            table_header h{(uint32_t)tr.nodes.size(), tr.leaves, tr.next_codebook_index, 0};
            char* at = blob + rec_first[t] * rec;
            std::memcpy(at, &h, sizeof(h));
            std::memcpy(at + sizeof(h), tr.nodes.data(), tr.nodes.size() * sizeof(node));
        }
        mine[B * T + n_rec] = 1; recs[B * T + n_rec].pad = local_ok ? 0u : 1u;   // the last entry
        void* d_staging = nullptr; uint64_t n_u64 = 0;
        if (!bu_hip_tsvq_exchange_pack(ctx, q, nodes.data(), mine.data(), recs.data(), (uint32_t)nodes.size(), &d_staging, &n_u64)) return false;
        if (!comm->stream_ordered && !bu_hip_sync(ctx)) return false;
        if (!comm->all_reduce_u64(comm->user, d_staging, n_u64)) return false;
        if (!bu_hip_tsvq_exchange_unpack(ctx, q, nodes.data(), mine.data(), recs.data(), (uint32_t)nodes.size())) return false;
        if (recs[B * T + n_rec].pad != 0) return false;   // a rank's trees failed: all ranks fail together
        blob = reinterpret_cast<char*>(recs.data() + B * T);
        for (size_t t = 0; t < T; t++) {
            if (owner[t] == comm->rank) continue;
            table_header h;
            const char* at = blob + rec_first[t] * rec;
            std::memcpy(&h, at, sizeof(h));
            if (!h.n_nodes || sizeof(h) + (size_t)h.n_nodes * sizeof(node) > (rec_first[t + 1] - rec_first[t]) * rec) return false;   // the owner never delivered
            tree& tr = subs[t];
            tr.nodes.resize(h.n_nodes);
            std::memcpy(tr.nodes.data(), at + sizeof(h), (size_t)h.n_nodes * sizeof(node));
            tr.leaves = h.leaves; tr.next_codebook_index = h.next_codebook_index;
        }
        return true;
    }

    template <class Groups>
    static bool build(bu_hip_context* ctx, bu_tsvq* q, const bu_tsvq_root& root, uint32_t n, const Groups& groups,
                      uint32_t max_codebook_size, uint32_t max_parent_codebook_size, std::vector<std::vector<uint32_t>>& codebook,
                      std::vector<std::vector<uint32_t>>& parent_codebook, stats* st, std::vector<uint32_t>* parent_of_unique = nullptr,
                      uint32_t* parent_count = nullptr, std::vector<uint32_t>* leaf_of_unique = nullptr, uint32_t* leaf_count = nullptr,
                      uint32_t* d_leaf_of_unique = nullptr, uint32_t* d_parent_of_unique = nullptr, const bu_comm* comm = nullptr,
                      uint32_t partitions = 1, const finish_extra* extra = nullptr) {
        if (parent_of_unique) parent_of_unique->clear();
        if (parent_count) *parent_count = 0;
        if (leaf_of_unique) leaf_of_unique->clear();
        if (leaf_count) *leaf_count = 0;
        struct guard { bu_hip_context* c; bu_tsvq* q; ~guard() { bu_hip_tsvq_destroy(c, q); } } g{ctx, q};
        // (the path choices are the context's tuning -- rank-agreed state under a communicator, not each process's environment; BU_TSVQ_VERIFY is a local debug aid)
        bu_hip_tuning tune; std::memset(&tune, 0, sizeof(tune));
        bu_hip_get_tuning(ctx, &tune, (uint32_t)sizeof(tune));
        const bool dbg_serial = (tune.debug & 2u) != 0, dbg_verify = std::getenv("BU_TSVQ_VERIFY") != nullptr;
        const uint32_t deep_levels = tune.tsvq_deep_levels;

        stats local;
        if (st) local.t_create = st->t_create;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        const auto t_loop0 = now();
        std::vector<bu_tsvq_split> cache;
        std::vector<int32_t> kid;

        // ---- the tree, or -- partitions = T > 1, generate_hierarchical_codebook_threaded_internal (enc.h:2086-2215) -- a T-leaf tree first and then
        //      one independent tree per leaf over that leaf's members: own root record (prepare_root over the span, bu_hip_tsvq_roots), own queue,
        //      ceil(K / T) leaves (all of the span's vectors when the codebook is not limited, enc.h:2150) and ceil(P / T) parents each
        const uint32_t T = partitions > 1 ? partitions : 1;
        tree top;
        top.reset(root, 0, 0, n, T > 1 ? T : max_codebook_size);
        std::vector<tree> subs;
        std::vector<tree*> active{&top};
        if (!run_trees(ctx, q, active, cache, kid, local, comm, dbg_serial, dbg_verify, deep_levels)) return false;
        std::vector<const tree*> final_trees{&top};
        std::vector<uint32_t> final_parents{max_parent_codebook_size};
        if (T > 1) {
            std::vector<bu_tsvq_node> spans;
            for (const node& nd : top.nodes)
                if (nd.left < 0) { bu_tsvq_node b; std::memset(&b, 0, sizeof(b)); b.buf = nd.buf; b.start = nd.start; b.count = nd.count; spans.push_back(b); }
            if (spans.size() >= T) {   // enc.h:2122: fewer leaves than threads -> the T-leaf tree is the result
                std::vector<bu_tsvq_root> roots(spans.size());
                const auto td = now();
                if (!bu_hip_tsvq_roots(ctx, q, spans.data(), (uint32_t)spans.size(), roots.data())) return false;
                local.t_device += secs(td, now());
                const bool limit = n > max_codebook_size;   // limit_clusterizers, enc.h:2305-2307
                subs.resize(spans.size());
                active.clear(); final_trees.clear(); final_parents.clear();
                for (size_t t = 0; t < spans.size(); t++) {
                    subs[t].reset(roots[t], spans[t].buf, spans[t].start, spans[t].count, limit ? (max_codebook_size + T - 1) / T : spans[t].count);
                    active.push_back(&subs[t]); final_trees.push_back(&subs[t]);
                    final_parents.push_back(max_parent_codebook_size ? (max_parent_codebook_size + T - 1) / T : 0);
                }
                if (comm && comm->world > 1 && spans.size() >= 2 && !dbg_serial) {
                    // Multi-GPU: ONE TREE PER RANK. The T trees are independent (own training sets, own queues: enc.h:2138-2215, where they are T host threads), so
                    // instead of sharing out every round's nodes with an all-reduce per round, every rank builds whole trees -- largest span first onto the least
                    // loaded rank, the same assignment everywhere -- without talking to anybody, and ONE exchange at the end hands every rank every tree: the member
                    // buffers' spans and the node tables. A tree is a pure function of its span and root record, so it does not matter who built it.
                    std::vector<uint32_t> order(spans.size()), owner(spans.size(), 0);
                    for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
                    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return spans[a].count > spans[b].count; });
                    std::vector<uint64_t> load(comm->world, 0);
                    for (uint32_t t : order) {
                        uint32_t r = 0;
                        for (uint32_t k = 1; k < comm->world; k++) if (load[k] < load[r]) r = k;
                        load[r] += spans[t].count; owner[t] = r;
                    }
                    std::vector<tree*> mine;
                    for (size_t t = 0; t < subs.size(); t++) if (owner[t] == comm->rank) mine.push_back(&subs[t]);
                    const bool mine_ok = mine.empty() || run_trees(ctx, q, mine, cache, kid, local, nullptr, dbg_serial, dbg_verify, deep_levels);
                    const auto tx = now();
                    if (!exchange_trees(ctx, q, comm, subs, spans, owner, mine_ok)) return false;
                    local.t_device += secs(tx, now());
                } else if (!run_trees(ctx, q, active, cache, kid, local, comm, dbg_serial, dbg_verify, deep_levels)) return false;
            }
        }
        const auto t_loop1 = now();
        local.t_replay = secs(t_loop0, t_loop1) - local.t_device;
        struct fin { stats* st; stats* local; std::chrono::steady_clock::time_point t; ~fin() { local->t_expand = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); if (st) *st = *local; } } fin_{st, &local, t_loop1};

        // ---- leaves in node order (enc.h:1573-1584), tree after tree (enc.h:2196-2210); parents = retrieve(max_clusters) of every tree
        //      (enc.h:1598-1628), likewise. Leaf segments are intact in their buffers and ascending.
        std::vector<bu_tsvq_span> spans;      // value: the leaf's parent (cut) index
        uint32_t cuts = 0;
        {
            std::vector<int32_t> leaf_id;
            std::vector<uint32_t> sub;
            for (size_t ti = 0; ti < final_trees.size(); ti++) {
                const tree& tr = *final_trees[ti];
                leaf_id.assign(tr.nodes.size(), -1);
                for (size_t ni = 0; ni < tr.nodes.size(); ni++) {
                    if (tr.nodes[ni].left >= 0) continue;
                    leaf_id[ni] = (int32_t)spans.size();
                    spans.push_back(bu_tsvq_span{tr.nodes[ni].buf, tr.nodes[ni].start, tr.nodes[ni].count, 0});
                }
                if (!final_parents[ti]) continue;
                tr.cuts(final_parents[ti], [&](uint32_t ni) {
                    sub.assign(1, ni);
                    while (!sub.empty()) {   // all leaves below this cut node
                        const uint32_t x = sub.back(); sub.pop_back();
                        if (tr.nodes[x].left < 0) spans[(size_t)leaf_id[x]].value = cuts;
                        else { sub.push_back((uint32_t)tr.nodes[x].left); sub.push_back((uint32_t)tr.nodes[x].right); }
                    }
                    cuts++;
                });
            }
        }
        if (leaf_count) *leaf_count = (uint32_t)spans.size();

        // ---- resident form: the leaf (and parent) of every distinct vector is written by the device from the leaves' spans of the member
        //      buffers; no member list leaves HBM.
        if (d_leaf_of_unique) {
            const bool parents = max_parent_codebook_size && d_parent_of_unique;
            if (parents && parent_count) *parent_count = cuts;
            // one pass: span i is leaf i, its value the parent cut it lies under (+ list positions and leaf sizes where the vectors' groups are given)
            if (!bu_hip_tsvq_finish_spans(ctx, q, spans.data(), (uint32_t)spans.size(), d_leaf_of_unique, parents ? d_parent_of_unique : nullptr,
                                          extra ? extra->d_group_offsets : nullptr, extra ? extra->d_first_pos : nullptr, extra ? extra->d_leaf_sizes : nullptr))
                return false;
            codebook.clear(); parent_codebook.clear();
            return true;
        }

        std::vector<uint32_t> perm[BU_TSVQ_BUFFERS];
        {
            bool used[BU_TSVQ_BUFFERS] = {};
            for (const bu_tsvq_span& sp : spans) { if (sp.buf >= BU_TSVQ_BUFFERS) return false; used[sp.buf] = true; }
            for (uint32_t b = 0; b < BU_TSVQ_BUFFERS; b++) {   // (a leaf's list lives in the buffer its depth put it in)
                if (!used[b]) continue;
                perm[b].resize(n);
                if (!bu_hip_tsvq_read_members(ctx, q, b, 0, n, perm[b].data())) return false;
            }
        }
        struct span { const uint32_t* p; uint32_t n; };
        std::vector<span> leaf_members;
        for (const bu_tsvq_span& s : spans) leaf_members.push_back(span{perm[s.buf].data() + s.start, s.count});
        // distinct vectors -> the training vectors behind them, list by list
        auto expand_one = [&](const uint32_t* us, size_t count, std::vector<uint32_t>& out) {
            size_t total = 0;
            for (size_t i = 0; i < count; i++) total += groups.size(us[i]);
            out.resize(total);
            uint32_t* dst = out.data();
            for (size_t i = 0; i < count; i++) {
                groups.copy(us[i], dst);
                dst += groups.size(us[i]);
            }
        };
        codebook.clear();
        if (leaf_of_unique) {  // the caller keeps "which cluster does every distinct vector belong to" and builds lists only when asked
            leaf_of_unique->resize(n);
            for (size_t l = 0; l < leaf_members.size(); l++)
                for (uint32_t i = 0; i < leaf_members[l].n; i++) (*leaf_of_unique)[leaf_members[l].p[i]] = (uint32_t)l;
        } else {
            codebook.resize(leaf_members.size());
            std::atomic<size_t> next{0};
            auto work = [&] { for (size_t i; (i = next.fetch_add(1)) < leaf_members.size();) expand_one(leaf_members[i].p, leaf_members[i].n, codebook[i]); };
            std::vector<std::thread> th;
            const unsigned TH = n > 65536 ? host_threads() : 1;
            for (unsigned t = 1; t < TH; t++) th.emplace_back(work);
            work();
            for (auto& x : th) x.join();
        }

        parent_codebook.clear();
        if (max_parent_codebook_size) {
            // Every cut node's member list is the ascending union of its leaves: sweep the vectors in order.
            std::vector<uint32_t> cut_of_vec(n);
            for (size_t l = 0; l < leaf_members.size(); l++)
                for (uint32_t i = 0; i < leaf_members[l].n; i++) cut_of_vec[leaf_members[l].p[i]] = spans[l].value;
            if (parent_of_unique) {  // the caller only wants "which parent does every distinct vector belong to": lists on demand (expand_parents)
                parent_of_unique->swap(cut_of_vec);
                if (parent_count) *parent_count = cuts;
                return true;
            }
            // counting sort of the distinct vectors by cut, ascending inside a cut
            std::vector<uint32_t> cut_ofs(cuts + 1, 0), sorted(n);
            for (uint32_t u = 0; u < n; u++) cut_ofs[cut_of_vec[u] + 1]++;
            for (uint32_t c = 0; c < cuts; c++) cut_ofs[c + 1] += cut_ofs[c];
            { std::vector<uint32_t> pos(cut_ofs.begin(), cut_ofs.end() - 1); for (uint32_t u = 0; u < n; u++) sorted[pos[cut_of_vec[u]]++] = u; }
            parent_codebook.resize(cuts);
            {
                std::atomic<uint32_t> next{0};
                auto work = [&] { for (uint32_t c; (c = next.fetch_add(1)) < cuts;) expand_one(sorted.data() + cut_ofs[c], cut_ofs[c + 1] - cut_ofs[c], parent_codebook[c]); };
                std::vector<std::thread> th;
                const unsigned TH = n > 65536 ? host_threads() : 1;
                for (unsigned t = 1; t < TH; t++) th.emplace_back(work);
                work();
                for (auto& x : th) x.join();
            }
        }
        return true;
    }
};

} // namespace bu
