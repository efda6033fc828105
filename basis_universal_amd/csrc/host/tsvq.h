// tsvq.h -- tree-structured vector quantiser used for the ETC1S endpoint (N = 6) and selector (N = 16) codebooks.
//
// Host side of row a8 of the hot-path table: the reference builds both codebooks with tree_vector_quant<> and
// generate_hierarchical_codebook_threaded() (encoder/basisu_enc.h:1546-2354). The result depends on the exact order and
// precision of every float/double operation (SURVEY hazard H2), so this restatement keeps the reference's operation order:
// running float sums in member order, double distances in difference form, 8 power iterations with the float early-out, and
// the same binary-heap tie behaviour. It is written for clarity around that constraint, not as a translation: vectors are
// kept as flat float rows, the node table owns index ranges, and de-duplication is done by the callers on integer keys.
//
// Compile with -ffp-contract=off and without -ffast-math / -march flags that enable FMA.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace bu {

// Max-heap on float priority with the reference's tie behaviour (enc.h:1455-1544): sift-up moves past parents with
// priority <= new; sift-down prefers the right child only when strictly larger and stops when the moving entry is
// strictly larger than the chosen child.
class variance_heap {
public:
    void reset(uint32_t first_index, float first_priority) { h_.assign(2, entry{0, 0.0f}); h_[1] = entry{first_index, first_priority}; n_ = 1; }
    uint32_t size() const { return n_; }
    uint32_t top_index() const { return h_[1].index; }
    void pop() {
        h_[1] = h_[n_];
        n_--;
        if (!n_) return;
        const entry moving = h_[1];
        uint32_t at = 1, child;
        while ((child = at << 1) <= n_) {
            if (child < n_ && h_[child].priority < h_[child + 1].priority) ++child;
            if (moving.priority > h_[child].priority) break;
            h_[at] = h_[child];
            at = child;
        }
        h_[at] = moving;
    }
    void push(uint32_t index, float priority) {
        n_++;
        if (n_ >= h_.size()) h_.resize(n_ + 1);
        uint32_t k = n_;
        for (;;) {
            const uint32_t parent = k >> 1;
            if (!parent || h_[parent].priority > priority) break;
            h_[k] = h_[parent];
            k = parent;
        }
        h_[k] = entry{index, priority};
    }
    // raw view of the live entries (1..size), for batch scheduling of speculative splits
    uint32_t entry_index(uint32_t i) const { return h_[i].index; }
    float entry_priority(uint32_t i) const { return h_[i].priority; }
private:
    struct entry { uint32_t index; float priority; };
    std::vector<entry> h_;
    uint32_t n_ = 0;
};

template <int N>
class tsvq {
public:
    // Training set: n rows of N floats plus a u64 weight each.
    void set_training(const float* rows, const uint64_t* weights, uint32_t n) {
        rows_.assign(rows, rows + (size_t)n * N);
        weights_.assign(weights, weights + n);
        n_ = n;
    }
    uint32_t training_size() const { return n_; }

    // enc.h:1616-1660
    bool generate(uint32_t max_leaves) {
        if (!n_) return false;
        next_codebook_index_ = 0;
        nodes_.clear();
        nodes_.reserve((size_t)max_leaves * 2 + 1);
        nodes_.push_back(make_root());
        variance_heap heap;
        heap.reset(0, nodes_[0].var);
        uint32_t leaves = 1;
        while (heap.size() && leaves < max_leaves) {
            const uint32_t ni = heap.top_index();
            heap.pop();
            if (nodes_[ni].members.size() > 1 && split(ni, heap)) leaves++;
        }
        return true;
    }

    // Leaves in node order (enc.h:1573-1584)
    void leaves(std::vector<std::vector<uint32_t>>& out) const {
        for (const node& n : nodes_)
            if (n.left < 0) out.push_back(n.members);
    }

    // The first `max_clusters` splits of the tree as a coarser partition (enc.h:1598-1628)
    void top_clusters(uint32_t max_clusters, std::vector<std::vector<uint32_t>>& out) const {
        out.clear();
        std::vector<uint32_t> stack;
        uint32_t ni = 0;
        for (;;) {
            const node& cur = nodes_[ni];
            if (cur.left < 0 || (2 + cur.codebook_index) > (int)max_clusters) {
                out.push_back(cur.members);
                if (stack.empty()) break;
                ni = stack.back();
                stack.pop_back();
                continue;
            }
            stack.push_back((uint32_t)cur.right);
            ni = (uint32_t)cur.left;
        }
    }

private:
    struct node {
        float var = 0.0f;
        uint64_t weight = 0;
        float origin[N];
        int32_t left = -1, right = -1;
        int codebook_index = -1;
        std::vector<uint32_t> members;
    };

    const float* row(uint32_t i) const { return &rows_[(size_t)i * N]; }

    static float dot(const float* a, const float* b) {
        float r = a[0] * b[0];
        for (int i = 1; i < N; i++) r += a[i] * b[i];
        return r;
    }
    static double dist2_d(const float* a, const float* b) {
        double d2 = 0;
        for (int i = 0; i < N; i++) { const double d = (double)a[i] - (double)b[i]; d2 += d * d; }
        return d2;
    }
    static bool rows_equal(const float* a, const float* b) {
        for (int i = 0; i < N; i++) if (a[i] != b[i]) return false;
        return true;
    }

    // enc.h:1708-1735
    node make_root() const {
        node root;
        for (int k = 0; k < N; k++) root.origin[k] = 0.0f;
        root.members.reserve(n_);
        double ttsum = 0.0;
        for (uint32_t i = 0; i < n_; i++) {
            const float* v = row(i);
            const uint64_t w = weights_[i];
            const float wf = static_cast<float>(w);
            root.members.push_back(i);
            for (int k = 0; k < N; k++) root.origin[k] += v[k] * wf;
            root.weight += w;
            ttsum += dot(v, v) * wf;
        }
        root.var = static_cast<float>(ttsum - (dot(root.origin, root.origin) / static_cast<float>(root.weight)));
        const float inv = 1.0f / static_cast<float>(root.weight);
        for (int k = 0; k < N; k++) root.origin[k] *= inv;
        return root;
    }

    // compute_split_axis + compute_pca_from_covar (enc.h:1802-1846, 605-648)
    void split_axis(const node& nd, float axis[N]) const {
        float cov[N][N];
        for (int x = 0; x < N; x++) for (int y = 0; y < N; y++) cov[x][y] = 0.0f;
        for (uint32_t m : nd.members) {
            const float* t = row(m);
            const float wf = static_cast<float>(weights_[m]);
            float v[N], w[N];
            for (int k = 0; k < N; k++) { v[k] = t[k] - nd.origin[k]; w[k] = wf * v[k]; }
            for (int x = 0; x < N; x++)
                for (int y = x; y < N; y++) cov[x][y] = cov[x][y] + v[x] * w[y];
        }
        const float renorm = 1.0f / static_cast<float>(nd.weight);
        for (int x = 0; x < N; x++) for (int y = x; y < N; y++) cov[x][y] *= renorm;
        for (int x = 0; x < N - 1; x++) for (int y = x + 1; y < N; y++) cov[y][x] = cov[x][y];

        float prev[N];
        for (int i = 0; i < N; i++) {
            const float t = (float)(uint32_t)i * (1.0f / (float)(N - 1 > 1 ? N - 1 : 1));
            axis[i] = .75f + (1.25f - .75f) * t;
            prev[i] = axis[i];
        }
        for (int iter = 0; iter < 8; iter++) {
            float trial[N];
            double max_sum = 0;
            for (int i = 0; i < N; i++) {
                double sum = 0;
                for (int j = 0; j < N; j++) sum += cov[i][j] * axis[j];
                trial[i] = static_cast<float>(sum);
                const double a = std::fabs(sum);
                if (a > max_sum) max_sum = a;
            }
            if (max_sum != 0.0f) {
                const float s = static_cast<float>(1.0f / max_sum);
                for (int i = 0; i < N; i++) trial[i] *= s;
            }
            float delta[N];
            for (int i = 0; i < N; i++) delta[i] = prev[i] - trial[i];
            for (int i = 0; i < N; i++) { prev[i] = axis[i]; axis[i] = trial[i]; }
            if (dot(delta, delta) < .0024f) break;
        }
        const float len = std::sqrt(dot(axis, axis));
        if (len != 0.0f) {
            const float s = 1.0f / len;
            for (int i = 0; i < N; i++) axis[i] *= s;
        }
    }

    // prep_split (enc.h:1848-1960): initial left/right centroids
    bool initial_children(const node& nd, float l_out[N], float r_out[N]) const {
        if (nd.members.size() == 2) {
            std::memcpy(l_out, row(nd.members[0]), sizeof(float) * N);
            std::memcpy(r_out, row(nd.members[1]), sizeof(float) * N);
            return true;
        }
        float axis[N], l_sum[N], r_sum[N];
        split_axis(nd, axis);
        for (int k = 0; k < N; k++) { l_sum[k] = 0.0f; r_sum[k] = 0.0f; }
        double l_w = 0.0, r_w = 0.0;
        for (uint32_t m : nd.members) {
            const float wf = (float)weights_[m];
            const float* v = row(m);
            float d[N];
            for (int k = 0; k < N; k++) d[k] = v[k] - nd.origin[k];
            const double t = dot(d, axis);
            if (t >= 0.0f) { for (int k = 0; k < N; k++) r_sum[k] += v[k] * wf; r_w += wf; }
            else           { for (int k = 0; k < N; k++) l_sum[k] += v[k] * wf; l_w += wf; }
        }
        if (l_w > 0.0f && r_w > 0.0f) {
            const float ls = static_cast<float>(1.0f / l_w), rs = static_cast<float>(1.0f / r_w);
            for (int k = 0; k < N; k++) { l_out[k] = l_sum[k] * ls; r_out[k] = r_sum[k] * rs; }
            return true;
        }
        // Degenerate projection: split the member list in half along the widest axis' bounding box (enc.h:1893-1957).
        float lo[N], hi[N];
        for (int k = 0; k < N; k++) { lo[k] = 1e+20f; hi[k] = -1e+20f; }
        for (uint32_t m : nd.members) {
            const float* v = row(m);
            for (int k = 0; k < N; k++) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); }
        }
        float widest = 0.0f; int widest_axis = -1;
        for (int k = 0; k < N; k++) { const float r = hi[k] - lo[k]; if (r > widest) { widest = r; widest_axis = k; } }
        if (widest_axis < 0) return false;
        for (int k = 0; k < N; k++) { l_sum[k] = 0.0f; r_sum[k] = 0.0f; }
        l_w = 0; r_w = 0;
        const size_t half = nd.members.size() / 2;
        for (size_t i = 0; i < nd.members.size(); i++) {
            const uint32_t m = nd.members[i];
            const float wf = (float)weights_[m];
            const float* v = row(m);
            if (i < half) { for (int k = 0; k < N; k++) l_sum[k] += v[k] * wf; l_w += wf; }
            else          { for (int k = 0; k < N; k++) r_sum[k] += v[k] * wf; r_w += wf; }
        }
        if (l_w > 0.0f && r_w > 0.0f) {
            const float ls = static_cast<float>(1.0f / l_w), rs = static_cast<float>(1.0f / r_w);
            for (int k = 0; k < N; k++) { l_out[k] = l_sum[k] * ls; r_out[k] = r_sum[k] * rs; }
        } else {
            for (int k = 0; k < N; k++) { l_out[k] = lo[k]; r_out[k] = hi[k]; }
        }
        return true;
    }

    struct side { float centroid[N]; uint64_t weight; float var; std::vector<uint32_t> members; };

    // refine_split (enc.h:1962-2077): up to 6 two-means iterations
    bool refine(const node& nd, side& L, side& R) const {
        float prev_total = 1e+10f;
        for (int iter = 0; iter < 6; iter++) {
            L.members.clear(); R.members.clear();
            float nl[N], nr[N];
            for (int k = 0; k < N; k++) { nl[k] = 0.0f; nr[k] = 0.0f; }
            double l_tt = 0.0, r_tt = 0.0;
            L.weight = 0; R.weight = 0;
            for (uint32_t m : nd.members) {
                const float* v = row(m);
                const uint64_t w = weights_[m];
                const float wf = static_cast<float>(w);
                const double dl = dist2_d(L.centroid, v), dr = dist2_d(R.centroid, v);
                if (dl >= dr) { for (int k = 0; k < N; k++) nr[k] += v[k] * wf; R.weight += w; r_tt += wf * dot(v, v); R.members.push_back(m); }
                else          { for (int k = 0; k < N; k++) nl[k] += v[k] * wf; L.weight += w; l_tt += wf * dot(v, v); L.members.push_back(m); }
            }
            if (!L.weight || !R.weight) {
                // everything fell on one side: peel off the vectors equal to the first one (enc.h:2014-2048)
                L.members.clear(); R.members.clear();
                for (int k = 0; k < N; k++) { nl[k] = 0.0f; nr[k] = 0.0f; }
                l_tt = 0.0; r_tt = 0.0; L.weight = 0; R.weight = 0;
                float first[N];
                for (int k = 0; k < N; k++) first[k] = 0.0f;
                for (size_t i = 0; i < nd.members.size(); i++) {
                    const uint32_t m = nd.members[i];
                    const float* v = row(m);
                    const uint64_t w = weights_[m];
                    const float wf = static_cast<float>(w);
                    if (!i || rows_equal(v, first)) {
                        std::memcpy(first, v, sizeof(first));
                        for (int k = 0; k < N; k++) nr[k] += v[k] * wf;
                        R.weight += w; r_tt += wf * dot(v, v); R.members.push_back(m);
                    } else {
                        for (int k = 0; k < N; k++) nl[k] += v[k] * wf;
                        L.weight += w; l_tt += wf * dot(v, v); L.members.push_back(m);
                    }
                }
                if (!L.weight || !R.weight) return false;
            }
            L.var = static_cast<float>(l_tt - (dot(nl, nl) / static_cast<float>(L.weight)));
            R.var = static_cast<float>(r_tt - (dot(nr, nr) / static_cast<float>(R.weight)));
            const float li = 1.0f / static_cast<float>(L.weight), ri = 1.0f / static_cast<float>(R.weight);
            for (int k = 0; k < N; k++) { L.centroid[k] = nl[k] * li; R.centroid[k] = nr[k] * ri; }
            const float total = L.var + R.var;
            if (total < .00001f) break;
            if (((prev_total - total) / total) < .00125f) break;
            prev_total = total;
        }
        return true;
    }

    bool all_members_equal(const std::vector<uint32_t>& m) const {
        for (size_t i = 1; i < m.size(); i++) if (!rows_equal(row(m[0]), row(m[i]))) return false;
        return true;
    }

    // split_node (enc.h:1737-1800)
    bool split(uint32_t ni, variance_heap& heap) {
        side L, R;
        if (!initial_children(nodes_[ni], L.centroid, R.centroid)) return false;
        if (!refine(nodes_[ni], L, R)) return false;
        const uint32_t li = (uint32_t)nodes_.size(), ri = li + 1;
        nodes_[ni].left = (int32_t)li;
        nodes_[ni].right = (int32_t)ri;
        nodes_[ni].codebook_index = (int)next_codebook_index_++;
        nodes_.resize(nodes_.size() + 2);
        auto fill = [&](node& c, side& s) {
            std::memcpy(c.origin, s.centroid, sizeof(c.origin));
            c.weight = s.weight; c.var = s.var; c.members.swap(s.members);
            if (c.var <= 0.0f && c.members.size() > 1 && !all_members_equal(c.members)) c.var = 1e-4f;
        };
        fill(nodes_[li], L);
        fill(nodes_[ri], R);
        if (nodes_[li].var > 0.0f && nodes_[li].members.size() > 1) heap.push(li, nodes_[li].var);
        if (nodes_[ri].var > 0.0f && nodes_[ri].members.size() > 1) heap.push(ri, nodes_[ri].var);
        return true;
    }

    std::vector<float> rows_;
    std::vector<uint64_t> weights_;
    uint32_t n_ = 0;
    std::vector<node> nodes_;
    uint32_t next_codebook_index_ = 0;
};

// The thread count generate_hierarchical_codebook_threaded hands to its _internal half (enc.h:2316): the T-way partition is only
// taken from 262,144 distinct vectors up. `max_threads` is what the frontend computes (frontend.cpp:2195-2198): min(hardware threads,
// 8, job pool size) when multithreaded, else 0.
constexpr uint32_t kThreadedCodebookMinUnique = 65536 * 4;
constexpr uint32_t kThreadedCodebookMaxThreads = 16;   // cMaxThreads, enc.h:2111
inline uint32_t codebook_partitions(uint32_t n_unique, uint32_t max_codebook_size, uint32_t max_threads, uint32_t min_unique = kThreadedCodebookMinUnique) {
    if (n_unique < min_unique) max_threads = 1;                                                          // enc.h:2316
    if (max_threads <= 1 || n_unique < 256 || max_codebook_size < max_threads * 16) return 1;           // enc.h:2097
    return std::min(max_threads, kThreadedCodebookMaxThreads);
}

// generate_hierarchical_codebook_threaded (enc.h:2218-2354): `unique_rows` are the DISTINCT training vectors in ascending
// lexicographic order with their summed weights; `groups[u]` lists the original training-vector indices that carry unique vector u,
// ascending. max_threads <= 1 is the single-threaded configuration. With T = codebook_partitions() > 1 the reference
// (generate_hierarchical_codebook_threaded_internal, enc.h:2086-2215) first builds a T-leaf tree, then one INDEPENDENT tree per leaf
// over that leaf's members in list order (own root, own variance queue, ceil(K / T) leaves and ceil(P / T) parents each) and
// concatenates the results in leaf order: deterministic for a given T, whatever the threads' timing.
template <int N>
bool hierarchical_codebook(const std::vector<float>& unique_rows, const std::vector<uint64_t>& unique_weights,
                           const std::vector<std::vector<uint32_t>>& groups, uint32_t max_codebook_size, uint32_t max_parent_codebook_size,
                           std::vector<std::vector<uint32_t>>& codebook, std::vector<std::vector<uint32_t>>& parent_codebook,
                           uint32_t max_threads = 0, uint32_t min_unique_for_threads = kThreadedCodebookMinUnique) {
    const uint32_t n = (uint32_t)unique_weights.size();
    const uint32_t T = codebook_partitions(n, max_codebook_size, max_threads, min_unique_for_threads);
    tsvq<N> q;
    q.set_training(unique_rows.data(), unique_weights.data(), n);
    if (!q.generate(T > 1 ? T : max_codebook_size)) return false;
    std::vector<std::vector<uint32_t>> group_codebook, group_parents;
    q.leaves(group_codebook);
    if (T > 1 && group_codebook.size() >= T) {
        const bool limit = n > max_codebook_size;   // limit_clusterizers, enc.h:2305-2307
        std::vector<std::vector<uint32_t>> initial;
        initial.swap(group_codebook);
        for (uint32_t t = 0; t < T; t++) {
            const std::vector<uint32_t>& members = initial[t];
            std::vector<float> rows(members.size() * N);
            std::vector<uint64_t> weights(members.size());
            for (size_t i = 0; i < members.size(); i++) {
                std::memcpy(&rows[i * N], &unique_rows[(size_t)members[i] * N], sizeof(float) * N);
                weights[i] = unique_weights[members[i]];
            }
            tsvq<N> lq;
            lq.set_training(rows.data(), weights.data(), (uint32_t)members.size());
            if (!lq.generate(limit ? (max_codebook_size + T - 1) / T : (uint32_t)members.size())) return false;
            std::vector<std::vector<uint32_t>> local, local_parents;
            lq.leaves(local);
            if (max_parent_codebook_size) lq.top_clusters((max_parent_codebook_size + T - 1) / T, local_parents);
            for (auto* lists : {&local, &local_parents})
                for (auto& l : *lists) for (uint32_t& m : l) m = members[m];
            for (auto& l : local) group_codebook.emplace_back(std::move(l));
            for (auto& l : local_parents) group_parents.emplace_back(std::move(l));
        }
    } else if (max_parent_codebook_size) q.top_clusters(max_parent_codebook_size, group_parents);
    auto expand = [&](const std::vector<std::vector<uint32_t>>& in, std::vector<std::vector<uint32_t>>& out) {
        out.clear();
        out.resize(in.size());
        for (size_t i = 0; i < in.size(); i++) {
            size_t total = 0;
            for (uint32_t g : in[i]) total += groups[g].size();
            out[i].reserve(total);
            for (uint32_t g : in[i]) out[i].insert(out[i].end(), groups[g].begin(), groups[g].end());
        }
    };
    expand(group_codebook, codebook);
    expand(group_parents, parent_codebook);
    return true;
}

} // namespace bu
