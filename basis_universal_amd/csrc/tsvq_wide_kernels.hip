// tsvq_wide_kernels.hip -- the codebook builder's LARGE nodes (row a8): the same split as tsvq_kernels.hip, bit for bit, but spread
// over the whole chip instead of one workgroup per node.
//
// tree_vector_quant<>::split_node (encoder/basisu_enc.h:1737-2077) is a handful of passes over the node's members; every pass
// classifies each member (independent work) and adds its contribution to 16..136 RUNNING float sums in member order. The
// one-workgroup kernel pays one dependent v_add_f32 per member per pass -- 18 ms for the root of the 4096^2 selector codebook
// (674,691 members, ~9 passes), on 1 of 256 CUs. Here a pass is cut into blocks of 256 members:
//
//   k_wide_sums      per block: classify, store the side, exact block sums of every chain (double) + the integer totals
//   k_wide_scan      per node:  prefix of the block sums -> the binade every chain's running sum will be in at every block start
//                               (a PREDICTION: it steers work, it never decides a result), left-count prefix, exact totals
//   k_wide_stretches per block: for each chain and two candidate binades, the block's addends folded into one parity map
//                               (fsum_scan.h: inside a binade a float add is k -> k + d[k & 1] on the significand)
//   k_wide_walk      per chain: one wave composes 64 blocks' maps at a time (prefix scan), applies the longest valid prefix to
//                               the running sum, and adds the members of a block one by one only where a map does not apply
//                               (binade crossings; stretches of those are taken block by block with the next blocks' members in
//                               flight). Integer-valued chains start behind the blocks whose running sum is <= 2^24 (exact)
//   k_wide_finish    per node:  the serial tail of the pass (centroids, variances, convergence test; PCA after the covariance pass)
//   k_wide_partition per block: the children's member lists (stable partition) + the result record
//
// The sums come out identical to the sequential ones because the walk applies a map only where fsum::applies() proves that
// every single add of the stretch stayed inside the binade the map was built for; everything else is added for real, in order.
// Selector vectors only (packed rows): that is where the large nodes are (the endpoint side has <= 2^18 distinct vectors).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include "tsvq_kernels.h"
#include "tsvq_common.h"
#include "tsvq_bufs.h"
#include "fsum_scan.h"

namespace bu {

namespace {

#include "tsvq_wide_common.h"   // (inside the unnamed namespace: internal linkage in each of the two translation units that use it)


enum { WM_ROOT = 0, WM_COV = 1, WM_PROJ = 2, WM_DIST = 3 };
#define W_IS_DIST(MODE, dist) ((MODE) == WM_DIST)
#define W_IS_PROJ(MODE, dist) ((MODE) == WM_PROJ)

template <int MODE> struct mode_traits;
template <> struct mode_traits<WM_ROOT> { static constexpr int NCH = 16; };
template <> struct mode_traits<WM_COV>  { static constexpr int NCH = 136; };
template <> struct mode_traits<WM_PROJ> { static constexpr int NCH = 32; };
template <> struct mode_traits<WM_DIST> { static constexpr int NCH = 32; };

// covariance chain -> (x, y >= x), the enumeration of tsvq_kernels.hip (row-major upper triangle)
__device__ __forceinline__ void cov_xy(int c, int& x, int& y) { x = 0; while (c >= 16 - x) { c -= 16 - x; x++; } y = x + c; }


struct member_info { uint32_t key; float wf; uint64_t w; bool valid; };

__device__ __forceinline__ member_info fetch_member(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint32_t* __restrict__ members,
                                                    uint32_t pos, uint32_t count) {
    member_info m;
    m.valid = pos < count;
    const uint32_t p = m.valid ? pos : count - 1;
    const uint32_t mi = members ? members[p] : p;
    m.key = keys[mi]; m.w = w64[mi]; m.wf = (float)m.w;
    return m;
}

// which child does a member go to (enc.h:1870-1871 projection sign, enc.h:1991 distance comparison)
template <int MODE>
__device__ __forceinline__ bool classify(uint32_t key, const float* s_origin, const float* s_axis, const double2 (*s_tab)[4], bool dist) {
    if (W_IS_DIST(MODE, dist)) {
        double dl = 0, dr = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) { const double2 t = s_tab[k][packed16_value(key, k)]; dl += t.x; dr += t.y; }
        return dl >= dr;
    } else if (W_IS_PROJ(MODE, dist)) {
        float dd[16];
#pragma unroll
        for (int k = 0; k < 16; k++) dd[k] = (float)packed16_value(key, k) - s_origin[k];
        return (double)dot_seq<16>(dd, s_axis) >= 0.0;
    }
    return false;
}

// ------------------------------------------------------------------------------------------------------------ per-block front end
// The LDS tiles both per-block kernels work from. Side modes / root: fa[k][m] = v_k * w, sd[m] = side (2: no member).
// Covariance: fa[k][m] = d_k, fb[k][m] = w * d_k.
template <int MODE>
struct tiles {
    float fa[16][WROW];
    float fb[MODE == WM_COV ? 16 : 1][WROW];
    uint8_t sd[WB];
};

// ------------------------------------------------------------------------------------------------------------ k_wide_sums
// (The per-phase bodies are device functions: the classic path wraps each in a kernel of its own, the fused kernel of the side passes calls them in turn with grid-wide
//  barriers in between. What other workgroups write between phases -- ctrl, the workspace, side -- is reached through plain pointers: no __restrict__ promise there.)
template <int MODE>
__device__ __forceinline__ void wide_sums_body(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint32_t* perm0,
                                               const uint32_t* perm1, uint8_t* side, const tsvq_wide_node* __restrict__ nodes,
                                               uint32_t n_nodes, const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb, uint2* pk, const uint32_t blk, const bool dist) {
    constexpr int NCH = mode_traits<MODE>::NCH;
    __shared__ tiles<MODE> T;
    __shared__ float s_origin[16], s_axis[16];
    __shared__ double2 s_tab[16][4];
    __shared__ double s_part[8][32];
    __shared__ double s_cpart[2][136];
    __shared__ uint32_t s_cnz[2][136];
    __shared__ uint64_t s_red[4][8];
    const wide_ws ws = carve(ws_base, tb);
    const int tid = threadIdx.x;
    const uint32_t ni = find_node(nodes, n_nodes, blk);
    const tsvq_wide_ctrl& ct = ctrl[ni];
    if (ct.done) return;
    const tsvq_wide_node& nd = nodes[ni];
    if (tid < 16) { s_origin[tid] = nd.origin[tid]; s_axis[tid] = ct.axis[tid]; }
    if (W_IS_DIST(MODE, dist) && tid < 64) {   // squared centroid differences per value (tsvq_kernels.hip, TQ_MODE_DIST)
        const int k = tid >> 2, val = tid & 3;
        const double a = (double)ct.l_c[k] - (double)(float)val, b = (double)ct.r_c[k] - (double)(float)val;
        s_tab[k][val] = make_double2(a * a, b * b);
    }
    __syncthreads();
    const uint32_t pos = (blk - nd.first_block) * WB + (uint32_t)tid;
    // (the root pass of the whole training set has no member list: perm0 == nullptr; the roots of member SPANS -- the independent trees of the partitioned
    //  build, enc.h:2137-2152 -- read their span and lay it out in list order like the covariance pass does)
    const uint32_t* members = (MODE == WM_ROOT && !perm0) ? nullptr : tsvq_list(perm0, perm1, nd.buf) + nd.start;
    const member_info m = fetch_member(keys, w64, members, pos, nd.count);
    if (MODE == WM_ROOT && pk && m.valid) pk[nd.start + pos] = make_uint2(m.key, __float_as_uint(m.wf));
    bool right = false;
    uint64_t red[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (MODE == WM_COV) {
        // the first pass of a split also lays the members out in list order (key, float weight): the later passes and the walks
        // read that instead of gathering through the member list
        if (m.valid) pk[nd.start + pos] = make_uint2(m.key, __float_as_uint(m.wf));
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float dk = (float)packed16_value(m.key, k) - s_origin[k];
            T.fa[k][tid] = m.valid ? dk : 0.0f;
            T.fb[k][tid] = m.valid ? m.wf * dk : 0.0f;
        }
    } else {
        if (MODE != WM_ROOT) right = classify<MODE>(m.key, s_origin, s_axis, s_tab, dist);
        float vsq = 0.0f;
        {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { v[k] = (float)packed16_value(m.key, k); T.fa[k][tid] = m.valid ? v[k] * m.wf : 0.0f; }
            vsq = dot_seq<16>(v, v);
        }
        T.sd[tid] = m.valid ? (right ? 1 : 0) : 2;
        if (m.valid) {
            if (MODE != WM_ROOT) side[nd.start + pos] = right ? 1 : 0;
            // the reference's double accumulators: l_weight / r_weight in the projection pass (enc.h:1873-1881), ttsum otherwise
            const float dvf = W_IS_PROJ(MODE, dist) ? m.wf : m.wf * vsq;
            exact_acc ex;
            const bool ok = ex.add(dvf);
            if (right) { red[1] = m.w; red[5] = ex.lo; red[6] = ex.hi; } else { red[0] = m.w; red[2] = 1; red[3] = ex.lo; red[4] = ex.hi; }
            red[7] = ok ? 0 : 1;
        }
    }
    __syncthreads();
    if (MODE == WM_COV) {
        for (int item = tid; item < 2 * 136; item += WB) {
            const int c = item % 136, h = item / 136;
            int x, y; cov_xy(c, x, y);
            double s = 0; uint32_t nz = 0;
            for (int j = h * 128; j < h * 128 + 128; j++) {
                const float p = T.fa[x][j] * T.fb[y][j];
                s += (double)p;
                nz |= __float_as_uint(p) << 1;
            }
            s_cpart[h][c] = s; s_cnz[h][c] = nz;
        }
        __syncthreads();
        if (tid < 136) {
            ws.bsum[ws.at(tid, blk)] = s_cpart[0][tid] + s_cpart[1][tid];
            ws.bzero[ws.at(tid, blk)] = (s_cnz[0][tid] | s_cnz[1][tid]) ? 0 : 1;
        }
        return;
    }
    {   // 32 chains x 8 slices of 32 members
        const int c = tid & 31, sl = tid >> 5;
        const int k = c & 15; const uint8_t want = (uint8_t)(c >> 4);
        double s = 0;
        for (int j = sl * 32; j < sl * 32 + 32; j++) s += (T.sd[j] == want) ? (double)T.fa[k][j] : 0.0;
        s_part[sl][c] = s;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {   // DPP wave sums (total in lane 63): the xor-shuffle tree this replaces was 96 trips through the LDS crossbar per thread
        const uint64_t v = wave_sum_u64(red[i]);
        if ((tid & 63) == 63) s_red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < NCH) {
        double s = 0;
        for (int i = 0; i < 8; i++) s += s_part[i][tid];
        ws.bsum[ws.at(tid, blk)] = s;
        ws.bzero[ws.at(tid, blk)] = s == 0.0 ? 1 : 0;   // addends are >= 0 here
    }
    if (tid >= 64 && tid < 72) ws.bex[(size_t)blk * 8 + (tid - 64)] = s_red[0][tid - 64] + s_red[1][tid - 64] + s_red[2][tid - 64] + s_red[3][tid - 64];
}
template <int MODE>
__global__ __launch_bounds__(WB) void k_wide_sums(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint32_t* perm0, const uint32_t* perm1, uint8_t* side,
                                                  const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes, const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb, uint2* pk) {
    wide_sums_body<MODE>(keys, w64, perm0, perm1, side, nodes, n_nodes, ctrl, ws_base, tb, pk, blockIdx.x, MODE == WM_DIST);
}


// ------------------------------------------------------------------------------------------------------------ k_wide_scan
// grid (node, y): y < NCH: one chain per workgroup -- thread t takes ceil(n_blocks / 256) consecutive blocks, the workgroup scans the threads' totals, and every
//                          thread walks its blocks again with the prefix in front of them (the sweep this replaces gave a chain ONE wave that took 64 blocks per
//                          step: 42 dependent wave scans for the 2,636 blocks of the 4096^2 root, 17-31 us per pass; the block sums are 21 KB per chain, L2-resident);
//                 y == NCH (not for the covariance pass): the integer totals and the left-count prefix of the node.
// The prefix is a PREDICTION aid for the signed covariance chains (association order immaterial) and exact for the integer-valued side / root chains (doubles < 2^53).
template <int MODE>
__device__ __forceinline__ void wide_scan_body(const tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb, const uint32_t ni, const uint32_t y) {
    constexpr int NCH = mode_traits<MODE>::NCH;
    __shared__ uint64_t s_tot[4][8];
    __shared__ uint32_t s_wl[4];
    __shared__ double s_wsum[4], s_wmax[4];
    __shared__ uint32_t s_wcnt[4];
    const wide_ws ws = carve(ws_base, tb);
    if (ctrl[ni].done) return;
    const tsvq_wide_node nd = nodes[ni];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)y < NCH) {
        const int c = (int)y;
        const uint32_t perb = (nd.n_blocks + 255) / 256;
        const uint32_t q0 = min((uint32_t)tid * perb, nd.n_blocks), q1 = min(q0 + perb, nd.n_blocks);
        const size_t at0 = ws.at(c, nd.first_block);
        double loc = 0;
        for (uint32_t b = q0; b < q1; b++) loc += ws.bsum[at0 + b];
        const double incl = wave_prefix_f64(loc);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        double P = incl - loc;   // the sum of everything in front of this thread's first block
        for (int w = 0; w < wave; w++) P += s_wsum[w];
        const double total = ((s_wsum[0] + s_wsum[1]) + s_wsum[2]) + s_wsum[3];
        // the leading blocks over which the (integer, non-decreasing) running sum stays <= 2^24: no rounding there, the walk starts behind them
        uint32_t ex_n = 0; double ex_sum = 0.0;
        for (uint32_t b = q0; b < q1; b++) {
            const size_t at = at0 + b;
            const double v = ws.bsum[at];
            const double Ps = P;
            P += v;
            if (MODE != WM_COV && P <= 16777216.0) { ex_n++; ex_sum = P; }
            uint16_t ep;
            if (ws.bzero[at]) ep = EP_ZERO;
            else {
                // the running float sum at this block's start is within (members so far) half-ulps of Ps: the binade of the lower
                // end, and whether the upper end is in the same one (then only one map is needed)
                const double eps = (double)((uint64_t)b * WB + 1) * 5.9604644775390625e-08;
                const double lo = fabs(Ps) * (1.0 - eps), hi = fabs(Ps) * (1.0 + eps);
                float lf = (float)(lo > 0.0 ? lo : 0.0), hf = (float)hi;
                if ((double)lf > lo) lf = __uint_as_float(__float_as_uint(lf) - 1u);
                if ((double)hf < hi) hf = __uint_as_float(__float_as_uint(hf) + 1u);
                const uint32_t e = (__float_as_uint(lf) >> 23) & 0xffu, eh = (__float_as_uint(hf) >> 23) & 0xffu;
                ep = (e >= 1u && e <= 252u) ? (uint16_t)(e | (Ps < 0.0 ? 0x100u : 0u) | (eh == e ? EP_SINGLE : 0u)) : EP_NONE;
            }
            ws.epred[at] = ep;
        }
        if (MODE != WM_COV) {   // (the sums are monotone: the blocks that qualify are a prefix of the list, the last of them carries the largest sum)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { ex_n += (uint32_t)__shfl_xor((int)ex_n, o, 64); ex_sum = fmax(ex_sum, __shfl_xor(ex_sum, o, 64)); }
            if (lane == 0) { s_wcnt[wave] = ex_n; s_wmax[wave] = ex_sum; }
            __syncthreads();
        }
        // Side / root chains add non-negative INTEGER-valued floats (value 0..3 times an integer weight). If the chain's total is below 2^24, every
        // partial sum of the sequential float chain is an integer below 2^24, i.e. exact: the chain's result is the total, in any order. (Block sums
        // are exact in double.) The other kernels of the pass skip such chains; the predictions written above are then never looked at.
        if (tid == 0) {
            const bool exact = MODE != WM_COV && total < 16777216.0;
            ctrl[ni].exact[c] = exact ? 1u : 0u; if (exact) ctrl[ni].sums[c] = (float)total;
            const uint32_t exact_blocks = MODE != WM_COV ? s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3] : 0u;
            const double exact_sum = MODE != WM_COV ? fmax(fmax(s_wmax[0], s_wmax[1]), fmax(s_wmax[2], s_wmax[3])) : 0.0;
            ctrl[ni].start_block[c] = exact_blocks; ctrl[ni].start_sum[c] = (float)exact_sum;
        }
        return;
    }
    // totals of the integer accumulators and the left-count prefix (block order)
    uint64_t tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t perb = (nd.n_blocks + 255) / 256;
    const uint32_t q0 = min((uint32_t)tid * perb, nd.n_blocks), q1 = min(q0 + perb, nd.n_blocks);
    for (uint32_t b = q0; b < q1; b++)
#pragma unroll
        for (int i = 0; i < 8; i++) tot[i] += ws.bex[(size_t)(nd.first_block + b) * 8 + i];
    const uint32_t my_left = (uint32_t)tot[2];
    uint32_t incl = my_left;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t v = wave_sum_u64(tot[i]);
        if (lane == 63) s_tot[tid >> 6][i] = v;
    }
    if (lane == 63) s_wl[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (tid >> 6); w++) base += s_wl[w];
    uint32_t run = base + incl - my_left;
    for (uint32_t b = q0; b < q1; b++) { ws.lpre[nd.first_block + b] = run; run += (uint32_t)ws.bex[(size_t)(nd.first_block + b) * 8 + 2]; }
    if (tid == 0) {
        uint64_t t[8];
        for (int i = 0; i < 8; i++) { t[i] = 0; for (int w = 0; w < 4; w++) t[i] += s_tot[w][i]; }
        tsvq_wide_ctrl& ct = ctrl[ni];
        ct.l_w = t[0]; ct.r_w = t[1]; ct.l_n = (uint32_t)t[2]; ct.r_n = nd.count - (uint32_t)t[2];
        double d0 = 0, d1 = 0;
        const bool ok = t[7] == 0 && exact_total(t[3], t[4], &d0) && exact_total(t[5], t[6], &d1);
        ct.dsum[0] = d0; ct.dsum[1] = d1;
        ct.ex_bad = ok ? 0u : 1u;
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_wide_scan(const tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb) {
    wide_scan_body<MODE>(nodes, ctrl, ws_base, tb, blockIdx.x, blockIdx.y);
}

// ------------------------------------------------------------------------------------------------------------ k_wide_stretches
// (key, float weight) of the member at list position pos: from the packed copy the covariance pass made, or -- root -- directly
template <int MODE>
__device__ __forceinline__ void fetch_packed(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint2* __restrict__ pk, uint32_t node_start,
                                             uint32_t pos, uint32_t count, uint32_t& key, float& wf, bool& valid) {
    valid = pos < count;
    const uint32_t p = valid ? pos : count - 1;
    if (MODE == WM_ROOT && !pk) { key = keys[p]; wf = (float)w64[p]; }
    else { const uint2 v = pk[node_start + p]; key = v.x; wf = __uint_as_float(v.y); }
}


// (320 threads -- two rounds over the covariance pass's 544 items instead of three -- measured slower: 307 against 250 us for the root.)
constexpr int ST_THREADS = WB;
// One item = (chain, slice of the block's members), both candidate binades; the Q slices of a chain sit in Q neighbouring lanes and are composed in member order by
// a segmented DPP scan (row_shr 1 / 2 / 4 / 8 inside aligned groups of Q lanes), so the folded maps never leave the registers -- the kernel's LDS is the tile alone
// (37 KB for the covariance pass instead of 59 KB with a staging array for the maps: four workgroups per CU instead of two for a kernel that waits on LDS reads).
// The tile's columns carry one float of padding per slice: the lanes of a group read the same rows at offsets 64 apart, which would be one bank.
template <int CTRL, int SHIFT>
__device__ __forceinline__ void seg_scan_step(fsum::stretch& st, const int q) {
    fsum::stretch f;
    f.d[0] = dpp_mov<CTRL, 0xf>(0, st.d[0]); f.d[1] = dpp_mov<CTRL, 0xf>(0, st.d[1]);
    f.lo[0] = dpp_mov<CTRL, 0xf>(fsum::D_SAT, st.lo[0]); f.lo[1] = dpp_mov<CTRL, 0xf>(fsum::D_SAT, st.lo[1]);
    f.hi[0] = dpp_mov<CTRL, 0xf>(-fsum::D_SAT, st.hi[0]); f.hi[1] = dpp_mov<CTRL, 0xf>(-fsum::D_SAT, st.hi[1]);
    if (q < SHIFT) f = fsum::identity();   // the lane SHIFT below belongs to another chain
    st = fsum::compose(f, st);
}
template <int Q>
__device__ __forceinline__ void seg_scan(fsum::stretch& st, const int q) {
    seg_scan_step<0x111, 1>(st, q);
    if (Q > 2) seg_scan_step<0x112, 2>(st, q);
    if (Q > 4) seg_scan_step<0x114, 4>(st, q);
    if (Q > 8) seg_scan_step<0x118, 8>(st, q);
}

constexpr int ST_ROW = 289;   // 256 members + one float of padding per slice (<= 16), and = 1 mod 32
template <int MODE>
struct st_tiles {
    float fa[16][ST_ROW];
    float fb[MODE == WM_COV ? 16 : 1][ST_ROW];
    uint8_t sd[WB];
};

template <int MODE>
__device__ __forceinline__ void wide_stretches_body(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint2* pk,
                                                    const uint8_t* side, const tsvq_wide_node* __restrict__ nodes,
                                                    uint32_t n_nodes, const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb, const uint32_t blk) {
    constexpr int NCH = mode_traits<MODE>::NCH;
    constexpr int Q = MODE == WM_COV ? 4 : (MODE == WM_ROOT ? 16 : 8);   // member slices per chain
    constexpr int ITEMS = NCH * Q;
    constexpr int SL = WB / Q;
    static_assert(ITEMS % Q == 0 && ST_THREADS % Q == 0 && 16 % Q == 0, "the slices of a chain are Q aligned lanes of one DPP row");
    __shared__ st_tiles<MODE> T;
    __shared__ float s_origin[16];
    const wide_ws ws = carve(ws_base, tb);
    const int tid = threadIdx.x;
    const uint32_t ni = find_node(nodes, n_nodes, blk);
    if (ctrl[ni].done) return;
    // every chain of this pass finished by the scan (integer totals below 2^24): nothing to fold
    if (MODE != WM_COV && __syncthreads_and(tid < NCH ? (ctrl[ni].exact[tid] != 0 ? 1 : 0) : 1)) return;
    const tsvq_wide_node& nd = nodes[ni];
    if (tid < 16) s_origin[tid] = nd.origin[tid];
    __syncthreads();
    const uint32_t pos = (blk - nd.first_block) * WB + (uint32_t)min(tid, WB - 1);
    uint32_t key; float wf; bool valid;
    fetch_packed<MODE>(keys, w64, pk, nd.start, pos, nd.count, key, wf, valid);
    const int col = tid + tid / SL;   // member tid's column
    if (tid >= WB) {
        // not a staging thread
    } else if (MODE == WM_COV) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float dk = (float)packed16_value(key, k) - s_origin[k];
            T.fa[k][col] = valid ? dk : 0.0f;
            T.fb[k][col] = valid ? wf * dk : 0.0f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) T.fa[k][col] = valid ? (float)packed16_value(key, k) * wf : 0.0f;
        T.sd[tid] = valid ? (MODE == WM_ROOT ? 0 : side[nd.start + pos]) : 2;
    }
    __syncthreads();
    for (int item = tid; item < ITEMS; item += ST_THREADS) {
        const int c = item / Q, q = item % Q;
        const bool skip = ctrl[ni].exact[c] != 0;
        const uint16_t ep = skip ? EP_NONE : ws.epred[ws.at(c, blk)];
        fsum::stretch st0 = fsum::identity(), st1 = fsum::identity();
        if (ep != EP_NONE && ep != EP_ZERO) {
            const int E = (int)(ep & 0xffu);
            const bool neg = (ep & 0x100u) != 0, two = (ep & EP_SINGLE) == 0;
            int x = c & 15, y = 0;
            if (MODE == WM_COV) cov_xy(c, x, y);
            const uint8_t want = (uint8_t)(c >> 4);
            bool bad0 = false, bad1 = false;
            const float* ra = &T.fa[x][q * SL + q];
            const float* rb = &T.fb[MODE == WM_COV ? y : 0][q * SL + q];
            for (int jj = 0; jj < SL; jj++) {
                uint32_t bits;
                if (MODE == WM_COV) bits = __float_as_uint(ra[jj] * rb[jj]);
                else { if (T.sd[q * SL + jj] != want) continue; bits = __float_as_uint(ra[jj]); }
                if ((bits << 1) == 0) continue;
                const fsum::parts pr = fsum::split(bits, neg);
                fsum::push_fast(st0, fsum::decode_fast(pr, E, bad0));
                if (two) fsum::push_fast(st1, fsum::decode_fast(pr, E + 1, bad1));
            }
            if (bad0) fsum::poison(st0);
            if (bad1 || E + 1 > 253) fsum::poison(st1);
        }
        // slices in member order: an inclusive scan over the Q lanes of the chain, the last one has the block's map
        seg_scan<Q>(st0, q);
        seg_scan<Q>(st1, q);
        if (q == Q - 1 && !skip) {
            int32_t* o = ws.summ + ws.at(c, blk) * 2 * 6;
            st_store(o, st0); st_store(o + 6, st1);
        }
    }
}
template <int MODE>
__global__ __launch_bounds__(ST_THREADS) void k_wide_stretches(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint2* pk, const uint8_t* side,
                                                       const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes, const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb) {
    wide_stretches_body<MODE>(keys, w64, pk, side, nodes, n_nodes, ctrl, ws_base, tb, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------ k_wide_windows
// The walk below is serial in the running sum, but what it does per window of 64 blocks -- compose the blocks' maps -- is not: here every (chain, window) gets
// a wave that composes the window's 64 maps for the two states the walk can arrive in (the binade the scan predicted for the window's first non-zero block, and the one
// above), so that the walk takes a window it finds in one of them with ONE applies() test. A window whose blocks do not all have a map for the state (a binade change
// predicted inside it, a block the stretches kernel gave up on) is recorded as never applying and walked block by block as before.
// Record (16 dwords): [0] predicted exponent | sign << 8 | all-zero << 9, [2..7] the map for that exponent (d0 d1 lo0 lo1 hi0 hi1), [8..13] for the exponent above.
__device__ __forceinline__ uint32_t window_index(const tsvq_wide_node& nd, uint32_t ni, uint32_t w) { return nd.first_block / 64 + ni + w; }

template <int MODE>
__global__ __launch_bounds__(64) void k_wide_windows(const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes, const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb) {
    constexpr int NCH = mode_traits<MODE>::NCH;
    const wide_ws ws = carve(ws_base, tb);
    const int lane = (int)threadIdx.x;
    const int c = (int)(blockIdx.x % NCH);
    const uint32_t g = blockIdx.x / NCH;   // window index over the batch: first_block / 64 + node + window of the node (strictly increasing with the node)
    uint32_t lo = 0, hi = n_nodes;         // last node whose first window index <= g
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (window_index(nodes[mid], mid, 0) <= g) lo = mid; else hi = mid; }
    const uint32_t ni = lo;
    const tsvq_wide_node& nd = nodes[ni];
    const uint32_t w = g - window_index(nd, ni, 0);
    if (w * 64 >= nd.n_blocks || ctrl[ni].done || (MODE != WM_COV && ctrl[ni].exact[c])) return;
    walk_window cur;
    load_window(ws, nd.first_block, nd.n_blocks, w * 64, lane, c, cur);
    const uint64_t nonzero = __ballot(cur.ep != EP_ZERO);
    int32_t* rec = ws.win + ((size_t)c * ws.tw + g) * 16;
    if (nonzero == 0ull) { if (lane == 0) rec[0] = 1 << 9; return; }
    const int first = __ffsll((long long)nonzero) - 1;
    const uint32_t ep0 = (uint32_t)__builtin_amdgcn_readlane((int)cur.ep, first);
    const int X0 = (int)(ep0 & 0xffu);
    const uint32_t sign0 = (ep0 >> 8) & 1u;
    if (lane == 0) rec[0] = (int32_t)((ep0 == EP_NONE ? 0u : (uint32_t)X0) | (sign0 << 8));   // exponent 0: no state has it -- never taken
#pragma unroll
    for (int cand = 0; cand < 2; cand++) {
        fsum::stretch st = fsum::identity();
        if (cur.ep != EP_ZERO) {
            const int dE = X0 + cand - (int)(cur.ep & 0xffu);
            const bool usable = cur.ep != EP_NONE && ep0 != EP_NONE && ((cur.ep >> 8) & 1u) == sign0 && (dE == 0 || (dE == 1 && !(cur.ep & EP_SINGLE)));
            if (!usable) fsum::poison(st);
            else {
                const bool up = dE != 0;
                st.d[0] = up ? cur.m[1][0] : cur.m[0][0]; st.d[1] = up ? cur.m[1][1] : cur.m[0][1];
                st.lo[0] = up ? cur.m[1][2] : cur.m[0][2]; st.lo[1] = up ? cur.m[1][3] : cur.m[0][3];
                st.hi[0] = up ? cur.m[1][4] : cur.m[0][4]; st.hi[1] = up ? cur.m[1][5] : cur.m[0][5];
            }
        }
        wave_scan(st);
        if (lane == 63) { int32_t* o = rec + 2 + cand * 6; o[0] = st.d[0]; o[1] = st.d[1]; o[2] = st.lo[0]; o[3] = st.lo[1]; o[4] = st.hi[0]; o[5] = st.hi[1]; }
    }
}

// ------------------------------------------------------------------------------------------------------------ k_wide_walk
// One wave per (node, chain). A window of 64 blocks is resident in registers with the maps of BOTH candidate binades, so that
// nothing has to be loaded again when the running sum changes binade inside the window; the following windows are loaded
// while the current one is walked. A block whose map does not apply is added member by member out of LDS.
// One WAVE per (node, chain) = `task`; s_add: 256 floats of LDS owned by that wave (the only synchronisation inside is between the lanes of the wave, whose LDS
// accesses execute in program order: a scheduling fence is all the member-by-member blocks need).
template <int MODE, bool USE_WIN = false>
__device__ __forceinline__ void wide_walk_body(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint2* pk,
                                               const uint8_t* side, const tsvq_wide_node* __restrict__ nodes,
                                               tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb, const uint32_t task, const int lane, float* s_add) {
    constexpr int NCH = mode_traits<MODE>::NCH;
    constexpr bool MONO = MODE != WM_COV;
    const wide_ws ws = carve(ws_base, tb);
    const uint32_t ni = task / NCH;
    const int c = (int)(task % NCH);
    if (ctrl[ni].done || ctrl[ni].exact[c]) return;
    const tsvq_wide_node& nd = nodes[ni];
    // what the chain adds, per member: everything about the chain is uniform over the wave
    int cx = 0, cy = 0;
    if (MODE == WM_COV) cov_xy(c, cx, cy); else cx = c & 15;
    const float ox = nd.origin[cx], oy = nd.origin[cy];
    const bool chain_right = (c >> 4) != 0;
    uint32_t s = 0;          // +0.0f

    // this lane's block of the current window against the running sum s: 0 = identity (all addends zero / past the end),
    // 1 = its map for s's binade exists (returned in st), 2 = no map for this state
    auto pick = [&](const walk_window& cur, fsum::stretch& st) -> int {
        st = fsum::identity();
        if (cur.ep == EP_ZERO) return 0;
        const int cand = fsum::state_exp(s) - (int)(cur.ep & 0xffu);
        const bool usable = cur.ep != EP_NONE && fsum::state_ok(s) && (cand == 0 || (cand == 1 && !(cur.ep & EP_SINGLE))) && (((cur.ep >> 8) & 1u) == (s >> 31));
        if (!usable) return 2;
        // selects, not cur.m[cand]: a dynamic index would put the windows into scratch
        const bool up = cand != 0;
        st.d[0] = up ? cur.m[1][0] : cur.m[0][0]; st.d[1] = up ? cur.m[1][1] : cur.m[0][1];
        st.lo[0] = up ? cur.m[1][2] : cur.m[0][2]; st.lo[1] = up ? cur.m[1][3] : cur.m[0][3];
        st.hi[0] = up ? cur.m[1][4] : cur.m[0][4]; st.hi[1] = up ? cur.m[1][5] : cur.m[0][5];
        return 1;
    };
    // the members of block `blk` (of the node) that this lane stages: positions blk * 256 + r * 64 + lane. Loading and turning them into
    // the chain's addends are separate steps so that the loads stay in flight until the addends are needed.
    struct staged { uint32_t key[4]; uint64_t w[4]; uint8_t sd[4]; };
    auto fetch_block = [&](uint32_t blk, staged& m) {
        const uint32_t p0 = min(blk, nd.n_blocks - 1) * WB;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t pos = min(p0 + (uint32_t)(r * 64 + lane), nd.count - 1);
            if (MODE == WM_ROOT && !pk) { m.key[r] = keys[pos]; m.w[r] = w64[pos]; }
            else { const uint2 v = pk[nd.start + pos]; m.key[r] = v.x; m.w[r] = v.y; }
            m.sd[r] = (MODE == WM_PROJ || MODE == WM_DIST) ? side[nd.start + pos] : (uint8_t)0;
        }
    };
    auto addends = [&](uint32_t blk, const staged& m, float (&a)[4]) {
        const uint32_t p0 = blk * WB;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool valid = p0 + (uint32_t)(r * 64 + lane) < nd.count;
            const uint32_t key = m.key[r];
            const float wf = (MODE == WM_ROOT && !pk) ? (float)m.w[r] : __uint_as_float((uint32_t)m.w[r]);
            float v;
            if (MODE == WM_COV) {
                const float dx = (float)packed16_value(key, cx) - ox;
                const float dy = (float)packed16_value(key, cy) - oy;
                const float wdy = wf * dy;
                v = dx * wdy;
            } else {
                v = (float)packed16_value(key, cx) * wf;
                if (MODE == WM_PROJ || MODE == WM_DIST) v = (m.sd[r] != 0) == chain_right ? v : 0.0f;
            }
            a[r] = valid ? v : -0.0f;   // past the node's end: leaves every sum as it is
        }
    };

    uint32_t n_scans = 0, n_raw = 0;
    // one window (64 blocks starting at b0) against the running sum
    int first_start = 0;
    auto process = [&](const walk_window& cur, uint32_t b0) {
        int start = first_start;       // lanes below are done
        first_start = 0;
        while (start < 64) {
            fsum::stretch st;
            n_scans++;
            int kind = pick(cur, st);
            if (lane < start) { kind = 0; st = fsum::identity(); }
            const int32_t k0 = fsum::state_k(s);
            const bool sok = fsum::state_ok(s);
            bool ok; int32_t d_sel;
            if (MONO) {
                mono m;
                m.d[0] = kind == 2 ? fsum::D_SAT : st.d[0]; m.d[1] = kind == 2 ? fsum::D_SAT : st.d[1];
                wave_scan(m);
                d_sel = (k0 & 1) ? m.d[1] : m.d[0];
                // nothing added so far (zero blocks only): fine for any state, +0 included
                ok = (m.d[0] == 0 && m.d[1] == 0) || (sok && k0 + d_sel < fsum::K_HI);
            } else {
                if (kind == 2) { st.lo[0] = st.lo[1] = -fsum::D_SAT; st.hi[0] = st.hi[1] = fsum::D_SAT; }
                wave_scan(st);
                d_sel = (k0 & 1) ? st.d[1] : st.d[0];
                const bool ident = st.lo[0] == fsum::D_SAT && st.lo[1] == fsum::D_SAT && st.hi[0] == -fsum::D_SAT && st.hi[1] == -fsum::D_SAT;
                if (ident) d_sel = 0;
                ok = ident || (sok && fsum::applies(st, k0));
            }
            const uint64_t good = __ballot(ok);
            const int first_fail = (~good == 0ull) ? 64 : __ffsll((long long)~good) - 1;
            if (first_fail > 0) {
                const int32_t d = __builtin_amdgcn_readlane(d_sel, first_fail - 1);
                if (d != 0) s = (s & 0xff800000u) | ((uint32_t)(k0 + d) & 0x7fffffu);
            }
            start = first_fail;
            if (first_fail == 64 || b0 + (uint32_t)first_fail >= nd.n_blocks) { start = 64; break; }
            // ---- a stretch of trouble (the sum changes binade inside blocks, or addends are as large as the sum): block by block, each
            //      either through its own map or member by member, until six blocks in a row went through their maps. The members of
            //      the next three blocks are always on their way (three rotating register sets), needed or not.
            const int lim = (int)min(64u, nd.n_blocks - b0);
            int j = first_fail, calm = 0;
            staged a0, a1, a2;
            // (unconditionally, past the end the last member again: with a fixed number of loads in flight the waits before a set's use
            //  cover that set only)
            fetch_block(b0 + (uint32_t)j, a0);
            fetch_block(b0 + (uint32_t)j + 1, a1);
            fetch_block(b0 + (uint32_t)j + 2, a2);
            auto step = [&](staged& m) -> bool {   // block j against s; true = leave this mode
                fsum::stretch t;
                const int kd = pick(cur, t);
                const int32_t k = fsum::state_k(s);
                const bool fits = kd == 0 || (kd == 1 && fsum::applies(t, k));
                const int32_t dsel = kd == 1 ? ((k & 1) ? t.d[1] : t.d[0]) : 0;
                const int ju = __builtin_amdgcn_readfirstlane(j);
                if ((__ballot(fits) >> ju) & 1ull) {
                    const int32_t d = __builtin_amdgcn_readlane(dsel, ju);
                    if (d != 0) s = (s & 0xff800000u) | ((uint32_t)(k + d) & 0x7fffffu);
                    calm++;
                } else {
                    // the 256 addends go through LDS so that every lane can run the same chain on them; the reads are issued sixteen
                    // (64 addends) ahead of the adds, which then follow each other at the VALU's own pace
                    float a[4];
                    addends(b0 + (uint32_t)j, m, a);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 4; r++) s_add[r * 64 + lane] = a[r];
                    __builtin_amdgcn_wave_barrier();
                    float f = __uint_as_float(s);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float4 v[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] = *reinterpret_cast<const float4*>(&s_add[r * 64 + q * 4]);
#pragma unroll
                        for (int q = 0; q < 16; q++) { f = f + v[q].x; f = f + v[q].y; f = f + v[q].z; f = f + v[q].w; }
                    }
                    s = __float_as_uint(f);
                    n_raw++;
                    calm = 0;
                }
                j++;
                const bool leave = j >= lim || calm >= 6;
                fetch_block(b0 + (uint32_t)j + 2, m);
                return leave;
            };
            for (;;) { if (step(a0)) break; if (step(a1)) break; if (step(a2)) break; }
            if (j >= lim) j = 64;
            start = j;
        }
    };
    // Four windows in flight: the maps of window w + 3 are requested before window w is walked, each buffer is only ever written by its own
    // loads (no register copies between them), so a window's wait covers its own (oldest) loads only and ~3 windows' worth of work hides
    // the trip to L2 / HBM.
    // the leading blocks whose running sum is exact were summed by the scan: start behind them (window-aligned; their lanes are skipped)
    const uint32_t skip = MONO ? min(ctrl[ni].start_block[c], nd.n_blocks) : 0u;
    if (skip) s = __float_as_uint(ctrl[ni].start_sum[c]);
    const uint32_t first_b0 = skip & ~63u;
    first_start = (int)(skip - first_b0);
    // The pre-composed windows (k_wide_windows): lane L holds the record of window wb + L of the node; a window whose record has a map for the state the walk is in,
    // and whose map applies, is taken in one step; everything else goes through process() as before.
    uint32_t wb = 0;              // first window of the batch of records in registers
    int32_t wrec[14];
    auto load_records = [&](uint32_t first_window) {   // (unconditional, outside the window loop: a load the compiler cannot count would make every wait below a wait for everything)
        wb = first_window;
        const uint32_t nw = (nd.n_blocks + 63) / 64;
        const int4* src = reinterpret_cast<const int4*>(ws.win + ((size_t)c * ws.tw + window_index(nd, ni, min(first_window + (uint32_t)lane, nw - 1))) * 16);
        const int4 a = src[0], b = src[1], d = src[2], e = src[3];
        wrec[0] = a.x; wrec[1] = a.y; wrec[2] = a.z; wrec[3] = a.w; wrec[4] = b.x; wrec[5] = b.y; wrec[6] = b.z; wrec[7] = b.w;
        wrec[8] = d.x; wrec[9] = d.y; wrec[10] = d.z; wrec[11] = d.w; wrec[12] = e.x; wrec[13] = e.y;
    };
    auto take_window = [&](uint32_t b0) -> bool {
        if (!USE_WIN || (b0 == first_b0 && first_start != 0)) return false;
        const int i = (int)(b0 / 64 - wb);
        const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane(wrec[0], i);
        if (meta & (1u << 9)) return true;                      // every block of the window adds +-0 only
        const int cand = fsum::state_exp(s) - (int)(meta & 0xffu);
        if (!fsum::state_ok(s) || (cand != 0 && cand != 1) || ((meta >> 8) & 1u) != (s >> 31)) return false;
        fsum::stretch st;
        const bool up = cand != 0;
        const int32_t a0 = __builtin_amdgcn_readlane(wrec[2], i), a1 = __builtin_amdgcn_readlane(wrec[3], i), a2 = __builtin_amdgcn_readlane(wrec[4], i),
                      a3 = __builtin_amdgcn_readlane(wrec[5], i), a4 = __builtin_amdgcn_readlane(wrec[6], i), a5 = __builtin_amdgcn_readlane(wrec[7], i);
        const int32_t c0 = __builtin_amdgcn_readlane(wrec[8], i), c1 = __builtin_amdgcn_readlane(wrec[9], i), c2 = __builtin_amdgcn_readlane(wrec[10], i),
                      c3 = __builtin_amdgcn_readlane(wrec[11], i), c4 = __builtin_amdgcn_readlane(wrec[12], i), c5 = __builtin_amdgcn_readlane(wrec[13], i);
        st.d[0] = up ? c0 : a0; st.d[1] = up ? c1 : a1; st.lo[0] = up ? c2 : a2; st.lo[1] = up ? c3 : a3; st.hi[0] = up ? c4 : a4; st.hi[1] = up ? c5 : a5;
        const int32_t k0 = fsum::state_k(s);
        const bool ident = st.lo[0] == fsum::D_SAT && st.lo[1] == fsum::D_SAT && st.hi[0] == -fsum::D_SAT && st.hi[1] == -fsum::D_SAT;
        if (!ident && !fsum::applies(st, k0)) return false;
        const int32_t d = ident ? 0 : ((k0 & 1) ? st.d[1] : st.d[0]);
        if (d != 0) s = (s & 0xff800000u) | ((uint32_t)(k0 + d) & 0x7fffffu);
        n_scans++;
        return true;
    };
    for (uint32_t batch0 = first_b0; batch0 < nd.n_blocks; batch0 += 64u * 64u) {   // 64 windows per batch of records
        if (USE_WIN) load_records(batch0 / 64);
        const uint32_t batch_end = min(batch0 + 64u * 64u, nd.n_blocks);
        walk_window w0, w1, w2, w3;
        load_window(ws, nd.first_block, nd.n_blocks, batch0, lane, c, w0);
        load_window(ws, nd.first_block, nd.n_blocks, batch0 + 64, lane, c, w1);
        load_window(ws, nd.first_block, nd.n_blocks, batch0 + 128, lane, c, w2);
        for (uint32_t b0 = batch0; b0 < batch_end; b0 += 256) {
            load_window(ws, nd.first_block, nd.n_blocks, b0 + 192, lane, c, w3);
            if (!take_window(b0)) process(w0, b0);
            if (b0 + 64 >= batch_end) break;
            load_window(ws, nd.first_block, nd.n_blocks, b0 + 256, lane, c, w0);
            if (!take_window(b0 + 64)) process(w1, b0 + 64);
            if (b0 + 128 >= batch_end) break;
            load_window(ws, nd.first_block, nd.n_blocks, b0 + 320, lane, c, w1);
            if (!take_window(b0 + 128)) process(w2, b0 + 128);
            if (b0 + 192 >= batch_end) break;
            load_window(ws, nd.first_block, nd.n_blocks, b0 + 384, lane, c, w2);
            if (!take_window(b0 + 192)) process(w3, b0 + 192);
        }
    }
    if (lane == 0) {
        ctrl[ni].sums[c] = __uint_as_float(s);
        if (MODE == WM_COV) { ctrl[ni].stat_cov_scans[c] = (uint16_t)min(n_scans, 65535u); ctrl[ni].stat_cov_raw[c] = (uint16_t)min(n_raw, 65535u); }
        else { ctrl[ni].stat_scans[c] = (uint16_t)min(n_scans, 65535u); ctrl[ni].stat_raw[c] = (uint16_t)min(n_raw, 65535u); }
    }
}
template <int MODE, bool WIN>
__global__ __launch_bounds__(64) void k_wide_walk(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ w64, const uint2* pk, const uint8_t* side,
                                                  const tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb) {
    __shared__ __align__(16) float s_add[WB];
    wide_walk_body<MODE, WIN>(keys, w64, pk, side, nodes, ctrl, ws_base, tb, blockIdx.x, (int)threadIdx.x, s_add);
}

// ------------------------------------------------------------------------------------------------------------ k_wide_finish
// The serial tail of a pass: what thread 0 of k_tsvq_split / k_tsvq_root does between passes.
// One WAVE per node (`lane` of it; the covariance tail, whole-wave work with a workgroup barrier inside, is only ever called from its own 64-thread kernel).
template <int MODE>
__device__ __forceinline__ void wide_finish_body(const tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* ctrl, tsvq_root_out* root_out, const uint32_t ni, const int lane, const bool dist) {
    tsvq_wide_ctrl& c = ctrl[ni];
    if (c.done) return;
    const tsvq_wide_node& nd = nodes[ni];
    constexpr int N = 16;
    if (MODE == WM_COV) {    // compute_split_axis (enc.h:1802-1846): the whole wave
        __shared__ float s_cov[16][16];
        if (lane == 0) {
            int ch = 0;
            for (int x = 0; x < N; x++) for (int y = x; y < N; y++) s_cov[x][y] = c.sums[ch++];
            const float renorm = 1.0f / (float)nd.weight;
            for (int x = 0; x < N; x++) for (int y = x; y < N; y++) s_cov[x][y] *= renorm;
            for (int x = 0; x < N - 1; x++) for (int y = x + 1; y < N; y++) s_cov[y][x] = s_cov[x][y];
        }
        __syncthreads();
        principal_axis_wave<N>(s_cov, c.axis);
        return;
    }
    if (lane != 0) return;
    if (MODE == WM_ROOT) {   // prepare_root (enc.h:1708-1735)
        root_out += nd.out_index;
        if (c.ex_bad) { root_out->pad = 1; c.done = 2; return; }
        float o[N];
        for (int k = 0; k < N; k++) o[k] = c.sums[k];
        const float wfl = (float)c.l_w;
        const float q = dot_seq<N>(o, o) / wfl;
        root_out->var = (float)(c.dsum[0] - (double)q);
        const float inv = 1.0f / wfl;
        for (int k = 0; k < N; k++) root_out->origin[k] = o[k] * inv;
        for (int k = N; k < 16; k++) root_out->origin[k] = 0.0f;
        root_out->weight = c.l_w;
        root_out->pad = 0;
        c.done = 1;
        return;
    }
    if (c.ex_bad) { c.done = 2; return; }
    if (W_IS_PROJ(MODE, dist)) {   // prep_split (enc.h:1887-1891); the degenerate projection (:1893-1957) is left to the one-workgroup kernel
        const double lw = c.dsum[0], rw = c.dsum[1];
        if (!(lw > 0.0 && rw > 0.0)) { c.done = 2; return; }
        const float ls = (float)(1.0 / lw), rs = (float)(1.0 / rw);
        for (int k = 0; k < N; k++) { c.l_c[k] = c.sums[k] * ls; c.r_c[k] = c.sums[N + k] * rs; }
        c.prev_total = 1e+10f; c.iter = 0;
        return;
    }
    // refine_split (enc.h:2047-2073); an empty child (:2008) is left to the one-workgroup kernel
    if (c.l_w == 0 || c.r_w == 0) { c.done = 2; return; }
    float nl[N], nr[N];
    for (int k = 0; k < N; k++) { nl[k] = c.sums[k]; nr[k] = c.sums[N + k]; }
    const float lwf = (float)c.l_w, rwf = (float)c.r_w;
    const float ql = dot_seq<N>(nl, nl) / lwf, qr = dot_seq<N>(nr, nr) / rwf;
    c.l_var = (float)(c.dsum[0] - (double)ql);
    c.r_var = (float)(c.dsum[1] - (double)qr);
    const float li = 1.0f / lwf, ri = 1.0f / rwf;
    for (int k = 0; k < N; k++) { c.l_c[k] = nl[k] * li; c.r_c[k] = nr[k] * ri; }
    const float total = c.l_var + c.r_var;
    bool stop = false;
    if (total < .00001f) stop = true;
    else {
        const float rel = (c.prev_total - total) / total;
        if (rel < .00125f) stop = true;
        else c.prev_total = total;
    }
    c.iter++;
    if (stop || c.iter == 6) c.done = 1;
}
template <int MODE>
__global__ __launch_bounds__(64) void k_wide_finish(const tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* ctrl, tsvq_root_out* root_out) {
    wide_finish_body<MODE>(nodes, ctrl, root_out, blockIdx.x, (int)threadIdx.x, MODE == WM_DIST);
}

// ------------------------------------------------------------------------------------------------------------ k_wide_partition
__global__ __launch_bounds__(WB) void k_wide_partition(uint32_t* perm0, uint32_t* perm1, const uint8_t* side, const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes,
                                                       const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb, tsvq_split_out* outs) {
    wide_partition_body(perm0, perm1, side, nodes, n_nodes, ctrl, ws_base, tb, outs, blockIdx.x);
}

__global__ __launch_bounds__(256) void k_wide_iota(uint32_t n, uint32_t* __restrict__ perm0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm0[i] = i;
}

} // namespace

size_t tsvq_wide_workspace_bytes(uint32_t total_blocks) {
    const size_t tb = total_blocks;
    return align256(tb * NCH_MAX * sizeof(double)) + align256(tb * 8 * sizeof(uint64_t)) + align256(tb * NCH_MAX * 2 * 6 * sizeof(int32_t)) +
           align256(tb * sizeof(uint32_t)) + align256(tb * NCH_MAX * sizeof(uint16_t)) + align256(tb * NCH_MAX) + align256((tb + tb / 64 + 1) * NCH_MAX * 16 * sizeof(int32_t));
}

template <int MODE>
static void launch_pass(hipStream_t st, const uint32_t* keys, const uint64_t* w64, uint32_t* perm0, uint32_t* perm1, uint8_t* side, uint2* pk, const tsvq_wide_node* nodes,
                        uint32_t n_nodes, uint32_t tb, tsvq_wide_ctrl* ctrl, void* ws, tsvq_root_out* root_out, int windows_knob, bool all_chains_exact = false) {
    constexpr int NCH = mode_traits<MODE>::NCH;
    hipLaunchKernelGGL((k_wide_sums<MODE>), dim3(tb), dim3(WB), 0, st, keys, w64, perm0, perm1, side, nodes, n_nodes, ctrl, ws, tb, pk);
    hipLaunchKernelGGL((k_wide_scan<MODE>), dim3(n_nodes, NCH + (MODE == WM_COV ? 0 : 1)), dim3(256), 0, st, nodes, ctrl, ws, tb);
    if (!(all_chains_exact && MODE != WM_COV)) {   // the caller knows that every chain total of the batch stays below 2^24: the scan finishes them all
        hipLaunchKernelGGL((k_wide_stretches<MODE>), dim3(tb), dim3(ST_THREADS), 0, st, keys, w64, pk, side, nodes, n_nodes, ctrl, ws, tb);
        // The pre-composed windows pay for nodes of millions of members (8192^2 q255, rounds 1-3: the covariance walk is 0.75 ms per round without them); for the
        // 4096^2 image the extra launch costs more than the walk gains (root covariance walk 107 -> 53 us, many-workgroup rounds 5.50 -> 5.62 ms per step, one box):
        // by default on from an average of 2,048 blocks per node of the batch (bu_hip_tuning::tsvq_windows: 0 = that rule, 1 = always, 2 = never; tests run both).
        const bool windows = windows_knob == 1 || (windows_knob == 0 && tb / (n_nodes ? n_nodes : 1u) >= 2048u);
        if (windows) {
            hipLaunchKernelGGL((k_wide_windows<MODE>), dim3((tb / 64 + n_nodes) * NCH), dim3(64), 0, st, nodes, n_nodes, ctrl, ws, tb);
            hipLaunchKernelGGL((k_wide_walk<MODE, true>), dim3(n_nodes * NCH), dim3(64), 0, st, keys, w64, pk, side, nodes, ctrl, ws, tb);
        } else
            hipLaunchKernelGGL((k_wide_walk<MODE, false>), dim3(n_nodes * NCH), dim3(64), 0, st, keys, w64, pk, side, nodes, ctrl, ws, tb);
    }
    hipLaunchKernelGGL((k_wide_finish<MODE>), dim3(n_nodes), dim3(64), 0, st, nodes, ctrl, root_out);
}

hipError_t launch_tsvq_wide_root(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0, const tsvq_wide_node* d_nodes,
                                 tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_root_out* d_out, int windows, bool ctrl_cleared) {
    hipError_t e = ctrl_cleared ? hipSuccess : hipMemsetAsync(d_ctrl, 0, sizeof(tsvq_wide_ctrl), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_wide_iota, dim3((n + 255) / 256), dim3(256), 0, st, n, d_perm0);
    launch_pass<WM_ROOT>(st, d_keys, d_w64, nullptr, nullptr, nullptr, nullptr, d_nodes, 1, total_blocks, d_ctrl, d_ws, d_out, windows);
    return hipGetLastError();
}

hipError_t launch_tsvq_wide_span_roots(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, const uint32_t* d_perm0, const uint32_t* d_perm1, void* d_packed,
                                       const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_root_out* d_outs, int windows) {
    if (!n_nodes) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_ctrl, 0, (size_t)n_nodes * sizeof(tsvq_wide_ctrl), st);
    if (e != hipSuccess) return e;
    launch_pass<WM_ROOT>(st, d_keys, d_w64, const_cast<uint32_t*>(d_perm0), const_cast<uint32_t*>(d_perm1), nullptr, static_cast<uint2*>(d_packed), d_nodes, n_nodes, total_blocks, d_ctrl, d_ws, d_outs, windows);
    return hipGetLastError();
}

hipError_t launch_tsvq_wide_split(hipStream_t st, const uint32_t* d_keys, const uint64_t* d_w64, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side, void* d_packed,
                                  const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_split_out* d_outs,
                                  bool chained_covariance, bool side_chains_exact, int windows, bool ctrl_cleared) {
    if (!n_nodes) return hipSuccess;
    hipError_t e = ctrl_cleared ? hipSuccess : hipMemsetAsync(d_ctrl, 0, (size_t)n_nodes * sizeof(tsvq_wide_ctrl), st);
    if (e != hipSuccess) return e;
    if (chained_covariance) {   // raw chain sums into ctrl[].sums (three workgroups per node), then the pass's own tail: renormalisation + principal axis
        if ((e = launch_tsvq_cov_axis(st, d_keys, d_w64, d_perm0, d_perm1, d_nodes, n_nodes, d_ctrl, d_packed)) != hipSuccess) return e;
        hipLaunchKernelGGL((k_wide_finish<WM_COV>), dim3(n_nodes), dim3(64), 0, st, d_nodes, d_ctrl, nullptr);
    }
    else launch_pass<WM_COV>(st, d_keys, d_w64, d_perm0, d_perm1, d_side, static_cast<uint2*>(d_packed), d_nodes, n_nodes, total_blocks, d_ctrl, d_ws, nullptr, windows);
    launch_pass<WM_PROJ>(st, d_keys, d_w64, d_perm0, d_perm1, d_side, static_cast<uint2*>(d_packed), d_nodes, n_nodes, total_blocks, d_ctrl, d_ws, nullptr, windows, side_chains_exact);
    for (int it = 0; it < 6; it++) launch_pass<WM_DIST>(st, d_keys, d_w64, d_perm0, d_perm1, d_side, static_cast<uint2*>(d_packed), d_nodes, n_nodes, total_blocks, d_ctrl, d_ws, nullptr, windows, side_chains_exact);
    hipLaunchKernelGGL(k_wide_partition, dim3(total_blocks), dim3(WB), 0, st, d_perm0, d_perm1, d_side, d_nodes, n_nodes, d_ctrl, d_ws, total_blocks, d_outs);
    return hipGetLastError();
}

} // namespace bu
