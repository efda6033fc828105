// tsvq_wide6_kernels.hip -- the ENDPOINT codebook builder's large nodes (row a8, 6-float training vectors) over the whole chip: the many-workgroup split of
// tsvq_wide_kernels.hip for rows that are not small integers.
//
// The endpoint tree of a 4096^2 image starts with one node of ~35,000 distinct vectors, then two, four, eight: in the one-workgroup kernel (tsvq_kernels.hip) every pass over
// such a node is one dependent float add per member (~4 ns), five to six passes per split, on 1 - 8 of 256 CUs -- 1.5 ms of a 19 ms step. Here a pass is the same five
// kernels as on the selector side (block sums -> binade prediction -> parity maps per block -> one wave per chain walks the maps -> the serial tail), with what is
// different for these rows:
//   * the addends v_k * w are not integers, so there is no "total below 2^24 = exact in any order" shortcut and no exact prefix: every chain is walked from block 0;
//   * the reference's DOUBLE accumulators of the two-means passes (l_ttsum / r_ttsum += w * |v|^2, encoder/basisu_enc.h:1996-2006) add floats that are not integers either,
//     so they cannot be integer reductions. A double has 29 bits more than the float it adds: the add is exact unless bits fall off the low end, which only happens when
//     the running sum has outgrown the smallest addend so far by more than 2^29. One wave per node (beside the chains' walks, in the walk kernel) therefore walks the blocks of the node in order with the block's sum
//     (exact when the test below holds) and the exponent of its smallest addend: while [lowest set bit that can be in the sum] >= ulp(sum after the block) every add of
//     the block is exact and the block is taken in one step; a block for which the test fails is added member by member, in order, with real double adds (tt_walk; the test itself is csrc/tt_exact.h,
//     shared with a host build the CPU tests run). It errs on the safe side only (a failing block costs ~1 us, never a wrong bit);
//   * l_weight / r_weight of the projection pass add integer-valued floats far below 2^53: integer sums, as on the selector side;
//   * the covariance pass stays CHAINED (k_tsvq_cov_axis6, tsvq_kernels.hip: one workgroup per node, 21 chains): its signed chains change binade in most blocks of
//     nodes this size, and a block that has to be added member by member costs twice what the plain chain costs (measured: 305 us through the maps for the 35,502-member
//     root of the bench image, 135 us chained). That kernel also lays out the addends of the passes that follow in list order (va = v_k * w, tta = w * |v|^2), so
//     the per-block kernels here read coalesced arrays; only the classification gathers rows through the member list;
//   * the side passes have 12 chains instead of 32;
//   * the root record of the tree (prepare_root) goes through the same passes with every vector on the "left" (W6_ROOT).
// Anything out of the ordinary (an empty child, a degenerate projection, non-finite data) hands the node back (ok == 2) and the one-workgroup kernel splits it.
// Parity: tests/test_gpu_tsvq.py runs trees through this path (BU_TSVQ_WIDE6_MIN lowered) against the host builder and the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tsvq_kernels.h"
#include "tsvq_common.h"
#include "tsvq_bufs.h"
#include "fsum_scan.h"
#include "tt_exact.h"

namespace bu {

namespace {

#include "tsvq_wide_common.h"   // (inside the unnamed namespace: internal linkage in each of the two translation units that use it)

constexpr int D6 = 6;
enum { W6_ROOT = 0, W6_COV = 1, W6_PROJ = 2, W6_DIST = 3 };   // W6_ROOT: prepare_root of the whole training set (every vector "left", list order = index order)
constexpr int NCH6 = 12;   // side passes: chain = side * 6 + component

struct member6 { float v[D6]; float wf; uint64_t w; bool valid; };
__device__ __forceinline__ member6 fetch6(const float* __restrict__ rows, const uint64_t* __restrict__ w64, const uint32_t* __restrict__ members, uint32_t pos, uint32_t count) {
    member6 m;
    m.valid = pos < count;
    const uint32_t p = m.valid ? pos : count - 1;
    const uint32_t mi = members ? members[p] : p;
    const float* r = rows + (size_t)mi * D6;
#pragma unroll
    for (int k = 0; k < D6; k++) m.v[k] = r[k];
    m.w = w64[mi]; m.wf = (float)m.w;
    return m;
}

// the addend of the reference's ttsum for one member: (double)(w * |v|^2), the product in float (enc.h:1998: `l_ttsum += weight * v.dot(v)` with float operands)
__device__ __forceinline__ float tt_addend(const member6& m) { return m.wf * dot_seq<D6>(m.v, m.v); }

struct tiles6 {
    float fa[D6][WROW];
    uint8_t sd[WB];
};

// ------------------------------------------------------------------------------------------------------------ sums
// per block: classify + store the side, the block sums of every chain in double (a prediction aid), the integer totals, and for the two-means passes the block's
// ttsum contributions per side with the exponent of their smallest non-zero addend. bex[blk]: 0 lw, 1 rw, 2 left count, 3 / 4 two-means passes: bits of the left / right
// ttsum block sums; projection pass: the integer sums of (float)w left / right, 5 the smallest addend's exponent field left | right << 32 (0xffff: no non-zero addend; 0: a denormal or non-finite one)
template <int MODE>
__global__ __launch_bounds__(WB) void k6_sums(const float* __restrict__ rows, const uint64_t* __restrict__ w64, const uint32_t* perm0, const uint32_t* perm1, uint8_t* side,
                                              const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes, const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb,
                                              float* va_out, double* tta_out, uint32_t n) {
    constexpr int NCH = NCH6;
    __shared__ tiles6 T;
    __shared__ float s_origin[D6], s_axis[D6], s_lc[D6], s_rc[D6];
    __shared__ double s_part[8][12];
    __shared__ uint8_t s_nz[8][12];   // some addend of the slice is not +-0
    __shared__ uint64_t s_red[4][5];
    __shared__ double s_tt[4][2];
    __shared__ uint32_t s_em[4][2];
    const wide_ws ws = carve(ws_base, tb);
    const int tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const uint32_t ni = find_node(nodes, n_nodes, blk);
    const tsvq_wide_ctrl& ct = ctrl[ni];
    if (ct.done) return;
    const tsvq_wide_node& nd = nodes[ni];
    if (tid < D6) { s_origin[tid] = nd.origin[tid]; s_axis[tid] = ct.axis[tid]; s_lc[tid] = ct.l_c[tid]; s_rc[tid] = ct.r_c[tid]; }
    __syncthreads();
    const uint32_t pos = (blk - nd.first_block) * WB + (uint32_t)tid;
    const uint32_t* members = MODE == W6_ROOT ? nullptr : tsvq_list(perm0, perm1, nd.buf) + nd.start;
    const member6 m = fetch6(rows, w64, members, pos, nd.count);
    // which child (enc.h:1870-1871 projection sign; enc.h:1991 distances in double, difference form)
    bool right;
    if (MODE == W6_ROOT) {   // what the covariance pass lays out for the splits, for the root's own passes
        right = false;
        if (m.valid) {
#pragma unroll
            for (int k = 0; k < D6; k++) va_out[(size_t)k * n + pos] = m.v[k] * m.wf;
            tta_out[pos] = (double)tt_addend(m);
        }
    } else if (MODE == W6_DIST) {
        double dl = 0, dr = 0;
#pragma unroll
        for (int k = 0; k < D6; k++) {
            const double a = (double)s_lc[k] - (double)m.v[k], b = (double)s_rc[k] - (double)m.v[k];
            dl += a * a; dr += b * b;
        }
        right = dl >= dr;
    } else {
        float dd[D6];
#pragma unroll
        for (int k = 0; k < D6; k++) dd[k] = m.v[k] - s_origin[k];
        right = (double)dot_seq<D6>(dd, s_axis) >= 0.0;
    }
#pragma unroll
    for (int k = 0; k < D6; k++) T.fa[k][tid] = m.valid ? m.v[k] * m.wf : 0.0f;
    T.sd[tid] = m.valid ? (right ? 1 : 0) : 2;
    if (m.valid) side[nd.start + pos] = right ? 1 : 0;
    uint64_t red[5] = {0, 0, 0, 0, 0};   // weight left / right (L.weight += w: the integer), left count, and -- projection pass -- l_weight / r_weight, which add (float)w
    double tt[2] = {0.0, 0.0};
    uint32_t em[2] = {0xffffu, 0xffffu};
    if (m.valid) {
        if (right) red[1] = m.w; else { red[0] = m.w; red[2] = 1; }
        if (MODE == W6_PROJ) red[right ? 4 : 3] = (uint64_t)m.wf;   // an integer-valued float below 2^64: exact
        if (MODE == W6_DIST || MODE == W6_ROOT) {
            const float af = tt_addend(m);
            const uint32_t bits = __float_as_uint(af);
            tt[right ? 1 : 0] = (double)af;
            em[right ? 1 : 0] = tt::addend_exp(bits);
        }
    }
    __syncthreads();
    for (int item = tid; item < NCH * 8; item += WB) {   // 12 chains x 8 slices of 32 members
        const int c = item % NCH, sl = item / NCH;
        const int k = c % D6; const uint8_t want = (uint8_t)(c / D6);
        double s = 0;
        uint32_t nz = 0;
        for (int j = sl * 32; j < sl * 32 + 32; j++) {
            const float a = (T.sd[j] == want) ? T.fa[k][j] : 0.0f;
            s += (double)a;
            nz |= __float_as_uint(a) << 1;   // anything but +-0
        }
        s_part[sl][c] = s;
        s_nz[sl][c] = nz ? 1 : 0;
    }
#pragma unroll
    for (int i = 0; i < (MODE == W6_PROJ ? 5 : 3); i++) red[i] = wave_sum_u64(red[i]);   // totals in lane 63
    if (MODE == W6_DIST || MODE == W6_ROOT) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            tt[i] = wave_prefix_f64(tt[i]);   // (a tree of double adds: exact whenever the block passes tt::block_is_exact, which is all that is asked of it)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) em[i] = min(em[i], (uint32_t)__shfl_xor((int)em[i], o, 64));
        }
    }
    if ((tid & 63) == 63) {
        for (int i = 0; i < 5; i++) s_red[tid >> 6][i] = red[i];
        for (int i = 0; i < 2; i++) { s_tt[tid >> 6][i] = tt[i]; s_em[tid >> 6][i] = em[i]; }
    }
    __syncthreads();
    if (tid < NCH) {
        double s = 0;
        uint32_t nz = 0;
        for (int i = 0; i < 8; i++) { s += s_part[i][tid]; nz |= s_nz[i][tid]; }
        ws.bsum[ws.at(tid, blk)] = s;
        ws.bzero[ws.at(tid, blk)] = nz ? 0 : 1;   // the block leaves the running sum alone only if EVERY addend is +-0 (signed rows can cancel to a zero sum)
    }
    if (tid >= 64 && tid < (MODE == W6_PROJ ? 69 : 67)) ws.bex[(size_t)blk * 8 + (tid - 64)] = s_red[0][tid - 64] + s_red[1][tid - 64] + s_red[2][tid - 64] + s_red[3][tid - 64];
    if ((MODE == W6_DIST || MODE == W6_ROOT) && tid >= 128 && tid < 130) {
        const int i = tid - 128;
        const double s = (s_tt[0][i] + s_tt[1][i]) + (s_tt[2][i] + s_tt[3][i]);
        ws.bex[(size_t)blk * 8 + 3 + i] = (uint64_t)__double_as_longlong(s);
    }
    if ((MODE == W6_DIST || MODE == W6_ROOT) && tid == 192) {
        const uint32_t e0 = min(min(s_em[0][0], s_em[1][0]), min(s_em[2][0], s_em[3][0])), e1 = min(min(s_em[0][1], s_em[1][1]), min(s_em[2][1], s_em[3][1]));
        ws.bex[(size_t)blk * 8 + 5] = (uint64_t)e0 | ((uint64_t)e1 << 32);
    }
}

// ------------------------------------------------------------------------------------------------------------ scan
// grid (node, y): y < NCW: four chains, one wave each: the binade each chain's running sum will be in at every block start (a prediction);
//                 y == NCW: the integer totals and the left-count prefix
template <int MODE>
__global__ __launch_bounds__(256) void k6_scan(const tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb) {
    constexpr int NCH = NCH6;
    constexpr int NCW = (NCH + 3) / 4;
    __shared__ uint64_t s_tot[4][5];
    __shared__ uint32_t s_wl[4];
    const wide_ws ws = carve(ws_base, tb);
    const uint32_t ni = blockIdx.x, y = blockIdx.y;
    if (ctrl[ni].done) return;
    const tsvq_wide_node nd = nodes[ni];
    const int tid = threadIdx.x, lane = tid & 63;
    if ((int)y < NCW) {
        const int c = (int)y * 4 + (tid >> 6);
        if (c >= NCH) return;
        double P = 0;   // sum of the blocks before the current 64
        for (uint32_t b0 = 0; b0 < nd.n_blocks; b0 += 64) {
            const uint32_t b = b0 + (uint32_t)lane;
            const bool have = b < nd.n_blocks;
            const size_t at = ws.at(c, nd.first_block + (have ? b : nd.n_blocks - 1));
            const double v = have ? ws.bsum[at] : 0.0;
            const uint32_t bz = have ? (uint32_t)ws.bzero[at] : 1u;
            const double incl = wave_prefix_f64(v);
            const double Ps = P + (incl - v);
            if (have) {
                uint16_t ep;
                if (bz) ep = EP_ZERO;
                else {
                    // the running float sum at this block's start is within (members so far) half-ulps of Ps (the block sums carry a rounding of their own here: a few
                    // parts in 2^53, far inside the margin): the binade of the lower end, and whether the upper end is in the same one (then one map is enough)
                    const double eps = (double)((uint64_t)b * WB + 2) * 5.9604644775390625e-08;
                    const double lo = fabs(Ps) * (1.0 - eps), hi = fabs(Ps) * (1.0 + eps);
                    float lf = (float)(lo > 0.0 ? lo : 0.0), hf = (float)hi;
                    if ((double)lf > lo) lf = __uint_as_float(__float_as_uint(lf) - 1u);
                    if ((double)hf < hi) hf = __uint_as_float(__float_as_uint(hf) + 1u);
                    const uint32_t e = (__float_as_uint(lf) >> 23) & 0xffu, eh = (__float_as_uint(hf) >> 23) & 0xffu;
                    ep = (e >= 1u && e <= 252u) ? (uint16_t)(e | (Ps < 0.0 ? 0x100u : 0u) | (eh == e ? EP_SINGLE : 0u)) : EP_NONE;
                }
                ws.epred[at] = ep;
            }
            P += __shfl(incl, 63, 64);
        }
        if (lane == 0) { ctrl[ni].exact[c] = 0u; ctrl[ni].start_block[c] = 0u; ctrl[ni].start_sum[c] = 0.0f; }
        return;
    }
    if ((int)y == NCW) {   // totals of the integer accumulators and the left-count prefix (block order)
        constexpr int NT = MODE == W6_PROJ ? 5 : 3;
        uint64_t tot[5] = {0, 0, 0, 0, 0};
        const uint32_t perb = (nd.n_blocks + 255) / 256;
        const uint32_t q0 = min((uint32_t)tid * perb, nd.n_blocks), q1 = min(q0 + perb, nd.n_blocks);
        for (uint32_t b = q0; b < q1; b++)
#pragma unroll
            for (int i = 0; i < NT; i++) tot[i] += ws.bex[(size_t)(nd.first_block + b) * 8 + i];
        const uint32_t my_left = (uint32_t)tot[2];
        uint32_t incl = my_left;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += t; }
#pragma unroll
        for (int i = 0; i < NT; i++) {
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
            uint64_t t[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < NT; i++) for (int w = 0; w < 4; w++) t[i] += s_tot[w][i];
            tsvq_wide_ctrl& ct = ctrl[ni];
            ct.l_w = t[0]; ct.r_w = t[1]; ct.l_n = (uint32_t)t[2]; ct.r_n = nd.count - (uint32_t)t[2];
            if (MODE == W6_PROJ) { ct.dsum[0] = (double)t[3]; ct.dsum[1] = (double)t[4]; if (t[3] >= (1ull << 53) || t[4] >= (1ull << 53)) ct.ex_bad = 1u; }   // l_weight / r_weight (doubles in the reference): exact below 2^53
        }
        return;
    }
}

// ------------------------------------------------------------------------------------------------------------ stretches
// per block, chain and up to two candidate binades: the block's addends folded into one parity map (fsum_scan.h)
__global__ __launch_bounds__(WB) void k6_stretches(const float* __restrict__ va, uint32_t n, const uint8_t* side, const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes,
                                                   const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb) {
    constexpr int NCH = NCH6;
    constexpr int Q = 8;   // member slices per chain; an item = (chain, slice), both candidate binades
    constexpr int ITEMS = NCH * Q;
    constexpr int SL = WB / Q;
    __shared__ tiles6 T;
    __shared__ int32_t s_st[ITEMS][2][6];
    const wide_ws ws = carve(ws_base, tb);
    const int tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const uint32_t ni = find_node(nodes, n_nodes, blk);
    if (ctrl[ni].done) return;
    const tsvq_wide_node& nd = nodes[ni];
    const uint32_t pos = (blk - nd.first_block) * WB + (uint32_t)tid;
    const bool valid = pos < nd.count;
    const size_t at = (size_t)nd.start + (valid ? pos : nd.count - 1);
#pragma unroll
    for (int k = 0; k < D6; k++) { const float a = va[(size_t)k * n + at]; T.fa[k][tid] = valid ? a : 0.0f; }
    T.sd[tid] = valid ? side[at] : 2;
    __syncthreads();
    for (int item = tid; item < ITEMS; item += WB) {
        const int c = item % NCH, q = item / NCH;
        const uint16_t ep = ws.epred[ws.at(c, blk)];
        fsum::stretch st0 = fsum::identity(), st1 = fsum::identity();
        if (ep != EP_NONE && ep != EP_ZERO) {
            const int E = (int)(ep & 0xffu);
            const bool neg = (ep & 0x100u) != 0, two = (ep & EP_SINGLE) == 0;
            const int x = c % D6;
            const uint8_t want = (uint8_t)(c / D6);
            bool bad0 = false, bad1 = false;
            for (int j = q * SL; j < q * SL + SL; j++) {
                if (T.sd[j] != want) continue;
                const uint32_t bits = __float_as_uint(T.fa[x][j]);
                if ((bits << 1) == 0) continue;
                const fsum::parts pr = fsum::split(bits, neg);
                fsum::push_fast(st0, fsum::decode_fast(pr, E, bad0));
                if (two) fsum::push_fast(st1, fsum::decode_fast(pr, E + 1, bad1));
            }
            if (bad0) fsum::poison(st0);
            if (bad1 || E + 1 > 253) fsum::poison(st1);
        }
        st_store(s_st[item][0], st0); st_store(s_st[item][1], st1);
    }
    __syncthreads();
    for (int item = tid; item < NCH * 2; item += WB) {   // slices in member order
        const int c = item % NCH, cand = item / NCH;
        fsum::stretch acc = st_load(s_st[c][cand]);
#pragma unroll
        for (int q = 1; q < Q; q++) acc = fsum::compose(acc, st_load(s_st[c + q * NCH][cand]));
        st_store(ws.summ + (ws.at(c, blk) * 2 + (size_t)cand) * 6, acc);
    }
}

// ---- tt_walk: the two double accumulators of a two-means pass for one node, on one wave (it runs beside the chains' walks, as one more "chain" of k6_walk). Lane 0
// carries l_ttsum, lane 1 r_ttsum (the other lanes shadow lane 0 and store nothing). See the file header for the test.
struct tt_lds { double a[WB]; double bs[2][64]; uint64_t em[64]; uint8_t sd[WB]; };
__device__ __forceinline__ void tt_walk(const double* __restrict__ tta, const uint8_t* side, const tsvq_wide_node& nd, tsvq_wide_ctrl* ctrl, const uint32_t ni, const wide_ws& ws,
                                        const int lane, tt_lds& sh) {
    double* s_a = sh.a; uint8_t* s_sd = sh.sd; double (*s_bs)[64] = sh.bs; uint64_t* s_em = sh.em;
    const int mine = lane == 1 ? 1 : 0;
    double s = 0.0;
    int L = tt::L_FREE;   // exponent (power of two) of the lowest bit that may be set in s
    bool bad = false;
    for (uint32_t b0 = 0; b0 < nd.n_blocks; b0 += 64) {
        // the records of the next 64 blocks, one per lane, into LDS: the loop below is a chain through s, and a trip to memory per block (0.3 us) was all of its time
        {
            const uint32_t bb = min(b0 + (uint32_t)lane, nd.n_blocks - 1);
            const uint64_t* bx = ws.bex + (size_t)(nd.first_block + bb) * 8;
            s_bs[0][lane] = __longlong_as_double((long long)bx[3]); s_bs[1][lane] = __longlong_as_double((long long)bx[4]);
            s_em[lane] = bx[5];
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t lim = min(64u, nd.n_blocks - b0);
        for (uint32_t i = 0; i < lim; i++) {
            const uint32_t b = b0 + i;
            const uint32_t summary = (uint32_t)(s_em[i] >> (mine ? 32 : 0));
            double s_end;
            const bool safe = tt::block_is_exact(s, s_bs[mine][i], L, summary, &s_end);   // tt_exact.h
            if (__ballot(lane < 2 && !safe) == 0ull) { s = s_end; L = min(L, tt::block_low(summary)); continue; }
            // member by member, in list order, with the double adds the reference does (the addends were laid out by the covariance pass)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t pos = b * WB + (uint32_t)(r * 64 + lane);
                const bool valid = pos < nd.count;
                s_a[r * 64 + lane] = valid ? tta[nd.start + pos] : 0.0;
                s_sd[r * 64 + lane] = valid ? side[nd.start + pos] : (uint8_t)2;
            }
            __builtin_amdgcn_wave_barrier();
            for (int j = 0; j < WB; j++) s = s + ((int)s_sd[j] == mine ? s_a[j] : 0.0);   // + 0.0 leaves a non-negative sum as it is
            __builtin_amdgcn_wave_barrier();
            if (((uint32_t)((uint64_t)__double_as_longlong(s) >> 52) & 0x7ffu) == 0x7ffu) bad = true;
            L = tt::low_bit(s);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < 2) ctrl[ni].dsum[lane] = s;
    if (__ballot(lane < 2 && bad) != 0ull && lane == 0) ctrl[ni].ex_bad = 1u;
}

// ------------------------------------------------------------------------------------------------------------ walk
// One wave per (node, chain): the walk of tsvq_wide_kernels.hip (64 blocks' maps composed per wave scan, the longest valid prefix applied, a block whose map does
// not apply added member by member out of LDS), from block 0 with a running sum of +0.
// (two-means passes: the workgroups behind the chains' -- one more per node -- run tt_walk)
template <bool WITH_TT>
__global__ __launch_bounds__(64) void k6_walk(const float* __restrict__ va, const double* __restrict__ tta, uint32_t n, const uint8_t* side, const tsvq_wide_node* __restrict__ nodes,
                                              uint32_t n_nodes, tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb) {
    constexpr int NCH = NCH6;
    __shared__ __align__(16) union { float add[WB]; tt_lds tt; } sh;
    float* s_add = sh.add;
    const wide_ws ws = carve(ws_base, tb);
    const int lane = (int)threadIdx.x;
    if (WITH_TT && blockIdx.x >= n_nodes * NCH) {
        const uint32_t nt = blockIdx.x - n_nodes * NCH;
        if (!ctrl[nt].done) tt_walk(tta, side, nodes[nt], ctrl, nt, ws, lane, sh.tt);
        return;
    }
    const uint32_t ni = blockIdx.x / NCH;
    const int c = (int)(blockIdx.x % NCH);
    if (ctrl[ni].done) return;
    const tsvq_wide_node& nd = nodes[ni];
    const float* chain_va = va + (size_t)(c % D6) * n + nd.start;   // this chain's addends in list order; the side array says whose they are
    const uint8_t* node_side = side + nd.start;
    const bool chain_right = (c / D6) != 0;
    uint32_t s = 0;          // +0.0f

    auto pick = [&](const walk_window& cur, fsum::stretch& st) -> int {   // 0 identity, 1 map for s's binade in st, 2 no map for this state
        st = fsum::identity();
        if (cur.ep == EP_ZERO) return 0;
        const int cand = fsum::state_exp(s) - (int)(cur.ep & 0xffu);
        const bool usable = cur.ep != EP_NONE && fsum::state_ok(s) && (cand == 0 || (cand == 1 && !(cur.ep & EP_SINGLE))) && (((cur.ep >> 8) & 1u) == (s >> 31));
        if (!usable) return 2;
        const bool up = cand != 0;   // selects, not cur.m[cand]: a dynamic index would put the window into scratch
        st.d[0] = up ? cur.m[1][0] : cur.m[0][0]; st.d[1] = up ? cur.m[1][1] : cur.m[0][1];
        st.lo[0] = up ? cur.m[1][2] : cur.m[0][2]; st.lo[1] = up ? cur.m[1][3] : cur.m[0][3];
        st.hi[0] = up ? cur.m[1][4] : cur.m[0][4]; st.hi[1] = up ? cur.m[1][5] : cur.m[0][5];
        return 1;
    };
    // the members of block `blk` (of the node) that this lane stages: positions blk * 256 + r * 64 + lane. Loading and turning them into the chain's addends are
    // separate steps so that the loads stay in flight until the addends are needed (three rotating register sets below: a wait before a set's use covers that set only)
    struct staged { float v[4]; uint8_t sd[4]; };
    auto fetch_block = [&](uint32_t blk, staged& m) {
        const uint32_t p0 = min(blk, nd.n_blocks - 1) * WB;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t p = min(p0 + (uint32_t)(r * 64 + lane), nd.count - 1);
            m.v[r] = chain_va[p]; m.sd[r] = node_side[p];
        }
    };
    auto addends = [&](uint32_t blk, const staged& m, float (&a)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool valid = blk * WB + (uint32_t)(r * 64 + lane) < nd.count;
            const float v = (m.sd[r] != 0) == chain_right ? m.v[r] : 0.0f;
            a[r] = valid ? v : -0.0f;   // past the node's end: leaves every sum as it is
        }
    };
    auto raw_block = [&](const float (&a)[4]) {   // the block's 256 addends (staged by `addends`) added to s one after the other
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
        __builtin_amdgcn_wave_barrier();
    };

    uint32_t n_scans = 0, n_raw = 0;
    // one window (64 blocks starting at b0) against the running sum
    auto process = [&](const walk_window& cur, uint32_t b0) {
        int start = 0;       // lanes below are done
        while (start < 64) {
            fsum::stretch st;
            n_scans++;
            int kind = pick(cur, st);
            if (lane < start) { kind = 0; st = fsum::identity(); }
            const int32_t k0 = fsum::state_k(s);
            const bool sok = fsum::state_ok(s);
            bool ok; int32_t d_sel;
            {
                mono mm;
                mm.d[0] = kind == 2 ? fsum::D_SAT : st.d[0]; mm.d[1] = kind == 2 ? fsum::D_SAT : st.d[1];
                wave_scan(mm);
                d_sel = (k0 & 1) ? mm.d[1] : mm.d[0];
                ok = (mm.d[0] == 0 && mm.d[1] == 0) || (sok && k0 + d_sel < fsum::K_HI);   // nothing added so far (zero blocks only): fine for any state, +0 included
            }
            const uint64_t good = __ballot(ok);
            const int first_fail = (~good == 0ull) ? 64 : __ffsll((long long)~good) - 1;
            if (first_fail > 0) {
                const int32_t d = __builtin_amdgcn_readlane(d_sel, first_fail - 1);
                if (d != 0) s = (s & 0xff800000u) | ((uint32_t)(k0 + d) & 0x7fffffu);
            }
            start = first_fail;
            if (first_fail == 64 || b0 + (uint32_t)first_fail >= nd.n_blocks) return;
            // a stretch of trouble (the sum changes binade inside blocks, or addends as large as the sum): block by block, each through its own map or member by
            // member, until six blocks in a row went through their maps
            const int lim = (int)min(64u, nd.n_blocks - b0);
            int j = first_fail, calm = 0;
            staged a0, a1, a2;   // the members of the next three blocks are always on their way, needed or not
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
                    float a[4];
                    addends(b0 + (uint32_t)j, m, a);
                    raw_block(a);
                    n_raw++;
                    calm = 0;
                }
                j++;
                const bool leave = j >= lim || calm >= 6;
                fetch_block(b0 + (uint32_t)j + 2, m);
                return leave;
            };
            for (;;) { if (step(a0)) break; if (step(a1)) break; if (step(a2)) break; }
            start = j >= lim ? 64 : j;
        }
    };
    // the maps of the next window are requested before the current one is walked (each buffer is only ever written by its own loads)
    walk_window w0, w1;
    load_window(ws, nd.first_block, nd.n_blocks, 0, lane, c, w0);
    for (uint32_t b0 = 0; b0 < nd.n_blocks; b0 += 128) {
        load_window(ws, nd.first_block, nd.n_blocks, b0 + 64, lane, c, w1);
        process(w0, b0);
        if (b0 + 64 >= nd.n_blocks) break;
        load_window(ws, nd.first_block, nd.n_blocks, b0 + 128, lane, c, w0);
        process(w1, b0 + 64);
    }
    if (lane == 0) {
        ctrl[ni].sums[c] = __uint_as_float(s);
        ctrl[ni].stat_scans[c] = (uint16_t)min(n_scans, 65535u); ctrl[ni].stat_raw[c] = (uint16_t)min(n_raw, 65535u);
    }
}

// ------------------------------------------------------------------------------------------------------------ finish
// the serial tail of a pass; one wave per node
template <int MODE>
__global__ __launch_bounds__(64) void k6_finish(const tsvq_wide_node* __restrict__ nodes, tsvq_wide_ctrl* ctrl, tsvq_root_out* root_out) {
    const uint32_t ni = blockIdx.x;
    const int lane = (int)threadIdx.x;
    tsvq_wide_ctrl& c = ctrl[ni];
    if (c.done) return;
    const tsvq_wide_node& nd = nodes[ni];
    constexpr int N = D6;
    if (MODE == W6_COV) {    // compute_split_axis (enc.h:1802-1846): the whole wave
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
    if (MODE == W6_ROOT) {   // prepare_root (enc.h:1708-1735)
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
    if (MODE == W6_PROJ) {   // prep_split (enc.h:1887-1891); the degenerate projection (:1893-1957) is left to the one-workgroup kernel
        const double lw = c.dsum[0], rw = c.dsum[1];   // sums of the integer-valued floats (float)w, below 2^53: what the reference's double accumulators hold
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

__global__ __launch_bounds__(WB) void k6_partition(uint32_t* perm0, uint32_t* perm1, const uint8_t* side, const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes,
                                                   const tsvq_wide_ctrl* ctrl, void* ws_base, uint32_t tb, tsvq_split_out* outs) {
    wide_partition_body(perm0, perm1, side, nodes, n_nodes, ctrl, ws_base, tb, outs, blockIdx.x);
}

template <int MODE>
void launch_pass6(hipStream_t st, const float* rows, const uint64_t* w64, uint32_t n, uint32_t* perm0, uint32_t* perm1, uint8_t* side, const tsvq_wide_node* nodes, uint32_t n_nodes,
                  uint32_t tb, tsvq_wide_ctrl* ctrl, void* ws, float* va, double* tta, tsvq_root_out* root_out = nullptr) {
    constexpr bool TT = MODE == W6_DIST || MODE == W6_ROOT;
    hipLaunchKernelGGL((k6_sums<MODE>), dim3(tb), dim3(WB), 0, st, rows, w64, perm0, perm1, side, nodes, n_nodes, ctrl, ws, tb, va, tta, n);
    hipLaunchKernelGGL((k6_scan<MODE>), dim3(n_nodes, (NCH6 + 3) / 4 + 1), dim3(256), 0, st, nodes, ctrl, ws, tb);
    hipLaunchKernelGGL(k6_stretches, dim3(tb), dim3(WB), 0, st, va, n, side, nodes, n_nodes, ctrl, ws, tb);
    hipLaunchKernelGGL((k6_walk<TT>), dim3(n_nodes * (NCH6 + (TT ? 1 : 0))), dim3(64), 0, st, va, tta, n, side, nodes, n_nodes, ctrl, ws, tb);
    hipLaunchKernelGGL((k6_finish<MODE>), dim3(n_nodes), dim3(64), 0, st, nodes, ctrl, root_out);
}

__global__ __launch_bounds__(256) void k6_iota(uint32_t n, uint32_t* __restrict__ perm0) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm0[i] = i;
}

} // namespace

hipError_t launch_tsvq_wide6_split(hipStream_t st, const float* d_rows, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0, uint32_t* d_perm1, uint8_t* d_side,
                                   const tsvq_wide_node* d_nodes, uint32_t n_nodes, tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_split_out* d_outs,
                                   float* d_va, double* d_tta, bool ctrl_cleared) {
    if (!n_nodes) return hipSuccess;
    hipError_t e = ctrl_cleared ? hipSuccess : hipMemsetAsync(d_ctrl, 0, (size_t)n_nodes * sizeof(tsvq_wide_ctrl), st);
    if (e != hipSuccess) return e;
    // covariance: chained sums, one workgroup per node (+ the list-order copies of the per-member addends), then the principal axis
    if ((e = launch_tsvq_cov_axis6(st, d_rows, d_w64, d_perm0, d_perm1, d_nodes, n_nodes, d_ctrl, d_va, d_tta, n)) != hipSuccess) return e;
    hipLaunchKernelGGL((k6_finish<W6_COV>), dim3(n_nodes), dim3(64), 0, st, d_nodes, d_ctrl, static_cast<tsvq_root_out*>(nullptr));
    launch_pass6<W6_PROJ>(st, d_rows, d_w64, n, d_perm0, d_perm1, d_side, d_nodes, n_nodes, total_blocks, d_ctrl, d_ws, d_va, d_tta);
    for (int it = 0; it < 6; it++) launch_pass6<W6_DIST>(st, d_rows, d_w64, n, d_perm0, d_perm1, d_side, d_nodes, n_nodes, total_blocks, d_ctrl, d_ws, d_va, d_tta);
    hipLaunchKernelGGL(k6_partition, dim3(total_blocks), dim3(WB), 0, st, d_perm0, d_perm1, d_side, d_nodes, n_nodes, d_ctrl, d_ws, total_blocks, d_outs);
    return hipGetLastError();
}

// prepare_root of the whole training set (one node: all n vectors, in index order) through the same passes; d_perm0 becomes 0..n-1
hipError_t launch_tsvq_wide6_root(hipStream_t st, const float* d_rows, const uint64_t* d_w64, uint32_t n, uint32_t* d_perm0, uint8_t* d_side, const tsvq_wide_node* d_nodes,
                                  tsvq_wide_ctrl* d_ctrl, void* d_ws, uint32_t total_blocks, tsvq_root_out* d_out, float* d_va, double* d_tta, bool ctrl_cleared) {
    hipError_t e = ctrl_cleared ? hipSuccess : hipMemsetAsync(d_ctrl, 0, sizeof(tsvq_wide_ctrl), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k6_iota, dim3((n + 255) / 256), dim3(256), 0, st, n, d_perm0);
    launch_pass6<W6_ROOT>(st, d_rows, d_w64, n, d_perm0, nullptr, d_side, d_nodes, 1, total_blocks, d_ctrl, d_ws, d_va, d_tta, d_out);
    return hipGetLastError();
}

} // namespace bu
