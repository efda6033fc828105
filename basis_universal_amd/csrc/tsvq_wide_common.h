// tsvq_wide_common.h -- what the two many-workgroup TSVQ paths share: tsvq_wide_kernels.hip (packed selector rows) and tsvq_wide6_kernels.hip (6-float endpoint rows).
// The workspace layout, the binade prediction codes, the parity-map records and their wave scans, the walk's block window and the partition pass.
// Included INSIDE `namespace bu { namespace { ... } }` of each of the two translation units (device code with internal linkage); needs tsvq_kernels.h, tsvq_common.h
// and fsum_scan.h in front of it.

constexpr int WB = 256;            // members per block = threads per workgroup of the per-block kernels
constexpr int WROW = WB + 1;       // LDS row stride (floats)
constexpr int NCH_MAX = TSVQ_WIDE_MAX_CHAINS;

struct wide_ws {                   // views into the workspace; per-(chain, block) arrays are CHAIN-major: [chain][TB blocks]
    double* bsum;                  // [NCH_MAX][TB]  sum of the block's addends per chain
    uint8_t* bzero;                // [NCH_MAX][TB]  1: every addend of the block is +-0 for that chain
    uint64_t* bex;                 // [TB][8]        lw, rw, ln, ex0.lo, ex0.hi, ex1.lo, ex1.hi, bad
    uint16_t* epred;               // [NCH_MAX][TB]  predicted exponent | sign << 8 | single << 9; EP_NONE / EP_ZERO
    uint32_t* lpre;                // [TB]           left members before the block (within its node)
    int32_t* summ;                 // [NCH_MAX][TB][2][6]
    int32_t* win;                  // [NCH_MAX][TW][16]  the maps of 64 consecutive blocks of a node composed into one (k_wide_windows): meta, candidate 0, candidate 1
    uint32_t tb, tw;               // tw = tb + tb / 64 + 1 >= the windows of any batch of tb blocks (window index: first_block / 64 + node index + window of the node)
    __host__ __device__ size_t at(int c, uint32_t blk) const { return (size_t)c * tb + blk; }
};
constexpr uint16_t EP_NONE = 0, EP_ZERO = 0xffff, EP_SINGLE = 0x200;

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
__host__ __device__ inline wide_ws carve(void* base, uint32_t tb) {
    char* p = static_cast<char*>(base);
    wide_ws w;
    w.tb = tb;
    w.bsum = reinterpret_cast<double*>(p);   p += align256((size_t)tb * NCH_MAX * sizeof(double));
    w.bex = reinterpret_cast<uint64_t*>(p);  p += align256((size_t)tb * 8 * sizeof(uint64_t));
    w.summ = reinterpret_cast<int32_t*>(p);  p += align256((size_t)tb * NCH_MAX * 2 * 6 * sizeof(int32_t));
    w.lpre = reinterpret_cast<uint32_t*>(p); p += align256((size_t)tb * sizeof(uint32_t));
    w.epred = reinterpret_cast<uint16_t*>(p); p += align256((size_t)tb * NCH_MAX * sizeof(uint16_t));
    w.bzero = reinterpret_cast<uint8_t*>(p); p += align256((size_t)tb * NCH_MAX);
    w.win = reinterpret_cast<int32_t*>(p);
    w.tw = tb + tb / 64 + 1;
    return w;
}

__device__ __forceinline__ uint32_t find_node(const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes, uint32_t blk) {
    uint32_t lo = 0, hi = n_nodes;   // last node whose first_block <= blk
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (nodes[mid].first_block <= blk) lo = mid; else hi = mid; }
    return lo;
}

// inclusive prefix sum of a double over the wave by DPP (rows of 16: row_shr 1/2/4/8, then row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3);
// lanes without a source add +0.0. A prediction aid (and exact for the integer-valued block sums): the association order does not matter.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_mov_f64(double src) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_prefix_f64(double v) {
    v += dpp_mov_f64<0x111, 0xf>(v); v += dpp_mov_f64<0x112, 0xf>(v); v += dpp_mov_f64<0x114, 0xf>(v); v += dpp_mov_f64<0x118, 0xf>(v);
    v += dpp_mov_f64<0x142, 0xa>(v); v += dpp_mov_f64<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ void st_store(int32_t* o, const fsum::stretch& a) { o[0] = a.d[0]; o[1] = a.d[1]; o[2] = a.lo[0]; o[3] = a.lo[1]; o[4] = a.hi[0]; o[5] = a.hi[1]; }
__device__ __forceinline__ fsum::stretch st_load(const int32_t* o) { fsum::stretch a; a.d[0] = o[0]; a.d[1] = o[1]; a.lo[0] = o[2]; a.lo[1] = o[3]; a.hi[0] = o[4]; a.hi[1] = o[5]; return a; }

struct walk_window { uint32_t ep; int32_t m[2][6]; };

__device__ __forceinline__ void load_window(const wide_ws& ws, uint32_t first_block, uint32_t n_blocks, uint32_t b0, int lane, int c, walk_window& w) {
    const uint32_t j = b0 + (uint32_t)lane;
    w.ep = EP_ZERO;   // past the node's end: identity
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int i = 0; i < 6; i++) w.m[k][i] = 0;
    if (j < n_blocks) {
        const size_t at = ws.at(c, first_block + j);
        w.ep = ws.epred[at];
        const int4* p = reinterpret_cast<const int4*>(ws.summ + at * 12);   // 48-byte records, 16-byte aligned
        const int4 a = p[0], b = p[1], d = p[2];
        w.m[0][0] = a.x; w.m[0][1] = a.y; w.m[0][2] = a.z; w.m[0][3] = a.w; w.m[0][4] = b.x; w.m[0][5] = b.y;
        w.m[1][0] = b.z; w.m[1][1] = b.w; w.m[1][2] = d.x; w.m[1][3] = d.y; w.m[1][4] = d.z; w.m[1][5] = d.w;
    }
}

// wave64 inclusive scans by DPP (rows of 16 lanes: row_shr 1/2/4/8, then row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3).
// Lanes without a source get `old`, which is the identity of the field, so every lane composes unconditionally.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int32_t dpp_mov(int32_t old, int32_t src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false); }

template <int CTRL, int ROW_MASK> __device__ __forceinline__ void scan_step(fsum::stretch& st) {
    fsum::stretch f;
    f.d[0] = dpp_mov<CTRL, ROW_MASK>(0, st.d[0]); f.d[1] = dpp_mov<CTRL, ROW_MASK>(0, st.d[1]);
    f.lo[0] = dpp_mov<CTRL, ROW_MASK>(fsum::D_SAT, st.lo[0]); f.lo[1] = dpp_mov<CTRL, ROW_MASK>(fsum::D_SAT, st.lo[1]);
    f.hi[0] = dpp_mov<CTRL, ROW_MASK>(-fsum::D_SAT, st.hi[0]); f.hi[1] = dpp_mov<CTRL, ROW_MASK>(-fsum::D_SAT, st.hi[1]);
    st = fsum::compose(f, st);
}
__device__ __forceinline__ void wave_scan(fsum::stretch& st) {
    scan_step<0x111, 0xf>(st); scan_step<0x112, 0xf>(st); scan_step<0x114, 0xf>(st); scan_step<0x118, 0xf>(st);
    scan_step<0x142, 0xa>(st); scan_step<0x143, 0xc>(st);
}
// monotone chains (addends >= 0 on a positive sum): the floor offsets never go below 0 and the greatest result offset is the last
// one, so a map is its two result offsets and applies iff k + d[k & 1] < 2^24. A map that must not apply carries D_SAT.
struct mono { int32_t d[2]; };
template <int CTRL, int ROW_MASK> __device__ __forceinline__ void scan_step(mono& st) {
    const int32_t f0 = dpp_mov<CTRL, ROW_MASK>(0, st.d[0]), f1 = dpp_mov<CTRL, ROW_MASK>(0, st.d[1]);
    const int32_t g0 = (f0 & 1) ? st.d[1] : st.d[0], g1 = (f1 & 1) ? st.d[0] : st.d[1];
    st.d[0] = fsum::sat_add(f0, g0); st.d[1] = fsum::sat_add(f1, g1);
}
__device__ __forceinline__ void wave_scan(mono& st) {
    scan_step<0x111, 0xf>(st); scan_step<0x112, 0xf>(st); scan_step<0x114, 0xf>(st); scan_step<0x118, 0xf>(st);
    scan_step<0x142, 0xa>(st); scan_step<0x143, 0xc>(st);
}


__device__ __forceinline__ void wide_partition_body(uint32_t* perm0, uint32_t* perm1, const uint8_t* side,
                                                    const tsvq_wide_node* __restrict__ nodes, uint32_t n_nodes, const tsvq_wide_ctrl* ctrl,
                                                    void* ws_base, uint32_t tb, tsvq_split_out* outs, const uint32_t blk) {
    __shared__ uint32_t s_wl[4];
    const wide_ws ws = carve(ws_base, tb);
    const int tid = threadIdx.x;
    const uint32_t ni = find_node(nodes, n_nodes, blk);
    const tsvq_wide_ctrl& c = ctrl[ni];
    const tsvq_wide_node& nd = nodes[ni];
    if (blk == nd.first_block && tid == 0) {
        tsvq_split_out* out = outs + nd.out_index;
        if (c.done == 1) {
            out->ok = 1; out->l_count = c.l_n; out->r_count = c.r_n; out->l_weight = c.l_w; out->r_weight = c.r_w;
            out->l_var = c.l_var; out->r_var = c.r_var;
            for (int k = 0; k < 16; k++) { out->l_centroid[k] = c.l_c[k]; out->r_centroid[k] = c.r_c[k]; }
        } else out->ok = 2;   // run this node through the one-workgroup kernel
    }
    if (c.done != 1) return;
    const uint32_t* members = tsvq_list(perm0, perm1, nd.buf) + nd.start;
    uint32_t* child = tsvq_child_list(perm0, perm1, nd.buf) + nd.start;
    const uint32_t pos = (blk - nd.first_block) * WB + (uint32_t)tid;
    const bool valid = pos < nd.count;
    const bool right = valid && side[nd.start + pos] != 0;
    const bool left = valid && !right;
    const uint64_t mL = __ballot(left);
    const int lane = tid & 63, wave = tid >> 6;
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) s_wl[wave] = (uint32_t)__popcll(mL);
    __syncthreads();
    uint32_t lbefore = ws.lpre[blk] + (uint32_t)__popcll(mL & below);
    for (int w = 0; w < wave; w++) lbefore += s_wl[w];
    if (left) child[lbefore] = members[pos];
    if (right) child[c.l_n + (pos - lbefore)] = members[pos];
}
