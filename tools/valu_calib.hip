// tools/valu_calib.hip -- gfx950 micro-benchmark: how many SIMD cycles one wave64 VALU instruction of a given opcode occupies, as a
// function of the instruction-level parallelism inside a wave (dependent chain vs 8 independent chains) and of the waves resident on a SIMD.
// This is what prices the instruction-bound ETC1S / UASTC kernels: their bound is VALU issue, and the cost of an
// instruction is not one number. What it measured (profiles/valu_calibration.json): two classes -- ~2.2 cycles for add / sub /
// logic / right shifts / mov / f32 add, mul, fma and ~4.1 cycles for every other opcode timed (multiplies of any width INCLUDING the 24-bit forms,
// left shifts, min / max, three-operand integer forms, compares, selects, converts, DPP, dot, packed 16-bit, double-precision add / mul / fma / min,
// v_mad_i64_i32) -- and ~8.1 for v_mad_u16. v_mul_lo_u32 is an ordinary member of the 4-cycle class, not an outlier.
//
// Method. One launch = (opcode, chains C, waves per SIMD W): 256-thread workgroups, W per CU (= W waves per SIMD; LDS sized so that no more fit), each
// wave executes ITER x 32 x C instructions `x[c] = op(x[c], a, b)` (C register chains, round robin: with C = 1 every instruction depends on the one
// before it, with C = 8 the nearest dependency is 8 instructions away). Every wave records absolute s_memtime at both ends and HW_ID / XCC_ID, so the host can
// (i) check how many waves really overlapped on a SIMD (the spans, not the launch geometry) and (ii) price an instruction from the launch's wall time:
//     cycles_per_instruction = SIMD-busy shader cycles / (waves on the SIMD x ITER x 32 x C)        [shader cycles a SIMD spends per wave-instruction]
// The kernel names carry (op, C, W) as template arguments, so a `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES ...`
// pass over this program gives the counters of every cell next to the measured cycles (tools/valu_table.py reads both).
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/bin/valu_calib tools/valu_calib.hip        Run: tools/bin/valu_calib > valu_calibration.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

enum op_id {
    OP_ADD_U32, OP_SUB_U32, OP_LSHRREV, OP_LSHL_ADD, OP_ADD3, OP_AND_OR, OP_MIN_U32, OP_MAX_I32, OP_MIN3_U32, OP_MED3_I32, OP_CNDMASK, OP_CMP_LT_U32,
    OP_MUL_I32_I24, OP_MUL_U32_U24, OP_MAD_U32_U24, OP_MAD_I32_I24, OP_MUL_LO_U32, OP_MUL_HI_U32, OP_MAD_U64_U32, OP_BFE_U32, OP_PERM_B32, OP_SAD_U8,
    OP_DOT2_I32_I16, OP_DOT4_I32_I8, OP_PK_MUL_LO_U16, OP_PK_MAD_U16, OP_PK_ADD_U16, OP_PK_MAX_I16, OP_MAD_U16, OP_ADD_F32, OP_FMA_F32, OP_FMA_F64, OP_ADD_F64, OP_MOV_DPP,
    OP_ADD_DPP, OP_CVT_F32_U32, OP_MBCNT, OP_MOV_B32, OP_AND_B32, OP_OR_B32, OP_XOR_B32, OP_LSHLREV, OP_ASHRREV, OP_MIN_I32, OP_MUL_F32, OP_MIN_F32, OP_MAX_F32,
    OP_CVT_U32_F32, OP_FMAC_F32, OP_ADD_CO_U32, OP_CMP_CNDMASK, OP_MAD_I64_I32, OP_LSHL_ADD_U64, OP_MIN_F64, OP_MUL_F64, OP_FMA_F64_SQUARE, OP_CMP_LT_F64, OP_CVT_F64_I32, OP_COUNT
};

static const char* const OP_NAME[OP_COUNT] = {
    "v_add_u32", "v_sub_u32", "v_lshrrev_b32", "v_lshl_add_u32", "v_add3_u32", "v_and_or_b32", "v_min_u32", "v_max_i32", "v_min3_u32", "v_med3_i32", "v_cndmask_b32", "v_cmp_lt_u32",
    "v_mul_i32_i24", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_bfe_u32", "v_perm_b32", "v_sad_u8",
    "v_dot2_i32_i16", "v_dot4_i32_i8", "v_pk_mul_lo_u16", "v_pk_mad_u16", "v_pk_add_u16", "v_pk_max_i16", "v_mad_u16", "v_add_f32", "v_fma_f32", "v_fma_f64", "v_add_f64", "v_mov_b32_dpp",
    "v_add_u32_dpp", "v_cvt_f32_u32", "v_mbcnt_lo_u32_b32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_min_i32", "v_mul_f32", "v_min_f32",
    "v_max_f32", "v_cvt_u32_f32", "v_fmac_f32", "v_add_co_u32", "v_cmp_lt_u32+v_cndmask_b32", "v_mad_i64_i32", "v_lshl_add_u64", "v_min_f64", "v_mul_f64", "v_fma_f64 (a, a, x)", "v_cmp_lt_f64",
    "v_cvt_f64_i32"};

// c64: a loop-invariant 64-bit VGPR pair for the second operand of the 64-bit opcodes (the first version of this file built it inside the loop with a v_or +
// v_mov per step, which it then timed along with v_fma_f64: 8.3 cycles reported for a 4.1-cycle instruction)
template <int OP> __device__ __forceinline__ void step(uint32_t& x, uint64_t& x64, uint32_t a, uint32_t b, uint64_t mask, uint64_t c64) {
    if constexpr (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x));
    else if constexpr (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MIN3_U32) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_MED3_I32) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(mask));   // the mask in an SGPR pair nobody writes
    else if constexpr (OP == OP_CMP_LT_U32) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : "+v"(x) : "v"(a) : "vcc");       // a chain of compares (nothing reads vcc)
    else if constexpr (OP == OP_CMP_CNDMASK) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");   // a select: 2 instructions
    else if constexpr (OP == OP_MOV_B32) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));
    else if constexpr (OP == OP_ASHRREV) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x));
    else if constexpr (OP == OP_MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_CVT_U32_F32) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x));
    else if constexpr (OP == OP_FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_ADD_CO_U32) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(a) : "vcc");
    else if constexpr (OP == OP_MUL_I32_I24) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_MAD_I32_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x64) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == OP_BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(x));
    else if constexpr (OP == OP_PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_SAD_U8) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_DOT2_I32_I16) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_DOT4_I32_I8) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_PK_MUL_LO_U16) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_PK_MAX_I16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MAD_U16) asm volatile("v_mad_u16 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (OP == OP_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x64) : "v"(c64));
    else if constexpr (OP == OP_ADD_F64) asm volatile("v_add_f64 %0, %0, %0" : "+v"(x64));
    else if constexpr (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
    else if constexpr (OP == OP_ADD_DPP) asm volatile("v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
    else if constexpr (OP == OP_MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(x) : "v"(a));
    else if constexpr (OP == OP_MAD_I64_I32) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x64) : "v"(a), "v"(b) : "vcc");
    else if constexpr (OP == OP_LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x64) : "v"(c64));
    else if constexpr (OP == OP_MIN_F64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x64) : "v"(c64));
    else if constexpr (OP == OP_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x64) : "v"(c64));
    else if constexpr (OP == OP_FMA_F64_SQUARE) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(x64) : "v"(c64));
    else if constexpr (OP == OP_CMP_LT_F64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x64), "v"(c64) : "vcc");
    else if constexpr (OP == OP_CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x64) : "v"(x));
}

struct wave_rec { uint64_t t0, t1; uint32_t hw_id, xcc_id; };

template <int OP, int CH, int W>
__global__ void __launch_bounds__(256) k_calib(uint32_t* out, wave_rec* rec, int iters, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = tid * 2654435761u + seed, b = (tid ^ seed) | 1u;
    if (OP == OP_ADD_F32 || OP == OP_FMA_F32 || OP == OP_MUL_F32 || OP == OP_MIN_F32 || OP == OP_MAX_F32 || OP == OP_FMAC_F32) { a = __float_as_uint(1.0f + (tid & 15) * 0.0625f); b = __float_as_uint(0.5f); }
    uint32_t x[CH]; uint64_t x64[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { x[c] = tid + c * 977u + seed; x64[c] = (uint64_t)x[c] << 20 | 0x3ff0000000000000ull; }
    const uint64_t mask = __ballot((tid & 3u) != 0u);   // wave-uniform: lives in an SGPR pair
    uint64_t c64 = 0x3ff0000000000001ull + tid;   // ~1.0 as a double
    asm volatile("" : "+v"(c64));
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
    __syncthreads();
    uint64_t t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++)
#pragma unroll
            for (int c = 0; c < CH; c++) step<OP>(x[c], x64[c], a, b, mask, c64);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) acc += x[c] + (uint32_t)x64[c] + (uint32_t)(x64[c] >> 32);
    out[tid] = acc;
    if ((threadIdx.x & 63) == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rec[tid >> 6] = wave_rec{t0, t1, hw, xcc};
    }
}

struct result { std::string op; int chains, waves; double cycles_per_inst, ns_per_inst, median_wave_cycles, wall_us, tick_ghz, concurrency; uint32_t simds_seen; };

// instructions one loop pass of `step` issues (the compare+select cell issues two)
template <int OP> constexpr int insts_per_step() { return OP == OP_CMP_CNDMASK ? 2 : 1; }

template <int OP, int CH, int W> static result run_cell(uint32_t* d_out, wave_rec* d_rec, int n_cu) {
    const int threads = 256, blocks = n_cu * W, waves = blocks * threads / 64;   // a 256-thread workgroup = one wave per SIMD; W workgroups per CU
    const int iters = (OP == OP_MUL_LO_U32 || OP == OP_MUL_HI_U32 || OP == OP_MAD_U64_U32 || OP == OP_FMA_F64 || OP == OP_ADD_F64 || OP >= OP_MAD_I64_I32) ? 64 : 128;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_calib<OP, CH, W>), dim3(blocks), dim3(threads), 0, 0, d_out, d_rec, 8, 1u);   // warm-up (clock ramp, code fetch)
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_calib<OP, CH, W>), dim3(blocks), dim3(threads), 0, 0, d_out, d_rec, iters, 2u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<wave_rec> rec(waves);
    CK(hipMemcpy(rec.data(), d_rec, sizeof(wave_rec) * waves, hipMemcpyDeviceToHost));
    // per SIMD: how many of its waves really overlapped = sum of wave durations / (latest end - earliest start)
    struct span { uint64_t lo = ~0ull, hi = 0, sum = 0; };
    std::vector<uint64_t> cyc; std::map<uint32_t, span> per_simd;
    for (const wave_rec& r : rec) {
        cyc.push_back(r.t1 - r.t0);
        // HW_ID: simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
        span& s = per_simd[((r.xcc_id & 15u) << 16) | (r.hw_id & 0xff30u)];
        s.lo = std::min(s.lo, r.t0); s.hi = std::max(s.hi, r.t1); s.sum += r.t1 - r.t0;
    }
    std::sort(cyc.begin(), cyc.end());
    std::vector<double> conc, busy;
    for (const auto& kv : per_simd) { conc.push_back((double)kv.second.sum / (double)(kv.second.hi - kv.second.lo)); busy.push_back((double)(kv.second.hi - kv.second.lo)); }
    std::sort(conc.begin(), conc.end()); std::sort(busy.begin(), busy.end());
    const double med = (double)cyc[cyc.size() / 2];
    const double insts_per_wave = (double)iters * 32 * CH * insts_per_step<OP>();
    const double waves_per_simd = (double)waves / (double)per_simd.size();
    result r;
    r.op = OP_NAME[OP]; r.chains = CH; r.waves = W; r.median_wave_cycles = med;
    // a SIMD's busy span (ticks) over the wave-instructions it executed
    r.cycles_per_inst = busy[busy.size() / 2] / (waves_per_simd * insts_per_wave);
    r.ns_per_inst = ms * 1e6 / (waves_per_simd * insts_per_wave);   // wall clock (includes launch latency and the tail: an upper bound, tight for the long cells)
    r.wall_us = ms * 1e3; r.tick_ghz = busy[busy.size() / 2] / (ms * 1e6); r.simds_seen = (uint32_t)per_simd.size(); r.concurrency = conc[conc.size() / 2];
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return r;
}

template <int OP> static void run_op(uint32_t* d_out, wave_rec* d_rec, int n_cu, std::vector<result>& out) {
    out.push_back(run_cell<OP, 1, 1>(d_out, d_rec, n_cu));
    out.push_back(run_cell<OP, 1, 2>(d_out, d_rec, n_cu));
    out.push_back(run_cell<OP, 1, 4>(d_out, d_rec, n_cu));
    out.push_back(run_cell<OP, 1, 8>(d_out, d_rec, n_cu));
    out.push_back(run_cell<OP, 8, 1>(d_out, d_rec, n_cu));
    out.push_back(run_cell<OP, 8, 2>(d_out, d_rec, n_cu));
    out.push_back(run_cell<OP, 8, 4>(d_out, d_rec, n_cu));
    out.push_back(run_cell<OP, 8, 8>(d_out, d_rec, n_cu));
}

static int g_first_op = 0;   // argv[1]: skip the opcodes before this index (adding opcodes without re-timing the table)
template <int OP> static void run_all(uint32_t* d_out, wave_rec* d_rec, int n_cu, std::vector<result>& out) {
    if constexpr (OP < OP_COUNT) { if (OP >= g_first_op) run_op<OP>(d_out, d_rec, n_cu, out); run_all<OP + 1>(d_out, d_rec, n_cu, out); }
}

int main(int argc, char** argv) {
    if (argc > 1) g_first_op = std::atoi(argv[1]);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    uint32_t* d_out; wave_rec* d_rec;
    const size_t max_threads = (size_t)n_cu * 8 * 256;
    CK(hipMalloc(&d_out, max_threads * 4)); CK(hipMalloc(&d_rec, max_threads / 64 * sizeof(wave_rec)));
    std::vector<result> res;
    run_all<0>(d_out, d_rec, n_cu, res);
    std::printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz_max\": %d, \"method\": \"W 256-thread workgroups per CU (one wave per SIMD each); every wave brackets ITER x 32 x chains instructions with s_memtime; cycles_per_inst = a SIMD's busy span in s_memtime ticks / the wave-instructions it executed; ns_per_inst = the launch's wall time (HIP events) over the same count; concurrency = waves of a SIMD that really overlapped\",\n \"cells\": [\n", p.gcnArchName, n_cu, p.clockRate / 1000);
    for (size_t i = 0; i < res.size(); i++) {
        const result& r = res[i];
        std::printf("  {\"op\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"cycles_per_inst\": %.3f, \"ns_per_inst\": %.4f, \"median_wave_cycles\": %.0f, \"wall_us\": %.1f, \"tick_ghz\": %.3f, \"simds_seen\": %u, \"concurrency\": %.2f}%s\n",
                    r.op.c_str(), r.chains, r.waves, r.cycles_per_inst, r.ns_per_inst, r.median_wave_cycles, r.wall_us, r.tick_ghz, r.simds_seen, r.concurrency, i + 1 < res.size() ? "," : "");
    }
    std::printf(" ]}\n");
    return 0;
}
