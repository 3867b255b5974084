// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the vector instructions the board-detection
// kernels are built from, on one MI355X.  Every kernel is the same skeleton: 8 independent accumulators per lane, the
// instruction applied to each of them 8 times per loop trip (64 back-to-back, dependency distance 8), 8 waves per SIMD
// on all 256 CUs, so the figure is the steady-state issue rate of that instruction alone.  Wall time comes from HIP events;
// "cyc@2.4GHz" converts it with the 2.4 GHz peak clock (DVFS can only run slower, so it is an upper bound on the true
// cycle count); the s_memtime ticks wave 0 of block 0 spent in its loop are printed per instruction of that wave as well.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o valu_rate_bench valu_rate_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define X8(I) I("%0") I("%1") I("%2") I("%3") I("%4") I("%5") I("%6") I("%7")
#define REP8(S) S S S S S S S S

// 32-bit accumulators; %8, %9 = two loop-invariant vector operands, %10 = scalar operand
#define KERNEL32(NAME, I)                                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, int iters, unsigned long long* ticks)              \
    {                                                                                                                  \
        unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4,                \
                 a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                                                  \
        unsigned b = threadIdx.x * 2654435761u + 77u, c = blockIdx.x * 40503u + 0x00010203u;                           \
        unsigned s = iters * 3 + 1;                                                                                    \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int it = 0; it < iters; it++)                                                                             \
            asm volatile(REP8(X8(I))                                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c), "s"(s)                                                                      \
                         : "vcc", "s20", "s21");                                                                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                   \
        if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;                                                   \
    }

// 64-bit accumulators (register pairs) for the packed-f32 / 64-bit instructions
#define KERNEL64(NAME, I)                                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, int iters, unsigned long long* ticks)              \
    {                                                                                                                  \
        unsigned long long a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4,      \
                           a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                                        \
        unsigned long long b = threadIdx.x * 2654435761ull + 77u, c = blockIdx.x * 40503ull + 0x00010203u;             \
        unsigned s = iters * 3 + 1;                                                                                    \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int it = 0; it < iters; it++)                                                                             \
            asm volatile(REP8(X8(I))                                                                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                         : "v"(b), "v"(c), "s"(s)                                                                      \
                         : "vcc", "s20", "s21");                                                                                     \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                       \
        if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;                                                   \
    }

// ---- f32
#define I_FMA_F32(d) "v_fma_f32 " d ", " d ", %8, %9\n"
#define I_MAC_F32(d) "v_fmac_f32 " d ", %8, %9\n"
#define I_ADD_F32(d) "v_add_f32 " d ", " d ", %8\n"
#define I_MUL_F32(d) "v_mul_f32 " d ", " d ", %8\n"
#define I_MIN_F32(d) "v_min_f32 " d ", " d ", %8\n"
#define I_MAX_F32(d) "v_max_f32 " d ", " d ", %8\n"
#define I_MED3_F32(d) "v_med3_f32 " d ", " d ", %8, %9\n"
#define I_MIN3_F32(d) "v_min3_f32 " d ", " d ", %8, %9\n"
#define I_MAX3_F32(d) "v_max3_f32 " d ", " d ", %8, %9\n"
#define I_CVT_F32_UB0(d) "v_cvt_f32_ubyte0 " d ", " d "\n"
#define I_CVT_F32_UB2(d) "v_cvt_f32_ubyte2 " d ", " d "\n"
#define I_CVT_U32_F32(d) "v_cvt_u32_f32 " d ", " d "\n"
#define I_CVT_F32_U32(d) "v_cvt_f32_u32 " d ", " d "\n"
#define I_CVT_PK_U8_F32(d) "v_cvt_pk_u8_f32 " d ", %8, 1, " d "\n"
#define I_PK_FMA_F32(d) "v_pk_fma_f32 " d ", " d ", %8, %9\n"
#define I_PK_ADD_F32(d) "v_pk_add_f32 " d ", " d ", %8\n"
#define I_PK_MUL_F32(d) "v_pk_mul_f32 " d ", " d ", %8\n"
// ---- packed f16 (0..255 and sums below 2048 are exact in f16)
#define I_PK_MIN_F16(d) "v_pk_min_f16 " d ", " d ", %8\n"
#define I_PK_MAX_F16(d) "v_pk_max_f16 " d ", " d ", %8\n"
#define I_PK_ADD_F16(d) "v_pk_add_f16 " d ", " d ", %8\n"
#define I_PK_FMA_F16(d) "v_pk_fma_f16 " d ", " d ", %8, %9\n"
#define I_MED3_F16(d) "v_med3_f16 " d ", " d ", %8, %9\n"
// ---- 32-bit integer / bit ops
#define I_MOV(d) "v_mov_b32 " d ", %8\n"
#define I_ADD_U32(d) "v_add_u32 " d ", " d ", %8\n"
#define I_SUB_U32(d) "v_sub_u32 " d ", " d ", %8\n"
#define I_ADD3_U32(d) "v_add3_u32 " d ", " d ", %8, %9\n"
#define I_AND(d) "v_and_b32 " d ", " d ", %8\n"
#define I_OR(d) "v_or_b32 " d ", " d ", %8\n"
#define I_XOR(d) "v_xor_b32 " d ", " d ", %8\n"
#define I_AND_OR(d) "v_and_or_b32 " d ", " d ", %8, %9\n"
#define I_OR3(d) "v_or3_b32 " d ", " d ", %8, %9\n"
#define I_LSHL_OR(d) "v_lshl_or_b32 " d ", " d ", 3, %9\n"
#define I_LSHL_ADD(d) "v_lshl_add_u32 " d ", " d ", 3, %9\n"
#define I_ADD_LSHL(d) "v_add_lshl_u32 " d ", " d ", %9, 3\n"
#define I_XAD(d) "v_xad_u32 " d ", " d ", %8, %9\n"
#define I_LSHLREV(d) "v_lshlrev_b32 " d ", 3, " d "\n"
#define I_LSHRREV(d) "v_lshrrev_b32 " d ", 3, " d "\n"
#define I_ASHRREV(d) "v_ashrrev_i32 " d ", 3, " d "\n"
#define I_BFE_U32(d) "v_bfe_u32 " d ", " d ", 5, 7\n"
#define I_BFI(d) "v_bfi_b32 " d ", %8, " d ", %9\n"
#define I_BCNT(d) "v_bcnt_u32_b32 " d ", " d ", %8\n"
#define I_CNDMASK(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n"
#define I_CMP_GT_U32(d) "v_cmp_gt_u32 vcc, " d ", %8\n"
#define I_CMP_GT_U32_S(d) "v_cmp_gt_u32 s[20:21], " d ", %8\n"
#define I_PERM(d) "v_perm_b32 " d ", " d ", %8, %9\n"
#define I_ALIGNBIT(d) "v_alignbit_b32 " d ", " d ", %8, 8\n"
#define I_ALIGNBYTE(d) "v_alignbyte_b32 " d ", " d ", %8, 1\n"
#define I_MIN_U32(d) "v_min_u32 " d ", " d ", %8\n"
#define I_MAX_U32(d) "v_max_u32 " d ", " d ", %8\n"
#define I_MIN_I32(d) "v_min_i32 " d ", " d ", %8\n"
#define I_MIN3_U32(d) "v_min3_u32 " d ", " d ", %8, %9\n"
#define I_MAX3_U32(d) "v_max3_u32 " d ", " d ", %8, %9\n"
#define I_MED3_U32(d) "v_med3_u32 " d ", " d ", %8, %9\n"
#define I_MED3_I32(d) "v_med3_i32 " d ", " d ", %8, %9\n"
#define I_MUL_U32_U24(d) "v_mul_u32_u24 " d ", " d ", %8\n"
#define I_MAD_U32_U24(d) "v_mad_u32_u24 " d ", " d ", %8, %9\n"
#define I_MAD_I32_I24(d) "v_mad_i32_i24 " d ", " d ", %8, %9\n"
#define I_MUL_LO_U32(d) "v_mul_lo_u32 " d ", " d ", %8\n"
#define I_MUL_HI_U32(d) "v_mul_hi_u32 " d ", " d ", %8\n"
#define I_MAD_U64_U32(d) "v_mad_u64_u32 " d ", vcc, %8, %9, " d "\n"
#define I_LSHLREV_B64(d) "v_lshlrev_b64 " d ", 3, " d "\n"
#define I_SAD_U8(d) "v_sad_u8 " d ", " d ", %8, %9\n"
#define I_SAD_U32(d) "v_sad_u32 " d ", " d ", %8, %9\n"
#define I_MSAD_U8(d) "v_msad_u8 " d ", " d ", %8, %9\n"
#define I_LERP_U8(d) "v_lerp_u8 " d ", " d ", %8, %9\n"
#define I_DOT4_U32_U8(d) "v_dot4_u32_u8 " d ", " d ", %8, %9\n"
#define I_DOT2_U32_U16(d) "v_dot2_u32_u16 " d ", " d ", %8, %9\n"
#define I_MBCNT_LO(d) "v_mbcnt_lo_u32_b32 " d ", " d ", %8\n"
#define I_FFBH(d) "v_ffbh_u32 " d ", " d "\n"
#define I_BFREV(d) "v_bfrev_b32 " d ", " d "\n"
// ---- 16-bit and packed 16-bit integer
#define I_ADD_U16(d) "v_add_u16 " d ", " d ", %8\n"
#define I_MIN_U16(d) "v_min_u16 " d ", " d ", %8\n"
#define I_MAX_U16(d) "v_max_u16 " d ", " d ", %8\n"
#define I_MED3_U16(d) "v_med3_u16 " d ", " d ", %8, %9\n"
#define I_MIN3_U16(d) "v_min3_u16 " d ", " d ", %8, %9\n"
#define I_MAD_U16(d) "v_mad_u16 " d ", " d ", %8, %9\n"
#define I_PK_ADD_U16(d) "v_pk_add_u16 " d ", " d ", %8\n"
#define I_PK_SUB_I16(d) "v_pk_sub_i16 " d ", " d ", %8\n"
#define I_PK_MIN_U16(d) "v_pk_min_u16 " d ", " d ", %8\n"
#define I_PK_MAX_U16(d) "v_pk_max_u16 " d ", " d ", %8\n"
#define I_PK_MAX_I16(d) "v_pk_max_i16 " d ", " d ", %8\n"
#define I_PK_MUL_LO_U16(d) "v_pk_mul_lo_u16 " d ", " d ", %8\n"
#define I_PK_MAD_U16(d) "v_pk_mad_u16 " d ", " d ", %8, %9\n"
#define I_PK_LSHRREV_B16(d) "v_pk_lshrrev_b16 " d ", 3, " d "\n"
#define I_PK_ASHRREV_I16(d) "v_pk_ashrrev_i16 " d ", 3, " d "\n"
// ---- operand-source variants: SDWA byte selects, DPP lane moves, scalar operand
#define I_ADD_U32_SDWA(d) "v_add_u32_sdwa " d ", " d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define I_MAX_U16_SDWA(d) "v_max_u16_sdwa " d ", " d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n"
#define I_MOV_DPP_SHR(d) "v_mov_b32_dpp " d ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD_DPP_SHR(d) "v_add_u32_dpp " d ", " d ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOV_DPP_BCAST(d) "v_mov_b32_dpp " d ", %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_ADD_U32_S(d) "v_add_u32 " d ", %10, " d "\n"
#define I_FMAC_F32_S(d) "v_fmac_f32 " d ", %10, %9\n"
#define I_FMA_F32_S(d) "v_fma_f32 " d ", " d ", %10, %9\n"
#define I_MUL_F32_S(d) "v_mul_f32 " d ", %10, " d "\n"
#define I_AND_S(d) "v_and_b32 " d ", %10, " d "\n"
#define I_AND_LIT(d) "v_and_b32 " d ", 0x7f7f7f7f, " d "\n"
#define I_ADD_LIT(d) "v_add_u32 " d ", 0x12345, " d "\n"
#define I_CNDMASK_E64(d) "v_cndmask_b32_e64 " d ", " d ", %8, s[20:21]\n"
#define I_CMP_CNDMASK(d) "v_cmp_gt_u32 vcc, " d ", %8\n v_cndmask_b32 " d ", " d ", %9, vcc\n"
#define I_BITOP3(d) "v_bitop3_b32 " d ", " d ", %8, %9 bitop3:0x96\n"
#define I_CVT_F32_U32_SDWA(d) "v_cvt_f32_u32_sdwa " d ", " d " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
#define I_LSHLREV_V(d) "v_lshlrev_b32 " d ", %8, " d "\n"
#define I_LSHRREV_V(d) "v_lshrrev_b32 " d ", %8, " d "\n"
#define I_SUB_U16(d) "v_sub_u16 " d ", " d ", %8\n"
#define I_MAX_I16(d) "v_max_i16 " d ", " d ", %8\n"
#define I_MUL_LO_U16(d) "v_mul_lo_u16 " d ", " d ", %8\n"
#define I_LSHRREV_B16(d) "v_lshrrev_b16 " d ", 3, " d "\n"
#define I_SUBREV_U32(d) "v_subrev_u32 " d ", %8, " d "\n"
#define I_MIN_F16(d) "v_min_f16 " d ", " d ", %8\n"
#define I_ADD_F16(d) "v_add_f16 " d ", " d ", %8\n"
#define I_READLANE(d) "v_readlane_b32 s20, " d ", 3\n"

KERNEL32(fma_f32, I_FMA_F32) KERNEL32(mac_f32, I_MAC_F32) KERNEL32(add_f32, I_ADD_F32) KERNEL32(mul_f32, I_MUL_F32)
KERNEL32(min_f32, I_MIN_F32) KERNEL32(max_f32, I_MAX_F32) KERNEL32(med3_f32, I_MED3_F32) KERNEL32(min3_f32, I_MIN3_F32)
KERNEL32(max3_f32, I_MAX3_F32) KERNEL32(cvt_f32_ub0, I_CVT_F32_UB0) KERNEL32(cvt_f32_ub2, I_CVT_F32_UB2)
KERNEL32(cvt_u32_f32, I_CVT_U32_F32) KERNEL32(cvt_f32_u32, I_CVT_F32_U32) KERNEL32(cvt_pk_u8_f32, I_CVT_PK_U8_F32)
KERNEL64(pk_fma_f32, I_PK_FMA_F32) KERNEL64(pk_add_f32, I_PK_ADD_F32) KERNEL64(pk_mul_f32, I_PK_MUL_F32)
KERNEL32(pk_min_f16, I_PK_MIN_F16) KERNEL32(pk_max_f16, I_PK_MAX_F16) KERNEL32(pk_add_f16, I_PK_ADD_F16)
KERNEL32(pk_fma_f16, I_PK_FMA_F16) KERNEL32(med3_f16, I_MED3_F16)
KERNEL32(mov, I_MOV) KERNEL32(add_u32, I_ADD_U32) KERNEL32(sub_u32, I_SUB_U32) KERNEL32(add3_u32, I_ADD3_U32)
KERNEL32(and_b32, I_AND) KERNEL32(or_b32, I_OR) KERNEL32(xor_b32, I_XOR) KERNEL32(and_or_b32, I_AND_OR) KERNEL32(or3_b32, I_OR3)
KERNEL32(lshl_or_b32, I_LSHL_OR) KERNEL32(lshl_add_u32, I_LSHL_ADD) KERNEL32(add_lshl_u32, I_ADD_LSHL) KERNEL32(xad_u32, I_XAD)
KERNEL32(lshlrev_b32, I_LSHLREV) KERNEL32(lshrrev_b32, I_LSHRREV) KERNEL32(ashrrev_i32, I_ASHRREV) KERNEL32(bfe_u32, I_BFE_U32)
KERNEL32(bfi_b32, I_BFI) KERNEL32(bcnt_u32_b32, I_BCNT) KERNEL32(cndmask_b32, I_CNDMASK) KERNEL32(cmp_gt_u32_vcc, I_CMP_GT_U32)
KERNEL32(cmp_gt_u32_sgpr, I_CMP_GT_U32_S)
KERNEL32(perm_b32, I_PERM) KERNEL32(alignbit_b32, I_ALIGNBIT) KERNEL32(alignbyte_b32, I_ALIGNBYTE)
KERNEL32(min_u32, I_MIN_U32) KERNEL32(max_u32, I_MAX_U32) KERNEL32(min_i32, I_MIN_I32) KERNEL32(min3_u32, I_MIN3_U32)
KERNEL32(max3_u32, I_MAX3_U32) KERNEL32(med3_u32, I_MED3_U32) KERNEL32(med3_i32, I_MED3_I32)
KERNEL32(mul_u32_u24, I_MUL_U32_U24) KERNEL32(mad_u32_u24, I_MAD_U32_U24) KERNEL32(mad_i32_i24, I_MAD_I32_I24)
KERNEL32(mul_lo_u32, I_MUL_LO_U32) KERNEL32(mul_hi_u32, I_MUL_HI_U32)
KERNEL64(lshlrev_b64, I_LSHLREV_B64)
KERNEL32(sad_u8, I_SAD_U8) KERNEL32(sad_u32, I_SAD_U32) KERNEL32(msad_u8, I_MSAD_U8) KERNEL32(lerp_u8, I_LERP_U8)
KERNEL32(dot4_u32_u8, I_DOT4_U32_U8) KERNEL32(dot2_u32_u16, I_DOT2_U32_U16)
KERNEL32(mbcnt_lo, I_MBCNT_LO) KERNEL32(ffbh_u32, I_FFBH) KERNEL32(bfrev_b32, I_BFREV)
KERNEL32(add_u16, I_ADD_U16) KERNEL32(min_u16, I_MIN_U16) KERNEL32(max_u16, I_MAX_U16) KERNEL32(med3_u16, I_MED3_U16)
KERNEL32(min3_u16, I_MIN3_U16) KERNEL32(mad_u16, I_MAD_U16)
KERNEL32(pk_add_u16, I_PK_ADD_U16) KERNEL32(pk_sub_i16, I_PK_SUB_I16) KERNEL32(pk_min_u16, I_PK_MIN_U16)
KERNEL32(pk_max_u16, I_PK_MAX_U16) KERNEL32(pk_max_i16, I_PK_MAX_I16) KERNEL32(pk_mul_lo_u16, I_PK_MUL_LO_U16)
KERNEL32(pk_mad_u16, I_PK_MAD_U16) KERNEL32(pk_lshrrev_b16, I_PK_LSHRREV_B16) KERNEL32(pk_ashrrev_i16, I_PK_ASHRREV_I16)
KERNEL32(add_u32_sdwa, I_ADD_U32_SDWA) KERNEL32(max_u16_sdwa, I_MAX_U16_SDWA) KERNEL32(mov_dpp_row_shr, I_MOV_DPP_SHR)
KERNEL32(add_u32_dpp_row_shr, I_ADD_DPP_SHR) KERNEL32(mov_dpp_row_bcast, I_MOV_DPP_BCAST) KERNEL32(add_u32_sgpr_src, I_ADD_U32_S)
KERNEL32(readlane_b32, I_READLANE)
KERNEL32(fmac_f32_sgpr, I_FMAC_F32_S) KERNEL32(fma_f32_sgpr, I_FMA_F32_S) KERNEL32(mul_f32_sgpr, I_MUL_F32_S) KERNEL32(and_b32_sgpr, I_AND_S)
KERNEL32(and_b32_literal, I_AND_LIT) KERNEL32(add_u32_literal, I_ADD_LIT) KERNEL32(cndmask_e64_sgpr, I_CNDMASK_E64)
KERNEL32(cmp_then_cndmask, I_CMP_CNDMASK) KERNEL32(bitop3_b32, I_BITOP3) KERNEL32(cvt_f32_u32_sdwa, I_CVT_F32_U32_SDWA)
KERNEL32(lshlrev_b32_vsrc, I_LSHLREV_V) KERNEL32(lshrrev_b32_vsrc, I_LSHRREV_V) KERNEL32(sub_u16, I_SUB_U16) KERNEL32(max_i16, I_MAX_I16)
KERNEL32(mul_lo_u16, I_MUL_LO_U16) KERNEL32(lshrrev_b16, I_LSHRREV_B16) KERNEL32(subrev_u32, I_SUBREV_U32) KERNEL32(min_f16, I_MIN_F16)
KERNEL32(add_f16, I_ADD_F16)

typedef void (*kern_t)(unsigned*, int, unsigned long long*);
struct Entry { const char* name; kern_t k; };
#define E(N) {#N, k_##N}
static const Entry entries[] = {
    E(fma_f32), E(mac_f32), E(add_f32), E(mul_f32), E(min_f32), E(max_f32), E(med3_f32), E(min3_f32), E(max3_f32),
    E(cvt_f32_ub0), E(cvt_f32_ub2), E(cvt_u32_f32), E(cvt_f32_u32), E(cvt_pk_u8_f32), E(pk_fma_f32), E(pk_add_f32), E(pk_mul_f32),
    E(pk_min_f16), E(pk_max_f16), E(pk_add_f16), E(pk_fma_f16), E(med3_f16),
    E(mov), E(add_u32), E(sub_u32), E(add3_u32), E(and_b32), E(or_b32), E(xor_b32), E(and_or_b32), E(or3_b32), E(lshl_or_b32),
    E(lshl_add_u32), E(add_lshl_u32), E(xad_u32), E(lshlrev_b32), E(lshrrev_b32), E(ashrrev_i32), E(bfe_u32), E(bfi_b32),
    E(bcnt_u32_b32), E(cndmask_b32), E(cmp_gt_u32_vcc), E(cmp_gt_u32_sgpr), E(perm_b32), E(alignbit_b32), E(alignbyte_b32),
    E(min_u32), E(max_u32), E(min_i32), E(min3_u32), E(max3_u32), E(med3_u32), E(med3_i32),
    E(mul_u32_u24), E(mad_u32_u24), E(mad_i32_i24), E(mul_lo_u32), E(mul_hi_u32), E(lshlrev_b64),
    E(sad_u8), E(sad_u32), E(msad_u8), E(lerp_u8), E(dot4_u32_u8), E(dot2_u32_u16), E(mbcnt_lo), E(ffbh_u32), E(bfrev_b32),
    E(add_u16), E(min_u16), E(max_u16), E(med3_u16), E(min3_u16), E(mad_u16),
    E(pk_add_u16), E(pk_sub_i16), E(pk_min_u16), E(pk_max_u16), E(pk_max_i16), E(pk_mul_lo_u16), E(pk_mad_u16),
    E(pk_lshrrev_b16), E(pk_ashrrev_i16),
    E(add_u32_sdwa), E(max_u16_sdwa), E(mov_dpp_row_shr), E(add_u32_dpp_row_shr), E(mov_dpp_row_bcast), E(add_u32_sgpr_src),
    E(readlane_b32),
    E(fmac_f32_sgpr), E(fma_f32_sgpr), E(mul_f32_sgpr), E(and_b32_sgpr), E(and_b32_literal), E(add_u32_literal), E(cndmask_e64_sgpr),
    E(cmp_then_cndmask), E(bitop3_b32), E(cvt_f32_u32_sdwa), E(lshlrev_b32_vsrc), E(lshrrev_b32_vsrc), E(sub_u16), E(max_i16),
    E(mul_lo_u16), E(lshrrev_b16), E(subrev_u32), E(min_f16), E(add_f16),
};

int main(int argc, char** argv)
{
    // blocks of 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
    const int blocks = 256 * waves_per_simd, iters = 2048;
    unsigned* d_out;
    unsigned long long* d_ticks;
    hipMalloc(&d_out, (size_t)blocks * 256 * sizeof(unsigned));
    hipMalloc(&d_ticks, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# %d waves per SIMD on 256 CUs, %d x 64 instructions per wave; cycles = shader cycles per wave64 instruction per SIMD\n",
           waves_per_simd, iters);
    printf("# cyc@2.4GHz assumes the 2.4 GHz peak clock: an UPPER bound on the true cycles per instruction per SIMD\n");
    printf("%-22s %9s %14s %10s %14s\n", "instruction", "ms", "Gwave-inst/s", "cyc@2.4GHz", "memtime/inst");
    for (const Entry& e : entries) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 16, d_ticks);
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_ticks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        if (hipGetLastError() != hipSuccess) { printf("%-22s launch failed\n", e.name); continue; }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)waves_per_simd * iters * 64;       // wave instructions issued by one SIMD
        const double rate = inst_per_simd / (ms * 1e-3);                        // per SIMD per second
        unsigned long long ticks = 0;
        hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
        printf("%-22s %9.3f %14.3f %10.2f %14.4f\n", e.name, ms, rate * 1e-9, 2.4e9 / rate, (double)ticks / ((double)iters * 64));
    }
    hipFree(d_out);
    hipFree(d_ticks);
    return 0;
}
