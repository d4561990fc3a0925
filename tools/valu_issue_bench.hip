// tools/valu_issue_bench.hip -- how many cycles does one wave64 integer VALU instruction hold a gfx950 SIMD?
//
// K3 (HT cleanup encoder) is VALU-issue bound; whether it sits at 40 % or 80 % of the issue roof depends on
// whether the integer / bit-manipulation ops it is made of issue at 2 cycles per wave64 instruction (SIMD-32,
// what MI355X_MICROARCH.md lists for v_fma_f32) or at 4 (SIMD-16, GCN).  This measures it: per op, a loop of
// independent (ILP 8) or dependent (ILP 1) instructions, timed with s_memtime inside the wave, at 1, 2, 4 waves
// per SIMD on ONE CU (a single workgroup of 256 / 512 / 1024 threads: a workgroup's waves are dealt to the four
// SIMDs in turn), and over the whole chip at 8 waves per SIMD (wall time).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue_bench tools/valu_issue_bench.hip && tools/valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIters = 512;       // loop trips
constexpr int kPerIter = 32;      // instructions of the measured kind per trip

enum Op { MIX_AND_LSHL, MIX_AND_AND_LSHL, MIX_ADD_BFE_OR, MIX_AND_SALU, MIX_LSHL_SALU, AND_SGPR, FMA, FMA3, MULF, ADDF, CVT, MOV, AND, ANDLIT, OR, XOR, NOT, ADD, SUB, ADDCO, LSHL, LSHLV, LSHR, ASHR, MIN, MAXI, FFBH, BFE, BFEI, BFEV, MUL24, MAD24, MULLO, PERM, ALIGNBIT, ADD3, OR3, MIN3, LSHL_ADD, LSHL_OR, AND_OR, BITOP3, LSHL64, LSHLADD64, PKMIN, PKADD, PKMAXI, PKLSHL, PKSUB, SDWA_ADD, SDWA_MIN16, CMP_VCC, CMP_SGPR, CND_VCC, CND_SGPR, CND_IMM, CMP_CND, MBCNT, DPP_ADD, DPP_QP, DPP_WSHR, DPP_WSHL, PERMLANE32, READLANE, READFIRST, SWIZZLE, BPERM, LDS_OR, LDS_RD32, LDS_RD64, LDS_WR32, LDS_WR64, NUM_OPS };
static const char* kNames[NUM_OPS] = {
    "mix: v_and + v_lshlrev (2 instr)",
    "mix: v_and + v_and + v_lshlrev (3 instr)",
    "mix: v_add + v_bfe + v_or (3 instr)",
    "mix: v_and + s_add (2 instr)",
    "mix: v_lshlrev + s_add (2 instr)",
    "v_and_b32 (sgpr operand)",
    "v_fma_f32",
    "v_fma_f32 (3 distinct srcs)",
    "v_mul_f32",
    "v_add_f32",
    "v_cvt_f32_u32",
    "v_mov_b32",
    "v_and_b32",
    "v_and_b32 (32-bit literal)",
    "v_or_b32",
    "v_xor_b32",
    "v_not_b32",
    "v_add_u32",
    "v_sub_u32",
    "v_add_co_u32 (writes vcc)",
    "v_lshlrev_b32",
    "v_lshlrev_b32 (vgpr amount)",
    "v_lshrrev_b32",
    "v_ashrrev_i32",
    "v_min_u32",
    "v_max_i32",
    "v_ffbh_i32",
    "v_bfe_u32",
    "v_bfe_i32",
    "v_bfe_u32 (vgpr width)",
    "v_mul_u32_u24",
    "v_mad_u32_u24",
    "v_mul_lo_u32",
    "v_perm_b32",
    "v_alignbit_b32",
    "v_add3_u32",
    "v_or3_b32",
    "v_min3_u32",
    "v_lshl_add_u32",
    "v_lshl_or_b32",
    "v_and_or_b32",
    "v_bitop3_b32",
    "v_lshlrev_b64",
    "v_lshl_add_u64",
    "v_pk_min_u16",
    "v_pk_add_u16",
    "v_pk_max_i16",
    "v_pk_lshlrev_b16",
    "v_pk_sub_i16",
    "v_add_u32_sdwa (BYTE_1 + WORD_1)",
    "v_min_u16_sdwa",
    "v_cmp_ne_u32 -> vcc",
    "v_cmp_ne_u32_e64 -> sgpr pair",
    "v_cndmask_b32 (vcc, set before the loop)",
    "v_cndmask_b32_e64 (sgpr pair)",
    "v_cndmask_b32_e64 0, 1 (sgpr pair)",
    "v_cmp_ne_u32 vcc + v_cndmask (2 instr)",
    "v_mbcnt_lo_u32_b32",
    "v_add_u32 dpp row_shr:1",
    "v_mov_b32 dpp quad_perm:[1,0,3,2]",
    "v_mov_b32 dpp wave_shr:1",
    "v_mov_b32 dpp wave_shl:1",
    "v_permlane32_swap_b32",
    "v_readlane_b32 (-> sgpr)",
    "v_readfirstlane_b32",
    "ds_swizzle_b32 (swap halves of 32? no: qdmode)",
    "ds_bpermute_b32",
    "ds_or_b32 (atomic)",
    "ds_read_b32",
    "ds_read_b64",
    "ds_write_b32",
    "ds_write_b64"};

// operands of the templates: the accumulator (chain register), a loop-invariant VGPR (also a valid LDS byte address /
// lane address), an SGPR pair (written by the -> sgpr forms), a 64-bit accumulator pair
template <int OP, int ILP>
__device__ __forceinline__ void one(uint32_t& a, uint32_t b, uint64_t& m, uint64_t& w, uint32_t& sc)
{
    // (a VALU write followed by a DPP read of the same register needs two wait states: the dependent chain pays them)
    if constexpr (ILP == 1 && (OP == DPP_ADD || OP == DPP_QP || OP == DPP_WSHR || OP == DPP_WSHL || OP == PERMLANE32)) asm volatile("s_nop 1");
    if constexpr (OP == MIX_AND_LSHL) asm volatile("v_and_b32 %0, %4, %0\n v_lshlrev_b32 %0, 1, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MIX_AND_AND_LSHL) asm volatile("v_and_b32 %0, %4, %0\n v_lshlrev_b32 %0, 1, %0\n v_or_b32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MIX_ADD_BFE_OR) asm volatile("v_add_u32 %0, %4, %0\n v_bfe_u32 %0, %0, 1, 31\n v_or_b32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MIX_AND_SALU) asm volatile("v_and_b32 %0, %4, %0\n s_add_u32 %3, %3, 3" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory", "scc");
    if constexpr (OP == MIX_LSHL_SALU) asm volatile("v_lshlrev_b32 %0, 1, %0\n s_add_u32 %3, %3, 3" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory", "scc");
    if constexpr (OP == AND_SGPR) asm volatile("v_and_b32 %0, %3, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %4, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == FMA3) asm volatile("v_fma_f32 %0, %0, %4, 2.0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MULF) asm volatile("v_mul_f32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == ADDF) asm volatile("v_add_f32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == CVT) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == AND) asm volatile("v_and_b32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == ANDLIT) asm volatile("v_and_b32 %0, 0x1ffffffc, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == OR) asm volatile("v_or_b32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == XOR) asm volatile("v_xor_b32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == NOT) asm volatile("v_not_b32 %0, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == ADD) asm volatile("v_add_u32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == SUB) asm volatile("v_sub_u32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == ADDCO) asm volatile("v_add_co_u32 %0, vcc, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LSHLV) asm volatile("v_lshlrev_b32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MIN) asm volatile("v_min_u32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MAXI) asm volatile("v_max_i32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == FFBH) asm volatile("v_ffbh_i32 %0, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == BFEI) asm volatile("v_bfe_i32 %0, %0, 1, 31" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == BFEV) asm volatile("v_bfe_u32 %0, %0, 0, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MUL24) asm volatile("v_mul_u32_u24 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MAD24) asm volatile("v_mad_u32_u24 %0, %0, %4, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %0, %4, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %4, 7" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %4, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == OR3) asm volatile("v_or3_b32 %0, %0, %4, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MIN3) asm volatile("v_min3_u32 %0, %0, %4, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %4, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %4, %4 bitop3:0xc8" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LSHL64) asm volatile("v_lshlrev_b64 %2, 1, %2" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LSHLADD64) asm volatile("v_lshl_add_u64 %2, %2, 0, %2" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == PKMIN) asm volatile("v_pk_min_u16 %0, %0, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == PKADD) asm volatile("v_pk_add_u16 %0, %0, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == PKMAXI) asm volatile("v_pk_max_i16 %0, %0, %4" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == PKLSHL) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == PKSUB) asm volatile("v_pk_sub_i16 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == SDWA_ADD) asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:WORD_1" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == SDWA_MIN16) asm volatile("v_min_u16_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == CMP_VCC) asm volatile("v_cmp_ne_u32 vcc, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == CMP_SGPR) asm volatile("v_cmp_ne_u32_e64 %1, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == CND_VCC) asm volatile("v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %4, %1" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == CND_IMM) asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == CMP_CND) asm volatile("v_cmp_ne_u32 vcc, %4, %0\n v_cndmask_b32 %0, %0, %4, vcc" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %4, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == DPP_ADD) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == DPP_QP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == DPP_WSHR) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == DPP_WSHL) asm volatile("v_mov_b32_dpp %0, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == PERMLANE32) asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == READLANE) asm volatile("v_readlane_b32 %3, %0, 3" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == READFIRST) asm volatile("v_readfirstlane_b32 %3, %0" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == SWIZZLE) asm volatile("ds_swizzle_b32 %0, %0 offset:0x8055\n s_waitcnt lgkmcnt(8)" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == BPERM) asm volatile("ds_bpermute_b32 %0, %4, %0\n s_waitcnt lgkmcnt(8)" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LDS_OR) asm volatile("ds_or_b32 %4, %0\n s_waitcnt lgkmcnt(8)" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LDS_RD32) asm volatile("ds_read_b32 %0, %4\n s_waitcnt lgkmcnt(8)" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LDS_RD64) asm volatile("ds_read_b64 %2, %4\n s_waitcnt lgkmcnt(8)" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LDS_WR32) asm volatile("ds_write_b32 %4, %0\n s_waitcnt lgkmcnt(8)" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
    if constexpr (OP == LDS_WR64) asm volatile("ds_write_b64 %4, %2\n s_waitcnt lgkmcnt(8)" : "+v"(a), "+s"(m), "+v"(w), "+s"(sc) : "v"(b) : "vcc", "memory");
}

// ILP = number of independent accumulators the kPerIter instructions of a trip are spread over
template <int OP, int ILP>
__global__ __launch_bounds__(1024) void bench(uint64_t* cycles, uint32_t* sink, uint32_t seed)
{
    __shared__ uint32_t lds[2048];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 1024] = seed;
    uint32_t acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = seed * (threadIdx.x + 3) + i;
    uint32_t b = (threadIdx.x * 8u) & 0x1F8u;         // a valid LDS byte address / bpermute lane address / operand
    uint64_t m = 0x5555555555555555ull, w[ILP];   // (the mix rows use the low word of w as a second, independent chain)
    uint32_t sc = seed;
#pragma unroll
    for (int i = 0; i < ILP; ++i) w[i] = acc[i];
    if (OP == LDS_OR) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = 1u << i;
    }
    asm volatile("v_cmp_gt_u32 vcc, 40, %0" :: "v"(b) : "vcc");
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int j = 0; j < kPerIter; ++j) one<OP, ILP>(acc[j % ILP], b, m, w[j % ILP], sc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s ^= acc[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
    s ^= (uint32_t)m ^ sc;
    if (s == 0x12345678u) sink[0] = s + lds[b >> 2];
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int ILP>
static int run(uint64_t* d_cycles, uint32_t* d_sink, double clock_ghz, bool& header)
{
    if (!header) {
        printf("%-26s %4s | %9s %9s %9s | %s\n", "op", "ILP", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD",
               "chip, 8 w/SIMD: cycles per wave-instruction and SIMD (from wall time)");
        header = true;
    }
    const double n = (double)kIters * kPerIter;
    double cpi[3];
    for (int k = 0; k < 3; ++k) {
        const int threads = 256 << k;
        hipLaunchKernelGGL((bench<OP, ILP>), dim3(1), dim3(threads), 0, 0, d_cycles, d_sink, 7u);     // warm
        hipLaunchKernelGGL((bench<OP, ILP>), dim3(1), dim3(threads), 0, 0, d_cycles, d_sink, 7u);
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> h(threads / 64);
        CHECK(hipMemcpy(h.data(), d_cycles, h.size() * 8, hipMemcpyDeviceToHost));
        // SIMD-side interval between two instructions of the kind: the slowest wave's cycles / (instructions x waves on the SIMD)
        const uint64_t worst = *std::max_element(h.begin(), h.end());
        cpi[k] = (double)worst / n / (double)(1 << k);
    }
    // whole chip: 256 CUs x 2 workgroups of 1024 threads resident = 8 waves per SIMD, 8 rounds
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256 * 2 * 8;
    hipLaunchKernelGGL((bench<OP, ILP>), dim3(grid), dim3(1024), 0, 0, d_cycles, d_sink, 7u);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((bench<OP, ILP>), dim3(grid), dim3(1024), 0, 0, d_cycles, d_sink, 7u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instr_per_simd = n * (double)grid * 16.0 / (256.0 * 4.0);
    const double chip = ms * 1e-3 * clock_ghz * 1e9 / wave_instr_per_simd;
    printf("%-26s %4d | %9.2f %9.2f %9.2f | %6.2f   (%.3f ms)\n", kNames[OP], ILP, cpi[0], cpi[1], cpi[2], chip, ms);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

__global__ void clock_probe(uint64_t* out)
{
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    while (wall_clock64() - r0 < 1000000) { }          // 10 ms of the 100 MHz constant clock
    out[0] = __builtin_readcyclecounter() - c0; out[1] = wall_clock64() - r0;
}

int main(int argc, char** argv)
{
    uint64_t* d_cycles; uint32_t* d_sink;
    CHECK(hipMalloc(&d_cycles, 8 * 65536 * 16)); CHECK(hipMalloc(&d_sink, 64));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(1), 0, 0, d_cycles);
    CHECK(hipDeviceSynchronize());
    uint64_t pr[2]; CHECK(hipMemcpy(pr, d_cycles, 16, hipMemcpyDeviceToHost));
    int wc_khz = 0; (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
    const double wall_hz = wc_khz ? wc_khz * 1e3 : 1e8;
    const double smem_ghz = (double)pr[0] / ((double)pr[1] / wall_hz) / 1e9;
    printf("device %s, %d CUs, clockRate %.0f MHz; s_memtime advances at %.3f GHz (vs wall clock %.0f MHz)\n",
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3, smem_ghz, wall_hz / 1e6);
    printf("columns: SIMD-side cycles (s_memtime ticks) between two wave64 instructions of the kind = slowest wave's ticks / (instructions x waves per SIMD)\n");
    // the chip column needs the shader clock under load; the in-wave columns are in s_memtime ticks whatever the clock is
    const double ghz = prop.clockRate / 1e6;
    if (smem_ghz < 0.5 * ghz) printf("NOTE: s_memtime is a constant clock here: multiply the in-wave columns by %.2f for shader cycles at %.0f MHz\n", ghz / smem_ghz, ghz * 1e3);
    bool header = false;
    const char* only = argc > 1 ? argv[1] : nullptr;
#define RUN(OP) do { if (!only || strstr(kNames[OP], only)) { if (run<OP, 8>(d_cycles, d_sink, ghz, header)) return 1; if (run<OP, 1>(d_cycles, d_sink, ghz, header)) return 1; } } while (0)
    RUN(MIX_AND_LSHL); RUN(MIX_AND_AND_LSHL); RUN(MIX_ADD_BFE_OR); RUN(MIX_AND_SALU); RUN(MIX_LSHL_SALU); RUN(AND_SGPR);
    RUN(FMA);
    RUN(FMA3);
    RUN(MULF);
    RUN(ADDF);
    RUN(CVT);
    RUN(MOV);
    RUN(AND);
    RUN(ANDLIT);
    RUN(OR);
    RUN(XOR);
    RUN(NOT);
    RUN(ADD);
    RUN(SUB);
    RUN(ADDCO);
    RUN(LSHL);
    RUN(LSHLV);
    RUN(LSHR);
    RUN(ASHR);
    RUN(MIN);
    RUN(MAXI);
    RUN(FFBH);
    RUN(BFE);
    RUN(BFEI);
    RUN(BFEV);
    RUN(MUL24);
    RUN(MAD24);
    RUN(MULLO);
    RUN(PERM);
    RUN(ALIGNBIT);
    RUN(ADD3);
    RUN(OR3);
    RUN(MIN3);
    RUN(LSHL_ADD);
    RUN(LSHL_OR);
    RUN(AND_OR);
    RUN(BITOP3);
    RUN(LSHL64);
    RUN(LSHLADD64);
    RUN(PKMIN);
    RUN(PKADD);
    RUN(PKMAXI);
    RUN(PKLSHL);
    RUN(PKSUB);
    RUN(SDWA_ADD);
    RUN(SDWA_MIN16);
    RUN(CMP_VCC);
    RUN(CMP_SGPR);
    RUN(CND_VCC);
    RUN(CND_SGPR);
    RUN(CND_IMM);
    RUN(CMP_CND);
    RUN(MBCNT);
    RUN(DPP_ADD);
    RUN(DPP_QP);
    RUN(DPP_WSHR);
    RUN(DPP_WSHL);
    RUN(PERMLANE32);
    RUN(READLANE);
    RUN(READFIRST);
    RUN(SWIZZLE);
    RUN(BPERM);
    RUN(LDS_OR);
    RUN(LDS_RD32);
    RUN(LDS_RD64);
    RUN(LDS_WR32);
    RUN(LDS_WR64);
    return 0;
}
