// grok_amd/csrc/kernels_t1dec.hip -- K8: Part-1 (EBCOT) Tier-1 block decoder + dequantisation, gfx950.
//
// Replaces T1Part1::decompress -> T1::decompress_cblk (t1/t1_part1/T1Part1.cpp:124-151,
// t1/t1_part1/T1.cpp:1262-1337) with its MQ decoder (mqc_dec.cpp:107-177, mqc_dec_inl.h), the raw (bypass)
// decoder (mqc_dec.cpp:156-160, mqc_dec_inl.h:55-76) and ShiftFilter / ScaleFilter
// (filters/PostDecompressFilters.h:26-35, :60-71).  Code-block styles as the reference decodes them: LAZY (raw
// sig-prop / mag-ref passes from the fifth plane on), RESET, TERMALL (both through the codeword-segment list),
// VSC, SEGSYM (decoded, a bad symbol is only a warning there) and PTERM (a check that does not change the result).
//
// EBCOT decoding is one dependent chain per code-block: every MQ decision renormalises the
// interval the next one uses and every context depends on the samples decoded so far.  As in K5a the
// parallelism is therefore ACROSS blocks:
//  K8a t1_dec_kernel  -- ONE code-block per wavefront, all 64 lanes running the same (uniform) program.  The lanes
//      are used as register-resident tables and storage: lane i of `cxv` / `tabv` holds MQ context state i / row i
//      of Table C.2 (v_readlane: both are on every decision's dependency chain, LDS would cost > 100 cycles each),
//      lane x of V[0..3] holds the decoded values of column x of the current stripe (one coalesced row load / store
//      per stripe and pass instead of a store / an atomic per sample), and the coded bytes come through a 16-byte
//      register window.  Significance / sign / visited / refined state is one 64-bit row bitmap each (code-blocks
//      are at most 64 wide) in LDS; while a stripe is processed it lives ON THE LANES: lane x holds column x's 3 x 6
//      significance neighbourhood, the same for the signs, and its four visited / refined bits, so that one v_readlane
//      per column gives every window of the column, the context tables (zero coding, sign) are indexed straight by
//      those bits, candidate columns are a ballot over a per-lane test, and a sample that turns significant is three
//      vector instructions on the lanes around it (r02: the uniform 64-bit rows in scalar registers -- most of them
//      spilled to vector registers -- cost 40 % of the kernel's time).  Measured in r01: ~110 executed
//      instructions and ~800 cycles per MQ decision with one wave alone on a SIMD; with many blocks resident the CU's
//      single scalar unit and its four vector units are about equally loaded, which is why this mixed scalar / vector
//      form beats an all-scalar one (a hand-scheduled 45-instruction scalar decoder was 1.5x slower on whole images).
//      (An earlier several-blocks-per-wave form -- branch divergence makes the lanes take turns -- measured 80 ms
//      with 16 lanes, 38.6 ms with 4 and 41 ms with 2 where this form takes 23 ms, on 12 288 blocks.)
//  The wave then dequantises and stores its block's rows (r02: a second kernel, K8b, from a global workspace).
#include "kernels.h"

namespace grk_amd {

namespace {

// T.800 Table C.2: Qe | NMPS << 16 | NLPS << 22 | SWITCH << 28
#define MQROW(qe, nm, nl, sw) ((uint32_t)(qe) | ((uint32_t)(nm) << 16) | ((uint32_t)(nl) << 22) | ((uint32_t)(sw) << 28))
__device__ const uint32_t g_mq_table[47] = {
    MQROW(0x5601, 1, 1, 1),  MQROW(0x3401, 2, 6, 0),  MQROW(0x1801, 3, 9, 0),  MQROW(0x0AC1, 4, 12, 0), MQROW(0x0521, 5, 29, 0),
    MQROW(0x0221, 38, 33, 0), MQROW(0x5601, 7, 6, 1),  MQROW(0x5401, 8, 14, 0), MQROW(0x4801, 9, 14, 0), MQROW(0x3801, 10, 14, 0),
    MQROW(0x3001, 11, 17, 0), MQROW(0x2401, 12, 18, 0), MQROW(0x1C01, 13, 20, 0), MQROW(0x1601, 29, 21, 0), MQROW(0x5601, 15, 14, 1),
    MQROW(0x5401, 16, 14, 0), MQROW(0x5101, 17, 15, 0), MQROW(0x4801, 18, 16, 0), MQROW(0x3801, 19, 17, 0), MQROW(0x3401, 20, 18, 0),
    MQROW(0x3001, 21, 19, 0), MQROW(0x2801, 22, 19, 0), MQROW(0x2401, 23, 20, 0), MQROW(0x2201, 24, 21, 0), MQROW(0x1C01, 25, 22, 0),
    MQROW(0x1801, 26, 23, 0), MQROW(0x1601, 27, 24, 0), MQROW(0x1401, 28, 25, 0), MQROW(0x1201, 29, 26, 0), MQROW(0x1101, 30, 27, 0),
    MQROW(0x0AC1, 31, 28, 0), MQROW(0x09C1, 32, 29, 0), MQROW(0x08A1, 33, 30, 0), MQROW(0x0521, 34, 31, 0), MQROW(0x0441, 35, 32, 0),
    MQROW(0x02A1, 36, 33, 0), MQROW(0x0221, 37, 34, 0), MQROW(0x0141, 38, 35, 0), MQROW(0x0111, 39, 36, 0), MQROW(0x0085, 40, 37, 0),
    MQROW(0x0049, 41, 38, 0), MQROW(0x0025, 42, 39, 0), MQROW(0x0015, 43, 40, 0), MQROW(0x0009, 44, 41, 0), MQROW(0x0005, 45, 42, 0),
    MQROW(0x0001, 45, 43, 0), MQROW(0x5601, 46, 46, 0)};
#undef MQROW

constexpr int kCtxZC = 0, kCtxAgg = 17, kCtxUni = 18, kNumCtx = 19;

// Zero-coding contexts (Table D.1) as a look-up by the eight neighbour significance bits: index = row above (x-1, x, x+1) in bits
// 0-2, left and right neighbour in bits 3-4, row below in bits 5-7; one table per sub-band orientation, 4 bits per entry, eight
// entries per dword -> 32 dwords that live across the lanes of ONE register (lane i: entries 8 i .. 8 i + 7) and are read with
// v_readlane like the MQ tables -- the arithmetic form was ~25 scalar instructions of every zero-coding decision's ~110.
constexpr int zc_context(int orient, uint32_t idx)
{
    const uint32_t w0 = idx & 7u, l = (idx >> 3) & 1u, r = (idx >> 4) & 1u, w2 = idx >> 5;
    int hh = (int)l + (int)r;
    int vv = (int)((w0 >> 1) & 1u) + (int)((w2 >> 1) & 1u);
    const int dd = (int)(w0 & 1u) + (int)((w0 >> 2) & 1u) + (int)(w2 & 1u) + (int)((w2 >> 2) & 1u);
    if (orient == 1) { const int t = hh; hh = vv; vv = t; }
    if (orient == 3) {
        const int hv = hh + vv;
        if (dd >= 3) return 8;
        if (dd == 2) return hv >= 1 ? 7 : 6;
        if (dd == 1) return hv >= 2 ? 5 : (hv == 1 ? 4 : 3);
        return hv >= 2 ? 2 : hv;
    }
    if (hh == 2) return 8;
    if (hh == 1) return vv >= 1 ? 7 : (dd >= 1 ? 6 : 5);
    if (vv == 2) return 4;
    if (vv == 1) return 3;
    return dd >= 2 ? 2 : dd;
}
// ... indexed by the NINE bits of a sample's 3 x 3 neighbourhood as they lie in a column's neighbourhood word (row above in bits
// 0-2, own row in 3-5 -- the centre bit does not matter --, row below in 6-8): 512 entries of 4 bits, 64 dwords, one per lane
struct ZcLut {
    uint32_t w[4][64];
    constexpr ZcLut() : w{}
    {
        for (int o = 0; o < 4; ++o)
            for (uint32_t i = 0; i < 512; ++i) {
                const uint32_t w0 = i & 7u, w1 = (i >> 3) & 7u, w2 = i >> 6;
                w[o][i >> 3] |= (uint32_t)zc_context(o, w0 | ((w1 & 1u) << 3) | ((w1 & 4u) << 2) | (w2 << 5)) << (4 * (i & 7u));
            }
    }
};
__device__ const ZcLut g_zc_lut{};
// Sign-coding context and XOR bit (Table D.3) by the significance and sign of the four horizontal / vertical neighbours.  The index
// takes the bits as they lie in the neighbourhood words: significant (up, left, right, down) in bits 0, 2, 4, 6 -- bits 1, 3, 5, 7
// of a 3 x 3 window shifted down by one --, negative in the bit above each; entry = context | xor << 4, one byte each, 64 dwords
// across the lanes of one register.
constexpr uint32_t sign_context(uint32_t idx)
{
    auto contrib = [&](int k) { return ((idx >> (2 * k)) & 1u) ? (((idx >> (2 * k + 1)) & 1u) ? -1 : 1) : 0; };
    int hc = contrib(1) + contrib(2), vc = contrib(0) + contrib(3);
    hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc); vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
    int cxn = 0, xr = 0;
    if (hc == 1)      { cxn = vc == 1 ? 13 : (vc == 0 ? 12 : 11); xr = 0; }
    else if (hc == 0) { cxn = vc == 0 ? 9 : 10; xr = vc == -1; }
    else              { cxn = vc == 1 ? 11 : (vc == 0 ? 12 : 13); xr = 1; }
    return (uint32_t)cxn | ((uint32_t)xr << 4);
}
struct SignLut {
    uint32_t w[64];
    constexpr SignLut() : w{}
    {
        for (uint32_t i = 0; i < 256; ++i) w[i >> 2] |= sign_context(i) << (8 * (i & 3u));
    }
};
__device__ const SignLut g_sign_lut{};

// One block per wave: everything the decoder touches is wave-uniform, so the compiler keeps it on the scalar unit, and
// the two lookups on every decision's dependency chain -- context state and Table C.2 -- come out of VGPRs whose LANE i
// holds entry i (v_readlane / v_writelane with a scalar index: a few cycles) instead of LDS (> 100).
struct MqDec {
    const uint8_t* d; uint32_t len, pos;        // pos = index of the byte the reference's `bp` points at
    uint32_t a, c, ct;
    uint32_t cxv, tabv;                          // lane i holds context byte i (state | mps << 7) / table entry i
    const uint8_t* lo; const uint8_t* hi;        // readable range of the coded buffer
    uint32_t win[4], wbase;                      // the 16 coded bytes [wbase, wbase + 16) of the segment
    __device__ __forceinline__ uint32_t ctx_get(int i) const { return (uint32_t)__builtin_amdgcn_readlane((int)cxv, i); }
    __device__ __forceinline__ void ctx_set(int i, uint32_t v) { cxv = threadIdx.x == (uint32_t)i ? v : cxv; }   // (every lane is active)
    __device__ __forceinline__ uint32_t tab_get(uint32_t i) const { return (uint32_t)__builtin_amdgcn_readlane((int)tabv, (int)i); }
    __device__ __forceinline__ void refill(uint32_t i)
    {   // 16 bytes from the dword-aligned address at or below d + i: one load per ~100 decisions instead of two
        // dependent byte loads per BYTEIN (each of which would also wait for every value store still in flight)
        const uint8_t* p = d + i;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
        const uint8_t* q = p - mis;
        wbase = i - mis;
        if (q >= lo && q + 16 <= hi) {
            const uint32_t* q4 = reinterpret_cast<const uint32_t*>(q);
            win[0] = q4[0]; win[1] = q4[1]; win[2] = q4[2]; win[3] = q4[3];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t v = 0;
                for (int b = 0; b < 4; ++b) { const uint8_t* r = q + k * 4 + b; v |= (uint32_t)((r >= lo && r < hi) ? *r : 0xFFu) << (8 * b); }
                win[k] = v;
            }
        }
    }
    __device__ __forceinline__ uint32_t byte_at(uint32_t i)
    {   // block bytes followed by the artificial 0xFF 0xFF terminator (mqc_dec.cpp:113-118)
        if (i >= len) return 0xFFu;
        if (i - wbase >= 16u) refill(i);
        const uint32_t o = i - wbase;
        const uint32_t lo2 = (o & 8u) ? win[2] : win[0], hi2 = (o & 8u) ? win[3] : win[1];
        return (((o & 4u) ? hi2 : lo2) >> ((o & 3u) * 8u)) & 0xFFu;
    }
    __device__ __forceinline__ void bytein()
    {
        const uint32_t cur = byte_at(pos), nxt = byte_at(pos + 1);
        if (cur == 0xFFu) {
            if (nxt > 0x8Fu) { c += 0xFF00u; ct = 8; }
            else { ++pos; c += nxt << 9; ct = 7; }
        } else { ++pos; c += nxt << 8; ct = 8; }
    }
    __device__ __forceinline__ void reset_states()                 // mqc_resetstates (mqc_dec.cpp:168-175)
    {
        for (int i = 0; i < kNumCtx; ++i) ctx_set(i, 0);
        ctx_set(kCtxUni, 46); ctx_set(kCtxAgg, 3); ctx_set(kCtxZC, 4);
    }
    __device__ __forceinline__ void init_segment()                 // mqc_init_dec (:140-154): the states are kept
    {
        pos = 0; wbase = 0x80000000u;
        c = (len == 0 ? 0xFFu : byte_at(0)) << 16;
        bytein();
        c <<= 7; ct -= 7; a = 0x8000u;
    }
    __device__ __forceinline__ void init_raw_segment() { pos = 0; c = 0; ct = 0; wbase = 0x80000000u; }     // mqc_raw_init_dec (:156-160)
    __device__ __forceinline__ uint32_t raw_decode()               // mqc_raw_decode (mqc_dec_inl.h:55-76)
    {
        if (ct == 0) {
            const uint32_t b = byte_at(pos);
            if (c == 0xFFu) {
                if (b > 0x8Fu) { c = 0xFFu; ct = 8; }              // the terminating marker: ones for ever
                else { c = b; ++pos; ct = 7; }
            } else { c = b; ++pos; ct = 8; }
        }
        --ct;
        return (c >> ct) & 1u;
    }
    // A RUN of decisions in ONE context (the dense mag-ref stripes of a long block: 256 decisions in context 16 in a row):
    // the context's byte and its table row stay in scalars between the decisions, no lane read / write per decision
    // registers a / c / ct through v_readfirstlane: the arithmetic of the run is the scalar unit's, whatever the compiler could
    // prove about the vector loads the bytes came from
    __device__ __forceinline__ static uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
    __device__ __forceinline__ uint32_t decode_run(uint32_t& sa, uint32_t& sc, uint32_t& sct, uint32_t& st, uint32_t& row)
    {
        const uint32_t qe = row & 0xFFFFu, mps = st >> 7;
        uint32_t dbit;
        sa -= qe;
        if ((sc >> 16) < qe) {
            const bool toM = sa < qe;
            dbit = toM ? mps : mps ^ 1u;
            st = (toM ? ((row >> 16) & 0x3Fu) : ((row >> 22) & 0x3Fu)) | ((toM ? mps : mps ^ (row >> 28)) << 7);
            sa = qe;
        } else {
            sc -= qe << 16;
            if (sa & 0x8000u) return mps;
            const bool toL = sa < qe;
            dbit = toL ? mps ^ 1u : mps;
            st = (toL ? ((row >> 22) & 0x3Fu) : ((row >> 16) & 0x3Fu)) | ((toL ? mps ^ (row >> 28) : mps) << 7);
        }
        row = tab_get(st & 0x7Fu);
        uint32_t left = (uint32_t)__builtin_clz(sa) - 16u;          // RENORMD, whole shifts at a time
        sa <<= left;
        while (left) {
            if (sct == 0) {                                          // BYTEIN
                const uint32_t cur = uni(byte_at(pos)), nxt = uni(byte_at(pos + 1));
                if (cur == 0xFFu) {
                    if (nxt > 0x8Fu) { sc += 0xFF00u; sct = 8; }
                    else { ++pos; sc += nxt << 9; sct = 7; }
                } else { ++pos; sc += nxt << 8; sct = 8; }
            }
            const uint32_t sh = left < sct ? left : sct;
            sc <<= sh; sct -= sh; left -= sh;
        }
        return dbit;
    }
    __device__ __forceinline__ uint32_t decode(int ctx)
    {
        const uint32_t st = ctx_get(ctx);
        const uint32_t row = tab_get(st & 0x7Fu);
        const uint32_t qe = row & 0xFFFFu, mps = st >> 7;
        uint32_t dbit;
        a -= qe;
        if ((c >> 16) < qe) {                              // LPS exchange (C.3.2)
            const bool toM = a < qe;
            dbit = toM ? mps : mps ^ 1u;
            const uint32_t nst = toM ? ((row >> 16) & 0x3Fu) : ((row >> 22) & 0x3Fu);
            const uint32_t nm = toM ? mps : mps ^ (row >> 28);
            ctx_set(ctx, nst | (nm << 7));
            a = qe;
        } else {
            c -= qe << 16;
            if (a & 0x8000u) return mps;
            const bool toL = a < qe;
            dbit = toL ? mps ^ 1u : mps;
            const uint32_t nst = toL ? ((row >> 22) & 0x3Fu) : ((row >> 16) & 0x3Fu);
            const uint32_t nm = toL ? mps ^ (row >> 28) : mps;
            ctx_set(ctx, nst | (nm << 7));
        }
        do {                                               // RENORMD
            if (ct == 0) bytein();
            a <<= 1; c <<= 1; --ct;
        } while (a < 0x8000u);
        return dbit;
    }
};

// bits (x-1, x, x+1) of a row bitmap at positions 0, 1, 2
__device__ __forceinline__ uint32_t win3(uint64_t s, uint32_t x)
{
    return (uint32_t)(x ? (s >> (x - 1)) : (s << 1)) & 7u;
}

// Where a block's decoded values live between the passes: a workspace in global memory, one row load / store per stripe row
// and pass.  A block's magnitudes need numbps + 1 bits and a sign, so a block of at most 14 bit planes -- every block of 8- to
// 12-bit content with the default guard bits -- keeps int16 there (half of r02's int32 traffic), deeper ones int32.  The wave
// dequantises and stores the block's rows itself at the end (r02's separate store kernel read the workspace once more).
// (Measured and not kept in r03: the same int16 values in LDS, 8 KB per block -- no workspace traffic at all, but 15 instead of
//  32 waves per CU: the scalar unit this kernel is bound by then idles, 45 -> 61 ms.)
// ---- a dense mag-ref stripe's 4 w decisions in ONE context, hand-scheduled on the scalar unit (r04) -------------------------
// The chains of a frame's longest blocks (the LL band: 15 of 17 bit-planes are nothing but such stripes) set the frame's decode
// time at (instructions per decision) x ~8.5 cycles; the compiler's form of MqDec::decode_run is ~45 instructions per decision,
// this one ~28: DECODE with the LPS / MPS exchange as two s_cselect pairs, the table row of the next state by v_readlane,
// RENORMD by s_flbit + whole shifts, BYTEIN out of an 8-byte scalar window refilled by s_load_dword (four bytes at a time,
// bytes past the segment's end as 0xFF: the reference's artificial terminator).  State in / out: a, c, ct, the context's
// state index / MPS / table row; bits: four 64-bit rows (row j: decisions of the stripe's row j, bit x).
struct DenseRun {
    uint32_t a, c, ct, idx, mps, row;          // MQ registers, context 16's state
    uint32_t wlo, whi, nv;                     // stream window: byte k of the next nv bytes (4 < nv <= 8) at bits 8 k
    uint32_t foff, flen;                       // offset of the next dword to fetch from fbase; the segment's end on the same scale
    uint32_t used;                             // bytes consumed (advance of MqDec::pos)
    const uint8_t* fbase;                      // dword-aligned
    uint64_t b0, b1, b2, b3;
};

#define T1_DENSE_DECISION(J, BJ)                                                                                        \
    "s_sub_u32 %[a], %[a], %[qe]\n\t"                                                                                   \
    "s_cmp_lt_u32 %[c], %[qe16]\n\t"                                                                                    \
    "s_cbranch_scc1 Llps" J "_%=\n\t"                                                                                   \
    "s_sub_u32 %[c], %[c], %[qe16]\n\t"                                                                                 \
    "s_bitcmp1_b32 %[a], 15\n\t"                                                                                        \
    "s_cbranch_scc1 Lplain" J "_%=\n\t"                                                                                 \
    "s_cmp_lt_u32 %[a], %[qe]\n\t"                       /* MPS path, renormalising: conditional exchange */          \
    "s_cselect_b32 %[sh], 22, 16\n\t"                                                                                   \
    "s_cselect_b32 %[fl], 1, 0\n\t"                                                                                     \
    "s_branch Lupd" J "_%=\n"                                                                                           \
    "Llps" J "_%=:\n\t"                                                                                                 \
    "s_cmp_lt_u32 %[a], %[qe]\n\t"                                                                                      \
    "s_cselect_b32 %[sh], 16, 22\n\t"                                                                                   \
    "s_cselect_b32 %[fl], 0, 1\n\t"                                                                                     \
    "s_mov_b32 %[a], %[qe]\n"                                                                                           \
    "Lupd" J "_%=:\n\t"                                                                                                 \
    "s_xor_b32 %[d], %[mps], %[fl]\n\t"                  /* the decision */                                            \
    "s_lshr_b32 %[t], %[row], 28\n\t"                                                                                   \
    "s_and_b32 %[t], %[t], %[fl]\n\t"                                                                                   \
    "s_xor_b32 %[mps], %[mps], %[t]\n\t"                 /* SWITCH */                                                  \
    "s_lshr_b32 %[t], %[row], %[sh]\n\t"                                                                                \
    "s_and_b32 %[idx], %[t], 63\n\t"                                                                                    \
    "s_nop 3\n\t"                                                                                                       \
    "v_readlane_b32 %[row], %[tab], %[idx]\n\t"                                                                         \
    "s_flbit_i32_b32 %[n], %[a]\n\t"                                                                                    \
    "s_sub_u32 %[n], %[n], 16\n\t"                                                                                      \
    "s_lshl_b32 %[a], %[a], %[n]\n\t"                                                                                   \
    "s_and_b32 %[qe], %[row], 0xffff\n\t"                                                                               \
    "s_lshl_b32 %[qe16], %[qe], 16\n"                                                                                   \
    "Lshift" J "_%=:\n\t"                                                                                               \
    "s_min_u32 %[sh], %[n], %[ct]\n\t"                                                                                  \
    "s_lshl_b32 %[c], %[c], %[sh]\n\t"                                                                                  \
    "s_sub_u32 %[ct], %[ct], %[sh]\n\t"                                                                                 \
    "s_sub_u32 %[n], %[n], %[sh]\n\t"                                                                                   \
    "s_cmp_eq_u32 %[n], 0\n\t"                                                                                          \
    "s_cbranch_scc1 Lrec" J "_%=\n\t"                                                                                   \
    /* BYTEIN (ct == 0): cur = byte 0 of the window, nxt = byte 1 */                                                   \
    "s_and_b32 %[t], %[wlo], 0xff\n\t"                                                                                  \
    "s_bfe_u32 %[fl], %[wlo], 0x80008\n\t"                                                                              \
    "s_cmp_eq_u32 %[t], 0xff\n\t"                                                                                       \
    "s_cbranch_scc1 Lff" J "_%=\n\t"                                                                                    \
    "s_lshl_b32 %[fl], %[fl], 8\n\t"                                                                                    \
    "s_add_u32 %[c], %[c], %[fl]\n\t"                                                                                   \
    "s_mov_b32 %[ct], 8\n\t"                                                                                            \
    "s_branch Ladv" J "_%=\n"                                                                                           \
    "Lff" J "_%=:\n\t"                                                                                                  \
    "s_cmp_gt_u32 %[fl], 0x8f\n\t"                                                                                      \
    "s_cbranch_scc1 Lmark" J "_%=\n\t"                                                                                  \
    "s_lshl_b32 %[fl], %[fl], 9\n\t"                                                                                    \
    "s_add_u32 %[c], %[c], %[fl]\n\t"                                                                                   \
    "s_mov_b32 %[ct], 7\n"                                                                                              \
    "Ladv" J "_%=:\n\t"                                   /* one byte consumed */                                       \
    "s_lshr_b32 %[wlo], %[wlo], 8\n\t"                                                                                  \
    "s_lshl_b32 %[t], %[whi], 24\n\t"                                                                                   \
    "s_or_b32 %[wlo], %[wlo], %[t]\n\t"                                                                                 \
    "s_lshr_b32 %[whi], %[whi], 8\n\t"                                                                                  \
    "s_add_u32 %[used], %[used], 1\n\t"                                                                                 \
    "s_sub_u32 %[nv], %[nv], 1\n\t"                                                                                     \
    "s_cmp_lg_u32 %[nv], 4\n\t"                                                                                         \
    "s_cbranch_scc1 Lshift" J "_%=\n\t"                                                                                 \
    /* four more bytes: the dword at fbase + foff; bytes at or past flen read as 0xFF */                               \
    "s_mov_b32 %[whi], -1\n\t"                                                                                          \
    "s_cmp_ge_u32 %[foff], %[flen]\n\t"                                                                                 \
    "s_cbranch_scc1 Lfed" J "_%=\n\t"                                                                                   \
    "s_load_dword %[whi], %[fbase], %[foff]\n\t"                                                                        \
    "s_sub_u32 %[t], %[flen], %[foff]\n\t"               /* bytes of it inside the segment: 1 .. */                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                          \
    "s_cmp_ge_u32 %[t], 4\n\t"                                                                                          \
    "s_cbranch_scc1 Lfed" J "_%=\n\t"                                                                                   \
    "s_lshl_b32 %[t], %[t], 3\n\t"                                                                                      \
    "s_lshl_b32 %[t], -1, %[t]\n\t"                                                                                     \
    "s_or_b32 %[whi], %[whi], %[t]\n"                                                                                   \
    "Lfed" J "_%=:\n\t"                                                                                                 \
    "s_add_u32 %[foff], %[foff], 4\n\t"                                                                                 \
    "s_mov_b32 %[nv], 8\n\t"                                                                                            \
    "s_branch Lshift" J "_%=\n"                                                                                         \
    "Lmark" J "_%=:\n\t"                                  /* 0xFF then > 0x8F: a marker -- ones for ever, nothing consumed */ \
    "s_add_u32 %[c], %[c], 0xff00\n\t"                                                                                  \
    "s_mov_b32 %[ct], 8\n\t"                                                                                            \
    "s_branch Lshift" J "_%=\n"                                                                                         \
    "Lplain" J "_%=:\n\t"                                                                                               \
    "s_mov_b32 %[d], %[mps]\n"                                                                                          \
    "Lrec" J "_%=:\n\t"                                                                                                 \
    "s_lshl_b32 %[t], %[d], %[xs]\n\t"                                                                                  \
    "s_or_b32 " BJ ", " BJ ", %[t]\n\t"

// 32 columns (or fewer) of a stripe: bits of row j into the low / high half picked by the caller (xs = x & 31)
__device__ __forceinline__ void dense_run_half(DenseRun& r, uint32_t tabv, uint32_t ncols, uint32_t& h0, uint32_t& h1, uint32_t& h2, uint32_t& h3)
{
    uint32_t qe = r.row & 0xFFFFu, qe16 = qe << 16, sh, fl, d, t, n, xs = 0;
    asm volatile(
        "Lcol_%=:\n\t"
        T1_DENSE_DECISION("0", "%[h0]")
        T1_DENSE_DECISION("1", "%[h1]")
        T1_DENSE_DECISION("2", "%[h2]")
        T1_DENSE_DECISION("3", "%[h3]")
        "s_add_u32 %[xs], %[xs], 1\n\t"
        "s_cmp_lt_u32 %[xs], %[ncols]\n\t"
        "s_cbranch_scc1 Lcol_%=\n\t"
        : [a] "+s"(r.a), [c] "+s"(r.c), [ct] "+s"(r.ct), [idx] "+s"(r.idx), [mps] "+s"(r.mps), [row] "+s"(r.row),
          [wlo] "+s"(r.wlo), [whi] "+s"(r.whi), [nv] "+s"(r.nv), [foff] "+s"(r.foff), [used] "+s"(r.used),
          [h0] "+s"(h0), [h1] "+s"(h1), [h2] "+s"(h2), [h3] "+s"(h3),
          [qe] "+s"(qe), [qe16] "+s"(qe16), [xs] "+s"(xs),
          [sh] "=&s"(sh), [fl] "=&s"(fl), [d] "=&s"(d), [t] "=&s"(t), [n] "=&s"(n)
        : [tab] "v"(tabv), [ncols] "s"(ncols), [flen] "s"(r.flen), [fbase] "s"(r.fbase)
        : "scc", "memory");
}

constexpr uint32_t kNarrowPlanes = 14;

// (the kernel's body as a device function: kernels_t1lanes.hip includes this file -- GRK_T1_FUSED_INCLUDE -- and runs it as the first
//  workgroups of ONE launch that also holds the lane decoder's waves: a frame's block decoding on one stream)
template <bool IRREV>
__device__ __forceinline__ void t1_dec_block(const T1DecArgs& a, const uint32_t bidx)
{
    __shared__ uint64_t bm_l[4][66];
    const uint32_t tabv0 = threadIdx.x < 47 ? g_mq_table[threadIdx.x] : 0u;       // Table C.2 across the lanes
    // all 64 lanes run the same (uniform) program -- lane-resident tables need every lane's registers to stay live
    // through the compiler's copies -- and only lane 0 performs the side effects
    const bool writer = threadIdx.x == 0;
    const uint32_t blk = a.list ? a.list[bidx] : bidx;
    if (blk >= a.nblocks) return;
    // beside the lane decoder this kernel holds the frame's longest chains: its waves win the issue arbitration of their SIMDs
    if (a.list) __builtin_amdgcn_s_setprio(3);
    const HtDecBlock in = a.table[blk];
    const HtBlockDesc bd = a.blocks[blk % a.blocks_per_tile];
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t orient = bd.pad;                        // 0 LL, 1 HL, 2 LH, 3 HH
    const uint32_t zcv = g_zc_lut.w[orient & 3u][threadIdx.x & 63u];           // this orientation's zero-coding contexts across the lanes
    const uint32_t sgv = g_sign_lut.w[threadIdx.x & 63u];                      // sign-coding contexts
    const uint32_t numbps = in.missing_msbs & 0xFFu, numpasses = in.missing_msbs >> 8;
    // value workspace of a deep block, [y * 64 + x] (the first pass writes every row)
    int32_t* ws = a.work + (size_t)blk * 4096u;
    if (in.length == 0 && in.missing_msbs == kSkipBlock) return;       // region decode: outside the decoded region
    const uint32_t tile = blk / a.blocks_per_tile;
    int32_t* const dst = a.mallat + ((size_t)tile * a.ncomp + bd.comp) * a.pitch + (size_t)bd.py * a.stride + bd.px;
    if (in.length == 0 || numpasses == 0 || numbps == 0 || numbps >= 25u) {   // absent (or beyond k_max_bit_planes, t1_common.h:70): zeros
        if (numbps >= 25u && in.length != 0 && numpasses != 0 && writer) atomicOr(a.status, 4u);
        if (threadIdx.x < w)
            for (uint32_t y = 0; y < h; ++y) dst[(size_t)y * a.stride + threadIdx.x] = 0;
        return;
    }
    const bool narrow = numbps <= kNarrowPlanes;
    int16_t* const ws16 = reinterpret_cast<int16_t*>(ws);

    // row bitmaps with one border row above and below (index y + 1)
    uint64_t* const sig = bm_l[0]; uint64_t* const neg = bm_l[1]; uint64_t* const pi = bm_l[2]; uint64_t* const mu = bm_l[3];
    for (int i = (int)threadIdx.x; i < 66; i += 64) { sig[i] = 0; neg[i] = 0; pi[i] = 0; mu[i] = 0; }
    __syncthreads();

    MqDec mq;
    mq.lo = a.coded; mq.hi = a.coded + a.coded_bytes;
    mq.tabv = tabv0; mq.cxv = 0;
    mq.reset_states();
    const bool lazy = (a.cblksty & 0x01u) != 0, reset = (a.cblksty & 0x02u) != 0, vsc = (a.cblksty & 0x08u) != 0,
               segsym = (a.cblksty & 0x20u) != 0;
    // codeword segments: the caller's list, or the whole block as one segment (T1.cpp:1280-1292)
    uint32_t sg = a.seg_first ? a.seg_first[blk] : 0u;
    const uint32_t sg_end = a.seg_first ? a.seg_first[blk + 1] : 1u;
    uint32_t seg_off = 0;

    // ---- the passes work stripe by stripe with the stripe's state on the LANES (nbv / nnv: significance / sign
    //      neighbourhoods of the lane's column over the stripe's rows + the rows above and below; pv / mv: its visited /
    //      refined bits).  Columns that cannot code anything are skipped with a candidate mask (a ballot).
    const uint64_t wmask = w >= 64 ? ~0ull : ((1ull << w) - 1ull);

    int bp = (int)numbps, type = 2;
    bool first_pass = true;                                 // nothing in the workspace yet
    const uint32_t tl = threadIdx.x;
    for (; sg < sg_end; ++sg) {
    const uint32_t seg_len = a.seg_first ? a.segs[sg].x : in.length, seg_passes = a.seg_first ? a.segs[sg].y : numpasses;
    const bool raw_seg = lazy && bp <= (int)numbps - 4 && type < 2;           // decided where the segment starts
    mq.d = a.coded + in.offset + seg_off; mq.len = seg_len;
    seg_off += seg_len;
    if (raw_seg) mq.init_raw_segment(); else mq.init_segment();
    for (uint32_t p = 0; p < seg_passes && bp >= 1; ++p) {
        const bool raw = raw_seg && type < 2;
        const int32_t one = 1 << bp, oph = one | (one >> 1), poshalf = one >> 1;
        for (uint32_t k = 0; k < h; k += 4) {
            // "coded in an earlier pass of this plane" (pi): lane x holds column x's four bits
            uint32_t pv = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) pv |= ((uint32_t)(pi[k + 1 + j] >> tl) & 1u) << j;
            // LANE x keeps column x's significance neighbourhood: bit 3 r + c = column x - 1 + c of row S[r] (r = 0: the row above
            // the stripe ... 5: the row below).  One v_readlane per column then gives every window of the column -- the scalar unit,
            // the scarce one here, would spend three 64-bit shifts with a select per SAMPLE on them -- and the vector unit keeps the
            // three lanes around a sample that turns significant up to date.
            uint32_t nbv = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {          // (vertically causal: a stripe never sees the one below, T1.cpp:198-221)
                const uint64_t sr6 = (vsc && r == 5) ? 0ull : sig[k + r];
                nbv |= ((uint32_t)(tl ? (sr6 >> (tl - 1u)) : (sr6 << 1)) & 7u) << (3 * r);
            }
            // the same for the signs (bit set: negative).  The sign rows are only read here: they stay in LDS, where lane 0 sets the
            // bit of a sample that turns out negative (one wave: its LDS operations keep their order, so the next stripe reads it)
            uint32_t nnv = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const uint64_t nr6 = (vsc && r == 5) ? 0ull : neg[k + r];
                nnv |= ((uint32_t)(tl ? (nr6 >> (tl - 1u)) : (nr6 << 1)) & 7u) << (3 * r);
            }
            const uint32_t nr = min(4u, h - k);
            // the stripe's decoded values live in registers, lane <-> column: one coalesced row load at the start (not
            // in the first pass) and one coalesced row store at the end instead of a store / an atomic per sample
            int32_t V[4] = {0, 0, 0, 0};
            if (!first_pass) {
                if (narrow) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if ((uint32_t)j < nr) V[j] = ws16[(k + j) * 64u + tl];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if ((uint32_t)j < nr) V[j] = ws[(k + j) * 64u + tl];
                }
            }
            // rows of the stripe that do not exist behave as "already coded"
#pragma unroll
            for (int j = 0; j < 4; ++j) if ((uint32_t)j >= nr) pv |= 1u << j;          // ("visited": never a candidate in any pass)

            // per-sample pieces (j is a compile-time constant after unrolling)
#define T1_SIGN_AND_SET(j, x)                                                                                     \
            {                                                                                                     \
                const uint32_t sidx = (((nbx >> (3 * (j))) & 0xAAu) >> 1) | ((nnx >> (3 * (j))) & 0xAAu);           \
                const uint32_t se = ((uint32_t)__builtin_amdgcn_readlane((int)sgv, (int)(sidx >> 2)) >> (8u * (sidx & 3u))) & 0xFFu; \
                const int cxn = (int)(se & 0xFu), xr = (int)(se >> 4);                                             \
                const uint32_t ng = raw ? mq.raw_decode() : (mq.decode(cxn) ^ (uint32_t)xr);                      \
                { const int32_t sm = -(int32_t)ng; V[(j)] = tl == (x) ? (oph ^ sm) - sm : V[(j)]; }                \
                if (ng && writer) neg[k + 1 + (j)] |= 1ull << (x);                                                \
                nbx |= 1u << (3 * ((j) + 1) + 1);                                                                 \
                nnx |= ng << (3 * ((j) + 1) + 1);                                                                 \
                { const uint32_t dl = tl - (x) + 1u, pat = dl < 3u ? (4u << (3 * ((j) + 1))) >> dl : 0u;           \
                  nbv |= pat; nnv |= ng ? pat : 0u; }                                                             \
            }
            auto zc_ctx9 = [&](uint32_t nine) -> int {             // Table D.1, looked up (zc_context above)
                return (int)(((uint32_t)__builtin_amdgcn_readlane((int)zcv, (int)(nine >> 3)) >> (4u * (nine & 7u))) & 0xFu);
            };

            if (type == 0) {                                           // significance propagation (T1.cpp:1024-1152)
                // candidate columns -- an uncoded, insignificant sample with a significant neighbour -- from the lanes' words;
                // looked for again only after a column in which a sample turned significant (nothing else makes new candidates)
                auto candidates = [&]() -> uint64_t {
                    uint32_t cand = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t t9 = nbv >> (3 * j);
                        cand |= ((t9 & 0x1EFu) != 0u ? 1u : 0u) & ~((t9 >> 4) | (pv >> j));
                    }
                    return __builtin_amdgcn_ballot_w64((cand & 1u) != 0) & wmask;
                };
                uint64_t cm = candidates();
                while (cm) {
                    const uint32_t x = (uint32_t)__ffsll((long long)cm) - 1u;
                    uint32_t nbx = (uint32_t)__builtin_amdgcn_readlane((int)nbv, (int)x), nnx = (uint32_t)__builtin_amdgcn_readlane((int)nnv, (int)x);
                    const uint32_t nbx0 = nbx;
                    const uint32_t pvx = (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)x);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (((pvx >> j) | (nbx >> (3 * j + 4))) & 1u) continue;
                        const uint32_t nine = (nbx >> (3 * j)) & 0x1FFu;
                        if (!(nine & 0x1EFu)) continue;                         // no significant neighbour
                        if (raw ? mq.raw_decode() : mq.decode(kCtxZC + zc_ctx9(nine))) T1_SIGN_AND_SET(j, x)
                        pv |= tl == x ? 1u << j : 0u;
                    }
                    const uint64_t beyond = (~1ull) << x;                  // columns right of x
                    cm = (nbx != nbx0 ? candidates() : cm) & beyond;
                }
            } else if (type == 1) {                                    // magnitude refinement (T1.cpp:1160-1255)
                // "refined before" (Table D.4) is this pass's own state: its rows stay in LDS, lane x holds column x's four bits
                uint32_t mv = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) mv |= ((uint32_t)(mu[k + 1 + j] >> tl) & 1u) << j;
                uint64_t cm = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) cm |= __builtin_amdgcn_ballot_w64((((nbv >> (3 * j + 4)) & ~(pv >> j)) & 1u) != 0);
                cm &= wmask;
                // a DENSE stripe -- every sample of it significant since an earlier plane and refined before (the lower planes of a
                // frame's LL band: 15 of 17 planes are nothing else): 4 w decisions in context 16 one after the other.  They run as one
                // scalar loop over the MQ registers, the bits collected in four 64-bit rows and applied to the lanes' values afterwards
                // -- ~25 instead of ~80 instructions per decision on the longest chains of the frame
                const bool dense = !raw && nr == 4 &&
                    __builtin_amdgcn_ballot_w64(tl >= w || (((nbv >> 4) & (nbv >> 7) & (nbv >> 10) & (nbv >> 13) & 1u) != 0 && pv == 0u && mv == 0xFu)) == ~0ull;
                if (dense) {
                    uint64_t rb0 = 0, rb1 = 0, rb2 = 0, rb3 = 0;
                    uint32_t st = mq.ctx_get(16), row = mq.tab_get(st & 0x7Fu);
                    // the scalar window reads whole dwords: the segment's last one must lie inside the coded buffer
                    const uint8_t* const seg_end4 = reinterpret_cast<const uint8_t*>((reinterpret_cast<uintptr_t>(mq.d + mq.len) + 3u) & ~(uintptr_t)3u);
                    if (seg_end4 <= mq.hi && mq.d >= mq.lo) {
                        DenseRun r;
                        r.a = MqDec::uni(mq.a); r.c = MqDec::uni(mq.c); r.ct = MqDec::uni(mq.ct);
                        r.idx = st & 0x7Fu; r.mps = st >> 7; r.row = row;
                        // bytes pos .. pos + nv - 1 into the window, nv such that the next fetch is dword-aligned
                        const uint32_t skew = (uint32_t)(reinterpret_cast<uintptr_t>(mq.d) & 3u);
                        const uint32_t p0 = mq.pos;
                        r.nv = 8u - ((p0 + skew) & 3u);                       // 5 .. 8
                        uint32_t wl = 0, wh = 0;
#pragma unroll
                        for (uint32_t k = 0; k < 8; ++k) {
                            const uint32_t bk = k < r.nv ? MqDec::uni(mq.byte_at(p0 + k)) : 0u;
                            if (k < 4) wl |= bk << (8u * k); else wh |= bk << (8u * (k - 4u));
                        }
                        r.wlo = wl; r.whi = wh;
                        r.fbase = mq.d - skew;                                // dword-aligned
                        r.foff = p0 + skew + r.nv;                            // (a multiple of four)
                        r.flen = mq.len + skew;
                        r.used = 0;
                        uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0, u0 = 0, u1 = 0, u2 = 0, u3 = 0;
                        dense_run_half(r, mq.tabv, w < 32u ? w : 32u, l0, l1, l2, l3);
                        if (w > 32u) dense_run_half(r, mq.tabv, w - 32u, u0, u1, u2, u3);
                        rb0 = l0 | ((uint64_t)u0 << 32); rb1 = l1 | ((uint64_t)u1 << 32);
                        rb2 = l2 | ((uint64_t)u2 << 32); rb3 = l3 | ((uint64_t)u3 << 32);
                        mq.a = r.a; mq.c = r.c; mq.ct = r.ct; mq.pos = p0 + r.used;
                        st = r.idx | (r.mps << 7);
                    } else {
                        uint32_t sa = MqDec::uni(mq.a), sc = MqDec::uni(mq.c), sct = MqDec::uni(mq.ct);
                        for (uint32_t x = 0; x < w; ++x) {
                            rb0 |= (uint64_t)mq.decode_run(sa, sc, sct, st, row) << x;
                            rb1 |= (uint64_t)mq.decode_run(sa, sc, sct, st, row) << x;
                            rb2 |= (uint64_t)mq.decode_run(sa, sc, sct, st, row) << x;
                            rb3 |= (uint64_t)mq.decode_run(sa, sc, sct, st, row) << x;
                        }
                        mq.a = sa; mq.c = sc; mq.ct = sct;
                    }
                    mq.ctx_set(16, st);
                    const uint64_t rbs[4] = {rb0, rb1, rb2, rb3};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t b = (uint32_t)(rbs[j] >> tl) & 1u, isneg = (nnv >> (3 * j + 4)) & 1u;
                        const int32_t dm = (int32_t)(b ^ isneg) - 1, dv = (poshalf ^ dm) - dm;
                        V[j] += tl < w ? dv : 0;
                    }
                    cm = 0;
                }
                while (cm) {
                    const uint32_t x = (uint32_t)__ffsll((long long)cm) - 1u;
                    cm &= cm - 1;
                    const uint32_t nbx = (uint32_t)__builtin_amdgcn_readlane((int)nbv, (int)x), nnx = (uint32_t)__builtin_amdgcn_readlane((int)nnv, (int)x);
                    const uint32_t mvx = (uint32_t)__builtin_amdgcn_readlane((int)mv, (int)x), pvx = (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)x);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!((nbx >> (3 * j + 4)) & ~(pvx >> j) & 1u)) continue;
                        const int cxn = ((mvx >> j) & 1u) ? 16 : (((nbx >> (3 * j)) & 0x1EFu) ? 15 : 14);    // Table D.4
                        const uint32_t b = raw ? mq.raw_decode() : mq.decode(cxn);
                        const uint32_t isneg = (nnx >> (3 * j + 4)) & 1u;        // the value's sign, without reading it back
                        const int32_t dm = (int32_t)(b ^ isneg) - 1, dv = (poshalf ^ dm) - dm;     // +half | -half
                        V[j] += tl == x ? dv : 0;
                        mv |= tl == x ? 1u << j : 0u;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {                          // the rows back to LDS: one ballot each
                    const uint64_t row = __builtin_amdgcn_ballot_w64(((mv >> j) & 1u) != 0);
                    if (writer) mu[k + 1 + j] = row;
                }
            } else {                                                   // cleanup (T1.cpp:854-1007)
                uint64_t cm = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) cm |= __builtin_amdgcn_ballot_w64(((~((nbv >> (3 * j + 4)) | (pv >> j))) & 1u) != 0);
                cm &= wmask;
                while (cm) {
                    const uint32_t x = (uint32_t)__ffsll((long long)cm) - 1u;
                    cm &= cm - 1;
                    uint32_t first = 0;                                // first row still to be coded normally
                    uint32_t nbx = (uint32_t)__builtin_amdgcn_readlane((int)nbv, (int)x), nnx = (uint32_t)__builtin_amdgcn_readlane((int)nnv, (int)x);
                    const uint32_t pvx = (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)x);
                    if (nr == 4) {                                     // run-length mode: whole column quiet (D.3.4) --
                        // nothing significant in the column's 3 x 6 neighbourhood (the lane's word is exactly that) and none coded
                        if (nbx == 0 && pvx == 0) {
                            if (!mq.decode(kCtxAgg)) continue;
                            uint32_t r = mq.decode(kCtxUni);
                            r = (r << 1) | mq.decode(kCtxUni);
#pragma unroll
                            for (int j = 0; j < 4; ++j) if ((uint32_t)j == r) T1_SIGN_AND_SET(j, x)
                            first = r + 1;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if ((uint32_t)j < first || (((pvx >> j) | (nbx >> (3 * j + 4))) & 1u)) continue;
                        if (mq.decode(kCtxZC + zc_ctx9((nbx >> (3 * j)) & 0x1FFu))) T1_SIGN_AND_SET(j, x)
                    }
                }
                pv = 0;                                                // the plane is complete
            }
#undef T1_SIGN_AND_SET
            if (narrow) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if ((uint32_t)j < nr) ws16[(k + j) * 64u + tl] = (int16_t)V[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if ((uint32_t)j < nr) ws[(k + j) * 64u + tl] = V[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t srow = __builtin_amdgcn_ballot_w64(((nbv >> (3 * j + 4)) & 1u) != 0);      // the row, from the lanes' centre bits
                const uint64_t prow = __builtin_amdgcn_ballot_w64(((pv >> j) & 1u) != 0);
                if (writer) { sig[k + 1 + j] = srow; pi[k + 1 + j] = prow; }
            }
        }
        if (type == 2 && segsym)                               // dec_clnpass_check_segsym (:977-993): 0xA expected, only warned about
            for (int i = 0; i < 4; ++i) (void)mq.decode(kCtxUni);
        if (reset && !raw_seg) mq.reset_states();
        first_pass = false;
        if (++type == 3) { type = 0; --bp; }
    }
    }
    // ---- the block leaves dequantised (ShiftFilter: v / 2 truncating toward zero; ScaleFilter: v x stepsize / 2 --
    //      filters/PostDecompressFilters.h:26-35, :60-71), lane <-> column, coalesced rows
    __syncthreads();
    if (threadIdx.x < w) {
        const float scale = bd.inv_step / 2;
        for (uint32_t y = 0; y < h; ++y) {
            const int32_t v = first_pass ? 0 : (narrow ? (int32_t)ws16[y * 64u + threadIdx.x] : ws[y * 64u + threadIdx.x]);
            int32_t o;
            if constexpr (IRREV) o = __float_as_int(__fmul_rn((float)v, scale));
            else o = v / 2;
            dst[(size_t)y * a.stride + threadIdx.x] = o;
        }
    }
}

#ifndef GRK_T1_FUSED_INCLUDE
template <bool IRREV>
__global__ void t1_dec_kernel(T1DecArgs a) { t1_dec_block<IRREV>(a, blockIdx.x); }
#endif

} // namespace

#ifndef GRK_T1_FUSED_INCLUDE
hipError_t launch_t1_decode(const T1DecArgs& a, hipStream_t s)
{
    const uint32_t n = a.list ? a.count : a.nblocks;
    if (!n) return hipSuccess;
    if (a.irreversible) hipLaunchKernelGGL(t1_dec_kernel<true>, dim3(n), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(t1_dec_kernel<false>, dim3(n), dim3(64), 0, s, a);
    return hipGetLastError();
}
#endif

} // namespace grk_amd
