// grok_amd/csrc/kernels_htdec.hip -- K5: HTJ2K cleanup-pass decoder + dequantisation, gfx950.
//
// Replaces T1HT::decompress -> ojph_decode_codeblock (t1/t1_ht/T1HT.cpp:129-179,
// t1/t1_ht/coding/ojph_block_decoder.cpp:989-1625, cleanup pass only -- Grok passes lengths2 = 0)
// and the ShiftHTFilter / ScaleHTFilter post-processing (filters/PostDecompressFilters.h:94-140).
//
// The reference decodes a block strictly serially.  The dependences are of two kinds and the
// work is split along them:
//
//  K5a ht_dec_vlc_kernel -- ONE LANE PER CODE-BLOCK.  MEL and VLC/UVLC decoding is a chain through
//      the whole block (every codeword's position depends on the previous one, every context on the
//      previous quad), so a wavefront gives it no parallelism; instead each lane walks the VLC and
//      MEL segments of its own block and emits one word per quad: the CxtVLC table entry
//      (rho, u_off, e_k, e_1) and u_q + 1.  Few lanes per wave are used on purpose when there are
//      few blocks, so that several waves per SIMD hide each other's latency.
//  K5b ht_dec_ms_kernel  -- ONE WAVEFRONT PER CODE-BLOCK.  The MagSgn segment is un-stuffed in
//      parallel into LDS (byte widths 8/7 -> prefix sum -> ds_or); then, one quad row per
//      iteration with lane <-> sample column, the exponent bound of the row above gives kappa,
//      a wave prefix sum of the bit counts m_n gives every sample its bit offset, and the
//      magnitudes are extracted, dequantised and stored to the component's Mallat plane as
//      coalesced rows.  Only the exponents of a row feed the next row.
//
// Results equal the reference's on every stream a conforming encoder produces; streams the
// reference rejects (bad Scup, U_q > missing_msbs) are rejected here as well.
#include <atomic>
#include <mutex>
#include "kernels.h"
#include "ht_vlc_tables.h"
#include <type_traits>

namespace grk_amd {

namespace {

constexpr uint32_t kQuadStride = 32;         // quads per row in the per-block quad-info array (blocks <= 64 wide)
constexpr uint32_t kQuadWords  = 32 * 32;

// ---- bit readers of K5a --------------------------------------------------------------------------------
// r03 / r04 ran a kernel of its own in front of K5a (K5p, one wave per block) that un-stuffed every block's VLC and MEL bytes into a
// global scratch K5a then read: 86 MB written and read back per 8K frame and 0.13 ms of kernel.  Since r05 the lane that walks a
// block un-stuffs its own bytes: a 64-bit shift register per stream, refilled FOUR coded bytes at a time -- the dword is taken as it
// is unless one of its bytes is a 7-bit byte (one SWAR test; on real streams one dword in ~70 has one) --, the VLC bytes of a quad
// row fetched in one batch at the row's start and handed on through the lane's LDS row (the pair loop must not wait on global
// loads: gfx9 counts loads and stores with one counter, see below), the MEL bytes -- a few per row -- a refill ahead of their use.
//   VLC (ojph_block_decoder.cpp rev_read / rev_init :380-421): read backwards from D[lcup - 3]; first the upper nibble of D[lcup - 2]
//       (3 bits if its low three are ones, else 4); a byte that follows a byte > 0x8F carries 7 bits when its low 7 are ones.
//       LSB first; zeros behind the end.
//   MEL (mel_read / mel_init :270-330): forward from D[lcup - scup], scup - 1 bytes, the last one with its low nibble set; a byte
//       after 0xFF carries 7 bits (its MSB is dropped).  MSB first; ones behind the end.
__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v)
{
    (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += dpp0<0x111, 0xF>(v);
    v += dpp0<0x112, 0xF>(v);
    v += dpp0<0x114, 0xF>(v);
    v += dpp0<0x118, 0xF>(v);
    v += dpp0<0x142, 0xA>(v);
    v += dpp0<0x143, 0xC>(v);
    return v;
}

// lcup, scup of a block as K5a takes them from its last two bytes (:1067-1090); false: the block is rejected
__device__ __forceinline__ bool ht_segments(const HtDecArgs& a, uint32_t blk, const HtDecBlock& in, int& lcup, int& scup)
{
    const uint8_t* D = a.coded + in.offset;
    lcup = (int)in.length - (a.refine ? (int)a.refine[blk].x : 0);      // (refinement passes: their segment follows the cleanup segment)
    if (in.missing_msbs > 29 || lcup < 2) return false;
    scup = ((int)D[lcup - 1] << 4) + (D[lcup - 2] & 0xF);
    return !(scup < 2 || scup > lcup || scup > 4079);
}

typedef uint32_t u32_unaligned __attribute__((aligned(1)));
// bytes q .. q + 3, little endian; a byte outside the coded buffer reads as 0 (the caller masks what lies outside its segment anyway)
__device__ __forceinline__ uint32_t ld4_guarded(const uint8_t* q, const uint8_t* lo, const uint8_t* hi)
{
    if (q >= lo && q + 4 <= hi) return *reinterpret_cast<const u32_unaligned*>(q);
    return (q + 0 >= lo && q + 0 < hi ? (uint32_t)q[0] : 0u) | (q + 1 >= lo && q + 1 < hi ? (uint32_t)q[1] << 8 : 0u) |
           (q + 2 >= lo && q + 2 < hi ? (uint32_t)q[2] << 16 : 0u) | (q + 3 >= lo && q + 3 < hi ? (uint32_t)q[3] << 24 : 0u);
}

// ---- K5a --------------------------------------------------------------------------------------------
// One lane per code-block walks the block's VLC and MEL bits and emits one word per
// quad, 16 bits: what K5b needs of the CxtVLC table entry (9 bits) | (u_q + 1) << 9.  The kernel is one dependent chain per lane, issue-bound at ~8.5 cycles per
// instruction while a SIMD holds <= 2 waves, so it is written for instruction COUNT: the pair's 32 VLC bits are the
// low word of the shift register, plain 32-bit shifts inside the pair, the
// neighbour-significance part of both quads' contexts out of one shift-or of the row above, table entries that carry what
// the chain needs next in place (the next quad's context bits at their position in the table ADDRESS, the quad's bottom-row
// significance for the row below), UVLC prefixes by v_perm from a register-resident 8-entry table.
__device__ uint2 g_vlc_dec2[2048];          // [0..1023] first quad row, [1024..2047] others; index (c_q << 7) | 7 bits;
                                            // .x = the 16-bit CxtVLC entry (len | u_off << 3 | rho << 4 | e_1 << 8 | e_k << 12),
                                            // .y = K5b's 9 bits of the entry | next context bits << 9 (address position) | bottom-row significance << 28

// The VLC bytes of one quad ROW pass through LDS: a quad row of <= 16 pairs consumes <= 16 x 30 bits and the register holds <= 63
// more, 543 bits -- 20 dwords even if every byte carried 7 bits (a conforming stream: every other byte at most) --, so they are
// fetched in one batch at the row's start (independent loads, one wait) and the pair loop refills its register from
// LDS.  A global load per pair would put `s_waitcnt vmcnt(0)` -- gfx9 counts loads and stores with one counter, and the
// compiler merges the wait over the loop's back edge -- and with it the round trip of the pair's own quad-info STORE on the
// serial chain (measured in r03: the kernel was no faster than r02's with half the instructions).
constexpr uint32_t kRowWords = 20, kRowStride = 21;       // (odd stride: the lanes' copies of word j lie in different banks)
struct VlcBits {      // LSB first
    const uint8_t* first;             // the byte read first: D + lcup - 3; byte k of the reading order is first[-k]
    const uint8_t* lo; const uint8_t* hi;
    uint32_t* row;                    // the lane's LDS row: the dwords of bytes k0 .. k0 + 4 * kRowWords - 1 as loaded (memory order)
    uint64_t acc; uint32_t n;         // n valid bits, the next one is bit 0
    uint32_t unst;                    // 0x80: the byte before byte k is > 0x8F
    uint32_t k, nv;                   // bytes handed to acc so far; bytes of the segment (behind them: zeros)
    uint32_t ri, nxt;                 // next dword of the row; the dword of bytes k .. k + 3 (read a refill ahead)
    __device__ __forceinline__ void init(const uint8_t* D, int lcup, int scup, uint32_t* lds_row, const uint8_t* buf_lo, const uint8_t* buf_hi)
    {
        first = D + lcup - 3; lo = buf_lo; hi = buf_hi; row = lds_row;
        const uint32_t d0 = D[lcup - 2];
        n = 4u - (((d0 >> 4) & 7u) == 7u ? 1u : 0u);
        acc = (d0 >> 4) & ((1u << n) - 1u);
        unst = (d0 | 0xFu) > 0x8Fu ? 0x80u : 0u;
        k = 0; nv = (uint32_t)scup - 2u; ri = 0; nxt = 0;
    }
    // the dwords of bytes k .. k + 4 kRowWords - 1 into the lane's LDS row, bytes behind the segment's end as zeros
    __device__ __forceinline__ void begin_row()
    {
        uint32_t t[kRowWords];
        const uint8_t* p = first - k - 3;                        // lowest address of the first dword
        if (k + 4u * kRowWords <= nv && p - 4 * (int)(kRowWords - 1) >= lo && p + 4 <= hi) {
#pragma unroll
            for (uint32_t j = 0; j < kRowWords; ++j) t[j] = *reinterpret_cast<const u32_unaligned*>(p - 4 * (int)j);
        } else {                                                 // the segment's last rows (or a block at the buffer's start)
#pragma unroll
            for (uint32_t j = 0; j < kRowWords; ++j) {
                const uint32_t kk = k + 4u * j;
                const uint32_t v = kk < nv ? min(nv - kk, 4u) : 0u;           // valid bytes: the HIGH addresses of the dword
                const uint32_t w = v ? ld4_guarded(p - 4 * (int)j, lo, hi) : 0u;
                t[j] = v >= 4u ? w : (v ? w & (0xFFFFFFFFu << (8u * (4u - v))) : 0u);
            }
        }
#pragma unroll
        for (uint32_t j = 1; j < kRowWords; ++j) row[j] = t[j];
        nxt = t[0]; ri = 1;
    }
    // four more bytes into the register (rev_read): at most once per pair, 28 .. 32 bits
    __device__ __forceinline__ void refill()
    {
        const uint32_t w = __builtin_bswap32(nxt);               // reading order: first byte in bits 7..0
        const uint32_t l7 = w & 0x7F7F7F7Fu;
        const uint32_t g = (l7 + 0x70707070u) & w & 0x80808080u;             // byte > 0x8F
        const uint32_t e = (l7 + 0x01010101u) & 0x80808080u;                 // 7 LSBs all ones
        const uint32_t s7 = e & ((g << 8) | unst);                           // bit 7 of a byte that carries 7 bits
        unst = g >> 24;
        uint32_t val = w, nb = 32u;
        if (s7) {                                                            // (rare: K5p's packing, the 7-bit bytes' MSBs dropped)
            const uint32_t m = w & ~s7;
            const uint32_t s0 = (s7 >> 7) & 1u, s1 = (s7 >> 15) & 1u, s2 = (s7 >> 23) & 1u, s3 = s7 >> 31;
            const uint32_t h1 = 8u - s0, h2 = h1 + 8u - s1, h3 = h2 + 8u - s2;
            val = (m & 0xFFu) | (((m >> 8) & 0xFFu) << h1) | (((m >> 16) & 0xFFu) << h2) | ((m >> 24) << h3);
            nb = h3 + 8u - s3;
        }
        acc |= (uint64_t)val << n;
        n += nb;
        k += 4u;
        nxt = row[ri];
        ri = min(ri + 1u, kRowWords - 1u);
    }
    __device__ __forceinline__ uint32_t peek() { if (n < 32u) refill(); return (uint32_t)acc; }   // >= 30 valid bits (a pair takes <= 30)
    __device__ __forceinline__ void advance(uint32_t used) { acc >>= used; n -= min(used, n); }
};

struct MelBits {      // MSB first
    const uint8_t* M; const uint8_t* lo; const uint8_t* hi;
    uint64_t acc; uint32_t n;         // n valid bits, the next one is bit 63
    uint32_t ff;                      // the byte before byte kb was 0xFF
    uint32_t kb, nm;                  // bytes handed over so far; bytes of the segment (behind them: 0xFF for ever)
    uint32_t nxt;                     // bytes kb .. kb + 3, the first one in bits 31..24, as the reader sees them
    int k, run;       // run: what is left of the current run, in the reference's coding (:196-235, :1101-1111):
                      // 2 * (zero events) + 1 if it ends with a one, 2 * (zero events - 1) otherwise
    __device__ __forceinline__ uint32_t fetch(uint32_t at) const
    {
        if (at >= nm) return 0xFFFFFFFFu;
        const uint32_t v = min(nm - at, 4u);
        uint32_t w = __builtin_bswap32(ld4_guarded(M + at, lo, hi));
        if (v < 4u) w |= 0xFFFFFFFFu >> (8u * v);                // behind the end: 0xFF
        if (at + v == nm) w |= 0x0Fu << (8u * (4u - v));         // the segment's last byte
        return w;
    }
    __device__ __forceinline__ void refill()                     // (mel_read)
    {
        const uint32_t w = nxt;
        const uint32_t f = ((w & 0x7F7F7F7Fu) + 0x01010101u) & w & 0x80808080u;   // byte == 0xFF
        const uint32_t s7 = (f >> 8) | (ff << 31);               // bit 7 of a byte that follows a 0xFF: 7 bits, its MSB dropped
        ff = (f >> 7) & 1u;
        uint32_t val = w, nb = 32u;
        if (s7) {
            const uint32_t f0 = s7 >> 31, f1 = (s7 >> 23) & 1u, f2 = (s7 >> 15) & 1u, f3 = (s7 >> 7) & 1u;     // byte 0 = the first read
            const uint32_t b0 = (w >> 24) & (0xFFu >> f0), b1 = (w >> 16) & 0xFFu & (0xFFu >> f1);
            const uint32_t b2 = (w >> 8) & 0xFFu & (0xFFu >> f2), b3 = w & 0xFFu & (0xFFu >> f3);
            const uint32_t w1 = 8u - f1, w2 = 8u - f2, w3 = 8u - f3;
            nb = 8u - f0 + w1 + w2 + w3;
            uint64_t t = b0;
            t = (t << w1) | b1; t = (t << w2) | b2; t = (t << w3) | b3;
            val = (uint32_t)(t << (32u - nb));
        }
        acc |= ((uint64_t)val << 32) >> n;
        n += nb;
        kb += 4u;
        nxt = fetch(kb);
    }
    __device__ __forceinline__ void init(const uint8_t* D, int lcup, int scup, const uint8_t* buf_lo, const uint8_t* buf_hi)
    {
        M = D + lcup - scup; lo = buf_lo; hi = buf_hi; nm = (uint32_t)scup - 1u;
        acc = 0; n = 0; ff = 0; kb = 0; k = 0; run = -1;
        nxt = fetch(0);
        refill();
        decode_run();
    }
    __device__ __forceinline__ void decode_run()                 // (:196-235)
    {
        const uint32_t e = (k < 8 ? 0x22111000u >> (4 * k) : 0x54332u >> (4 * (k - 8))) & 0xFu;   // MEL exponents (:196)
        const uint32_t top = (uint32_t)(acc >> 32);
        const bool one = (top >> 31) != 0;                       // '1': 2^e zero events; '0' + e bits: that many, then a one
        run = one ? (int)((2u << e) - 2u) : (int)((((top >> (31u - e)) & ((1u << e) - 1u)) << 1) | 1u);
        k = one ? (k < 12 ? k + 1 : 12) : (k > 0 ? k - 1 : 0);
        const uint32_t used = one ? 1u : e + 1u;
        acc <<= used; n -= used;
        if (n < 32u) refill();
    }
    // One MEL event if `need` (:1101-1111): returns 1 if the run ends here with a one.  The decode of the next run sits
    // behind a branch the whole wave skips when no lane has used its run up: in dense blocks (contexts rarely zero) and in
    // empty ones (long runs) that is most of the time.
    __device__ __forceinline__ bool event(bool need)
    {
        run -= need ? 2 : 0;
        const bool ev = run == -1;
        if (run < 0) decode_run();
        return ev;
    }
};

// UVLC prefix: '1' -> 1, '01' -> 2, '001' -> 3 + 1-bit suffix, '000' -> 5 + 5-bit suffix (:706-716), looked up by
// the three next bits: entry = prefix_len | suffix_len << 2 | base << 5, eight entries in two registers, picked by v_perm
__device__ __forceinline__ uint32_t uvlc_entry(uint32_t bits)
{
    // entries for bits & 7 = 0 .. 7: 000 -> (3, 5, 5), xx1 -> (1, 0, 1), x10 -> (2, 0, 2), 100 -> (3, 1, 3)
    return __builtin_amdgcn_perm(0x21422167u, 0x214221B7u, (bits & 7u) | 0x0C0C0C00u);
}

__global__ void ht_dec_vlc_kernel(HtDecArgs a)
{
    // CxtVLC decode tables in LDS: one dependent lookup per quad sits on the serial chain
    __shared__ uint2 tbl_l[2048];
    __shared__ uint32_t rows_l[64 * kRowStride];
    for (uint32_t i = threadIdx.x; i < 2048; i += blockDim.x) tbl_l[i] = g_vlc_dec2[i];
    __syncthreads();
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= a.nactive) return;
    const uint32_t blk = a.active ? a.active[li] : li;
    const HtDecBlock in = a.table[blk];
    const HtBlockDesc bd = a.blocks[blk % a.blocks_per_tile];
    uint16_t* qi = reinterpret_cast<uint16_t*>(a.quads) + (size_t)blk * kQuadWords;     // per quad: 9 table bits | (u_q + 1) << 9
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t QW = (w + 1) >> 1, QH = (h + 1) >> 1;
    const uint32_t mm = in.missing_msbs;

    if (in.length == 0) return;                            // absent (K5b writes the zeros) or outside the decoded region
    int lcup, scup;
    if (!ht_segments(a, blk, in, lcup, scup)) { atomicOr(a.status, 4u); a.ms_len[blk] = 0xFFFFFFFFu; return; }
    a.ms_len[blk] = (uint32_t)(lcup - scup);

    const uint8_t* const D = a.coded + in.offset;
    const uint8_t* const buf_hi = a.coded + a.coded_bytes;
    MelBits mel;
    mel.init(D, lcup, scup, a.coded, buf_hi);
    VlcBits vlc;
    vlc.init(D, lcup, scup, rows_l + threadIdx.x * kRowStride, a.coded, buf_hi);
    const uint32_t NP = (QW + 1) >> 1;  // quad pairs per row
    uint64_t sa = 0;                     // significance of the bottom sample row of the quad row above: bit x
    uint32_t umax = 0;                   // largest u_q + 1 of the block (:1194: > missing_msbs rejects it)
    // ---- the first quad row: its own contexts (the quad to the left only), table and u-value rules (:1093-1190)
    {
        const uint2* tbl = tbl_l;
        uint16_t* qrow = qi;
        uint64_t sn = 0;
        uint32_t caddr = 0;                                // context of the next quad, << 9
        vlc.begin_row();
        for (uint32_t q0 = 0; q0 < QW; q0 += 2) {
            const bool has1 = q0 + 1 < QW;
            uint32_t v = vlc.peek(), used = 0;
            // ---- quad q0
            uint2 t0 = tbl[(caddr >> 2) | (v & 0x7Fu)];
            bool need = caddr == 0;
            bool zero = need && !mel.event(need);
            if (zero) t0 = make_uint2(0, 0);
            uint32_t len = t0.x & 7u;
            v >>= len; used += len;
            caddr = t0.y & 0xE00u;
            // ---- quad q0 + 1 (absent when the row has an odd number of quads)
            uint2 t1 = tbl[(caddr >> 2) | (v & 0x7Fu)];
            need = has1 && caddr == 0;
            zero = !has1 || (need && !mel.event(need));
            if (zero) t1 = make_uint2(0, 0);
            len = t1.x & 7u;
            v >>= len; used += len;
            caddr = t1.y & 0xE00u;
            sn = (sn >> 4) | ((uint64_t)((t0.y & 0x30000000u) | ((t1.y & 0x30000000u) << 2)) << 32);
            // ---- u values of the pair (:668-777): prefix0, prefix1, suffix0, suffix1, each present only if its quad has u_off
            const uint32_t uo0 = (t0.x >> 3) & 1u, uo1 = (t1.x >> 3) & 1u;
            uint32_t d = uo0 ? uvlc_entry(v) : 0u;
            const uint32_t pl = d & 3u, sl = (d >> 2) & 7u, base = d >> 5;
            v >>= pl;
            const uint32_t both = uo0 & uo1;               // both quads: a MEL event picks the variant
            const uint32_t e2 = mel.event(both != 0) ? 1u : 0u;
            const uint32_t add = (both & e2) ? 3u : 1u;
            const uint32_t onebit = both & (e2 ^ 1u) & (pl > 2 ? 1u : 0u);    // second quad is a single bit
            d = uo1 ? uvlc_entry(v) : 0u;
            uint32_t pl2 = d & 3u, sl2 = (d >> 2) & 7u, base2 = d >> 5;
            pl2 = onebit ? 1u : pl2; sl2 = onebit ? 0u : sl2; base2 = onebit ? (v & 1u) + 1u : base2;
            v >>= pl2;
            const uint32_t U0 = base + (v & ((1u << sl) - 1u)) + (uo0 ? add : 1u);
            v >>= sl;
            const uint32_t U1 = base2 + (v & ((1u << sl2) - 1u)) + (uo1 ? add : 1u);
            vlc.advance(used + pl + pl2 + sl + sl2);
            umax = max(umax, max(U0, U1));
            *reinterpret_cast<uint32_t*>(&qrow[q0]) = ((t0.y & 0x1FFu) | (U0 << 9)) | (((t1.y & 0x1FFu) | (U1 << 9)) << 16);
        }
        sa = sn >> (64u - 4u * NP);
    }
    // ---- the other quad rows (:1192-1330)
    const uint2* tbl = tbl_l + 1024;
    for (uint32_t qy = 1; qy < QH; ++qy) {
        uint16_t* qrow = qi + qy * kQuadStride;
        uint64_t sw = sa, sn = 0;
        uint32_t west = 0, chain = 0;    // sample 2 * q0 - 1 of the row above; the west quad's contribution to the context, << 9
        vlc.begin_row();
        for (uint32_t q0 = 0; q0 < QW; q0 += 2) {
            const bool has1 = q0 + 1 < QW;
            uint32_t v = vlc.peek(), used;
            // the row above over the pair: x = (sample 2 q0 - 1) | samples 2 q0 .. 2 q0 + 4 << 1; y = x | x >> 1 has
            // (nw | n) of quad q0 at bit 0, (ne | nf) at bit 2, the same of quad q0 + 1 two bits up
            const uint32_t up = (uint32_t)sw;
            const uint32_t x = ((up << 1) | west) & 0x3Fu;
            const uint32_t y9 = (x | (x >> 1)) << 9;
            west = (up >> 3) & 1u; sw >>= 4;
            // ---- quad q0
            uint32_t caddr = chain | (y9 & 0xA00u);
            uint2 t0 = tbl[(caddr >> 2) | (v & 0x7Fu)];
            bool need = caddr == 0;
            bool zero = need && !mel.event(need);
            if (zero) t0 = make_uint2(0, 0);
            uint32_t len = t0.x & 7u;
            v >>= len; used = len;
            // ---- quad q0 + 1
            caddr = (t0.y & 0x400u) | ((y9 >> 2) & 0xA00u);
            uint2 t1 = tbl[(caddr >> 2) | (v & 0x7Fu)];
            need = has1 && caddr == 0;
            zero = !has1 || (need && !mel.event(need));
            if (zero) t1 = make_uint2(0, 0);
            len = t1.x & 7u;
            v >>= len; used += len;
            chain = t1.y & 0x400u;
            sn = (sn >> 4) | ((uint64_t)((t0.y & 0x30000000u) | ((t1.y & 0x30000000u) << 2)) << 32);
            // ---- u values: prefix0, prefix1, suffix0, suffix1
            uint32_t d = ((t0.x >> 3) & 1u) ? uvlc_entry(v) : 0u;
            const uint32_t pl = d & 3u, sl = (d >> 2) & 7u, base = d >> 5;
            v >>= pl;
            d = ((t1.x >> 3) & 1u) ? uvlc_entry(v) : 0u;
            const uint32_t pl2 = d & 3u, sl2 = (d >> 2) & 7u, base2 = d >> 5;
            v >>= pl2;
            const uint32_t U0 = base + __builtin_amdgcn_ubfe(v, 0u, sl) + 1u;
            v >>= sl;
            const uint32_t U1 = base2 + __builtin_amdgcn_ubfe(v, 0u, sl2) + 1u;
            vlc.advance(used + pl + pl2 + sl + sl2);
            umax = max(umax, max(U0, U1));
            *reinterpret_cast<uint32_t*>(&qrow[q0]) = ((t0.y & 0x1FFu) | (U0 << 9)) | (((t1.y & 0x1FFu) | (U1 << 9)) << 16);
        }
        sa = sn >> (64u - 4u * NP);
    }
    if (umax > mm) { atomicOr(a.status, 4u); a.ms_len[blk] = 0xFFFFFFFFu; }      // :1194
}

// ---- K5b --------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bperm(int addr, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v);
}

// the block carries SigProp (/ MagRef) data that the reference's decoder would use (ojph_block_decoder.cpp:1014-1019:
// the passes need p = 30 - missing_msbs >= 2 and a non-empty second segment)
__device__ __forceinline__ bool ht_block_refined(const HtDecArgs& a, uint32_t blk, uint32_t mm)
{
    return a.refine && a.refine[blk].x > 0 && a.refine[blk].y >= 2 && mm <= 28;
}

template <bool IRREV>
__global__ __launch_bounds__(64) void ht_dec_ms_kernel(HtDecArgs a, uint32_t raw_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t raw[];       // un-stuffed MagSgn bits + 4 words of ones
    const int lane = threadIdx.x;
    const uint32_t blk = a.ms_count ? blockIdx.y * a.ms_bpc + a.ms_first + blockIdx.x : blockIdx.x;
    const HtDecBlock in = a.table[blk];
    const HtBlockDesc bd = a.blocks[blk % a.blocks_per_tile];
    const uint32_t tile = blk / a.blocks_per_tile;
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t QH = (h + 1) >> 1;
    const size_t dst_at = ((size_t)tile * a.ncomp + bd.comp) * a.pitch + (size_t)bd.py * a.stride + bd.px;
    int32_t* dst = a.mallat + dst_at;
    int16_t* dst16 = reinterpret_cast<int16_t*>(a.mallat) + dst_at;       // (a.h16: the same planes with int16 elements)
    const bool h16 = !IRREV && a.h16 != 0;
    const uint32_t ms_len = a.ms_len[blk];
    const uint32_t x = lane;                              // sample column of this lane
    const bool col_ok = x < w;

    if (in.length == 0 && in.missing_msbs == kSkipBlock) return;   // region decode: nobody reads this block's samples
    if (in.length == 0 || ms_len == 0xFFFFFFFFu) {        // absent or rejected block: zeros
        for (uint32_t y = 0; y < h; ++y)
            if (col_ok) { if (h16) dst16[(size_t)y * a.stride + x] = 0; else dst[(size_t)y * a.stride + x] = 0; }
        return;
    }
    // ---- un-stuff the MagSgn segment: byte i contributes 8 bits, or 7 if byte i-1 is 0xFF ----------
    // FOUR bytes per lane and step (256 per wave): the bytes i-1 .. i+3 come out of three aligned dwords with two
    // v_alignbit, which of them follow a 0xFF is one SWAR test on (prev, b0, b1, b2), the lane packs its <= 32 bits and
    // a wave prefix sum of the lanes' bit counts places them
    const uint8_t* D = a.coded + in.offset;
    for (uint32_t i = lane; i < raw_words; i += 64) raw[i] = 0;
    __syncthreads();
    uint32_t base_bits = 0;
    {
        const uint8_t* buf_hi = a.coded + a.coded_bytes;
        // (an aligned dword that starts inside the buffer lies in a mapped page; its bytes past ms_len are masked)
        auto ld = [&](const uint8_t* q) { return (q >= a.coded && q < buf_hi) ? *reinterpret_cast<const uint32_t*>(q) : 0u; };
        for (uint32_t i0 = 0; i0 < ms_len; i0 += 256) {
            const uint32_t i = i0 + 4u * (uint32_t)lane;
            uint32_t val = 0, tot = 0;
            if (i < ms_len) {
                const uint8_t* P = D + i - 1;                                  // the byte before this lane's four
                const uint8_t* A = P - ((uintptr_t)P & 3u);
                const uint32_t sh = (uint32_t)((uintptr_t)P & 3u) * 8u;
                const uint32_t w0 = ld(A), w1 = ld(A + 4), w2 = ld(A + 8);
                const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh);     // prev, b0, b1, b2
                const uint32_t hi = __builtin_amdgcn_alignbit(w2, w1, sh);     // b3, ...
                uint32_t b = (lo >> 8) | (hi << 24);
                const uint32_t nv = min(ms_len - i, 4u);
                b &= 0xFFFFFFFFu >> (32u - 8u * nv);                           // bytes past the segment: dropped
                uint32_t st = ((lo & 0x7F7F7F7Fu) + 0x01010101u) & lo & 0x80808080u;   // bit 7 of byte k: byte k follows a 0xFF
                st &= (i == 0) ? 0xFFFFFF00u : 0xFFFFFFFFu;                    // (the first byte follows nothing)
                b &= ~st;
                const uint32_t s0 = (st >> 7) & 1u, s1 = (st >> 15) & 1u, s2 = (st >> 23) & 1u, s3 = st >> 31;
                const uint32_t h1 = 8u - s0, h2 = h1 + 8u - s1, h3 = h2 + 8u - s2;
                val = (b & 0xFFu) | (((b >> 8) & 0xFFu) << h1) | (((b >> 16) & 0xFFu) << h2) | ((b >> 24) << h3);
                const uint32_t full = h3 + 8u - s3;                            // bits of four bytes
                tot = nv == 4 ? full : (nv == 3 ? h3 : (nv == 2 ? h2 : h1));
            }
            const uint32_t incl = wave_incl_scan(tot);
            const uint32_t pos = base_bits + incl - tot;
            if (tot) {
                const uint64_t v = (uint64_t)val << (pos & 31);
                lds_or(&raw[pos >> 5], (uint32_t)v);
                lds_or(&raw[(pos >> 5) + 1], (uint32_t)(v >> 32));
            }
            base_bits += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    __syncthreads();
    // an exhausted MagSgn segment reads as ones (frwd_read<0xFF>, :823-850)
    if (lane < 4) {
        const uint32_t wi = (base_bits >> 5) + lane;
        const uint32_t ones = lane == 0 ? (0xFFFFFFFFu << (base_bits & 31)) : 0xFFFFFFFFu;
        if (wi < raw_words) lds_or(&raw[wi], ones);
    }
    __syncthreads();

    // ---- quad rows ---------------------------------------------------------------------------------
    const uint32_t mm = in.missing_msbs;
    const uint32_t p = 30u - mm;
    const bool refined = ht_block_refined(a, blk, mm);
    // K5a's 16 bits per quad: the block's 2 KB in LDS first (two 16-byte loads per lane, all in flight at once) -- fetched row by row,
    // one row ahead, every row of the dependent chain waited for an L2 round trip
    __shared__ __attribute__((aligned(16))) uint16_t qi[kQuadWords];
    {
        const uint4* const src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(a.quads) + (size_t)blk * kQuadWords);
        const uint32_t n16 = (QH * kQuadStride * 2u + 15u) / 16u;            // 16-byte pieces that hold rows 0 .. QH - 1
        uint4* const dq = reinterpret_cast<uint4*>(qi);
        if ((uint32_t)lane < n16) dq[lane] = src[lane];
        if ((uint32_t)lane + 64u < n16) dq[lane + 64] = src[lane + 64];
    }
    __syncthreads();
    const uint32_t q = x >> 1, right = x & 1u;
    uint32_t Eprev = 0;                                   // exponent of this column's bottom sample, row above
    uint32_t bitpos = 0;
    uint32_t range = 0;
    uint32_t info = col_ok ? qi[q] : 0u;
    for (uint32_t qy = 0; qy < QH; ++qy) {
        const uint32_t cur = info;
        if (qy + 1 < QH) info = col_ok ? qi[(qy + 1) * kQuadStride + q] : 0u;     // prefetch next row's quad info
        // this lane's two samples (top: i = 2 * right, bottom: the next): their states -- 0 insignificant, 1 significant, 2 with
        // e_k, 3 with e_k and e_1 --, whether the quad has two or more significant samples, and u_q + 1
        const uint32_t nib = cur >> (4u * right);
        const uint32_t s_t = nib & 3u, s_b = (nib >> 2) & 3u;
        const bool many = (cur & 0x100u) != 0;
        uint32_t U = cur >> 9;
        // kappa: max exponent over columns 2q-1 .. 2q+2 of the row above, when more than one sample is significant.
        // With lane shifts (DPP wave_shl / wave_shr, zero beyond the wave's ends = outside the block): P[x] = max(E[x], E[x+1]);
        // the quad's left lane (x = 2q) takes max(P[x-1], P[x+1]), the right lane that lane's value.
        {
            const uint32_t Pm = max(Eprev, dpp0<0x130, 0xF>(Eprev));            // wave_shl:1 -> lane x reads x + 1
            const uint32_t Al = max(max(dpp0<0x138, 0xF>(Pm), dpp0<0x130, 0xF>(Pm)), Eprev);  // wave_shr:1 -> lane x reads x - 1
                                                                                   // (E[x] itself: lane 0 has no P[-1])
            // (selected with a mask, not a branch: a lane shift inside a predicated region reads zeros from the disabled lanes)
            const uint32_t Ar = dpp0<0x138, 0xF>(Al);
            const uint32_t E = Al ^ ((Al ^ Ar) & (0u - right));
            if (qy > 0 && many) U += E > 2 ? E - 2 : 0;
        }
        const uint32_t st = s_t ? 1u : 0u, sb = s_b ? 1u : 0u;
        const uint32_t mt = st ? U - (s_t >> 1) : 0u;
        const uint32_t mb = sb ? U - (s_b >> 1) : 0u;
        const uint32_t incl = wave_incl_scan(mt + mb);
        const uint32_t pos = bitpos + incl - (mt + mb);
        bitpos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // up to 62 bits from the raw stream
        const uint32_t wi = pos >> 5, sh = pos & 31;
        const uint32_t r0 = raw[wi], r1 = raw[wi + 1], r2 = raw[wi + 2];
        const uint32_t win0 = __builtin_amdgcn_alignbit(r1, r0, sh), win1 = __builtin_amdgcn_alignbit(r2, r1, sh);   // 64 bits from pos on
        const uint32_t bt = win0 & ((1u << mt) - 1u);                                                          // mt, mb <= 31
        const uint32_t bb = __builtin_amdgcn_alignbit(win1, win0, mt) & ((1u << mb) - 1u);
        const uint32_t vt = bt | (((s_t + 1u) >> 2) << mt) | 1u;        // (state 3: e_1 set)
        const uint32_t vb = bb | (((s_b + 1u) >> 2) << mb) | 1u;
        const uint32_t wt = st ? ((bt << 31) | ((vt + 2u) << (p - 1u))) : 0u;
        const uint32_t wb = sb ? ((bb << 31) | ((vb + 2u) << (p - 1u))) : 0u;
        Eprev = sb ? 32u - (uint32_t)__clz((int)vb) : 0u;
        // dequantise and store (PostDecompressFilters.h: ShiftHTFilter shift = 31 - (k_msbs + 1) = p); a block with
        // refinement passes keeps the decoder's words: K5c refines them and dequantises
        int32_t ot, ob;
        if (refined) { ot = (int32_t)wt; ob = (int32_t)wb; }
        else if constexpr (IRREV) {
            const float ft = (float)(int32_t)(wt & 0x7FFFFFFFu) * bd.inv_step;       // inv_step holds the decode scale here
            const float fb = (float)(int32_t)(wb & 0x7FFFFFFFu) * bd.inv_step;
            ot = __float_as_int((wt & 0x80000000u) ? -ft : ft);
            ob = __float_as_int((wb & 0x80000000u) ? -fb : fb);
        } else {
            // ((v + 2) << (p - 1)) >> p is (v + 2) >> 1 whatever p: the magnitude comes straight from v, the sign is bit 0 of
            // the MagSgn value, applied arithmetically
            const int32_t mgt = st ? (int32_t)((vt + 2u) >> 1) : 0, mgb = sb ? (int32_t)((vb + 2u) >> 1) : 0;
            const int32_t sgt = -(int32_t)(bt & 1u), sgb = -(int32_t)(bb & 1u);
            ot = (mgt ^ sgt) - sgt;
            ob = (mgb ^ sgb) - sgb;
        }
        if (col_ok) {
            const uint32_t y0 = 2 * qy;
            if (h16) {
                dst16[(size_t)y0 * a.stride + x] = (int16_t)ot;
                if (y0 + 1 < h) dst16[(size_t)(y0 + 1) * a.stride + x] = (int16_t)ob;
                range |= (uint32_t)(ot + a.h16_bias) | (uint32_t)(ob + a.h16_bias);          // >= 2 bias: outside [-bias, bias)
            } else {
                dst[(size_t)y0 * a.stride + x] = ot;
                if (y0 + 1 < h) dst[(size_t)(y0 + 1) * a.stride + x] = ob;
            }
        }
    }
    if (h16 && __builtin_amdgcn_ballot_w64(range >= 2u * (uint32_t)a.h16_bias) != 0 && lane == 0) atomicOr(a.status, 8u);
}

// ---- K5c: the refinement passes, ONE WAVEFRONT PER CODE-BLOCK ---------------------------------------------------------
// Replaces the SigProp / MagRef half of ojph_decode_codeblock (t1/t1_ht/coding/ojph_block_decoder.cpp:1627-2100; bit
// readers :466-550 rev_*_mrp, :875-945 frwd_* with X = 0).  Grok never reaches that code (T1HT.cpp:158-166 passes
// lengths2 = 0): this is reachable through grk_amd_set_decode_segments only.  K5b left the cleanup pass's words in the
// block's place in the Mallat plane; here, lane <-> column, one stripe of 4 rows at a time:
//   * both segments are un-stuffed in parallel into LDS bit arrays, as K5b does with MagSgn (SigProp: forward, the byte
//     after 0xFF has 7 bits; MagRef: backward from the end, a byte whose 7 low bits are ones has 7 bits after a byte > 0x8F);
//   * MagRef is data-parallel: a sample's bit is at (significant samples before it in scan order) -- popcount of the lane's
//     significance nibble, a wave prefix sum per stripe;
//   * SigProp is one chain through the block (a newly significant sample makes its later neighbours members, and every bit's
//     position depends on the members before it): the wave runs it as a uniform program on 64-bit row bitmaps (ballots), the
//     way K8 runs its passes -- the column loop hops from member to member with find-first-set;
//   * the stripe is dequantised (ShiftHTFilter / ScaleHTFilter) and stored as K5b would have.
template <bool IRREV>
__global__ __launch_bounds__(64) void ht_dec_refine_kernel(HtDecArgs a, uint32_t seg_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* spp_raw = lds;                    // un-stuffed SigProp bits (zeros beyond the end)
    uint32_t* mrp_raw = lds + seg_words;        // un-stuffed MagRef bits, in reading order
    const int lane = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const HtDecBlock in = a.table[blk];
    if (in.length == 0 || !ht_block_refined(a, blk, in.missing_msbs) || a.ms_len[blk] == 0xFFFFFFFFu) return;
    const HtBlockDesc bd = a.blocks[blk % a.blocks_per_tile];
    const uint32_t tile = blk / a.blocks_per_tile;
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t len2 = a.refine[blk].x, npasses = a.refine[blk].y;
    const uint32_t p = 30u - in.missing_msbs;
    const uint8_t* seg = a.coded + in.offset + (in.length - len2);
    int32_t* dst = a.mallat + ((size_t)tile * a.ncomp + bd.comp) * a.pitch + (size_t)bd.py * a.stride + bd.px;
    const uint32_t x = lane;
    const bool col_ok = x < w;

    for (uint32_t i = lane; i < 2 * seg_words; i += 64) lds[i] = 0;
    __syncthreads();
    {   // un-stuff: byte i (SigProp: seg[i]; MagRef: seg[len2 - 1 - i]) contributes 8 or 7 bits
        uint32_t base_s = 0, base_m = 0;
        for (uint32_t i0 = 0; i0 < len2; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint32_t bs = 0, ws = 0, bm = 0, wm = 0;
            if (i < len2) {
                bs = seg[i];
                const bool st = i > 0 && seg[i - 1] == 0xFFu;
                ws = st ? 7u : 8u;
                bs &= st ? 0x7Fu : 0xFFu;
                bm = seg[len2 - 1 - i];
                const bool un = i == 0 || seg[len2 - i] > 0x8Fu;
                const bool sm = un && (bm & 0x7Fu) == 0x7Fu;
                wm = sm ? 7u : 8u;
                bm &= sm ? 0x7Fu : 0xFFu;
            }
            const uint32_t incl = wave_incl_scan(ws | (wm << 16));
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (i < len2) {
                const uint32_t ps = base_s + (incl & 0xFFFFu) - ws, pm = base_m + (incl >> 16) - wm;
                const uint64_t vs = (uint64_t)bs << (ps & 31), vm = (uint64_t)bm << (pm & 31);
                lds_or(&spp_raw[ps >> 5], (uint32_t)vs); lds_or(&spp_raw[(ps >> 5) + 1], (uint32_t)(vs >> 32));
                lds_or(&mrp_raw[pm >> 5], (uint32_t)vm); lds_or(&mrp_raw[(pm >> 5) + 1], (uint32_t)(vm >> 32));
            }
            base_s += tot & 0xFFFFu; base_m += tot >> 16;
        }
    }
    __syncthreads();

    const uint64_t wmask = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
    auto dilh = [&](uint64_t v) { return (v | (v << 1) | (v >> 1)) & wmask; };
    uint32_t spp_pos = 0, mrp_pos = 0;          // bits consumed (wave-uniform)
    auto spp_bit = [&]() -> uint32_t {
        const uint32_t v = spp_pos < seg_words * 32u ? (spp_raw[spp_pos >> 5] >> (spp_pos & 31)) & 1u : 0u;
        ++spp_pos;
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    };
    uint64_t above = 0;                         // bottom row of the stripe above: significant after both passes
    uint32_t nxt[4];                            // the next stripe's words (its top row feeds this stripe's membership)
#pragma unroll
    for (int j = 0; j < 4; ++j) nxt[j] = (col_ok && (uint32_t)j < h) ? (uint32_t)dst[(size_t)j * a.stride + x] : 0u;
    for (uint32_t y0 = 0; y0 < h; y0 += 4) {
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = nxt[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) nxt[j] = (col_ok && y0 + 4 + j < h) ? (uint32_t)dst[(size_t)(y0 + 4 + j) * a.stride + x] : 0u;
        const uint32_t nr = min(4u, h - y0);
        uint64_t S[4];                           // cleanup significance of the stripe's rows
#pragma unroll
        for (int j = 0; j < 4; ++j) S[j] = __ballot(v[j] != 0u);
        const uint64_t below = __ballot(nxt[0] != 0u);
        // ---- MagRef: one bit per cleanup-significant sample, column by column
        if (npasses >= 3) {
            const uint32_t cnt = (v[0] != 0u) + (v[1] != 0u) + (v[2] != 0u) + (v[3] != 0u);
            const uint32_t incl = wave_incl_scan(cnt);
            uint32_t at = mrp_pos + incl - cnt;
            mrp_pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] != 0u) {
                    const uint32_t sym = at < seg_words * 32u ? (mrp_raw[at >> 5] >> (at & 31)) & 1u : 0u;
                    ++at;
                    v[j] = (v[j] ^ ((1u - sym) << (p - 1u))) | (1u << (p - 2u));
                }
        }
        // ---- SigProp: members = insignificant samples with a significant neighbour; scanned in groups of 4 columns
        uint64_t M[4], NEW[4] = {0, 0, 0, 0}, SG[4] = {0, 0, 0, 0};
        {
            const uint64_t rows_ok[4] = {wmask, nr > 1 ? wmask : 0ull, nr > 2 ? wmask : 0ull, nr > 3 ? wmask : 0ull};
            const uint64_t d0 = dilh(S[0]), d1 = dilh(S[1]), d2 = dilh(S[2]), d3 = dilh(S[3]);
            M[0] = (dilh(above) | d0 | d1) & ~S[0] & rows_ok[0];
            M[1] = (d0 | d1 | d2) & ~S[1] & rows_ok[1];
            M[2] = (d1 | d2 | d3) & ~S[2] & rows_ok[2];
            M[3] = (d2 | d3 | dilh(below)) & ~S[3] & rows_ok[3];
            for (uint32_t g0 = 0; g0 < w; g0 += 4) {
                const uint64_t gm = (0xFull << g0) & wmask;
                uint32_t xs = g0;
                while (true) {                   // columns of the group that hold a member, left to right
                    const uint64_t cm = (M[0] | M[1] | M[2] | M[3]) & gm & (~0ull << xs);
                    if (!cm) break;
                    const uint32_t xc = (uint32_t)__ffsll((long long)cm) - 1u;
                    const uint64_t bitx = 1ull << xc, nb = (bitx << 1) & wmask;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!(M[j] & bitx)) continue;
                        if (!spp_bit()) continue;
                        NEW[j] |= bitx;
                        // its later neighbours become members: the sample below, the three in the next column
                        if (j < 3) M[j + 1] |= bitx & ~S[j + 1] & rows_ok[j + 1];
                        if (j > 0) M[j - 1] |= nb & ~S[j - 1] & rows_ok[j - 1];
                        M[j] |= nb & ~S[j] & rows_ok[j];
                        if (j < 3) M[j + 1] |= nb & ~S[j + 1] & rows_ok[j + 1];
                    }
                    xs = xc + 1;
                    if (xs >= 64) break;
                }
                uint64_t nm = (NEW[0] | NEW[1] | NEW[2] | NEW[3]) & gm;         // then the signs of the group's new samples
                while (nm) {
                    const uint32_t xc = (uint32_t)__ffsll((long long)nm) - 1u;
                    const uint64_t bitx = 1ull << xc;
                    nm &= nm - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (NEW[j] & bitx) { if (spp_bit()) SG[j] |= bitx; }
                }
            }
        }
        above = (S[nr - 1] | NEW[nr - 1]);
        // ---- the stripe leaves dequantised
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((NEW[j] >> x) & 1ull) v[j] = ((uint32_t)((SG[j] >> x) & 1ull) << 31) | (3u << (p - 2u));
            int32_t o;
            if constexpr (IRREV) {
                const float f = (float)(int32_t)(v[j] & 0x7FFFFFFFu) * bd.inv_step;
                o = __float_as_int((v[j] & 0x80000000u) ? -f : f);
            } else {
                const int32_t mg = (int32_t)((v[j] & 0x7FFFFFFFu) >> p);
                o = (v[j] & 0x80000000u) ? -mg : mg;
            }
            if (col_ok && (uint32_t)j < nr) dst[(size_t)(y0 + j) * a.stride + x] = o;
        }
    }
}

} // namespace

namespace {

// The per-call tables of a decode (code-block rows, K5's scratch index) come out of pinned host memory by a kernel of the call's own
// stream: two engine copies + a fill cost the 8K decode 63 us per call (13 + 21 + 3.5 us and four ~9 us hand-overs between the copy
// engine and the compute queue, tools/dec_timeline.sh), a kernel that reads the same bytes over the link starts where the last
// kernel of the call before ended.  The status word of the call is cleared on the way.
typedef unsigned int up_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void dec_upload_kernel(const up_u32x4* __restrict__ src, up_u32x4* __restrict__ dst, uint32_t n, up_u32x4* status)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = __builtin_nontemporal_load(src + i);
    if (i == 0 && status) *status = up_u32x4{0u, 0u, 0u, 0u};
}

} // namespace

hipError_t launch_dec_upload(const void* pinned, void* dst, size_t bytes, void* status, hipStream_t s)
{
    const uint32_t n = (uint32_t)((bytes + 15) / 16);
    hipLaunchKernelGGL(dec_upload_kernel, dim3(n ? (n + 255) / 256 : 1), dim3(256), 0, s, (const up_u32x4*)pinned, (up_u32x4*)dst, n, (up_u32x4*)status);
    return hipGetLastError();
}

static std::atomic<bool> g_dec_tables_ready[16];             // (zero-initialised: false)
static std::mutex g_dec_tables_mu;                           // first use from several host threads at once

hipError_t launch_ht_decode(const HtDecArgs& a, uint32_t max_ms_bytes, hipStream_t s)
{
    const hipError_t e = launch_ht_decode_front(a, s);
    return e != hipSuccess ? e : launch_ht_decode_ms(a, max_ms_bytes, s);
}

hipError_t launch_ht_decode_front(const HtDecArgs& a, hipStream_t s)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 16 && !g_dec_tables_ready[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_dec_tables_mu);
        if (!g_dec_tables_ready[dev].load(std::memory_order_relaxed)) {
        // the CxtVLC decode tables with what K5a's chain needs next to each entry (kernels above: g_vlc_dec2)
        static uint2 tab[2048];
        for (uint32_t i = 0; i < 2048; ++i) {
            const uint32_t t = i < 1024 ? HT_VLC_DEC0[i] : HT_VLC_DEC1[i - 1024];
            const uint32_t rho = (t >> 4) & 0xFu;
            const uint32_t next = i < 1024 ? ((rho & 1u) | (rho >> 1))                        // first row: c_q of the quad to the right (:1117)
                                           : ((((rho >> 2) | (rho >> 3)) & 1u) << 1);         // other rows: its west part (:1216)
            const uint32_t sb = ((rho >> 1) & 1u) | (((rho >> 3) & 1u) << 1);                 // the quad's bottom samples
            // what K5b needs of the entry, in 9 bits: per sample rho_i + e_k,i + e_1,i (e_1 <= e_k <= rho holds for every codeword of
            // T.814 Annex C: 0 insignificant, 1 significant, 2 with e_k, 3 with e_k and e_1), and "two or more samples significant"
            const uint32_t e1 = (t >> 8) & 0xFu, ek = (t >> 12) & 0xFu;
            uint32_t pay = (rho & (rho - 1u)) ? 0x100u : 0u;
            for (uint32_t k = 0; k < 4; ++k) pay |= (((rho >> k) & 1u) + ((ek >> k) & 1u) + ((e1 >> k) & 1u)) << (2 * k);
            tab[i].x = t;
            tab[i].y = pay | (next << 9) | (sb << 28);
        }
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_vlc_dec2), tab, sizeof(tab), 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        g_dec_tables_ready[dev].store(true, std::memory_order_release);
        }
    }
    if (a.nactive == 0) return hipSuccess;
    // K5a is one serial chain per lane.  A wave costs the same issue slots however many lanes are
    // live, so few lanes per wave multiply the instruction count, while few waves per SIMD leave the
    // chain's own latency exposed: aim for ~1.5-2 waves per SIMD (1024 SIMDs), measured optimum.
    // (r02, 8K, 49 152 blocks: 32, 48 and 64 lanes per wave take the same time -- what lasts is the chain of one wave, and a
    //  second wave on the SIMD interleaves for free; 16 and 24 lanes are slower.  Running this kernel BESIDE the wide ones of
    //  another frame does not pay either: next to K5b both take twice as long, next to the inverse DWT 1.4x / 1.5x -- the
    //  chain's next instruction waits behind whatever holds the SIMD's ALU for its four cycles, wave priority or not)
    uint32_t lanes = 64;
    while (lanes > 16 && (a.nactive + lanes - 1) / lanes < 1280) lanes >>= 1;
    if (const char* e = getenv("GRK_AMD_K5A_LANES")) {              // (experiments; anything but a wave-sized power of two is ignored)
        const int v = atoi(e);
        if (v == 16 || v == 32 || v == 64) lanes = (uint32_t)v;
    }
    hipLaunchKernelGGL(ht_dec_vlc_kernel, dim3((a.nactive + lanes - 1) / lanes), dim3(lanes), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ht_decode_ms(const HtDecArgs& a, uint32_t max_ms_bytes, hipStream_t s)
{
    // (runs even when no block has data: absent blocks are zeros in the planes)
    const uint32_t raw_words = (max_ms_bytes * 8u) / 32u + 8u;
    const dim3 grid = a.ms_count ? dim3(a.ms_count, a.nblocks / a.ms_bpc) : dim3(a.nblocks);
    // (measured and not kept: the side-stream part capped at 6 / 5 / 4 waves per SIMD so that the four-wave workgroups of the small
    //  inverse levels find room beside it -- they do, 227 us for the four instead of 290, but K5b wants its eight waves: 267 -> 320 us
    //  at 5, the call 0.833 -> 0.852 / 0.861 / 0.880 ms)
    if (a.irreversible)
        hipLaunchKernelGGL(ht_dec_ms_kernel<true>, grid, dim3(64), raw_words * 4, s, a, raw_words);
    else
        hipLaunchKernelGGL(ht_dec_ms_kernel<false>, grid, dim3(64), raw_words * 4, s, a, raw_words);
    if (a.refine && a.max_refine_bytes) {          // some block carries SigProp / MagRef data
        const uint32_t seg_words = (a.max_refine_bytes * 8u) / 32u + 4u;
        if (a.irreversible)
            hipLaunchKernelGGL(ht_dec_refine_kernel<true>, dim3(a.nblocks), dim3(64), seg_words * 8, s, a, seg_words);
        else
            hipLaunchKernelGGL(ht_dec_refine_kernel<false>, dim3(a.nblocks), dim3(64), seg_words * 8, s, a, seg_words);
    }
    return hipGetLastError();
}

} // namespace grk_amd
