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
#include "kernels.h"
#include "ht_vlc_tables.h"
#include <type_traits>

namespace grk_amd {

namespace {

__device__ uint16_t g_vlc_dec[2048];         // [0..1023] first quad row, [1024..2047] others; index (c_q<<7)|7 bits

constexpr uint32_t kQuadStride = 32;         // quads per row in the per-block quad-info array (blocks <= 64 wide)
constexpr uint32_t kQuadWords  = 32 * 32;

// ---- K5a --------------------------------------------------------------------------------------------
// Both readers keep the aligned 8-byte word under the cursor and the next one in registers (fetched one
// word ahead of use, so the serial decoder never waits for memory) and refill FOUR bytes at a time with
// straight-line SWAR code: which bytes are bit-stuffed depends only on the byte read just before, so the
// four widths are computed at once and the bytes packed with three shifts.  One refill check per quad
// pair is enough for both (a pair consumes <= 31 VLC bits and <= 18 MEL bits).
struct WordWindow {
    const uint8_t* lo; const uint8_t* hi;      // readable range [lo, hi) of the coded buffer
    __device__ __forceinline__ uint64_t load(const uint8_t* p) const
    {
        // an aligned word that overlaps the buffer lies in a mapped page (words do not straddle pages);
        // its bytes outside [lo, hi) are never consumed (the readers count the bytes they may use)
        return (p + 8 > lo && p < hi) ? *reinterpret_cast<const uint64_t*>(p) : 0ull;
    }
};

struct RevReader : WordWindow {   // VLC: backward, LSB first; after a byte > 0x8F a byte whose 7 LSBs are ones carries 7 bits
    const uint8_t* bp;                         // next byte to read (addresses go down)
    uint64_t cur, prv;                         // the aligned word holding *bp and the word below it
    int left;                                  // bytes of the segment still unread; beyond it zeros are fed
    uint64_t acc; int n; uint32_t unstuff;
    __device__ __forceinline__ void init(const uint8_t* first, int count, const uint8_t* lo_, const uint8_t* hi_)
    {
        lo = lo_; hi = hi_; bp = first; left = count;
        const uint8_t* wp = first - ((uintptr_t)first & 7u);
        cur = load(wp); prv = load(wp - 8);
    }
    __device__ __forceinline__ void fill()
    {
        if (n > 32) return;
        const uint32_t o = (uint32_t)((uintptr_t)bp & 7u);
        // bytes bp-3 .. bp as a little-endian word: byte 3 (= *bp) is read first, so byte i follows byte i+1
        uint32_t w = o >= 3 ? (uint32_t)(cur >> (8 * (o - 3))) : (uint32_t)((cur << (8 * (3 - o))) | (prv >> (8 * (o + 5))));
        const int v = left < 0 ? 0 : (left > 4 ? 4 : left);
        w &= (uint32_t)(0xFFFFFFFF00000000ull >> (8 * v));                          // bytes past the segment read as 0
        const uint32_t l7 = w & 0x7F7F7F7Fu;
        const uint32_t g = (l7 + 0x70707070u) & w & 0x80808080u;                    // byte > 0x8F
        const uint32_t e = (l7 + 0x01010101u) & 0x80808080u;                        // 7 LSBs all ones
        const uint32_t st = e & ((g >> 8) | (unstuff << 31));                       // byte carries 7 bits
        unstuff = (g >> 7) & 1u;
        const uint32_t s3 = st >> 31, s2 = (st >> 23) & 1u, s1 = (st >> 15) & 1u;
        const uint32_t sh2 = 8u - s3, sh1 = sh2 + 8u - s2, sh0 = sh1 + 8u - s1;
        const uint32_t val = (w >> 24) | (((w >> 16) & 0xFFu) << sh2) | (((w >> 8) & 0xFFu) << sh1) | ((w & 0xFFu) << sh0);
        acc |= (uint64_t)val << n;
        n += 32 - (int)__builtin_popcount(st);
        bp -= 4; left -= 4;
        if (o < 4) {                                       // moved into the word below: fetch the next one ahead
            cur = prv;
            prv = load(bp - ((uintptr_t)bp & 7u) - 8);
        }
    }
    __device__ __forceinline__ uint32_t peek() const { return (uint32_t)acc; }      // valid after fill(): > 32 bits
    __device__ __forceinline__ void skip(uint32_t nb) { acc >>= nb; n -= (int)nb; }
};

struct MelReader : WordWindow {   // MEL: forward, MSB first; byte after 0xFF carries 7 bits; last byte |= 0x0F; then 0xFF
    const uint8_t* bp; uint64_t cur, nxt; int left; uint32_t unstuff;
    uint64_t tmp; int bits;             // un-stuffed bits, next bit at the MSB
    int k;
    int run;                            // what is left of the current run, in the reference's coding (:196-235, :1101-1111):
                                        // 2 * (zero events) + 1 if it ends with a one, 2 * (zero events - 1) otherwise
    __device__ __forceinline__ void init(const uint8_t* p, int size, const uint8_t* lo_, const uint8_t* hi_)
    {
        lo = lo_; hi = hi_; bp = p; left = size;
        const uint8_t* wp = p - ((uintptr_t)p & 7u);
        cur = load(wp); nxt = load(wp + 8);
        unstuff = 0; tmp = 0; bits = 0; k = 0; run = -1;
        fill();
        (void)event(false);                                // run < 0: decodes the first run
    }
    __device__ __forceinline__ void fill()
    {
        if (bits > 32) return;
        const uint32_t o = (uint32_t)((uintptr_t)bp & 7u);
        uint32_t w = o <= 4 ? (uint32_t)(cur >> (8 * o)) : (uint32_t)((cur >> (8 * o)) | (nxt << (64 - 8 * o)));   // byte 0 first
        const int v = left < 0 ? 0 : (left > 4 ? 4 : left);
        const uint32_t valid = (uint32_t)((1ull << (8 * v)) - 1ull);
        w = (w & valid) | ~valid;                                                   // past the segment: 0xFF
        w |= (left >= 1 && left <= 4) ? 0x0Fu << (8 * (left - 1)) : 0u;             // the segment's last byte
        const uint32_t ff = ((w & 0x7F7F7F7Fu) + 0x01010101u) & w & 0x80808080u;    // byte == 0xFF
        const uint32_t st = (ff << 8) | (unstuff << 7);                             // byte follows a 0xFF: 7 bits, MSB dropped
        unstuff = ff >> 31;
        w &= ~st;
        const uint32_t f0 = (st >> 7) & 1u, f1 = (st >> 15) & 1u, f2 = (st >> 23) & 1u, f3 = st >> 31;
        const uint32_t h0 = 24u + f0, h1 = h0 - 8u + f1, h2 = h1 - 8u + f2, h3 = h2 - 8u + f3;
        const uint32_t val = ((w & 0xFFu) << h0) | (((w >> 8) & 0xFFu) << h1) | (((w >> 16) & 0xFFu) << h2) | ((w >> 24) << h3);
        tmp |= (uint64_t)val << (32 - bits);
        bits += 32 - (int)h3;
        bp += 4; left -= 4;
        if (o >= 4) {
            cur = nxt;
            nxt = load(bp - ((uintptr_t)bp & 7u) + 8);
        }
    }
    // One MEL event if `need` (:1101-1111): returns 1 if the run ends here with a one.  The decode of the next run sits
    // behind a branch the whole wave skips when no lane has used its run up: in dense blocks (contexts rarely zero) and in
    // empty ones (long runs) that is most of the time, and every instruction on this chain costs the wave ~8 cycles.
    __device__ __forceinline__ uint32_t event(bool need)
    {
        run -= need ? 2 : 0;
        const uint32_t ev = run == -1;
        if (run < 0) {                                                              // decode the next run (:196-235)
            const uint32_t e = (k < 8 ? 0x22111000u >> (4 * k) : 0x54332u >> (4 * (k - 8))) & 0xFu;   // MEL exponents (:196)
            const uint32_t top = (uint32_t)(tmp >> 32);
            const bool one = (top >> 31) != 0;             // '1': 2^e zero events; '0' + e bits: that many, then a one
            run = one ? (int)((2u << e) - 2u) : (int)((((top >> (31u - e)) & ((1u << e) - 1u)) << 1) | 1u);
            k = one ? (k < 12 ? k + 1 : 12) : (k > 0 ? k - 1 : 0);
            const uint32_t used = one ? 1u : e + 1u;
            tmp <<= used; bits -= (int)used;
        }
        return ev;
    }
};

// UVLC prefix: '1' -> 1, '01' -> 2, '001' -> 3 + 1-bit suffix, '000' -> 5 + 5-bit suffix (:706-716), looked up by
// the three next bits in a table packed into one 64-bit constant: entry = prefix_len | suffix_len << 2 | base << 5
__device__ __forceinline__ void uvlc_prefix(uint32_t bits, uint32_t& pl, uint32_t& sl, uint32_t& base)
{
    const uint32_t d = (uint32_t)(0x21422167214221B7ull >> (8 * (bits & 7u))) & 0xFFu;
    pl = d & 3u; sl = (d >> 2) & 7u; base = d >> 5;
}

__global__ void ht_dec_vlc_kernel(HtDecArgs a)
{
    // CxtVLC decode tables in LDS: one dependent lookup per quad sits on the serial chain
    __shared__ uint16_t tbl_l[2048];
    for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x)
        reinterpret_cast<uint32_t*>(tbl_l)[i] = reinterpret_cast<const uint32_t*>(g_vlc_dec)[i];
    __syncthreads();
    const uint32_t blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= a.nblocks) return;
    const HtDecBlock in = a.table[blk];
    const HtBlockDesc bd = a.blocks[blk % a.blocks_per_tile];
    uint32_t* qi = a.quads + (size_t)blk * kQuadWords;
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t QW = (w + 1) >> 1, QH = (h + 1) >> 1;
    const uint32_t mm = in.missing_msbs;
    const uint8_t* D = a.coded + in.offset;
    // (blocks with refinement passes: their SigProp / MagRef segment follows the cleanup segment)
    const int lcup = (int)in.length - (a.refine ? (int)a.refine[blk].x : 0);

    if (in.length == 0) return;                            // absent (K5b writes the zeros) or outside the decoded region
    bool bad = mm > 29 || lcup < 2;
    int scup = 0;
    if (!bad) {
        scup = ((int)D[lcup - 1] << 4) + (D[lcup - 2] & 0xF);
        bad = scup < 2 || scup > lcup || scup > 4079;
    }
    if (bad) { atomicOr(a.status, 4u); a.ms_len[blk] = 0xFFFFFFFFu; return; }
    a.ms_len[blk] = (uint32_t)(lcup - scup);

    const uint8_t* buf_hi = a.coded + a.coded_bytes;
    MelReader mel;
    mel.init(D + (lcup - scup), scup - 1, a.coded, buf_hi);
    RevReader vlc;
    {
        const uint32_t d0 = D[lcup - 2];
        vlc.init(D + (lcup - 3), scup - 2, a.coded, buf_hi);
        vlc.acc = d0 >> 4; vlc.n = 4 - (((d0 >> 4) & 7u) == 7u ? 1 : 0);
        vlc.unstuff = (d0 | 0xFu) > 0x8Fu;
    }
    const uint32_t NP = (QW + 1) >> 1;  // quad pairs per row
    uint64_t sa = 0;                     // significance of the bottom sample row of the quad row above: bit x
    // One quad row; the first row has its own contexts, table and u-value rules, so it gets its own instance.
    auto quad_row = [&](auto first_tag, uint32_t qy) -> bool {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint16_t* tbl = tbl_l + (FIRST ? 0 : 1024);
        uint32_t* qrow = qi + qy * kQuadStride;
        uint64_t sw = sa, sn = 0;
        uint32_t west = 0, chain = 0;    // sample 2 * q0 - 1 of the row above; the west quad's contribution to c_q
        for (uint32_t q0 = 0; q0 < QW; q0 += 2) {
            vlc.fill();                  // > 32 bits: a quad pair consumes at most 7 + 7 + 17
            mel.fill();                  // > 18 bits: at most three runs of 6 bits
            const bool has1 = q0 + 1 < QW;
            const uint32_t up = (uint32_t)sw;            // samples 2 * q0 ... of the row above
            const uint32_t nb0 = ((up << 1) | west) & 0xFu, nb1 = (up >> 1) & 0xFu;   // nw, n, ne, nf of each quad
            west = (up >> 3) & 1u; sw >>= 4;
            // ---- quad q0
            uint32_t c = chain;
            if (!FIRST) c |= ((nb0 & 3u) ? 1u : 0u) | ((nb0 & 12u) ? 4u : 0u);
            uint32_t t0 = tbl[(c << 7) | (vlc.peek() & 0x7Fu)];
            uint32_t ev = mel.event(c == 0);
            t0 = (c == 0 && !ev) ? 0u : t0;
            vlc.skip(t0 & 7u);
            const uint32_t rho0 = (t0 >> 4) & 0xFu;
            chain = FIRST ? ((rho0 & 1u) | (rho0 >> 1)) : ((((rho0 >> 2) | (rho0 >> 3)) & 1u) << 1);
            // ---- quad q0 + 1 (absent when the row has an odd number of quads)
            c = chain;
            if (!FIRST) c |= ((nb1 & 3u) ? 1u : 0u) | ((nb1 & 12u) ? 4u : 0u);
            uint32_t t1 = tbl[(c << 7) | (vlc.peek() & 0x7Fu)];
            ev = mel.event(has1 && c == 0);
            t1 = (!has1 || (c == 0 && !ev)) ? 0u : t1;
            vlc.skip(t1 & 7u);
            const uint32_t rho1 = (t1 >> 4) & 0xFu;
            chain = FIRST ? ((rho1 & 1u) | (rho1 >> 1)) : ((((rho1 >> 2) | (rho1 >> 3)) & 1u) << 1);
            const uint32_t sb = ((rho0 >> 1) & 1u) | (((rho0 >> 3) & 1u) << 1) | (((rho1 >> 1) & 1u) << 2) | (((rho1 >> 3) & 1u) << 3);
            sn = (sn >> 4) | ((uint64_t)sb << 60);
            // ---- u values of the pair (:668-777), written without branches: prefix0, prefix1, suffix0, suffix1,
            // each present only if its quad has u_off set
            // (r02: the same by ONE look-up in a 256-entry LDS table -- u_off of both quads + six bits -> prefix / suffix
            //  lengths and bases -- removed ~25 instructions from the chain and was 3 % SLOWER: the look-up's latency sits on
            //  the chain as well)
            const uint32_t uo0 = (t0 >> 3) & 1u, uo1 = (t1 >> 3) & 1u;
            uint32_t add = 1, onebit = 0;
            uint32_t v = vlc.peek();
            uint32_t pl, sl, base, pl2, sl2, base2;
            uvlc_prefix(v, pl, sl, base);
            pl = uo0 ? pl : 0; sl = uo0 ? sl : 0; base = uo0 ? base : 0;
            v >>= pl;
            if (FIRST) {                                           // both quads: a MEL event picks the variant
                const uint32_t both = uo0 & uo1;
                const uint32_t e2 = mel.event(both != 0);
                add = (both & e2) ? 3u : 1u;
                onebit = both & (e2 ^ 1u) & (pl > 2 ? 1u : 0u);    // second quad is a single bit
            }
            uvlc_prefix(v, pl2, sl2, base2);
            pl2 = uo1 ? pl2 : 0; sl2 = uo1 ? sl2 : 0; base2 = uo1 ? base2 : 0;
            if (FIRST) { pl2 = onebit ? 1u : pl2; sl2 = onebit ? 0u : sl2; base2 = onebit ? (v & 1u) + 1u : base2; }
            v >>= pl2;
            const uint32_t U0 = base + (v & ((1u << sl) - 1u)) + (uo0 ? add : 1u);
            v >>= sl;
            const uint32_t U1 = base2 + (v & ((1u << sl2) - 1u)) + (uo1 ? add : 1u);
            vlc.skip(pl + pl2 + sl + sl2);
            if (U0 > mm || U1 > mm) return false;                  // :1194
            *reinterpret_cast<uint2*>(&qrow[q0]) = make_uint2(t0 | (U0 << 16), t1 | (U1 << 16));
        }
        sa = sn >> (64u - 4u * NP);
        return true;
    };
    bool ok = quad_row(std::true_type{}, 0);
    for (uint32_t qy = 1; ok && qy < QH; ++qy) ok = quad_row(std::false_type{}, qy);
    if (!ok) { atomicOr(a.status, 4u); a.ms_len[blk] = 0xFFFFFFFFu; }
}

// ---- K5b --------------------------------------------------------------------------------------------
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
    const uint32_t blk = blockIdx.x;
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
    const uint32_t* qi = a.quads + (size_t)blk * kQuadWords;
    const uint32_t q = x >> 1, right = x & 1u;
    uint32_t Eprev = 0;                                   // exponent of this column's bottom sample, row above
    uint32_t bitpos = 0;
    uint32_t range = 0;
    uint32_t info = col_ok ? qi[q] : 0u;
    for (uint32_t qy = 0; qy < QH; ++qy) {
        const uint32_t cur = info;
        if (qy + 1 < QH) info = col_ok ? qi[(qy + 1) * kQuadStride + q] : 0u;     // prefetch next row's quad info
        const uint32_t rho = (cur >> 4) & 0xFu, e1 = (cur >> 8) & 0xFu, ek = (cur >> 12) & 0xFu;
        uint32_t U = cur >> 16;
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
            if (qy > 0 && (rho & (rho - 1))) U += E > 2 ? E - 2 : 0;
        }
        // this lane's two samples: i = 2*right (top) and 2*right + 1 (bottom)
        const uint32_t it = 2 * right, ib = it + 1;
        const uint32_t st = (rho >> it) & 1u, sb = (rho >> ib) & 1u;
        const uint32_t mt = st ? U - ((ek >> it) & 1u) : 0u;
        const uint32_t mb = sb ? U - ((ek >> ib) & 1u) : 0u;
        const uint32_t incl = wave_incl_scan(mt + mb);
        const uint32_t pos = bitpos + incl - (mt + mb);
        bitpos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // up to 62 bits from the raw stream
        const uint32_t wi = pos >> 5, sh = pos & 31;
        const uint32_t r0 = raw[wi], r1 = raw[wi + 1], r2 = raw[wi + 2];
        const uint32_t win0 = __builtin_amdgcn_alignbit(r1, r0, sh), win1 = __builtin_amdgcn_alignbit(r2, r1, sh);   // 64 bits from pos on
        const uint32_t bt = win0 & ((1u << mt) - 1u);                                                          // mt, mb <= 31
        const uint32_t bb = __builtin_amdgcn_alignbit(win1, win0, mt) & ((1u << mb) - 1u);
        const uint32_t vt = bt | (((e1 >> it) & 1u) << mt) | 1u;
        const uint32_t vb = bb | (((e1 >> ib) & 1u) << mb) | 1u;
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

static bool g_dec_tables_ready[16] = {false};

hipError_t launch_ht_decode(const HtDecArgs& a, uint32_t max_ms_bytes, hipStream_t s)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 16 && !g_dec_tables_ready[dev]) {
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_vlc_dec), HT_VLC_DEC0, sizeof(HT_VLC_DEC0), 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_vlc_dec), HT_VLC_DEC1, sizeof(HT_VLC_DEC1), sizeof(HT_VLC_DEC0), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        g_dec_tables_ready[dev] = true;
    }
    // K5a: few lanes per wave when blocks are few, so that >= ~4 waves per SIMD overlap their latencies
    // K5a is one serial chain per lane.  A wave costs the same issue slots however many lanes are
    // live, so few lanes per wave multiply the instruction count, while few waves per SIMD leave the
    // chain's own latency exposed: aim for ~1.5-2 waves per SIMD (1024 SIMDs), measured optimum.
    // (r02, 8K, 49 152 blocks: 32, 48 and 64 lanes per wave take the same time -- what lasts is the chain of one wave, and a
    //  second wave on the SIMD interleaves for free; 16 and 24 lanes are slower.  Running this kernel BESIDE the wide ones of
    //  another frame does not pay either: next to K5b both take twice as long, next to the inverse DWT 1.4x / 1.5x -- the
    //  chain's next instruction waits behind whatever holds the SIMD's ALU for its four cycles, wave priority or not)
    uint32_t lanes = 64;
    while (lanes > 16 && (a.nblocks + lanes - 1) / lanes < 1280) lanes >>= 1;
    hipLaunchKernelGGL(ht_dec_vlc_kernel, dim3((a.nblocks + lanes - 1) / lanes), dim3(lanes), 0, s, a);
    const uint32_t raw_words = (max_ms_bytes * 8u) / 32u + 8u;
    if (a.irreversible)
        hipLaunchKernelGGL(ht_dec_ms_kernel<true>, dim3(a.nblocks), dim3(64), raw_words * 4, s, a, raw_words);
    else
        hipLaunchKernelGGL(ht_dec_ms_kernel<false>, dim3(a.nblocks), dim3(64), raw_words * 4, s, a, raw_words);
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
