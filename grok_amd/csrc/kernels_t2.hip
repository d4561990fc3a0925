// grok_amd/csrc/kernels_t2.hip -- Tier-2 on the device: packet headers and the assembly of finished tile-parts in HBM.
//
// Replaces, for the single-layer HTJ2K packets this encoder makes, T2Compress::compressPacket (t2/T2Compress.cpp:123-333: the
// header's inclusion / zero-bit-plane tag trees t1/TagTree.cpp:170-218, the pass count and Lblock / length coding), the header's
// bit stuffing (t1/BitIO.cpp:46-175), the copy of the code-blocks' bytes behind it (T2Compress.cpp:300-333) and the frame of a
// tile-part (SOT markers/SOTMarker.cpp:41-72, PLT markers/LengthMarkers.cpp:313-374, SOD).  The host writer (t2_writer.cpp) is the
// oracle of this file's tests.
//
// KT1 t2_header_kernel: one workgroup per (packet, tile).
//   1. Every code-block's share of the header is a bit string that depends on its place in the band's grid and on its length
//      alone (t2_writer.cpp, "tag trees as this writer meets them": both trees are uniform): `nn` ones, (root: Kmax - 1 zeros,) `nn`
//      ones, the pass bit 0, Lblock's comma code, the length.  A workgroup-wide prefix sum over (bits, bytes) gives every block its
//      bit position in the RAW header and its byte offset in the packet's body; the bits are OR-ed into zeroed scratch words.
//   2. Stuffing -- a byte that follows 0xFF carries seven bits -- makes every byte's position depend on the bytes before it:
//      from a byte start p the next one is p + 8, or p + 15 behind an 0xFF (the 0xFF and the 7-bit byte as one step).  The chain is
//      cut into chunks of 256 raw bits; a chain enters a chunk at one of 15 offsets, and per (chunk, entry offset) a lane walks
//      the chunk -- from 0xFF candidate to 0xFF candidate of its phase (a mask per chunk and phase) --: exit offset + bytes produced.  Sixteen chunks make a super-chunk with a table of the same kind,
//      one lane walks the super-chunks, then the chunks of every super-chunk and the steps of every chunk are walked again from
//      their now known entry, the last walk writing the bytes.  The raw bits pass through LDS in windows of 16 KB.
// KT1b t2_frame_kernel: one lane per tile -- the tile-part's frame (SOT with Psot, the PLT marker segments with every packet's length as
//   a base-128 number, at most 65 532 bytes of them per segment, SOD; markers/SOTMarker.cpp:41-72, markers/LengthMarkers.cpp:313-374,
//   TileProcessor.cpp:719-734), the tile-part's length, and -- after a prefix sum over the call's tiles -- where every frame and every
//   packet goes.  Nothing comes to the host: an exchange can send the finished tile-parts with their sizes straight from the device.
// KT2 t2_gather_kernel: one wavefront per item -- a code-block's bytes, a packet's header (+ SOP / EPH), a tile-part's frame -- copies
//   it to its place in the output: 16-byte stores on the destination's alignment, unaligned 16-byte loads.
#include "kernels.h"

namespace grk_amd {
namespace {

constexpr uint32_t kChunkBits = 256;                              // raw bits per chunk
constexpr uint32_t kSup = 16;                                     // chunks per super-chunk
// raw words per LDS window: 4096 (16 KB; 512 chunks, 32 super-chunks) for the instance of large packets, 512 for the others -- a
// workgroup of a small packet does not claim 37 KB of LDS
constexpr uint64_t kByteMask = (1ull << 40) - 1;                  // the scan's packing: bits << 40 | bytes

__device__ __forceinline__ void put_bits(uint32_t* u, uint32_t pos, uint64_t v, uint32_t n)     // n <= 64 bits of v, MSB first, at bit `pos`
{
    if (!n) return;
    const uint64_t a = v << (64 - n);                             // left-aligned
    const uint32_t o = pos & 31u;
    uint32_t* w = u + (pos >> 5);
    const uint32_t w0 = (uint32_t)(a >> 32) >> o, w1 = (uint32_t)(a >> o), w2 = o ? (uint32_t)(a << (32 - o)) : 0u;
    // (workgroup scope: a packet's raw bits are written and read by ONE workgroup, and stay in its XCD's L2)
    if (w0) __hip_atomic_fetch_or(w, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (w1) __hip_atomic_fetch_or(w + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (w2) __hip_atomic_fetch_or(w + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The chain's steps over the window in LDS, through a 64-bit register that is refilled a word at a time (one LDS read per ~3 steps).
// A step takes 24 bits at the position: an 0xFF and the seven bits behind it (15 raw bits, two bytes out); else, when the next byte
// starts inside the chunk and is no 0xFF either, two bytes (16 bits); else one (8).
struct ChainWalk {
    const uint32_t* win; uint64_t buf; uint32_t wi, off;
    __device__ __forceinline__ void start(const uint32_t* w, uint32_t p)
    {
        win = w; wi = p >> 5; off = p & 31u;
        buf = ((uint64_t)w[wi] << 32) | w[wi + 1];
    }
    // -> raw bits consumed; b0 / b1: the bytes made (b1 only when two: the return value is 15 or 16)
    __device__ __forceinline__ uint32_t step(uint32_t p, uint32_t end, uint32_t& b0, uint32_t& b1)
    {
        if (off >= 32u) { buf = (buf << 32) | win[wi + 2]; ++wi; off -= 32u; }
        const uint32_t t = (uint32_t)(buf >> (40u - off)) & 0xFFFFFFu;
        b0 = t >> 16;
        const uint32_t n1 = (t >> 8) & 0xFFu;
        const bool ff = b0 == 0xFFu, two = !ff && n1 != 0xFFu && p + 8u < end;
        b1 = ff ? (t >> 9) & 0x7Fu : n1;
        const uint32_t adv = ff ? 15u : two ? 16u : 8u;
        off += adv;
        return adv;
    }
};

struct __attribute__((aligned(1))) U128 { uint32_t x, y, z, w; };
__device__ __forceinline__ void wave_copy(uint8_t* d, const uint8_t* s, uint64_t n, uint32_t lane)
{
    const uint64_t head = min(n, (uint64_t)((0 - (uintptr_t)d) & 15u));
    if (lane < head) d[lane] = s[lane];
    d += head; s += head; n -= head;
    const uint64_t nv = n >> 4;
    for (uint64_t i = lane; i < nv; i += 64) {
        U128 v;
        __builtin_memcpy(&v, s + 16 * i, 16);
        *reinterpret_cast<uint4*>(d + 16 * i) = make_uint4(v.x, v.y, v.z, v.w);
    }
    const uint64_t t0 = nv << 4;
    if (t0 + lane < n) d[t0 + lane] = s[t0 + lane];
}

} // namespace

template <uint32_t kWinWords>
__global__ __launch_bounds__(1024) void t2_header_kernel(T2HeaderArgs a)
{
    constexpr uint32_t kWinBits = kWinWords * 32, kWinChunks = kWinBits / kChunkBits, kWinSups = kWinChunks / kSup;
    __shared__ uint32_t win[kWinWords + 2];
    __shared__ uint16_t chunk_tab[kWinChunks * 16];               // [chunk][entry]: exit offset << 12 | bytes
    __shared__ uint32_t sup_tab[kWinSups * 16];                   // [super-chunk][entry]: exit offset << 16 | bytes
    __shared__ uint32_t ev_mask[kWinChunks * 8];                  // [chunk][phase]: bit k = the byte at chunk bit 8 k + phase is 0xFF (eight ones there)
    __shared__ uint32_t chunk_in[kWinChunks];                     // entry offset << 28 | first output byte (relative to the window's)
    __shared__ uint32_t sup_in[kWinSups + 1];
    __shared__ uint64_t wave_sum[16];
    __shared__ uint32_t carry_state[2];                           // between windows: entry offset, bytes so far

    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63u, wave = tid >> 6, nwaves = nthr >> 6;
    const uint32_t pk = blockIdx.x, tile = blockIdx.y;
    // (the packet's descriptor once, into scalar registers: indexed by a lane's band it would be a load per use)
    const T2Packet* const Pp = a.packets + pk;
    struct { uint32_t nblocks, h_at; } P = {Pp->nblocks, Pp->h_at};
    const uint32_t nbands = Pp->nbands;
    const uint32_t gw0 = Pp->gw[0], gw1 = Pp->gw[1], gw2 = Pp->gw[2], fb0 = Pp->first_block[0], fb1 = Pp->first_block[1], fb2 = Pp->first_block[2];
    const uint32_t ht0 = Pp->height[0], ht1 = Pp->height[1], ht2 = Pp->height[2], km0 = Pp->kmax[0], km1 = Pp->kmax[1], km2 = Pp->kmax[2];
    uint32_t* const u = a.ubits + (size_t)tile * a.u_words + Pp->u_at;
    const size_t row_base = (size_t)tile * a.bpt + Pp->row0;
    const uint32_t n0 = nbands > 0 ? gw0 * Pp->gh[0] : 0u, n1 = nbands > 1 ? gw1 * Pp->gh[1] : 0u;

    // ---- 1. the blocks' bit strings at their places in the raw header ------------------------------------------------------
    uint64_t carry = 1ull << 40;                                  // the packet's first bit: "not empty"
    if (tid == 0) __hip_atomic_fetch_or(u, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool bad = false;
    for (uint32_t base = 0; base < P.nblocks; base += nthr) {
        const uint32_t j = base + tid;
        const bool valid = j < P.nblocks;
        uint32_t b = 0, k = 0, len = 0, nn = 0, inc = 0, nbits = 0, zeros = 0;
        size_t row = 0;
        if (valid) {
            b = (j >= n0) + (j >= n0 + n1);
            k = j - (b == 0 ? 0u : b == 1 ? n0 : n0 + n1);
            const uint32_t gw = b == 0 ? gw0 : b == 1 ? gw1 : gw2, height = b == 0 ? ht0 : b == 1 ? ht1 : ht2;
            const uint32_t x = k % gw, y = k / gw, m = x | y;
            row = row_base + (b == 0 ? fb0 : b == 1 ? fb1 : fb2) + k;
            len = a.lengths[row];
            if (len >> kT2MaxLenBits) { bad = true; len = 0; }
            nn = m ? min((uint32_t)__builtin_ctz(m) + 1u, height) : height;
            const int fl = len ? 31 - __builtin_clz(len) : 0;
            inc = fl + 1 > 3 ? (uint32_t)(fl + 1 - 3) : 0u;
            zeros = k == 0 ? (b == 0 ? km0 : b == 1 ? km1 : km2) - 1u : 0u;      // the zero-bit-plane tree's root: Kmax - 1
            nbits = 2 * nn + zeros + 1 + (inc + 1) + (3 + inc);
        }
        const uint64_t v = ((uint64_t)nbits << 40) | len;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += o;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        uint64_t before = carry, total = 0;
        for (uint32_t w = 0; w < nwaves; ++w) { const uint64_t s = wave_sum[w]; if (w < wave) before += s; total += s; }
        __syncthreads();
        if (valid) {
            const uint64_t excl = before + incl - v;
            uint32_t pos = (uint32_t)(excl >> 40);
            a.rel[row] = (uint32_t)(excl & kByteMask);
            put_bits(u, pos, (1ull << nn) - 1ull, nn);
            pos += nn + zeros;
            // `nn` ones and the pass bit 0 | Lblock: `inc` ones and a zero | the length in 3 + inc bits
            const uint64_t tail = ((((((1ull << nn) - 1ull) << 1) << (inc + 1)) | (((1ull << inc) - 1ull) << 1)) << (3 + inc)) | len;
            put_bits(u, pos, tail, nn + 1 + inc + 1 + 3 + inc);
        }
        carry += total;
    }
    if (bad) atomicOr(a.status, 4u);
    const uint32_t nbits_total = (uint32_t)(carry >> 40);
    const uint32_t nwords = (nbits_total + 31u) >> 5;
    // the raw bits are complete before anybody reads them back.  A WORKGROUP-scope fence: an agent-scope one (__threadfence) writes the
    // XCD's L2 back on this machine -- 1 152 workgroups of a 64-tile call spent 70 of their 94 us in it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    // ---- 2. stuffing, window by window --------------------------------------------------------------------------------------
    uint8_t* const hdr = a.hdr + (size_t)tile * a.h_bytes + P.h_at;
    if (tid == 0) { carry_state[0] = 0; carry_state[1] = 0; }
    for (uint32_t w0 = 0; w0 * 32u < nbits_total; w0 += kWinWords) {
        __syncthreads();
        for (uint32_t i = tid; i < kWinWords + 2; i += nthr)
            win[i] = w0 + i < nwords ? __hip_atomic_load(u + w0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
        const uint32_t lim = min(nbits_total - w0 * 32u, kWinBits);          // raw bits of this window
        const uint32_t nch = (lim + kChunkBits - 1) / kChunkBits, nsup = (nch + kSup - 1) / kSup;
        __syncthreads();
        // Where the 0xFF bytes COULD be: for every raw word the bit positions at which eight ones start (three shift-and steps over
        // the word and its successor), sorted by phase -- position modulo 8 -- into a 32-bit mask per (chunk, phase).  A chain that
        // enters a chunk in phase f meets byte starts 8 k + f only, until an 0xFF moves it to phase f - 1: it steps from 0xFF to 0xFF.
        for (uint32_t i = tid; i < nch * 8u; i += nthr) ev_mask[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < nch * 8u; i += nthr) {                  // word i of the window = word j of chunk c
            const uint64_t x = ((uint64_t)win[i] << 32) | win[i + 1];
            uint64_t y = x & (x << 1); y &= y << 2; y &= y << 4;           // bit q: x's bits q .. q - 7 are ones
            const uint32_t e8 = (uint32_t)(y >> 32);                       // bit 31 - o: eight ones from bit o of the word on
            if (e8) {
                const uint32_t c = i >> 3, j = i & 7u;
#pragma unroll
                for (uint32_t f = 0; f < 8; ++f) {
                    const uint32_t v = (e8 >> (7u - f)) & 0x01010101u;     // the word's four byte starts of phase f: bits 24, 16, 8, 0
                    const uint32_t nib = ((v >> 24) & 1u) | ((v >> 15) & 2u) | ((v >> 6) & 4u) | ((v << 3) & 8u);
                    if (nib) atomicOr(&ev_mask[c * 8u + f], nib << (4u * j));
                }
            }
        }
        __syncthreads();
        // per (chunk, entry offset): where the chain leaves the chunk and how many bytes it makes on the way
        for (uint32_t job = tid; job < nch * 16u; job += nthr) {
            const uint32_t c = job >> 4, e = job & 15u;
            if (e == 15u) continue;
            const uint32_t limc = min(kChunkBits, lim - c * kChunkBits);       // raw bits of this chunk
            uint32_t f = e & 7u, k = e >> 3, cnt = 0;
            for (;;) {
                const uint32_t ns = limc > f ? (limc - f + 7u) >> 3 : 0u;      // byte starts of phase f inside the chunk
                if (k >= ns) break;
                const uint32_t m = ev_mask[c * 8u + f] >> k;
                if (!m) { cnt += ns - k; k = ns; break; }
                const uint32_t j = (uint32_t)__builtin_ctz(m);                  // j plain bytes, then the 0xFF and its 7-bit byte: 15 raw bits
                cnt += j + 2u;
                k += j + (f ? 2u : 1u);
                f = (f + 7u) & 7u;
            }
            // (a full chunk is left at bit 8 k + f >= 256: offset 0 .. 14 into the next one; the last chunk's exit is not used)
            const uint32_t p = 8u * k + f;
            chunk_tab[job] = (uint16_t)((min(p - min(p, kChunkBits), 14u) << 12) | cnt);
        }
        __syncthreads();
        for (uint32_t job = tid; job < nsup * 16u; job += nthr) {
            const uint32_t s = job >> 4;
            uint32_t e = job & 15u, cnt = 0;
            if (e == 15u) continue;
            for (uint32_t c = s * kSup; c < min((s + 1) * kSup, nch); ++c) { const uint32_t t = chunk_tab[c * 16u + e]; e = t >> 12; cnt += t & 0xFFFu; }
            sup_tab[job] = (e << 16) | cnt;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t e = carry_state[0], at = 0;
            for (uint32_t s = 0; s < nsup; ++s) { sup_in[s] = (e << 28) | at; const uint32_t t = sup_tab[s * 16u + e]; e = t >> 16; at += t & 0xFFFFu; }
            sup_in[nsup] = (e << 28) | at;
        }
        __syncthreads();
        if (tid < nsup) {
            uint32_t e = sup_in[tid] >> 28, at = sup_in[tid] & 0x0FFFFFFFu;
            for (uint32_t c = tid * kSup; c < min((tid + 1) * kSup, nch); ++c) { chunk_in[c] = (e << 28) | at; const uint32_t t = chunk_tab[c * 16u + e]; e = t >> 12; at += t & 0xFFFu; }
        }
        __syncthreads();
        const uint32_t out0 = carry_state[1];
        for (uint32_t c = tid; c < nch; c += nthr) {
            const uint32_t end = min((c + 1) * kChunkBits, lim);
            uint32_t p = c * kChunkBits + (chunk_in[c] >> 28), b0, b1;
            uint8_t* o = hdr + out0 + (chunk_in[c] & 0x0FFFFFFFu);
            ChainWalk cw;
            cw.start(win, p);
            while (p < end) {
                const uint32_t adv = cw.step(p, end, b0, b1);
                *o++ = (uint8_t)b0;
                if (adv > 8u) *o++ = (uint8_t)b1;
                p += adv;
            }
        }
        __syncthreads();
        if (tid == 0) {
            // (a window that is not the last is full: its exit offset is the next one's entry)
            carry_state[0] = sup_in[nsup] >> 28;
            carry_state[1] = out0 + (sup_in[nsup] & 0x0FFFFFFFu);
        }
    }
    __syncthreads();
    if (tid == 0) {
        a.pk_hdr[(size_t)tile * a.npackets + pk] = carry_state[1];
        a.pk_body[(size_t)tile * a.npackets + pk] = carry & kByteMask;
    }
}

__global__ __launch_bounds__(256) void t2_frame_kernel(T2FrameArgs a)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t t = tid; t < a.ntiles; t += blockDim.x) {
        uint8_t* const f = a.lit + (size_t)t * a.lit_stride;
        const uint32_t* const hl = a.pk_hdr + (size_t)t * a.npackets;
        const uint64_t* const bl = a.pk_body + (size_t)t * a.npackets;
        const uint32_t isot = a.tile_index[t];
        uint32_t pos = 0;
        f[pos++] = 0xFF; f[pos++] = 0x90; f[pos++] = 0; f[pos++] = 10; f[pos++] = (uint8_t)(isot >> 8); f[pos++] = (uint8_t)isot;
        pos += 4;                                                   // Psot: below
        f[pos++] = 0; f[pos++] = 1;                                 // TPsot, TNsot
        uint64_t sum = 0;
        if (a.plt) {
            uint32_t seg = pos, z = 0, fill = 0;                    // the open marker segment, its Zplt, its bytes of lengths
            f[pos++] = 0xFF; f[pos++] = 0x58; pos += 2; f[pos++] = 0;
            for (uint32_t i = 0; i < a.npackets; ++i) {
                const uint64_t v = (uint64_t)a.extra + hl[i] + bl[i];
                sum += v;
                uint32_t k = 1;
                for (uint64_t w = v >> 7; w; w >>= 7) ++k;
                if (fill + k > 65532u) {                            // a length is never split over two segments
                    f[seg + 2] = (uint8_t)((3 + fill) >> 8); f[seg + 3] = (uint8_t)(3 + fill);
                    if (++z > 255u) atomicOr(a.status, 8u);         // (Zplt is one byte)
                    seg = pos; fill = 0;
                    f[pos++] = 0xFF; f[pos++] = 0x58; pos += 2; f[pos++] = (uint8_t)z;
                }
                for (uint32_t j = k; j-- > 0;) f[pos++] = (uint8_t)(((v >> (7 * j)) & 0x7Fu) | (j ? 0x80u : 0u));
                fill += k;
            }
            f[seg + 2] = (uint8_t)((3 + fill) >> 8); f[seg + 3] = (uint8_t)(3 + fill);
        } else {
            for (uint32_t i = 0; i < a.npackets; ++i) sum += (uint64_t)a.extra + hl[i] + bl[i];
        }
        f[pos++] = 0xFF; f[pos++] = 0x93;
        const uint64_t len = pos + sum;
        if (len >> 32) atomicOr(a.status, 8u);                      // (Psot is 32 bits)
        f[6] = (uint8_t)(len >> 24); f[7] = (uint8_t)(len >> 16); f[8] = (uint8_t)(len >> 8); f[9] = (uint8_t)len;
        a.lit_len[t] = pos;
        a.part_len[t] = (uint32_t)len;
    }
    __syncthreads();
    if (tid == 0) {
        uint64_t at = a.dst_offset;
        for (uint32_t t = 0; t < a.ntiles; ++t) { a.tile_dst[t] = at; at += a.part_len[t]; }
        a.total[0] = at - a.dst_offset; a.total[1] = at;
    }
    __syncthreads();
    for (uint32_t t = tid; t < a.ntiles; t += blockDim.x) {
        uint64_t at = a.tile_dst[t] + a.lit_len[t];
        const uint32_t* const hl = a.pk_hdr + (size_t)t * a.npackets;
        const uint64_t* const bl = a.pk_body + (size_t)t * a.npackets;
        uint64_t* const dst = a.pk_dst + (size_t)t * a.npackets;
        for (uint32_t i = 0; i < a.npackets; ++i) { dst[i] = at; at += (uint64_t)a.extra + hl[i] + bl[i]; }
    }
}

__global__ __launch_bounds__(256) void t2_gather_kernel(T2GatherArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t nblk = (uint64_t)a.ntiles * a.bpt, npk = (uint64_t)a.ntiles * a.npackets;
    if (item < nblk) {
        const uint64_t tile = item / a.bpt;
        const uint32_t row = (uint32_t)(item - tile * a.bpt);
        const uint64_t q = tile * a.npackets + a.packet_of_block[row];
        wave_copy(a.out + a.pk_dst[q] + a.sop + a.pk_hdr[q] + a.eph + a.rel[item], a.arena + a.offsets[item], a.lengths[item], lane);
    } else if (item < nblk + npk) {
        const uint64_t q = item - nblk;
        const uint64_t tile = q / a.npackets;
        const uint32_t pk = (uint32_t)(q - tile * a.npackets);
        uint8_t* d = a.out + a.pk_dst[q];
        const uint32_t n = a.pk_hdr[q];
        if (a.sop) {                    // SOP marker segment: Lsop 4, Nsop = the packet's number in the tile modulo 65536 (T2Compress.cpp:149-164)
            if (lane < 6) d[lane] = lane == 0 ? 0xFF : lane == 1 ? 0x91 : lane == 2 ? 0 : lane == 3 ? 4 : lane == 4 ? (uint8_t)(pk >> 8) : (uint8_t)pk;
            d += 6;
        }
        wave_copy(d, a.hdr + tile * a.h_bytes + a.packets[pk].h_at, n, lane);
        if (a.eph && lane < 2) d[n + lane] = lane ? 0x92 : 0xFF;
    } else if (item < nblk + npk + a.ntiles) {
        const uint64_t t = item - nblk - npk;
        wave_copy(a.out + a.tile_dst[t], a.lit + t * a.lit_stride, a.lit_len[t], lane);
    }
}

hipError_t launch_t2_header(const T2HeaderArgs& a, uint32_t max_blocks_per_packet, hipStream_t s)
{
    if (!a.npackets || !a.ntiles) return hipSuccess;
    // (a packet of a few blocks -- small precincts, low resolutions -- does not need sixteen waves' barriers)
    const uint32_t threads = max_blocks_per_packet > 1024 ? 1024u : max_blocks_per_packet > 128 ? 256u : 64u;
    if (threads == 1024u) hipLaunchKernelGGL(t2_header_kernel<4096>, dim3(a.npackets, a.ntiles), dim3(threads), 0, s, a);
    else hipLaunchKernelGGL(t2_header_kernel<512>, dim3(a.npackets, a.ntiles), dim3(threads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_t2_frame(const T2FrameArgs& a, hipStream_t s)
{
    if (!a.ntiles) return hipSuccess;
    hipLaunchKernelGGL(t2_frame_kernel, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_t2_gather(const T2GatherArgs& a, hipStream_t s)
{
    const uint64_t items = (uint64_t)a.ntiles * a.bpt + (uint64_t)a.ntiles * a.npackets + a.ntiles;
    if (!items) return hipSuccess;
    hipLaunchKernelGGL(t2_gather_kernel, dim3((uint32_t)((items + 3) / 4)), dim3(256), 0, s, a);
    return hipGetLastError();
}

} // namespace grk_amd
