// grok_amd/csrc/kernels.h -- host-callable launchers of the gfx950 kernels.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include "geometry.h"

namespace grk_amd {

// ---- K1: ingest + DC shift + forward colour transform (kernels_ingest.hip) -------------------
struct IngestArgs {
    const void* pixels;      // tiles back to back, component-major planar, tight
    int32_t*    planes;      // [tile][comp] planes, `stride` elements per row, `pitch` per plane
    uint32_t w, h, stride;
    uint64_t pitch;
    uint32_t ncomp, ntiles;
    uint32_t bytes_per_sample;   // 1 or 2
    int32_t  dc;                 // 2^(prec-1) or 0
    int32_t  sext;               // signed samples: 1 << (8*bytes_per_sample - 1) (stored as int8/int16), else 0
    int      mct;                // apply RCT/ICT to components 0..2
    int      irreversible;
};
hipError_t launch_ingest(const IngestArgs& a, hipStream_t s);

// ---- K2: one forward DWT level, vertical + horizontal fused (kernels_dwt.hip) ----------------
struct DwtLevelArgs {
    const int32_t* in;  uint32_t in_stride;  uint64_t in_pitch;    // current LL, cw x ch
    int32_t* ll;        uint32_t ll_stride;  uint64_t ll_pitch;    // next LL (sw x sh)
    int32_t* mallat;    uint32_t m_stride;   uint64_t m_pitch;     // HL/LH/HH go to their Mallat slots
    uint32_t cw, ch;      // size of the level being transformed
    uint32_t px, py;      // parity of the level's origin on its grid: 1 = the first column / row is a high-pass sample
    uint32_t nplanes;
    uint32_t seg_pairs;   // row pairs per workgroup
    int      irreversible;
    // level 0 fused with K1 (launch_dwt_level0_fused): rows come from the caller's pixel planes
    const void* pixels;   // tiles back to back, component-major planar, tight (as IngestArgs::pixels)
    uint32_t px_bytes;    // 1 or 2 bytes per sample
    int32_t  dc;          // 2^(prec-1) or 0
    int32_t  sext;        // signed samples: sign bit of the stored word (see IngestArgs), else 0
    uint32_t ncomp;       // components per tile
    uint32_t comp0, zdiv; // set by the launcher: first component of a z slot, z slots per tile
    // level 0 fused with K1 in a pipelined encode also resets K3's arena allocator (its first workgroup; kernels_ht.hip ht_alloc_reset):
    // one launch less on the chain that bounds frames of 4096 x 4096 and below
    unsigned long long* alloc_reset; uint32_t alloc_chunk_units;
    int      h16;         // reversible, 8-bit pixels: every plane (in, ll, mallat) holds int16 instead of int32
    int      xcd;         // XCD-aware workgroup order (kernels_dwt.hip)
    int      pk;          // h16 and every intermediate of this level within 16 bits: arithmetic on packed int16 pairs
};
hipError_t launch_dwt_level(const DwtLevelArgs& a, hipStream_t s);
hipError_t launch_dwt_level0_fused(const DwtLevelArgs& a0, uint32_t ntiles, uint32_t ncomp, int mct, hipStream_t s);

// ---- K3: HT cleanup encoder, one wavefront per code-block (kernels_ht.hip) -------------------
struct HtBlockDesc {        // one per code-block of a tile-component set (all comps of one tile)
    uint32_t px, py;        // origin in the plane
    uint16_t w, h;
    uint16_t comp;
    uint8_t  kmax;
    uint8_t  pad;
    float    inv_step;      // 1/stepsize (irreversible)
};
constexpr uint32_t kHtMaxClasses = 24;      // (resolution, LDS need): up to 10 levels + 1, two each
struct HtClass {
    const uint32_t* sel;      // device: indices (within a tile) of the blocks of this class; nullptr = all blocks in order
    uint32_t count;
    uint32_t max_kmax, max_samples, max_quads;   // extents that size the class's worst-case LDS buffers
    uint32_t ovf_base;        // first entry of the class's part of HtArgs::ovf_list
    uint32_t cap_kmax;        // the exponent most of the class's samples have: sizes the capped LDS buffers
};
struct HtArgs {
    const int32_t* mallat; uint32_t stride; uint64_t pitch;   // planes [tile][comp] (stride, pitch in elements)
    int h16;                                                  // the planes hold int16 coefficients (reversible only)
    int room;                                                 // the launch runs beside the next frame's DWT level 0: the instance that leaves it registers (kernels_ht.hip)
    const HtBlockDesc* blocks; uint32_t blocks_per_tile; uint32_t ncomp; uint32_t ntiles;
    uint8_t*  arena; uint64_t arena_bytes;      // coded bytes of all blocks (chunked region allocator, kernels_ht.hip)
    unsigned long long* alloc;                  // kHtAllocBytes of allocator state: [0] status flags (bit 0 arena
                                                // overflow, bit 1 magnitude out of contract), [1] bytes used, region words
    uint32_t* lengths;                          // [ntiles*blocks_per_tile]
    unsigned long long* offsets;                // [ntiles*blocks_per_tile]
    const uint32_t* sel; uint32_t sel_count;    // set per launch by launch_ht_encode from `classes`
    uint32_t* ovf_list; uint32_t ovf_base;      // [ntiles * blocks_per_tile] blocks whose raw streams outgrew the capped LDS
                                                // buffers (per class from ovf_base; count in alloc[2 + class]): coded again
                                                // by the fallback launch.  nullptr: worst-case buffers, no fallback
    HtClass classes[kHtMaxClasses]; uint32_t num_classes;   // block classes of a tile (= resolutions, finest first), each launched on its own
    uint32_t region_mask;         // (power of two <= kHtAllocRegions) - 1: block i allocates from region i & mask
    uint32_t chunk_units;         // 16-byte units a region takes from the shared cursor at a time (kHtAllocChunk / 16, or half of it for small jobs)
    const uint32_t* vlc_tab;      // the CxtVLC encode table on the device (set by launch_ht_classes)
    int irreversible;
};
size_t ht_lds_bytes(uint32_t samples, uint32_t quads, uint32_t kmax);
// r03: 64 region words and 64 KiB chunks (r01 / r02: 16 and 256 KiB -- the same 4 MiB of slack at most).  An atomic on a region word
// executes at the memory side, one after the other per word, and a block coder waits for its answer: with 16 words the round trip
// was 10.8 % of K3's time (counters of a build that stops behind it), with 64 the 8K frame's K3 takes 0.30 instead of 0.315 ms and
// the pipelined step 0.422 instead of 0.437 (128 / 256 words: the same; two words: 0.77 ms)
#ifndef GRK_HT_ALLOC_REGIONS
#define GRK_HT_ALLOC_REGIONS 64
#endif
constexpr uint32_t kHtAllocRegions = GRK_HT_ALLOC_REGIONS;           // region words available; a launch uses region_mask + 1 of them
constexpr uint32_t kHtAllocChunk = 64u << 10;      // bytes a region takes from the shared cursor at a time (> twice the largest block)
constexpr uint32_t kHtAllocChunkSmall = 32u << 10; // ... in a job of few blocks (the slack of half-used chunks counts there)
constexpr size_t   kHtAllocBytes = 256u * (1u + kHtAllocRegions);   // 32 status / cursor / class words, then one 256-byte line per region word

// the allocator's initial state, written by `nthreads` lanes of one workgroup (ht_alloc_init_kernel; the fused level 0 of a pipelined
// encode): [0] status flags, [1] cursor (bytes), [2 + class] blocks handed to the fallback launch = 0; every region word "chunk full"
// so that the first allocation refills -- the start field holds a value no real chunk has, otherwise waves waiting for the refill
// could not tell the first chunk (start 0) from this state
__device__ __forceinline__ void ht_alloc_reset(unsigned long long* flagbuf, uint32_t chunk_units, uint32_t t, uint32_t nthreads)
{
    if (t < 32) flagbuf[t] = 0;
    for (uint32_t r = t; r < kHtAllocRegions; r += nthreads) flagbuf[32 * (1 + r)] = (0xFFFFFFFFFFull << 24) | chunk_units;
}
hipError_t launch_ht_encode(const HtArgs& a, hipStream_t s);          // allocator reset + every class
hipError_t launch_ht_alloc_init(const HtArgs& a, hipStream_t s);
hipError_t launch_ht_classes(const HtArgs& a, uint32_t first, uint32_t last, hipStream_t s);   // classes [first, last)
// sum of q^2 over each block of the planes an encode left (kernels_ingest.hip: the rate-control hook)
hipError_t launch_block_energy(const void* mallat, int h16, int irreversible, uint32_t stride, uint64_t pitch, const HtBlockDesc* blocks,
                               uint32_t blocks_per_tile, uint32_t ncomp, uint64_t nblocks, unsigned long long* out, hipStream_t s);
uint32_t   dwt_strip_cols();      // output columns a K2 workgroup owns
uint32_t   dwt_level_strip_cols(const DwtLevelArgs& a);   // ... for this level (the packed 5/3 kernel's strips are wider)
uint32_t   idwt_strip_pairs();    // coefficient pairs a K6 workgroup owns

// ---- K5: HT cleanup decoder + dequantisation (kernels_htdec.hip) --------------------------------
constexpr uint32_t kSkipBlock = 0xFFFFFFFFu;   // missing_msbs of a zero-length row: the block lies outside the decoded region
struct HtDecBlock {          // one per code-block, same layout as grk_amd_coded_block
    uint64_t offset;         // first byte of the block's cleanup pass inside `coded`
    uint32_t length;         // Lcup (0: block has no data -> all samples zero)
    uint32_t missing_msbs;   // band numbps - block numbps (T1DecompressScheduler.cpp:59)
};
struct HtDecArgs {
    const HtDecBlock* table;                   // [ntiles * blocks_per_tile], device
    const HtBlockDesc* blocks;                 // geometry per block of one tile; inv_step = decode scale (irreversible)
    uint32_t blocks_per_tile, nblocks, ncomp;
    const uint8_t* coded; uint64_t coded_bytes; // device buffer holding every block's bytes
    const uint32_t* active;                    // [nactive] the blocks with data, K5a's lanes (null: all nblocks)
    uint32_t nactive;
    uint32_t* quads;                           // K5a -> K5b: 16 bits per quad, [nblocks][32 * 32] (kernels_htdec.hip: 9 bits of the CxtVLC entry | (u_q + 1) << 9)
    uint32_t* ms_len;                          // [nblocks] MagSgn bytes (0xFFFFFFFF: block rejected)
    unsigned int* status;                      // bit 2: a block was rejected; bit 3: a value did not fit the 16-bit planes
    int32_t* mallat; uint32_t stride; uint64_t pitch;
    int irreversible;
    int h16;                                   // reversible: the planes hold int16 (strides / pitches in elements all the same)
    int h16_bias;                              // ... and a coefficient outside [-bias, bias) raises bit 3 of *status (32768: what fits; 2048:
                                               // what the packed inverse transform takes, pk16.h)
    const uint2* refine;                       // [nblocks] {bytes of the SigProp / MagRef segment at the end of the block's
                                               // data, coding passes in total (1..3)}, or null: cleanup passes only
    uint32_t max_refine_bytes;                 // largest such segment
    uint32_t ms_first, ms_count, ms_bpc;       // K5b of a part of the blocks: [ms_first, ms_first + ms_count) of every component's
                                               // ms_bpc blocks (ms_count = 0: all nblocks)
};
// K5a + K5b (+ K5c) on one stream, or in two parts: front = tables + K5a, ms = K5b (+ K5c) of a.ms_first / ms_count
hipError_t launch_ht_decode(const HtDecArgs& a, uint32_t max_ms_bytes, hipStream_t s);
hipError_t launch_ht_decode_front(const HtDecArgs& a, hipStream_t s);
hipError_t launch_ht_decode_ms(const HtDecArgs& a, uint32_t max_ms_bytes, hipStream_t s);
// a decode call's tables: `bytes` of pinned host memory (device-visible address) -> dst, the 16-byte status block cleared
hipError_t launch_dec_upload(const void* pinned, void* dst, size_t bytes, void* status, hipStream_t s);

// ---- K8: Part-1 (EBCOT) block decoder + dequantisation (kernels_t1dec.hip) -----------------------
struct T1DecArgs {
    const HtDecBlock* table;                   // missing_msbs field carries numbps | numpasses << 8
    const HtBlockDesc* blocks;                 // pad = band orientation, inv_step = band step size (irreversible)
    uint32_t blocks_per_tile, nblocks, ncomp;
    const uint8_t* coded; uint64_t coded_bytes;
    int32_t* work;                             // [nblocks][64*64] decoded values
    unsigned int* status;
    int32_t* mallat; uint32_t stride; uint64_t pitch;
    int irreversible;
    uint32_t cblksty;                          // COD code-block style bits (LAZY 1, RESET 2, TERMALL 4, VSC 8, PTERM 16, SEGSYM 32)
    const uint32_t* seg_first;                 // [nblocks + 1] first codeword segment of each block, or null: one segment
    const uint2* segs;                         // {bytes, passes} per segment
    const uint32_t* list; uint32_t count;      // the blocks this launch decodes (one wave each, in this order), or null: all nblocks
};
hipError_t launch_t1_decode(const T1DecArgs& a, hipStream_t s);

// ---- K8L: the same decoder with one code-block per LANE + reconstruction from bit-plane bitmaps (kernels_t1lanes.hip) ----
constexpr uint32_t kT1WorkBytes = 16384;       // a block's share of the Part-1 workspace (K8: 64 x 64 values; K8L: t1_lanes.h)
constexpr uint32_t kT1LaneMaxPlanes = 14;      // bit-planes whose bitmaps fit behind a block's state there
constexpr uint32_t kT1LaneMinRows = 9;         // a lane block has at least three stripes (t1_lanes.h: stripe hand-over)
struct T1LaneArgs {
    const HtDecBlock* table;                   // as T1DecArgs
    const HtBlockDesc* blocks;
    uint32_t blocks_per_tile, ncomp;
    const uint32_t* list; uint32_t count;      // the blocks of this launch: lane i of wave k decodes list[64 k + i]
    const uint8_t* coded; uint64_t coded_bytes;
    uint64_t* work;                            // [nblocks][kT1WorkBytes / 8]
    int32_t* mallat; uint32_t stride; uint64_t pitch;
    int irreversible;
    int pass_sync;                             // the waves hold blocks of equal bit-plane / pass counts and run them pass by pass
};
constexpr uint32_t kT1NoBlock = 0xFFFFFFFFu;   // list entry of a lane without a block
hipError_t launch_t1_lanes(const T1LaneArgs& a, hipStream_t s);       // t1_lanes_kernel, then t1_recon_kernel
// both decoders in ONE launch (the blocks of d.list first, then the lane waves), then t1_recon_kernel: a frame's block decoding on one stream
hipError_t launch_t1_fused(const T1DecArgs& d, const T1LaneArgs& a, hipStream_t s);

// ---- K6: one inverse DWT level, horizontal + vertical fused (kernels_idwt.hip) ------------------
struct IdwtLevelArgs {
    const int32_t* ll;     uint32_t ll_stride;  uint64_t ll_pitch;   // LL of the level (sw x sh)
    const int32_t* mallat; uint32_t m_stride;   uint64_t m_pitch;    // HL/LH/HH read from their Mallat slots
    int32_t* out;          uint32_t out_stride; uint64_t out_pitch;  // synthesised level, cw x ch
    uint32_t cw, ch;
    uint32_t px, py;      // parity of the level's origin (as DwtLevelArgs)
    uint32_t nplanes;
    uint32_t seg_pairs;
    int      irreversible;
    // last level fused with K7 (launch_idwt_level0_fused): the rows leave as the caller's pixels
    void*    pixels;      // tiles back to back, component-major planar, tight (as EgressArgs::pixels)
    uint32_t px_bytes;    // 1 or 2 bytes per sample
    int32_t  dc, lo, hi;  // DC shift and clamp range (EgressArgs)
    int      mct;
    uint32_t ncomp;       // components per tile
    uint32_t comp0, zdiv; // set by the launcher: first component of a z slot, z slots per tile
    uint32_t wx0, wy0, wx1, wy1;   // window of the tile the pixels are for (the whole tile: 0, 0, cw, ch)
    // region decode: only the strips [strip0, strip0 + nstrips) x row segments [seg0, seg0 + nsegs) (0 = all)
    uint32_t strip0, nstrips, seg0, nsegs;
    int      xcd;         // XCD-aware workgroup order (as DwtLevelArgs)
    int      h16;         // reversible: ll / mallat / out hold int16; a synthesised value that does not fit sets bit 3 of *status
    int      pk;          // h16 and every coefficient within +-kPkDecodeBound (pk16.h): arithmetic on packed int16 pairs
    unsigned int* status;
};
hipError_t launch_idwt_level(const IdwtLevelArgs& a, hipStream_t s);
hipError_t launch_idwt_level0_fused(const IdwtLevelArgs& a0, uint32_t ntiles, uint32_t ncomp, hipStream_t s);
uint32_t   idwt_level_strip_pairs(const IdwtLevelArgs& a);   // coefficient pairs a K6 workgroup owns at this level

// ---- K7: inverse colour transform + DC shift + clamp + store as pixels (kernels_idwt.hip) --------
struct EgressArgs {
    const int32_t* planes;   // [tile][comp] planes (float bit patterns when irreversible)
    void*    pixels;         // tiles back to back, component-major planar, tight
    uint32_t w, h, stride;
    uint64_t pitch;
    uint32_t ncomp, ntiles;
    uint32_t bytes_per_sample;   // 1, 2 or 4 (int32 out)
    int32_t  dc, lo, hi;
    int      mct, irreversible;
};
hipError_t launch_egress(const EgressArgs& a, hipStream_t s);

// ---- KT1 / KT2: Tier-2 on the device (kernels_t2.hip) -------------------------------------------------------------------------
struct T2HeaderArgs {
    const T2Packet* packets; uint32_t npackets;      // a tile's packets in progression order (geometry.h)
    const uint32_t* lengths;                         // the call's table: [tile][row]
    uint32_t bpt, ntiles;                            // rows per tile
    uint32_t* ubits; uint32_t u_words;               // raw header bits, per tile, ZEROED by the caller
    uint8_t*  hdr;   uint32_t h_bytes;               // stuffed headers, per tile
    uint32_t* rel;                                   // [tile][row]: the block's offset in its packet's body
    uint32_t* pk_hdr; uint64_t* pk_body;             // [tile][packet]: header / body bytes
    unsigned int* status;                            // bit 2: a block length the writer does not take (>= 2^kT2MaxLenBits)
};
hipError_t launch_t2_header(const T2HeaderArgs& a, uint32_t max_blocks_per_packet, hipStream_t s);
struct T2FrameArgs {
    uint32_t npackets, ntiles;
    const uint32_t* pk_hdr; const uint64_t* pk_body;
    const uint32_t* tile_index;                      // [tile]: Isot
    uint32_t extra;                                  // bytes per packet beside header and body (SOP 6, EPH 2)
    uint32_t plt;
    uint64_t dst_offset;                             // where the call's first tile-part goes in the output
    uint8_t* lit; uint32_t lit_stride;               // the frames, lit_stride bytes apart
    uint32_t* lit_len;                               // [tile]
    uint64_t* pk_dst;                                // [tile][packet]
    uint32_t* part_len; unsigned long long* tile_dst;   // [tile]: the tile-part's length and place (what a host or an exchange reads)
    unsigned long long* total;                       // [0] bytes assembled by this call, [1] the end of the output
    unsigned int* status;                            // bit 3: a tile-part beyond 4 GB / PLT beyond 256 marker segments
};
hipError_t launch_t2_frame(const T2FrameArgs& a, hipStream_t s);
struct T2GatherArgs {
    const T2Packet* packets; uint32_t npackets;
    const uint32_t* packet_of_block;                 // [row]
    const uint32_t* lengths; const uint64_t* offsets; const uint8_t* arena;
    uint32_t bpt, ntiles;
    const uint8_t* hdr; uint32_t h_bytes;
    const uint32_t* rel; const uint32_t* pk_hdr;
    const uint64_t* pk_dst;                          // [tile][packet]: where the packet starts in `out`
    const uint8_t* lit; uint32_t lit_stride; const uint32_t* lit_len;    // the tile-parts' frames (SOT .. SOD) as KT1b left them ...
    const unsigned long long* tile_dst;              // ... and where the tile-parts start in `out`
    uint8_t* out;
    uint32_t sop, eph;                               // 6 / 2 when the markers are written, else 0
};
hipError_t launch_t2_gather(const T2GatherArgs& a, hipStream_t s);

} // namespace grk_amd
