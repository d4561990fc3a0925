/* include/grok_amd.h -- the C-ABI of the MI355X tile processor (libgrok_amd.so).
 *
 * This is the thin C layer SURVEY.md §8(b) asks for underneath the Grok plugin entry points:
 * plain pointers and sizes only (no torch / C++ types), one opaque context per GPU.
 * The plugin shim (libgrokj2k_plugin.so, include/grk_plugin_abi.h) wraps these calls into the
 * grk_plugin_* protocol; bench.py and the parity tests call them through ctypes.
 *
 * Every entry point names the reference interface it replaces (paths relative to the Grok
 * 8.0.2 tree, src/lib/jp2/...):
 *
 *   grk_amd_encode_tiles        TileProcessor::do_compress()  tile/TileProcessor.cpp:665-699
 *                               = dc_level_shift_encode :922, mct_encode :945, dwt_encode :977,
 *                                 t1_encode :992 (T1HT::compress, t1/t1_ht/T1HT.cpp:102-128)
 *   grk_amd_stage_ingest_mct    TileProcessor.cpp:1166-1216 + :922-944 + mct::compress_rev/irrev
 *                               (point_transform/mct.cpp:48-105, :469-554)
 *   grk_amd_stage_dwt_fwd       WaveletFwdImpl::compress (transform/WaveletFwd.cpp:612-620)
 *   grk_amd_stage_ht_encode     T1CompressScheduler::scheduleCompress (t1/T1CompressScheduler.cpp:33-90)
 *                               + ojph_encode_codeblock (t1/t1_ht/coding/ojph_block_encoder.cpp:463)
 *   grk_amd_tile_layout         TileComponent::init geometry (tile/TileComponent.cpp:69-170),
 *                               Quantizer::setBandStepSizeAndBps (codestream/Quantizer.cpp:26-66),
 *                               param_qcd::set_rev_quant/set_irrev_quant (codestream/HTParams.cpp:248-312)
 *   grk_amd_write_codestream    CodeStreamCompress main header + tile parts + T2 packets, restricted
 *                               to 1 layer / LRCP / 1 precinct per resolution
 *                               (codestream/CodeStreamCompress.cpp:722-753, t2/T2Compress.cpp:123-333)
 *
 * Error convention (mirrors the plugin boundary, grok.h:1781): 0 = success, negative = failure /
 * "not handled" so that the host can fall back to its CPU path.  No exceptions cross this ABI.
 */
#ifndef GROK_AMD_H
#define GROK_AMD_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRK_AMD_OK                0
#define GRK_AMD_ERR_NO_DEVICE    -1   /* no HIP device / HIP runtime failure           */
#define GRK_AMD_ERR_UNSUPPORTED  -2   /* parameter combination outside the hot path     */
#define GRK_AMD_ERR_INVALID      -3   /* bad argument                                   */
#define GRK_AMD_ERR_NOMEM        -4
#define GRK_AMD_ERR_OVERFLOW     -5   /* coded data did not fit the arena               */
#define GRK_AMD_ERR_RANGE        -6   /* decode: a value left the 16-bit planes (see grk_amd_set_decode_planes16) */

#define GRK_AMD_MAX_LEVELS 10

typedef struct grk_amd_ctx grk_amd_ctx;      /* one per process per GPU */

/* What the hot path needs out of grk_cparameters / grk_image (grok.h:451-573, :866-929). */
typedef struct grk_amd_tile_params {
    uint32_t tile_w, tile_h;     /* tile-component size (dx=dy=1)                                 */
    uint16_t num_comps;          /* 1..4 ; MCT needs >=3                                        */
    uint8_t  prec;               /* bits per sample, 1..16                                      */
    uint8_t  sgnd;               /* signed samples (DC shift 0)                                 */
    uint8_t  irreversible;       /* 0: RCT + 5/3 (lossless)   1: ICT + 9/7 + dead-zone quantiser*/
    uint8_t  mct;                /* tcp_mct: component transform on components 0..2             */
    uint8_t  num_levels;         /* numresolution - 1                                           */
    uint8_t  cblk_w_exp;         /* log2 code-block width  (6)                                  */
    uint8_t  cblk_h_exp;         /* log2 code-block height (6)                                  */
    uint8_t  reserved[3];        /* [0]: decode only -- 1 = the blocks are Part-1 (EBCOT/MQ) instead of HT;
                                    then missing_msbs of a table row carries numbps | numpasses << 8
                                    (block bit-planes coded, coding passes in total)
                                    [1]: Part-1 decode -- the COD code-block style bits as the reference
                                    names them (grok.h GRK_CBLKSTY_*): LAZY 0x01, RESET 0x02, TERMALL 0x04,
                                    VSC 0x08, PTERM 0x10, SEGSYM 0x20                                 */
    uint32_t tile_x0, tile_y0;   /* the tile's origin on the canonical grid (grk_image x0/y0, tile grid: the tile's x0/y0 as
                                    TileProcessor::init computes it).  0, 0 for an image at the origin with one tile; any
                                    other value changes the sub-band sizes, the code-block partition and, where a
                                    resolution starts on an odd coordinate, the lifting variant (WaveletFwd.cpp:884-905) */
    uint8_t  precinct_exp[12];   /* [r] = PPx | PPy << 4 of resolution r (0 = coarsest), as the COD marker carries them
                                    (grk_compress -c; TileComponentCodingParams::precinctWidthExp / HeightExp); 0 = not set
                                    = 15 | 15 << 4: one precinct per resolution (so PPx = PPy = 0, legal for r = 0 only, cannot
                                    be asked for).  Precincts cut the code-block partition
                                    (code-block exponent <= precinct exponent of the band) and make one packet each, in
                                    any of the five progression orders */
} grk_amd_tile_params;

/* One code-block of the tile, in the reference's enumeration order
 * comp -> resolution -> band -> precinct(1) -> raster (T1CompressScheduler.cpp:44-85). */
typedef struct grk_amd_block {
    uint32_t x0, y0, x1, y1;     /* band coordinates (what grk_plugin_code_block.x0.. carries) */
    uint32_t px, py;             /* origin inside the component's Mallat plane                 */
    uint16_t comp;
    uint8_t  res;                /* 0 = coarsest                                               */
    uint8_t  band;               /* orientation 0 LL, 1 HL, 2 LH, 3 HH                         */
    uint8_t  kmax;               /* band->numbps (= QCD exponent for HT, numgbits 1)           */
    uint8_t  reserved[3];
    float    stepsize;           /* band->stepsize (1.0 reversible)                            */
    uint32_t precinct;           /* index of its precinct in the resolution's precinct grid (raster)  */
} grk_amd_block;

/* Result row per coded block. `offset` is relative to the start of the coded arena. */
typedef struct grk_amd_coded_block {
    uint64_t offset;
    uint32_t length;             /* MagSgn | MEL | VLC bytes of the single HT cleanup pass      */
    uint32_t missing_msbs;       /* band numbps - block numbps: what the decoder needs besides the
                                    bytes (Kmax - 1 for blocks of this encoder, T1HT.cpp:123)    */
} grk_amd_coded_block;

/* ---- lifecycle ---------------------------------------------------------------------------- */
/* plugin_init(grk_plugin_init_info{deviceId,verbose}) -> grk_amd_create (grok.h:1749-1757) */
int  grk_amd_create(int device_id, int verbose, grk_amd_ctx** out);
void grk_amd_destroy(grk_amd_ctx* ctx);
const char* grk_amd_version(void);
/* sha1 over the sources (grok_amd/csrc, include) the library was built from, as __graft_entry__.source_stamp() computes it for a
 * tree: the test suite compares the two, so that a stale prebuilt library cannot pass for the tree it travels with */
const char* grk_amd_source_stamp(void);
const char* grk_amd_last_error(grk_amd_ctx* ctx);
/* use an externally owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL = own */
int  grk_amd_set_stream(grk_amd_ctx* ctx, void* hip_stream);
/* Pinned (page-locked) host memory for the buffers handed to the host-pointer entry points below (on_device = 0): such a
 * buffer crosses the link in one DMA at the link's rate (~50 GB/s each way on MI355X hosts).  Any other host memory works
 * too -- it is moved through context-owned pinned chunks by a few copy threads, at those threads' memcpy rate.  What the
 * reference would do in grk_plugin_compress's image reader: the plugin's own loader reads files into such memory. */
void* grk_amd_host_alloc(grk_amd_ctx* ctx, uint64_t bytes);
void  grk_amd_host_free(grk_amd_ctx* ctx, void* p);

/* ---- geometry (host only, no GPU needed) ---------------------------------------------------- */
/* Number of code-blocks in one tile (all components). */
int64_t grk_amd_tile_num_blocks(const grk_amd_tile_params* p);
/* Fill `blocks` (capacity cap) in enumeration order; returns count or negative error.
 * Also returns per-band QCD words: reversible -> expn<<3 (u8), irreversible -> (expn<<11)|mant. */
int64_t grk_amd_tile_layout(const grk_amd_tile_params* p, grk_amd_block* blocks, uint64_t cap,
                            uint16_t* qcd_words /* [3*levels+1] or NULL */);
/* precincts of every resolution (counts[r], r = 0 coarsest .. num_levels): the packets a tile-component has per layer */
int grk_amd_tile_precincts(const grk_amd_tile_params* p, uint32_t* counts);
/* int32 elements between consecutive rows / planes of the device working planes */
uint32_t grk_amd_plane_stride(const grk_amd_tile_params* p);
uint64_t grk_amd_plane_elems(const grk_amd_tile_params* p);

/* ---- whole hot path -------------------------------------------------------------------------
 * Encode `num_tiles` tiles of one geometry (grk_amd_same_tile_geometry; the batch is coded with *p) in one batch.  `pixels` holds the tiles back to back,
 * each tile component-major planar, row-major, tightly packed, ceil(prec/8) bytes per sample,
 * host endian -- the layout grk_compress_tile() takes (TileProcessor.cpp:1177-1213).
 * pixels_on_device != 0: `pixels` is a device pointer (HBM-resident input, what bench.py times).  Lifetime of device pixels: they
 * are read in the order of the context's stream (grk_amd_set_stream) -- whatever the caller queues on that stream behind the call (the
 * next frame's pixels into the same buffer, a stream-ordered free) comes after the read, on every path, pipelined or not
 * (grk_amd_set_pixel_hold below relaxes this for callers that rotate their input buffers).
 *
 * On return the coded bytes of all blocks live in the context's device arena; `table` (host,
 * num_tiles * blocks_per_tile rows, tile-major) describes them.  Pass table == NULL to leave the
 * table on the device as well (fully asynchronous; fetch later with grk_amd_fetch_table). */
int grk_amd_encode_tiles(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles,
                         const void* pixels, int pixels_on_device,
                         grk_amd_coded_block* table, uint64_t* total_bytes);
/* Pixel lifetime, relaxed (default off).  Pipelined encodes of small frames run a frame's whole chain on a stream of the context's own
 * (GRK_AMD_FRAME_STREAMS), so keeping the rule above costs a wait of the context's stream per call -- consecutive 512 x 512 frames:
 * 0.057 instead of 0.038 ms per call.  on != 0: the caller PROMISES not to touch a call's device pixels before a stream of its own has
 * passed grk_amd_stream_wait_pixels (or grk_amd_synchronize returned); the wait is then left out.  A ring of input buffers with
 * one grk_amd_stream_wait_pixels before a buffer is refilled is the intended use. */
int grk_amd_set_pixel_hold(grk_amd_ctx* ctx, int on);
/* Pipelined encodes run on three streams (the caller's + two of the context's) whose kernels have to be DISPATCHED side by side; whether
 * they can depends on the hardware queues the runtime gave them, i.e. on the streams the process made before (profiles/r06_hw_queues.txt).
 * Before the first pipelined encode on a given main stream the context probes its streams pairwise (~2 ms, the main stream is
 * synchronised once) and replaces a side stream that has to take turns (GRK_AMD_STREAM_PROBE=0: never).  Returns the number of side
 * streams replaced so far, -1 with the probe off. */
int grk_amd_stream_probe_result(grk_amd_ctx* ctx);
/* the same probe at once (e.g. before a host vets its own streams against the context's) */
int grk_amd_probe_streams(grk_amd_ctx* ctx);
/* ... and for a host's own streams: 1 when kernels of HIP streams a and b are dispatched side by side (both directions), 0 when one waits
 * for the other's grid.  A stream that WAITS for the encoder's results (an exchange's) should pass this against grk_amd_internal_stream(ctx,
 * 0 / 1 / 2) -- the main stream and the two side streams as they are now. */
int grk_amd_streams_side_by_side(grk_amd_ctx* ctx, void* a, void* b);
void* grk_amd_internal_stream(grk_amd_ctx* ctx, int which);
/* Makes `hip_stream` wait until the LATEST grk_amd_encode_tiles call has read its device pixels (not for its results). */
int grk_amd_stream_wait_pixels(grk_amd_ctx* ctx, void* hip_stream);
int grk_amd_fetch_table(grk_amd_ctx* ctx, grk_amd_coded_block* table, uint64_t* total_bytes);
/* The rate-control hook (the host reads grk_plugin_pass::distortionDecrease when it makes quality layers,
 * plugin/plugin_bridge.cpp:214-226, tile/TileProcessor.cpp:320-458): for every code-block of the LATEST grk_amd_encode_tiles call,
 * in table order, the distortion decrease of coding its single HT cleanup pass, in the units T1::getwmsedec gives the passes of the
 * reference's own Tier-1 (t1/t1_part1/T1.cpp:394-414): (w_mct * w_band * stepsize)^2 * sum of q^2 over the block, q the quantised
 * integer magnitude that was coded.  (The reference's own HT encoder leaves the field unset, T1HT.cpp:102-127: on its CPU path an
 * HTJ2K job with several layers has no slopes to work with.)  Not for pipelined sequences: reads the planes of the latest call. */
int grk_amd_block_distortion(grk_amd_ctx* ctx, double* distortion, uint64_t capacity);
/* copy coded bytes [0,total) of the arena to host memory */
int grk_amd_fetch_coded(grk_amd_ctx* ctx, uint8_t* dst, uint64_t nbytes);
/* The same without waiting: `dst` must be pinned (grk_amd_host_alloc); the copy is queued on the context's stream and complete after
 * grk_amd_synchronize -- a host that runs Tier-2 over the table (grk_amd_plan_tile_part) while the bytes cross the link. */
int grk_amd_fetch_coded_async(grk_amd_ctx* ctx, uint8_t* dst, uint64_t nbytes);
/* device pointers for zero-copy consumers (RCCL gather of tile parts, tests) */
void* grk_amd_coded_device_ptr(grk_amd_ctx* ctx);
void* grk_amd_plane_device_ptr(grk_amd_ctx* ctx, int which /*0: ingest planes, 1: Mallat planes*/);
/* (after grk_amd_encode_tiles of 8-bit reversible content the Mallat planes hold int16 coefficients -- same strides
 *  and pitches in elements -- unless the environment says GRK_AMD_PLANES16=0; the stage entry points are int32) */
/* the block table of the last encode where it was produced, for exchanges that never touch the host:
 * which 0: uint64 offsets[nblocks], 1: uint32 lengths[nblocks], 2: uint64 bytes used in the arena,
 * 3: uint64[24] diagnostics -- per block class, how many blocks outgrew the capped LDS buffers of the HT encoder and
 *    were coded by its fallback launch (0 on natural images; environment GRK_AMD_LDS_CAP=0 gives every block
 *    worst-case buffers instead) */
void* grk_amd_table_device_ptr(grk_amd_ctx* ctx, int which);
int  grk_amd_synchronize(grk_amd_ctx* ctx);
/* The sub-band coefficients of the latest grk_amd_encode_tiles (one tile), component `comp`, as the reference holds them
 * after its DWT: Mallat layout, int32 (float32 bit patterns when irreversible), dst_stride elements per row.  What the
 * plugin hands the host in its self-check mode (GRK_PLUGIN_STATE_DEBUG, grok.h:1719-1739: the host then runs its own
 * Tier-1 over the plugin's coefficients and compares every code-block). */
int  grk_amd_fetch_coefficients(grk_amd_ctx* ctx, uint32_t comp, int32_t* dst, uint32_t dst_stride);

/* ---- stage entry points (parity tests and per-kernel benchmarks call these) ------------------
 * All pointers are DEVICE pointers; planes are int32 (or float32 bit patterns for 9/7) with
 * row stride grk_amd_plane_stride() and plane pitch grk_amd_plane_elems(). */
int grk_amd_stage_ingest_mct(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles,
                             const void* d_pixels, void* d_planes);
/* forward DWT of num_planes planes: d_in (ingest planes) -> d_out (Mallat layout); d_in is
 * left untouched (intermediate LL planes live in context-owned ping-pong buffers). */
int grk_amd_stage_dwt_fwd(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_planes,
                          void* d_in, void* d_out);
/* HT cleanup-encode every block of num_tiles tiles from Mallat planes into the context arena. */
int grk_amd_stage_ht_encode(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles,
                            const void* d_mallat);

/* ---- decode: the inverse hot path --------------------------------
 * TileProcessor::decompress T1 + post-T1 for HT blocks: T1HT::decompress (t1/t1_ht/T1HT.cpp:129-179,
 * ojph_decode_codeblock t1/t1_ht/coding/ojph_block_decoder.cpp:989), dequantisation
 * (filters/PostDecompressFilters.h:94-140), inverse DWT (transform/WaveletReverse.cpp:852-936,
 * :1360-1439), inverse MCT + DC shift + clamp (point_transform/mct.cpp:109-465).
 * `table` (host) has one row per code-block in the encoder's enumeration order with the block's
 * byte range inside `coded` and its missing_msbs -- exactly what the host's Tier-2 parser knows
 * after decompress_synch_plugin_with_host (plugin/plugin_bridge.cpp:63-76); length 0 = no data.
 * `pixels` receives the tiles back to back, component-major planar, tight, ceil(prec/8) bytes per
 * sample.  With pixels_on_device != 0 the call is asynchronous: query grk_amd_decode_status().
 * Returns GRK_AMD_ERR_INVALID for a block the reference decoder would reject. */
int grk_amd_decode_tiles(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles,
                         const grk_amd_coded_block* table, const void* coded, uint64_t coded_bytes,
                         int coded_on_device, void* pixels, int pixels_on_device);
int grk_amd_decode_status(grk_amd_ctx* ctx);
/* A SEQUENCE of frames: with frames_in_flight = n in 2..8, consecutive grk_amd_decode_tiles calls whose coded bytes and pixels
 * are device buffers are decoded on n internal buffer / stream sets in turn, each queued behind what the caller has on the
 * context's stream at the time of the call, so that frame f + 1's block decoding (serial chains that leave most of the GPU
 * idle) runs beside frame f's dequantisation and inverse transform.  The reference decodes a frame's tiles as pooled tasks
 * (codestream/CodeStreamDecompress.cpp:450-519); this is the same idea across frames.  A call returns when its kernels are
 * queued; a frame's pixels are complete after grk_amd_synchronize (all frames) -- grk_amd_decode_status reports the worst
 * status of all of them.  0 or 1: off (every call on the context's own stream, as before).  Calls with host buffers and
 * grk_amd_decode_region always run on the context itself.
 * Streams and hardware queues: the HIP runtime puts a process's streams on (by default) 4 hardware queues per priority level
 * and kernels that share a queue run in turn; the internal sets therefore use two streams each, and for Part-1 frames (two
 * long kernels per frame) streams of both priority levels in turn (8192x8192x3 12-bit: 15.0 ms one frame at a time, 9.1 with
 * two in flight, 7.2 with six).  A host that decodes such sequences gains a little more from GPU_MAX_HW_QUEUES=8 in its
 * environment before its first HIP call (6.2-6.8 with six); the library leaves the variable alone: it is process-wide, and the
 * encode pipeline beside an RCCL exchange measured 25 % slower on anything but the default. */
int grk_amd_set_decode_pipelining(grk_amd_ctx* ctx, int frames_in_flight);
/* Buffer lifetime in a decode sequence.  With n frames in flight a call's device buffers -- the coded bytes it reads, the pixels
 * it writes -- are in use until THAT frame has been decoded, which is up to n calls later and on streams of the library's own:
 * work the caller queues on the context's stream (or any other) is NOT ordered behind it.  A caller therefore rotates n coded /
 * pixel buffers (as many as frames in flight) and, before it overwrites or reads the buffers it handed over n calls ago -- the
 * ones the NEXT grk_amd_decode_tiles call will reuse the same internal set for --, makes its stream wait:
 *     grk_amd_decode_stream_wait_slot(ctx, stream);   upload frame f's coded bytes on `stream`;   grk_amd_decode_tiles(...);
 * (no host synchronisation: a stream-side wait for the event behind that set's last frame; without a sequence it is
 * grk_amd_stream_wait_results).  grk_amd_synchronize(ctx) remains the blunt form: all frames of all sets. */
int grk_amd_decode_stream_wait_slot(grk_amd_ctx* ctx, void* hip_stream);
/* 8-bit reversible HT tiles are decoded with int16 planes between the block decoder and the inverse DWT (default on; half
 * the bytes of the two HBM-bound halves of the decode), and the inverse 5/3 runs on packed pairs of them, which takes every
 * coefficient and every synthesised LL sample within +-2047 (an 8-bit image's are: |HH| <= 1020 at the top resolution, the
 * LL of every level is image-sized); a stream whose values are not is never decoded to other pixels: a synchronous call (host
 * pixels) decodes it again with int32 planes by itself, an asynchronous one reports GRK_AMD_ERR_RANGE from
 * grk_amd_decode_status() and the caller repeats the call after grk_amd_set_decode_planes16(ctx, 0). */
int grk_amd_set_decode_planes16(grk_amd_ctx* ctx, int on);
/* Bytes per coefficient of the LL / Mallat planes that grk_amd_encode_tiles (decode = 0, device pixels) or grk_amd_decode_tiles
 * (decode = 1) keeps for tiles of these parameters: 2 for 8-bit reversible content (int16 planes: every coefficient provably
 * fits, results identical to the int32 path), else 4.  *packed_levels (may be NULL) = the forward DWT levels that run on packed
 * int16 pairs.  What a roofline figure has to count the plane traffic with (bench.py). */
int grk_amd_plane_sample_bytes(grk_amd_ctx* ctx, const grk_amd_tile_params* p, int decode, uint32_t* packed_levels);
/* Region (windowed) decode of ONE tile -- what grk_decompress_set_window() + grk_decompress() do on the host
 * (grok.h; partial synthesis: transform/WaveletReverse.cpp:1466-2213, tile/SparseBuffer.h): the pixels of the window
 * [x0, x1) x [y0, y1) of the tile, component-major planar, tight, (x1 - x0) * (y1 - y0) samples per component --
 * bit-identical to the same crop of grk_amd_decode_tiles' output.  Only the code-blocks a sample of the window depends
 * on are entropy-decoded and only the strips / row segments of each DWT level that lead to it are synthesised, so
 * the cost follows the window, not the image.  table / coded describe the whole tile, as for grk_amd_decode_tiles.
 * Needs at least one DWT level and <= 16-bit pixels (GRK_AMD_ERR_UNSUPPORTED otherwise). */
int grk_amd_decode_region(grk_amd_ctx* ctx, const grk_amd_tile_params* p,
                          const grk_amd_coded_block* table, const void* coded, uint64_t coded_bytes, int coded_on_device,
                          uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, void* pixels, int pixels_on_device);
/* Irreversible streams of another encoder: the SPqcd words (expn << 11 | mant, one per sub-band in QCD
 * order) its QCD marker carries, from which the decode-side step sizes are derived
 * (codestream/Quantizer.cpp:41-63).  count = 0 returns to the exponents this library's encoder writes. */
int grk_amd_set_decode_qcd(grk_amd_ctx* ctx, const uint16_t* words, uint32_t count);
/* The same for a host that already holds the band step sizes: steps[comp * (3 * levels + 1) + band] = the
 * TileBand::stepsize Grok's decoder computed (codestream/Quantizer.cpp:26-66, HT fix-up :54-63 included; note that
 * decompress_synch_plugin_with_host stores HALF of it in the plugin tree, plugin_bridge.cpp:40).  Takes precedence
 * over the QCD words; count = 0 drops it. */
int grk_amd_set_decode_steps(grk_amd_ctx* ctx, const float* steps, uint32_t count);
/* Part-1 blocks with more than one codeword segment (LAZY, TERMALL -- T1::decompress_cblk's segment loop,
 * t1/t1_part1/T1.cpp:1280-1318; Grok's own plugin bridge refuses those, plugin_bridge.cpp:50-61, so this is
 * reachable through this C ABI only).  Segments of block i are first_segment[i] .. first_segment[i+1]-1, their
 * bytes lie end to end at the block's offset.  Applies to the following decode calls; nblocks = 0 returns to
 * one segment per block (length and pass count of the table row).
 * HT blocks (reserved[0] = 0) with the refinement passes of T.814 -- SigProp, MagRef; ojph_decode_codeblock with
 * lengths2 != 0, t1/t1_ht/coding/ojph_block_decoder.cpp:1627-2100, which Grok itself never reaches (T1HT.cpp:158-166) --
 * use the same list: segment 0 = the cleanup pass {Lcup, 1}, segment 1 = {bytes of SigProp + MagRef, 1 or 2 passes}; the
 * table row's length is their sum. */
typedef struct grk_amd_segment { uint32_t length; uint32_t numpasses; } grk_amd_segment;
int grk_amd_set_decode_segments(grk_amd_ctx* ctx, const uint32_t* first_segment, const grk_amd_segment* segments,
                                uint32_t nblocks);
/* HT cleanup decode + dequantisation of every block into Mallat planes (device pointers) */
int grk_amd_stage_ht_decode(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles,
                            const grk_amd_coded_block* table, const void* d_coded, uint64_t coded_bytes,
                            void* d_mallat);

/* ---- decode-side stages ------------------------------------------
 * inverse DWT of num_planes Mallat planes -> image-domain planes
 *   (transform/WaveletReverse.cpp:852-936 decompress_tile_53, :1360-1439 decompress_tile_97) */
int grk_amd_stage_dwt_inv(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_planes,
                          const void* d_mallat, void* d_out);
/* inverse RCT/ICT + DC level shift + clamp, planes -> tightly packed pixels (ceil(prec/8) bytes)
 *   (point_transform/mct.cpp:109-177, :186-294, :297-364, :369-465) */
int grk_amd_stage_egress(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles,
                         const void* d_planes, void* d_pixels);

/* average duration (ms) of the named kernel family over the launches since the last reset,
 * measured with HIP events on the context's stream when timing is enabled.
 * which: 0 ingest+mct, 1 dwt (all levels), 2 ht encode kernel (launches on the context's stream), 3 whole
 *        encode_tiles call, 4 ht encode kernel, top resolution (side stream, beside DWT levels >= 1),
 *        5 ht decode, 6 inverse dwt (all levels), 7 egress, 8 ht encode kernel, large-LDS classes (second side stream).
 * One encode runs the ht encode kernel up to three times (2, 4, 8): its time per step is their sum.
 * Pipelined encodes of small frames (a frame's whole chain on one of the side streams, GRK_AMD_FRAME_STREAMS): families 1, 2 and 3
 * are then measured on that stream -- the stream that carries the call -- and 4 / 8 stay empty. */
int    grk_amd_enable_timing(grk_amd_ctx* ctx, int on);
/* K3 of the top resolution beside DWT levels >= 1 on side streams (default on; environment GRK_AMD_OVERLAP=0/1
 * sets the default).  Off = every kernel alone on the GPU, one after the other: what per-kernel durations
 * (roofline figures, rocprofv3 summaries) should be measured with, since co-running kernels stretch each other. */
int    grk_amd_set_overlap(grk_amd_ctx* ctx, int on);
/* Pipelining of consecutive grk_amd_encode_tiles calls (a sequence of frames; default off): the per-encode buffers
 * (Mallat planes, coded arena, block table, allocator state) exist twice and the call returns without joining its side
 * streams, so the next call's DWT runs while the blocks of this one are still being coded.  The results of a call stay
 * valid until the second next call.  Every grk_amd_* function that reads them (fetch_table, fetch_coded, synchronize,
 * decode, the stage entry points) joins first; a caller that consumes grk_amd_coded_device_ptr / _table_device_ptr on
 * its own stream must call grk_amd_synchronize before.  Needs the overlap (above) to be on.
 * on = n >= 2 (up to 7): n + 1 buffer sets in rotation -- the results of a call stay valid until the (n + 1)-th next call, for a
 * consumer that works behind the encoder: n = 2 for the tile-part gather of a tile-sharded job one frame behind
 * (grok_amd/dist.py: the receive sizes have to pass through the host, and with two sets that round trip would sit between
 * consecutive frames), n = k + 1 for k gathers in flight at once (each towards another writer, over another xGMI link). */
int    grk_amd_set_pipelining(grk_amd_ctx* ctx, int on);
/* The number of OTHER buffer sets consecutive encodes of tiles with at least one DWT level rotate through right now (0: every
 * encode writes the same arena -- pipelining off, or the overlap switched off). */
int    grk_amd_get_pipelining(grk_amd_ctx* ctx);
/* Makes `hip_stream` (the caller's, e.g. the one its RCCL collectives run on) wait for the results of the latest encode
 * -- the context's stream and, when pipelined, its side streams -- without blocking the context's own stream: the
 * consumer of grk_amd_coded_device_ptr / grk_amd_table_device_ptr in a pipelined sequence. */
int    grk_amd_stream_wait_results(grk_amd_ctx* ctx, void* hip_stream);
double grk_amd_kernel_ms(grk_amd_ctx* ctx, int which, uint32_t* launches);

/* ---- codestream assembly (host) ----------------------------------
 * Writes a complete Part-15 codestream (SOC SIZ CAP COD QCD COM, then per tile SOT SOD + LRCP
 * packets, EOC) for an image of img_w x img_h cut into tiles of p->tile_w x p->tile_h, given the
 * per-tile block tables + coded bytes produced above (tiles in raster order).
 * Byte-identical to the reference's output for the same parameters. Returns length or <0. */
int64_t grk_amd_write_codestream(const grk_amd_tile_params* p, uint32_t img_w, uint32_t img_h,
                                 const grk_amd_coded_block* table, const uint8_t* coded,
                                 uint8_t* out, uint64_t cap);

/* A tile-part as a PLAN instead of bytes: what the writer writes itself (SOT .. SOD, packet headers, SOP / EPH) in `literal`, and the
 * tile-part as a list of segments in output order -- kind 0: `len` bytes of `literal` from `src`; kind 1: a code-block's bytes, `len`
 * bytes of the coded buffer from `src` (the table row's offset) -- each with its place `dst` in the tile-part.  Tier-2 is O(#blocks) of
 * host work (~1 ms for the 49 152 blocks of an 8K tile); moving the ~100 MB of coded bytes is then the caller's to spread over host
 * threads or to do on the device (grk_amd_assemble_device), once, into the tile-part's final place.  Returns the tile-part's length;
 * with literal / segments NULL only the sizes (*literal_len, *num_segments). */
typedef struct grk_amd_tp_segment { uint64_t dst, src; uint32_t len, kind; } grk_amd_tp_segment;
int64_t grk_amd_plan_tile_part(const grk_amd_tile_params* p, uint32_t tile_index, uint32_t flags,
                               const grk_amd_coded_block* tile_table, uint8_t* literal, uint64_t literal_cap, uint64_t* literal_len,
                               grk_amd_tp_segment* segments, uint64_t segment_cap, uint64_t* num_segments);

/* Tier-2 on the device (replaces T2Compress::compressPacket t2/T2Compress.cpp:123-333, the tag trees t1/TagTree.cpp:170-218 and the
 * header's bit stuffing t1/BitIO.cpp:46-175 for this encoder's single-layer HT packets; SOT / PLT / SOD as markers/SOTMarker.cpp:41-72,
 * markers/LengthMarkers.cpp:313-374): the finished tile-parts of the LATEST grk_amd_encode_tiles call (num_tiles tiles of *p; no
 * table needs to be fetched for it), tile i numbered tile_index[i], in call order back to back in a device buffer of the context's,
 * starting at dst_offset (0, or the end of what earlier calls assembled: batches of several geometries append).  flags: GRK_AMD_CS_PLT /
 * SOP / EPH / PROG as grk_amd_write_tile_part, whose bytes these are.  part_bytes[i] (optional) = tile-part i's length (what TLM and
 * the caller's placement need).  Returns the bytes assembled by this call, or < 0.  Three kernels (packet headers; the tile-parts' frames
 * and everybody's place; the gather) and one wait for the sizes; the ~100 MB of an 8K frame's code-blocks are moved once, on the device,
 * to where the file has them. */
int64_t grk_amd_assemble_device(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles, const uint32_t* tile_index,
                                uint32_t flags, uint64_t dst_offset, uint32_t* part_bytes);
void* grk_amd_assembled_device_ptr(grk_amd_ctx* ctx);
/* The same without the host: queued on `hip_stream`, which is first made to wait for the encode's results (grk_amd_stream_wait_results);
 * nothing is waited for, nothing comes to the host.  The tile-parts (grk_amd_assembled_device_ptr), their places and lengths and the total
 * (grk_amd_assembled_table_ptr -- 0: uint64[tiles] offsets, 1: uint32[tiles] lengths, 2: uint64[2] {bytes assembled, end}) stay where they are
 * for as many further calls as the encoder rotates buffer sets (grk_amd_set_pipelining): what an exchange needs to send FINISHED tile-parts
 * from device memory ("gather of coded tile-parts over xGMI").  Every asynchronous call of a context uses the same stream. */
int grk_amd_assemble_device_async(grk_amd_ctx* ctx, const grk_amd_tile_params* p, uint32_t num_tiles, const uint32_t* tile_index,
                                  uint32_t flags, void* hip_stream);
void* grk_amd_assembled_table_ptr(grk_amd_ctx* ctx, int which);
/* bytes [offset, offset + nbytes) of the assembled tile-parts to host memory (pinned: one DMA; pageable: through pinned chunks on
 * several copy threads); complete on return */
int grk_amd_fetch_assembled(grk_amd_ctx* ctx, uint64_t offset, uint64_t nbytes, uint8_t* dst);
/* the same queued on the context's stream (dst pinned: grk_amd_host_alloc); complete after grk_amd_synchronize */
int grk_amd_fetch_assembled_async(grk_amd_ctx* ctx, uint64_t offset, uint64_t nbytes, uint8_t* dst);

/* ---- images of any tile layout (tiles / images off the origin, ragged edge tiles) ----------------------------------
 * What SIZ says about the image (ISO 15444-1 B.2, B.3; grok.h grk_image x0..y1, grk_cparameters tx0 ty0 t_width t_height):
 * the image area [x0, x1) x [y0, y1) on the canonical grid and the tile grid anchored at (tx0, ty0) <= (x0, y0).  Tile t
 * (raster order) is its grid cell clipped to the image area (tile/TileProcessor.cpp:100-170). */
typedef struct grk_amd_image_layout {
    uint32_t x0, y0, x1, y1;
    uint32_t tx0, ty0, t_width, t_height;
} grk_amd_image_layout;
int64_t grk_amd_layout_num_tiles(const grk_amd_image_layout* im);
/* *out = *base with tile_w / tile_h / tile_x0 / tile_y0 of tile `tile_index` */
int grk_amd_layout_tile(const grk_amd_image_layout* im, const grk_amd_tile_params* base, uint32_t tile_index,
                        grk_amd_tile_params* out);
/* 1 when two tiles can share one grk_amd_encode_tiles / grk_amd_decode_tiles batch: same sub-band and code-block partition,
 * same lifting variant at every level (0: they cannot, < 0: error) */
int grk_amd_same_tile_geometry(const grk_amd_tile_params* a, const grk_amd_tile_params* b);
/* The codestream of such an image: table = the tiles' rows one tile after the other, tile t having the
 * grk_amd_tile_num_blocks() rows of ITS parameters (grk_amd_layout_tile).  grk_amd_write_codestream(_ex) is this with the
 * layout {origin of p, img_w x img_h, tiles of p->tile_w x p->tile_h}. */
int64_t grk_amd_write_codestream_layout(const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                        const grk_amd_coded_block* table, const uint8_t* coded, uint32_t flags,
                                        uint8_t* out, uint64_t cap);
int64_t grk_amd_write_main_header_layout(const grk_amd_image_layout* im, const grk_amd_tile_params* base, uint32_t flags,
                                         const uint32_t* tile_part_bytes, uint8_t* out, uint64_t cap);
/* Whole image -> codestream: `pixels` (host) is the image area, component-major planar, row-major, tight, ceil(prec/8)
 * bytes per sample.  Tiles are grouped by geometry and every group is coded as one grk_amd_encode_tiles batch (one group
 * for an image at the origin whose tile size is a multiple of 2^levels x the code-block size); returns the length. */
int64_t grk_amd_encode_image(grk_amd_ctx* ctx, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                             const void* pixels, uint32_t flags, uint8_t* out, uint64_t cap);

/* ---- sub-sampled components (4:2:2, 4:2:0 ...; SIZ XRsiz / YRsiz, grok.h grk_image_comp::dx / dy) ------------------------------------
 * Component c of a tile [x0, x1) x [y0, y1) of the reference grid covers [ceil(x0 / dx_c), ceil(x1 / dx_c)) x [ceil(y0 / dy_c),
 * ceil(y1 / dy_c)) of its own samples (tile/TileProcessor.cpp:605-612): grk_amd_layout_tile_comp gives *base with that rectangle.
 * grk_amd_encode_image_subsampled: `pixels` (host) holds the components back to back, component c as ceil(x1 / dx_c) - ceil(x0 / dx_c)
 * columns x ceil(y1 / dy_c) - ceil(y0 / dy_c) rows, tight (the planar layout of a .yuv / raw file).  Runs of consecutive components
 * with equal factors are coded together (the colour transform applies only where components 0..2 are such a run; with
 * other factors base->mct is switched off, as the reference does: CodeStreamCompress.cpp:434-447); the codestream == grk_compress's for the same image (one precinct per resolution or
 * any precinct sizes, the five progression orders, SOP / EPH / TLM / PLT as for any image).
 * grk_amd_write_codestream_subsampled: the table's rows tile by tile, within a tile component by component, each component with the
 * grk_amd_tile_num_blocks() rows of ITS rectangle (num_comps = 1). */
int grk_amd_layout_tile_comp(const grk_amd_image_layout* im, const grk_amd_tile_params* base, uint32_t dx, uint32_t dy,
                             uint32_t tile_index, grk_amd_tile_params* out);
int64_t grk_amd_write_codestream_subsampled(const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                            const uint8_t* comp_dx, const uint8_t* comp_dy,
                                            const grk_amd_coded_block* table, const uint8_t* coded, uint32_t flags,
                                            uint8_t* out, uint64_t cap);
int64_t grk_amd_encode_image_subsampled(grk_amd_ctx* ctx, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                        const uint8_t* comp_dx, const uint8_t* comp_dy, const void* pixels, uint32_t flags,
                                        uint8_t* out, uint64_t cap);

/* The same with the optional pointer marker segments of the reference's encoder (grk_compress -L / -X, grok.h
 * grk_cparameters::writePLT / writeTLM; codestream/markers/LengthMarkers.cpp): TLM in the main header (one-byte tile index +
 * four-byte tile-part length, <= 255 tiles), PLT (packet lengths) in every tile-part header. */
#define GRK_AMD_CS_TLM 1u
#define GRK_AMD_CS_PLT 2u
/* SOP marker segments in front of the packets / EPH markers after the packet headers (grk_cparameters::csty bits 0x02 / 0x04,
 * grk_compress -S / -E; t2/T2Compress.cpp:149-164, :321-327) */
#define GRK_AMD_CS_SOP 4u
#define GRK_AMD_CS_EPH 8u
/* progression order (grk_cparameters::prog_order, GRK_PROG_ORDER: 0 LRCP, 1 RLCP, 2 RPCL, 3 PCRL, 4 CPRL) << 8 */
#define GRK_AMD_CS_PROG_SHIFT 8
#define GRK_AMD_CS_PROG(order) ((uint32_t)(order) << GRK_AMD_CS_PROG_SHIFT)
int64_t grk_amd_write_codestream_ex(const grk_amd_tile_params* p, uint32_t img_w, uint32_t img_h,
                                    const grk_amd_coded_block* table, const uint8_t* coded, uint32_t flags,
                                    uint8_t* out, uint64_t cap);
/* The pieces of the above, for a tile-sharded job whose ranks write their own tile-parts (parallel writer: every rank sizes
 * its tile-parts, the sizes are exchanged, each rank then knows its offsets in the file -- CodeStreamCompress.cpp:535-603
 * writes the tile-parts in index order).  out == NULL: only the size is returned.
 *   grk_amd_write_main_header  SOC SIZ CAP COD QCD [TLM with tile_part_bytes[tile], needed when flags has GRK_AMD_CS_TLM] COM
 *   grk_amd_write_tile_part    SOT [PLT] SOD + LRCP packets of tile `tile_index`; tile_table = that tile's rows
 *   (the codestream ends with EOC, 0xFFD9) */
int64_t grk_amd_write_main_header(const grk_amd_tile_params* p, uint32_t img_w, uint32_t img_h, uint32_t flags,
                                  const uint32_t* tile_part_bytes, uint8_t* out, uint64_t cap);
int64_t grk_amd_write_tile_part(const grk_amd_tile_params* p, uint32_t tile_index, uint32_t flags,
                                const grk_amd_coded_block* tile_table, const uint8_t* coded, uint8_t* out, uint64_t cap);
/* Random access (the reader's side of markers/LengthMarkers.cpp:91-164): offset, length and tile index of every tile-part
 * of a codestream -- from its TLM marker segments when present (*used_tlm = 1: no byte of a tile-part is read), else by
 * hopping over Psot.  Returns the number of tile-parts (entries beyond `cap` are counted, not stored) or < 0. */
int64_t grk_amd_locate_tile_parts(const uint8_t* cs, uint64_t len, uint64_t* offsets, uint32_t* lengths,
                                  uint16_t* tile_index, uint64_t cap, int* used_tlm);

/* ---- one image over the GPUs of a node (node.cpp) -------------------------------------------------------
 * Replaces the reference's tile-level task pool (codestream/CodeStreamCompress.cpp:535-603: tiles are independent tasks whose
 * tile-parts are written in index order) with one grk_amd_ctx + one host thread per device: tile t is coded on device
 * t mod R, and the coded tile-parts are brought together into ONE codestream, byte-identical to the single-GPU file
 *   - flags without GRK_AMD_NODE_GATHER ("parallel writers"): every worker fetches its own coded bytes over its own PCIe
 *     link and runs Tier-2 for its own tiles; the calling thread writes main header (TLM from the sizes) + tile-parts + EOC;
 *   - flags | GRK_AMD_NODE_GATHER: the workers' coded bytes are copied device to device (hipMemcpyPeer: xGMI between GPUs)
 *     into the frame's writer device -- rotating with the frame number -- which brings them to the host in one piece and
 *     runs Tier-2 for all tiles.
 * `devices` lists the HIP devices to use (NULL / 0: all of the node); an entry may repeat (several contexts on one GPU).
 * `pixels` is the whole image, component-major planar, tight, in host memory (pinned or not).
 * In gather mode a worker whose tiles fall into more than one geometry group rotates four buffer sets, so that a group's bytes
 * travel to the writer while the worker's next groups are coded (an event per group, one wait behind the last); the rotation is
 * switched on by the first such encode and multiplies that worker's device memory for planes, arena and tables by four (several
 * GB for 8K tiles).  Parallel writers, and contexts taken through grk_amd_node_ctx before any gather, keep one set.  The rotation
 * STAYS on afterwards: a context taken through grk_amd_node_ctx after such a gather, and every later encode of that worker (gather or
 * not), works with four sets and with the "results valid until the 4th next call" rule of grk_amd_set_pipelining(ctx, 3); a caller
 * that wants the memory back calls grk_amd_set_pipelining(grk_amd_node_ctx(node, i), 0) between images. */
typedef struct grk_amd_node grk_amd_node;
#define GRK_AMD_NODE_GATHER 0x80000000u
int  grk_amd_device_count(void);
int  grk_amd_node_create(const int* devices, uint32_t num_devices, int verbose, grk_amd_node** out);
void grk_amd_node_destroy(grk_amd_node* node);
uint32_t grk_amd_node_size(const grk_amd_node* node);
grk_amd_ctx* grk_amd_node_ctx(grk_amd_node* node, uint32_t i);
const char* grk_amd_node_last_error(grk_amd_node* node);
int64_t grk_amd_node_encode_image(grk_amd_node* node, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                  const void* pixels, uint32_t flags, uint8_t* out, uint64_t cap);
/* The same with the image resident in the memory of HIP device `pixels_device` (same layout): every worker cuts its tiles out of it
 * with 2-D device-to-device copies -- from another GPU's memory over xGMI, peer access is switched on by grk_amd_node_create --,
 * so that no pixel crosses PCIe (the host-pixels entry is PCIe-bound by construction).  Same codestream, byte for byte. */
int64_t grk_amd_node_encode_image_device(grk_amd_node* node, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                         const void* pixels, int pixels_device, uint32_t flags, uint8_t* out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* GROK_AMD_H */
