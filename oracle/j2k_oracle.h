/* oracle/j2k_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's per-tile hot path, used exclusively
 * as the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product (grok_amd/) never links, imports or calls anything declared here.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_*.py) against
 *   - the known-answer vectors of SURVEY.md Appendix C.1/C.2 (committed under tests/golden/), and
 *   - the real reference built from its own sources by oracle/Makefile (oracle/_ref), call by call.
 */
#ifndef J2K_ORACLE_H
#define J2K_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1/a2: widen + DC level shift  (tile/TileProcessor.cpp:1166-1216, :922-944) */
/*     bytes_per_sample < 0: signed samples (int8 / int16), dc_shift is then 0 */
void orc_ingest(const void* src, int bytes_per_sample, int32_t* dst,
                uint32_t w, uint32_t h, uint32_t stride, int32_t dc_shift);

/* ---- a3/a4: forward colour transforms, in place (point_transform/mct.cpp:48-105, :469-554) */
void orc_rct_fwd(int32_t* c0, int32_t* c1, int32_t* c2, size_t n);
void orc_ict_fwd(int32_t* c0, int32_t* c1, int32_t* c2, size_t n); /* writes float bit patterns */

/* ---- a5-a7: forward DWT over `levels` decompositions, in place, Mallat layout
 *            (transform/WaveletFwd.cpp:434-609; 5/3 :625-906; 9/7 :134-214, :911-994) */
void orc_dwt53_fwd(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels);
void orc_dwt97_fwd(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels);
void orc_dwt53_fwd_1d(int32_t* x, uint32_t n);           /* [s | d] out, even start */
void orc_dwt97_fwd_1d(float* x, uint32_t n);
/* inverse transforms (transform/WaveletReverse.cpp:852-936, :1360-1439) for round-trip checks */
void orc_dwt53_inv(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels);
void orc_rct_inv(int32_t* c0, int32_t* c1, int32_t* c2, size_t n);

/* ---- a8: quantiser exponents.  Reversible HT: codestream/HTParams.cpp:248-268 (RCT bit NOT
 *          added: defect D4).  Fills 3*levels+1 exponents in QCD order; Kmax(band)=expn (a8). */
void orc_ht_rev_exponents(uint32_t prec, uint32_t levels, uint8_t* expn /* [3*levels+1] */);
/* irreversible HT: HTParams.cpp:269-312 -> (expn<<11)|mant per band, and the float step of
 * Quantizer.cpp:41-45 */
void orc_ht_irrev_stepsizes(uint32_t prec, uint32_t levels, uint16_t* spqcd, float* delta);

/* ---- a9: code-block enumeration (comp-local; order res -> band -> precinct(1) -> raster) */
typedef struct {
    uint32_t x, y;        /* origin inside the component's Mallat plane (samples) */
    uint32_t w, h;        /* block size */
    uint8_t  res, band;   /* resolution 0..levels ; orientation 0=LL 1=HL 2=LH 3=HH */
    uint8_t  kmax;        /* band->numbps for HT = QCD exponent */
    uint8_t  pad;
    uint32_t bx, by;      /* block coordinates inside the band's code-block grid */
} orc_block;
/* returns number of blocks (writes at most cap of them). cblk_exp = log2 nominal size (6). */
uint32_t orc_enumerate_blocks(uint32_t w, uint32_t h, uint32_t levels, uint32_t cblk_exp,
                              const uint8_t* expn, orc_block* out, uint32_t cap);

/* ---- a10: sign-magnitude conversion (t1/t1_ht/T1HT.cpp:58-84), a11: HT cleanup encoder
 *           (t1/t1_ht/coding/ojph_block_encoder.cpp:463-938). Returns coded length. */
void     orc_ht_signmag_rev(const int32_t* src, uint32_t stride, uint32_t w, uint32_t h,
                            uint32_t kmax, uint32_t* dst /* w*h */);
int32_t  orc_ht_encode_sm(const uint32_t* sm, uint32_t kmax, uint32_t w, uint32_t h,
                          uint8_t* out, uint32_t cap);
int32_t  orc_ht_encode_block_rev(const int32_t* src, uint32_t stride, uint32_t w, uint32_t h,
                                 uint32_t kmax, uint8_t* out, uint32_t cap);
/* irreversible ("intended" dead-zone quantiser of SURVEY.md A.5, NOT the reference's D1 bug) */
void     orc_ht_signmag_irrev(const float* src, uint32_t stride, uint32_t w, uint32_t h,
                              uint32_t kmax, float inv_delta, uint32_t* dst);
/* N3: distortion decrease of an HT block's single pass in T1::getwmsedec's units (t1/t1_part1/T1.cpp:394-414) */
double   orc_ht_block_distortion(const uint32_t* sm, uint32_t n, uint32_t kmax, uint32_t orient, uint32_t level, int reversible,
                                 int mct, uint32_t comp, double stepsize);

/* ---- whole tile, reversible: pixels (C planes, tight) -> per-block coded bytes.
 * blocks_out[i] describes block i (enumeration order comp -> res -> band -> raster), lens[i] its
 * coded length, bytes are appended to `coded` (capacity cap). Returns total number of blocks or <0. */
int32_t orc_encode_tile_rev(const void* pixels, int bytes_per_sample, uint32_t ncomp,
                            uint32_t w, uint32_t h, uint32_t prec, uint32_t levels, int mct,
                            orc_block* blocks_out, uint32_t* lens, uint32_t max_blocks,
                            uint8_t* coded, uint64_t cap, uint64_t* total_bytes);

/* ======================= decode half (oracle/j2k_decode_oracle.c) ================================ */
/* ---- a14: HT cleanup-pass block decoder (t1/t1_ht/coding/ojph_block_decoder.cpp:989-1625, cleanup
 *           only).  `missing_msbs` = band numbps - block numbps as Grok passes it
 *           (T1DecompressScheduler.cpp:59; Kmax-1 for streams of the encoder above).  out: w x h
 *           words `sign<<31 | (2*mu+1) << (29-missing_msbs)`, row stride `stride`.  0 ok, -1 bad stream. */
int32_t orc_ht_decode_block(const uint8_t* coded, uint32_t len, uint32_t missing_msbs,
                            uint32_t w, uint32_t h, uint32_t* out, uint32_t stride);
/* ---- tiles / images on odd origins (SURVEY.md Appendix A.3 "odd-start variant"; WaveletFwd.cpp:884-905, :812-815,
 *      band coordinates util/util.cpp:49-58): the same functions with the tile-component's origin (x0, y0) on the
 *      canonical grid -- the parity of ceil(origin / 2^l) decides at every level which samples are predicted / updated */
void orc_dwt53_fwd_1d_par(int32_t* x, uint32_t n, uint32_t par);
void orc_dwt97_fwd_1d_par(float* x, uint32_t n, uint32_t par);
void orc_dwt53_fwd_at(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0);
void orc_dwt97_fwd_at(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0);
void orc_dwt53_inv_at(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0);
void orc_dwt97_inv_at(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0);
uint32_t orc_enumerate_blocks_at(uint32_t w, uint32_t h, uint32_t levels, uint32_t cblk_exp, uint32_t x0, uint32_t y0,
                                 const uint8_t* expn, orc_block* out, uint32_t cap);
int32_t orc_encode_tile_rev_at(const void* pixels, int bps, uint32_t ncomp, uint32_t w, uint32_t h,
                               uint32_t prec, uint32_t levels, int mct, uint32_t x0, uint32_t y0, orc_block* blocks_out,
                               uint32_t* lens, uint32_t max_blocks, uint8_t* coded, uint64_t cap, uint64_t* total_bytes);
/* ---- precincts (grk_compress -c; COD precinct sizes): prc[r] = PPx | PPy << 4 for resolution r, NULL / 0 = 15, 15 ---- */
uint32_t orc_enumerate_blocks_prc(uint32_t w, uint32_t h, uint32_t levels, uint32_t cblk_exp, uint32_t x0, uint32_t y0,
                                  const uint8_t* prc, const uint8_t* expn, orc_block* out, uint32_t cap);
int32_t orc_encode_tile_rev_prc(const void* pixels, int bps, uint32_t ncomp, uint32_t w, uint32_t h,
                                uint32_t prec, uint32_t levels, int mct, uint32_t x0, uint32_t y0, const uint8_t* prc,
                                orc_block* blocks_out, uint32_t* lens, uint32_t max_blocks, uint8_t* coded, uint64_t cap,
                                uint64_t* total_bytes);
/* ---- N3: the HT refinement passes (SigProp, MagRef) on top of the cleanup pass's output -- oracle/ht_refine_oracle.c
 *           (t1/t1_ht/coding/ojph_block_decoder.cpp:1627-2100; bit readers :466-550, :875-945) and an encoder of them
 *           that makes the test vectors (Grok's encoder never emits the passes) */
int32_t orc_ht_refine_encode(const uint32_t* mag, const uint8_t* sign, uint32_t w, uint32_t h, uint32_t npasses,
                             uint8_t* out, uint32_t cap, uint32_t* spp_len);
int32_t orc_ht_refine_decode(uint32_t* words, uint32_t w, uint32_t h, uint32_t stride, uint32_t missing_msbs,
                             const uint8_t* seg, uint32_t len2, uint32_t npasses);
/* ---- a15: dequantisation (filters/PostDecompressFilters.h:94-106 ShiftHTFilter, :128-140 ScaleHTFilter) */
void orc_ht_dequant_rev(const uint32_t* sm, uint32_t n, uint32_t k_msbs, int32_t* out);
void orc_ht_dequant_irrev(const uint32_t* sm, uint32_t n, float scale, float* out);
/* ---- a16: inverse 9/7 over `levels` decompositions, in place, Mallat layout in, image out
 *           (transform/WaveletReverse.cpp:938-1074, :1360-1439); 5/3: orc_dwt53_inv above */
void orc_dwt97_inv(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels);
/* ---- a17: inverse colour transform + DC level shift + clamp to [lo,hi], in place
 *           (point_transform/mct.cpp:369-465 rev, :186-294 irrev (float bit patterns in), :297-364,
 *           :109-177 single component) */
void orc_rct_inv_store(int32_t* c0, int32_t* c1, int32_t* c2, size_t n, int32_t shift, int32_t lo, int32_t hi);
void orc_ict_inv_store(int32_t* c0, int32_t* c1, int32_t* c2, size_t n, int32_t shift, int32_t lo, int32_t hi);
void orc_dc_store_rev(int32_t* c, size_t n, int32_t shift, int32_t lo, int32_t hi);
void orc_dc_store_irrev(int32_t* c, size_t n, int32_t shift, int32_t lo, int32_t hi);

/* ---- a13: Part-1 (EBCOT) Tier-1 block decoder (oracle/ebcot_oracle.c; t1/t1_part1/T1.cpp:1262-1337,
 *           code-block style 0).  numbps = magnitude bit-planes coded for the block (band numbps minus
 *           the zero bit-planes Tier-2 signals), orient 0 LL / 1 HL / 2 LH / 3 HH.  out: w*h int32 in
 *           the decoder's representation (one extra fractional bit).  Then ShiftFilter / ScaleFilter. */
int32_t orc_t1_decode_block(const uint8_t* coded, uint32_t len, uint32_t numpasses, uint32_t numbps,
                            uint32_t orient, uint32_t w, uint32_t h, int32_t* out);
/* general form: codeword segments + code-block styles (LAZY 1, RESET 2, TERMALL 4, VSC 8, PTERM 16, SEGSYM 32);
 * returns the number of bad segmentation symbols (the reference only warns) or -1 */
int32_t orc_t1_decode_block_sty(const uint8_t* coded, uint32_t nsegs, const uint32_t* seg_len, const uint32_t* seg_passes,
                                uint32_t numbps, uint32_t orient, uint32_t cblksty, uint32_t w, uint32_t h, int32_t* out);
void orc_t1_dequant_rev(const int32_t* v, uint32_t n, int32_t* out);
void orc_t1_dequant_irrev(const int32_t* v, uint32_t n, float stepsize, float* out);

#ifdef __cplusplus
}
#endif
#endif
