/* oracle/ht_refine_oracle.c -- TEST INFRASTRUCTURE ONLY (see j2k_oracle.h).
 *
 * The two refinement passes of an HTJ2K code-block (ITU-T T.814 §7.4 / §7.5):
 *   SigProp (SPP)  -- forward bit stream, LSB first, the byte after 0xFF carries 7 bits, zeros when exhausted
 *                     (the reference's frwd_struct with X = 0: t1/t1_ht/coding/ojph_block_decoder.cpp:875-913, :916-945)
 *   MagRef  (MRP)  -- read BACKWARDS from the end of the refinement segment, LSB first; a byte whose 7 low bits are ones
 *                     carries 7 bits when the byte read before it was > 0x8F (initially: as if it was)
 *                     (rev_read_mrp / rev_init_mrp, ojph_block_decoder.cpp:466-550)
 * as ojph_decode_codeblock applies them to the output of the cleanup pass (ojph_block_decoder.cpp:1627-2100).
 * Grok itself never reaches that code (T1HT.cpp:158-166 passes lengths2 = 0) and its encoder never emits the passes, so
 * this file also holds a small ENCODER for them: it makes the test vectors, which are pinned by feeding them to the
 * reference's own decoder with lengths2 != 0 (oracle/ref_harness: ref_ht_decode_block_passes).
 *
 * Restated from the definitions, with plain per-sample arrays instead of the reference's nibble-packed words:
 *   scan     stripes of 4 rows; inside a stripe groups of 4 columns; inside a group column by column, top to bottom
 *   SPP      a sample is a MEMBER when the cleanup pass left it insignificant and one of its 8 neighbours is significant,
 *            where "significant" means: by the cleanup pass (any neighbour: same stripe, the row above the stripe, the
 *            row below it), or by this pass if the neighbour was scanned earlier (same stripe: the column before, or the
 *            row above in the same column; the stripe above: its bottom row).  Each member takes one bit (1 = becomes
 *            significant); after the significance bits of a 4-column group come the sign bits of its new samples.
 *            A new sample decodes to sign << 31 | 3 << (p - 2): magnitude bit p - 1 plus half a bin.
 *   MRP      every sample the cleanup pass made significant takes one bit, in scan order; sym: word ^= (1 - sym) << (p - 1)
 *            (the cleanup pass had put its bin centre there), word |= 1 << (p - 2).
 *   p = 30 - missing_msbs is the cleanup pass's bit-plane; the refinement passes need p >= 2.
 */
#include "j2k_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- bit writers (test-vector side) ------------------------------------------------------------- */
typedef struct { uint8_t* d; uint32_t n, cap; uint32_t acc, nb, last_ff; } spp_wr;
static void spp_put(spp_wr* s, uint32_t bit)
{
    s->acc |= bit << s->nb;
    s->nb++;
    if (s->nb == (s->last_ff ? 7u : 8u)) {
        if (s->n < s->cap) s->d[s->n] = (uint8_t)s->acc;
        s->n++;
        s->last_ff = s->acc == 0xFF;
        s->acc = 0; s->nb = 0;
    }
}
static void spp_flush(spp_wr* s)
{
    if (s->nb) { if (s->n < s->cap) s->d[s->n] = (uint8_t)s->acc; s->n++; s->last_ff = s->acc == 0xFF; s->acc = 0; s->nb = 0; }
    if (s->last_ff) { if (s->n < s->cap) s->d[s->n] = 0; s->n++; s->last_ff = 0; }    /* a segment does not end in 0xFF */
}
/* MRP bytes in the order a decoder reads them (the first one ends up LAST in the segment) */
typedef struct { uint8_t* d; uint32_t n, cap; uint32_t acc, nb, prev_gt8f; } mrp_wr;
static void mrp_put(mrp_wr* s, uint32_t bit)
{
    s->acc |= bit << s->nb;
    s->nb++;
    const int seven = s->prev_gt8f && s->nb == 7 && s->acc == 0x7F;
    if (s->nb == 8 || seven) {
        if (s->n < s->cap) s->d[s->n] = (uint8_t)s->acc;
        s->n++;
        s->prev_gt8f = s->acc > 0x8F;
        s->acc = 0; s->nb = 0;
    }
}
static void mrp_flush(mrp_wr* s)
{
    if (s->nb) { if (s->n < s->cap) s->d[s->n] = (uint8_t)s->acc; s->n++; s->acc = 0; s->nb = 0; }
}

/* ---- bit readers (decoder side) ------------------------------------------------------------------ */
typedef struct { const uint8_t* d; int pos, end; int unstuff; uint32_t cur; int nb; } spp_rd;
static uint32_t spp_get(spp_rd* s)
{
    if (s->nb == 0) {
        const uint32_t b = s->pos < s->end ? s->d[s->pos] : 0u;       /* exhausted: zeros */
        s->pos++;
        s->nb = 8 - s->unstuff;
        s->cur = b;                                                    /* (the skipped MSB is simply never reached) */
        s->unstuff = b == 0xFF;
    }
    const uint32_t v = s->cur & 1u;
    s->cur >>= 1; s->nb--;
    return v;
}
typedef struct { const uint8_t* d; int pos, left; int unstuff; uint32_t cur; int nb; } mrp_rd;
static uint32_t mrp_get(mrp_rd* s)
{
    if (s->nb == 0) {
        const uint32_t b = s->left > 0 ? s->d[s->pos] : 0u;
        s->pos--; s->left--;
        s->nb = 8 - ((s->unstuff && (b & 0x7F) == 0x7F) ? 1 : 0);
        s->cur = b;
        s->unstuff = b > 0x8F;
    }
    const uint32_t v = s->cur & 1u;
    s->cur >>= 1; s->nb--;
    return v;
}

/* significance maps with a one-sample border; sc = after the cleanup pass, sn = made significant by the SPP */
#define AT(m, x, y) (m)[((y) + 1) * mw + (x) + 1]

/* One walk over the block in SPP order, shared by the encoder (bit_src != NULL: take the decisions from the samples and
 * write them) and the decoder (read them).  refine[] (encoder): the bit of plane p - 1 of every sample; sign[]: its sign. */
static void spp_walk(uint32_t w, uint32_t h, const uint8_t* sc, uint8_t* sn, uint32_t mw,
                     spp_rd* rd, spp_wr* wr, const uint8_t* refine, const uint8_t* sign,
                     uint32_t* words, uint32_t stride, uint32_t p)
{
    for (uint32_t y0 = 0; y0 < h; y0 += 4) {
        const uint32_t y1 = y0 + 4 < h ? y0 + 4 : h;
        for (uint32_t x0 = 0; x0 < w; x0 += 4) {
            const uint32_t x1 = x0 + 4 < w ? x0 + 4 : w;
            for (uint32_t x = x0; x < x1; ++x)
                for (uint32_t y = y0; y < y1; ++y) {
                    if (AT(sc, (int)x, (int)y)) continue;
                    int mem = 0;
                    for (int dy = -1; dy <= 1 && !mem; ++dy)
                        for (int dx = -1; dx <= 1 && !mem; ++dx) {
                            if (!dx && !dy) continue;
                            const int nx = (int)x + dx, ny = (int)y + dy;
                            if (AT(sc, nx, ny)) { mem = 1; break; }
                            /* made significant by this pass AND scanned before (x, y) */
                            const int earlier = ny < (int)y0 ? 1                                   /* the stripe above */
                                              : ny >= (int)y1 ? 0                                  /* the stripe below: not yet */
                                              : (dx < 0 || (dx == 0 && dy < 0));                   /* same stripe */
                            if (earlier && AT(sn, nx, ny)) mem = 1;
                        }
                    if (!mem) continue;
                    uint32_t bit;
                    if (wr) { bit = refine[y * w + x]; spp_put(wr, bit); } else bit = spp_get(rd);
                    if (bit) AT(sn, (int)x, (int)y) = 1;
                }
            for (uint32_t x = x0; x < x1; ++x)
                for (uint32_t y = y0; y < y1; ++y) {
                    if (!AT(sn, (int)x, (int)y) || AT(sc, (int)x, (int)y)) continue;
                    if (wr) spp_put(wr, sign[y * w + x]);
                    else words[y * stride + x] = (spp_get(rd) << 31) | (3u << (p - 2));
                }
        }
    }
}

/* mag: the samples' magnitudes INCLUDING bit-plane p - 1 as their LSB (the cleanup pass codes mag >> 1); sign: 0 / 1.
 * Writes the refinement segment (SPP bytes, then the MRP bytes in file order) and returns its length; *spp_len gets the
 * SPP part.  npasses: 2 = SPP only, 3 = SPP + MRP. */
int32_t orc_ht_refine_encode(const uint32_t* mag, const uint8_t* sign, uint32_t w, uint32_t h, uint32_t npasses,
                             uint8_t* out, uint32_t cap, uint32_t* spp_len)
{
    const uint32_t mw = w + 2;
    uint8_t* sc = (uint8_t*)calloc((size_t)mw * (h + 2), 1);
    uint8_t* sn = (uint8_t*)calloc((size_t)mw * (h + 2), 1);
    uint8_t* refine = (uint8_t*)malloc((size_t)w * h);
    uint8_t* tmp = (uint8_t*)malloc((size_t)w * h + 16);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            AT(sc, (int)x, (int)y) = (mag[y * w + x] >> 1) != 0;
            refine[y * w + x] = (uint8_t)(mag[y * w + x] & 1u);
        }
    spp_wr sw = {out, 0, cap, 0, 0, 0};
    spp_walk(w, h, sc, sn, mw, NULL, &sw, refine, sign, NULL, 0, 2);
    spp_flush(&sw);
    if (spp_len) *spp_len = sw.n;
    uint32_t total = sw.n;
    if (npasses >= 3) {
        mrp_wr mr = {tmp, 0, w * h + 16, 0, 0, 1};
        for (uint32_t y0 = 0; y0 < h; y0 += 4)
            for (uint32_t x = 0; x < w; ++x)
                for (uint32_t y = y0; y < y0 + 4 && y < h; ++y)
                    if (AT(sc, (int)x, (int)y)) mrp_put(&mr, refine[y * w + x]);
        mrp_flush(&mr);
        for (uint32_t i = 0; i < mr.n; ++i)
            if (total + i < cap) out[total + i] = tmp[mr.n - 1 - i];
        total += mr.n;
    }
    free(sc); free(sn); free(refine); free(tmp);
    return total <= cap ? (int32_t)total : -1;
}

/* words: the block as the cleanup pass decoded it (orc_ht_decode_block: sign << 31 | (2 mu + 1) << (p - 1)), refined in
 * place by the passes in seg[0, len2).  npasses: 2 = SPP, 3 = SPP + MRP.  Returns 0, or -1 when p < 2. */
int32_t orc_ht_refine_decode(uint32_t* words, uint32_t w, uint32_t h, uint32_t stride, uint32_t missing_msbs,
                             const uint8_t* seg, uint32_t len2, uint32_t npasses)
{
    if (missing_msbs > 28) return -1;
    const uint32_t p = 30 - missing_msbs;
    if (npasses < 2 || len2 == 0) return 0;
    const uint32_t mw = w + 2;
    uint8_t* sc = (uint8_t*)calloc((size_t)mw * (h + 2), 1);
    uint8_t* sn = (uint8_t*)calloc((size_t)mw * (h + 2), 1);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) AT(sc, (int)x, (int)y) = words[y * stride + x] != 0;
    if (npasses >= 3) {
        mrp_rd mr = {seg, (int)len2 - 1, (int)len2, 1, 0, 0};
        for (uint32_t y0 = 0; y0 < h; y0 += 4)
            for (uint32_t x = 0; x < w; ++x)
                for (uint32_t y = y0; y < y0 + 4 && y < h; ++y)
                    if (AT(sc, (int)x, (int)y)) {
                        const uint32_t sym = mrp_get(&mr);
                        words[y * stride + x] ^= (1u - sym) << (p - 1);
                        words[y * stride + x] |= 1u << (p - 2);
                    }
    }
    spp_rd sr = {seg, 0, (int)len2, 0, 0, 0};
    spp_walk(w, h, sc, sn, mw, &sr, NULL, NULL, NULL, words, stride, p);
    free(sc); free(sn);
    return 0;
}
