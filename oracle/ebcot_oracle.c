/* oracle/ebcot_oracle.c -- TEST INFRASTRUCTURE ONLY (see j2k_oracle.h).
 *
 * CPU restatement of row a13 of SURVEY.md §8: the Part-1 (EBCOT) Tier-1 block DEcoder
 *   T1::decompress_cblk                 t1/t1_part1/T1.cpp:1262-1337   (segments; code-block styles LAZY = raw
 *                                        sig-prop/mag-ref passes after four planes, RESET, TERMALL, VSC,
 *                                        PTERM [a check only], SEGSYM)
 *   raw (bypass) decoder                 mqc_dec.cpp:156-160, mqc_dec_inl.h:55-76; raw passes T1.cpp:1009-1021, :1155-1165
 *   cleanup / sig-prop / mag-ref passes  T1.cpp:854-1007, :1024-1152, :1160-1255
 *   MQ decoder                           t1/t1_part1/mqc_dec.cpp:107-177, mqc_dec_inl.h:26-150
 *   dequantisation                       filters/PostDecompressFilters.h:26-35 (ShiftFilter: v/2),
 *                                        :60-71 (ScaleFilter: (float)v * stepsize/2)
 * written from ITU-T T.800 Annex C (MQ coder, Table C.2) and Annex D (coding passes, Tables D.1-D.4),
 * with one flag byte per sample instead of the reference's packed 4-row flag words and contexts
 * computed from the neighbourhood rules instead of lookup tables.
 * Pinned against the real T1 (oracle/_ref, ref_t1_decode_block) in tests/test_oracle_ebcot.py.
 */
#include "j2k_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- MQ decoder (T.800 C.3, software conventions of the reference: C.3.5 INITDEC) ---------------- */
typedef struct { uint16_t qe; uint8_t nmps, nlps, sw; } mq_row;
static const mq_row MQ[47] = {          /* Table C.2 */
    {0x5601, 1, 1, 1},  {0x3401, 2, 6, 0},  {0x1801, 3, 9, 0},  {0x0AC1, 4, 12, 0}, {0x0521, 5, 29, 0},
    {0x0221, 38, 33, 0}, {0x5601, 7, 6, 1},  {0x5401, 8, 14, 0}, {0x4801, 9, 14, 0}, {0x3801, 10, 14, 0},
    {0x3001, 11, 17, 0}, {0x2401, 12, 18, 0}, {0x1C01, 13, 20, 0}, {0x1601, 29, 21, 0}, {0x5601, 15, 14, 1},
    {0x5401, 16, 14, 0}, {0x5101, 17, 15, 0}, {0x4801, 18, 16, 0}, {0x3801, 19, 17, 0}, {0x3401, 20, 18, 0},
    {0x3001, 21, 19, 0}, {0x2801, 22, 19, 0}, {0x2401, 23, 20, 0}, {0x2201, 24, 21, 0}, {0x1C01, 25, 22, 0},
    {0x1801, 26, 23, 0}, {0x1601, 27, 24, 0}, {0x1401, 28, 25, 0}, {0x1201, 29, 26, 0}, {0x1101, 30, 27, 0},
    {0x0AC1, 31, 28, 0}, {0x09C1, 32, 29, 0}, {0x08A1, 33, 30, 0}, {0x0521, 34, 31, 0}, {0x0441, 35, 32, 0},
    {0x02A1, 36, 33, 0}, {0x0221, 37, 34, 0}, {0x0141, 38, 35, 0}, {0x0111, 39, 36, 0}, {0x0085, 40, 37, 0},
    {0x0049, 41, 38, 0}, {0x0025, 42, 39, 0}, {0x0015, 43, 40, 0}, {0x0009, 44, 41, 0}, {0x0005, 45, 42, 0},
    {0x0001, 45, 43, 0}, {0x5601, 46, 46, 0}};

enum { CTX_ZC = 0, CTX_SC = 9, CTX_MAG = 14, CTX_AGG = 17, CTX_UNI = 18, NCTX = 19 };

typedef struct {
    const uint8_t* d; uint32_t len, pos;       /* pos = index of the byte "bp" points at */
    uint32_t a, c, ct;
    uint8_t idx[NCTX], mps[NCTX];
} mq_t;

/* byte at index i of the block's data followed by the artificial 0xFF 0xFF terminator (mqc_dec.cpp:113-118) */
static uint32_t mq_byte(const mq_t* m, uint32_t i) { return i < m->len ? m->d[i] : 0xFFu; }
static void mq_bytein(mq_t* m)
{
    const uint32_t cur = mq_byte(m, m->pos), nxt = mq_byte(m, m->pos + 1);
    if (cur == 0xFF) {
        if (nxt > 0x8F) { m->c += 0xFF00; m->ct = 8; }                /* marker: feed 1s, do not advance */
        else { m->pos++; m->c += nxt << 9; m->ct = 7; }
    } else { m->pos++; m->c += nxt << 8; m->ct = 8; }
}
static void mq_resetstates(mq_t* m)
{   /* mqc_resetstates (mqc_dec.cpp:168-175) */
    memset(m->idx, 0, sizeof m->idx); memset(m->mps, 0, sizeof m->mps);
    m->idx[CTX_UNI] = 46; m->idx[CTX_AGG] = 3; m->idx[CTX_ZC] = 4;
}
static void mq_init_dec(mq_t* m, const uint8_t* d, uint32_t len)
{   /* mqc_init_dec (mqc_dec.cpp:140-154): a segment start; the context states are NOT touched */
    m->d = d; m->len = len; m->pos = 0;
    m->c = (len == 0 ? 0xFFu : d[0]) << 16;
    mq_bytein(m);
    m->c <<= 7; m->ct -= 7; m->a = 0x8000;
}
static void mq_init(mq_t* m, const uint8_t* d, uint32_t len)
{
    mq_resetstates(m);
    mq_init_dec(m, d, len);
}
/* raw (bypass) segments: mqc_raw_init_dec (mqc_dec.cpp:156-160), mqc_raw_decode (mqc_dec_inl.h:55-76) */
static void raw_init_dec(mq_t* m, const uint8_t* d, uint32_t len)
{
    m->d = d; m->len = len; m->pos = 0; m->c = 0; m->ct = 0;
}
static uint32_t raw_decode(mq_t* m)
{
    if (m->ct == 0) {
        const uint32_t b = mq_byte(m, m->pos);
        if (m->c == 0xFF) {
            if (b > 0x8F) { m->c = 0xFF; m->ct = 8; }                /* the terminating marker: ones for ever */
            else { m->c = b; m->pos++; m->ct = 7; }
        } else { m->c = b; m->pos++; m->ct = 8; }
    }
    m->ct--;
    return (m->c >> m->ct) & 1u;
}
static uint32_t mq_decode(mq_t* m, int cx)
{
    const mq_row* r = &MQ[m->idx[cx]];
    uint32_t d;
    m->a -= r->qe;
    if ((m->c >> 16) < r->qe) {                                      /* LPS exchange (C.3.2) */
        if (m->a < r->qe) { d = m->mps[cx]; m->idx[cx] = r->nmps; }
        else { d = m->mps[cx] ^ 1u; if (r->sw) m->mps[cx] ^= 1u; m->idx[cx] = r->nlps; }
        m->a = r->qe;
    } else {
        m->c -= (uint32_t)r->qe << 16;
        if (m->a & 0x8000) return m->mps[cx];
        if (m->a < r->qe) { d = m->mps[cx] ^ 1u; if (r->sw) m->mps[cx] ^= 1u; m->idx[cx] = r->nlps; }
        else { d = m->mps[cx]; m->idx[cx] = r->nmps; }
    }
    do {                                                             /* RENORMD */
        if (m->ct == 0) mq_bytein(m);
        m->a <<= 1; m->c <<= 1; m->ct--;
    } while (m->a < 0x8000);
    return d;
}

/* ---- coding passes --------------------------------------------------------------------------------- */
enum { F_SIG = 1, F_NEG = 2, F_PI = 4, F_MU = 8 };          /* significant, negative, visited this plane, refined */

typedef struct { uint8_t* f; int32_t* v; uint32_t w, h, fs; int orient; int vsc, raw; mq_t mq; } t1_t;
#define FL(t, x, y) ((t)->f[((y) + 1) * (t)->fs + (x) + 1])
/* Neighbour (x + dx, y + dy) as sample (x, y) sees it.  Vertically causal contexts (VSC): the last row of a
 * stripe never learns about the stripe below -- update_flags skips the northward update for ci == 0
 * (T1.cpp:198-221) -- so those three neighbours read as insignificant. */
static uint8_t NB(const t1_t* t, uint32_t x, uint32_t y, int dx, int dy)
{
    if (t->vsc && dy == 1 && (y & 3u) == 3u) return 0;
    return FL(t, x + dx, y + dy);
}

static int zc_ctx(const t1_t* t, uint32_t x, uint32_t y)
{   /* Table D.1 */
    int hh = (NB(t, x, y, -1, 0) & F_SIG) + (NB(t, x, y, 1, 0) & F_SIG);
    int vv = (NB(t, x, y, 0, -1) & F_SIG) + (NB(t, x, y, 0, 1) & F_SIG);
    int dd = (NB(t, x, y, -1, -1) & F_SIG) + (NB(t, x, y, 1, -1) & F_SIG) +
             (NB(t, x, y, -1, 1) & F_SIG) + (NB(t, x, y, 1, 1) & F_SIG);
    if (t->orient == 1) { int s = hh; hh = vv; vv = s; }            /* HL: horizontal and vertical swap roles */
    if (t->orient == 3) {                                            /* HH */
        int hv = hh + vv;
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
static int any_sig_neighbour(const t1_t* t, uint32_t x, uint32_t y)
{
    return (NB(t, x, y, -1, 0) | NB(t, x, y, 1, 0) | NB(t, x, y, 0, -1) | NB(t, x, y, 0, 1) | NB(t, x, y, -1, -1) |
            NB(t, x, y, 1, -1) | NB(t, x, y, -1, 1) | NB(t, x, y, 1, 1)) & F_SIG;
}
static int contrib(uint8_t a, uint8_t b)
{   /* Table D.2: sign contribution of two opposite neighbours */
    int s = 0;
    if (a & F_SIG) s += (a & F_NEG) ? -1 : 1;
    if (b & F_SIG) s += (b & F_NEG) ? -1 : 1;
    return s > 1 ? 1 : (s < -1 ? -1 : s);
}
static uint32_t decode_sign(t1_t* t, uint32_t x, uint32_t y)
{   /* Table D.3 */
    const int hc = contrib(NB(t, x, y, -1, 0), NB(t, x, y, 1, 0)), vc = contrib(NB(t, x, y, 0, -1), NB(t, x, y, 0, 1));
    int cx, xr;
    if (hc == 1)      { cx = vc == 1 ? 13 : (vc == 0 ? 12 : 11); xr = 0; }
    else if (hc == 0) { cx = vc == 0 ? 9 : 10; xr = vc == -1; }
    else              { cx = vc == 1 ? 11 : (vc == 0 ? 12 : 13); xr = 1; }
    return mq_decode(&t->mq, cx) ^ (uint32_t)xr;
}
static void become_significant(t1_t* t, uint32_t x, uint32_t y, int32_t oneplushalf)
{
    const uint32_t neg = t->raw ? raw_decode(&t->mq) : decode_sign(t, x, y);       /* raw: the sign bit itself (T1.cpp:1016-1018) */
    t->v[y * t->w + x] = neg ? -oneplushalf : oneplushalf;
    FL(t, x, y) |= (uint8_t)(F_SIG | (neg ? F_NEG : 0));
}

static void sigpass(t1_t* t, int bp)
{
    const int32_t one = 1 << bp, oph = one | (one >> 1);
    for (uint32_t k = 0; k < t->h; k += 4)
        for (uint32_t x = 0; x < t->w; ++x)
            for (uint32_t y = k; y < k + 4 && y < t->h; ++y) {
                if ((FL(t, x, y) & (F_SIG | F_PI)) || !any_sig_neighbour(t, x, y)) continue;
                if (t->raw ? raw_decode(&t->mq) : mq_decode(&t->mq, CTX_ZC + zc_ctx(t, x, y))) become_significant(t, x, y, oph);
                FL(t, x, y) |= F_PI;
            }
}
static void refpass(t1_t* t, int bp)
{
    const int32_t poshalf = (1 << bp) >> 1;
    for (uint32_t k = 0; k < t->h; k += 4)
        for (uint32_t x = 0; x < t->w; ++x)
            for (uint32_t y = k; y < k + 4 && y < t->h; ++y) {
                if ((FL(t, x, y) & (F_SIG | F_PI)) != F_SIG) continue;
                const int cx = (FL(t, x, y) & F_MU) ? 16 : (any_sig_neighbour(t, x, y) ? 15 : 14);   /* Table D.4 */
                const uint32_t b = t->raw ? raw_decode(&t->mq) : mq_decode(&t->mq, cx);
                int32_t* p = &t->v[y * t->w + x];
                *p += (b ^ (uint32_t)(*p < 0)) ? poshalf : -poshalf;
                FL(t, x, y) |= F_MU;
            }
}
static void clnpass(t1_t* t, int bp)
{
    const int32_t one = 1 << bp, oph = one | (one >> 1);
    for (uint32_t k = 0; k < t->h; k += 4)
        for (uint32_t x = 0; x < t->w; ++x) {
            uint32_t y = k;
            if (k + 4 <= t->h) {                                     /* run-length mode: full stripe column, all quiet (D.3.4) */
                int quiet = 1;
                for (uint32_t j = 0; j < 4; ++j)
                    if ((FL(t, x, k + j) & (F_SIG | F_PI)) || any_sig_neighbour(t, x, k + j)) { quiet = 0; break; }
                if (quiet) {
                    if (!mq_decode(&t->mq, CTX_AGG)) continue;
                    uint32_t r = mq_decode(&t->mq, CTX_UNI);
                    r = (r << 1) | mq_decode(&t->mq, CTX_UNI);
                    become_significant(t, x, k + r, oph);             /* the first significant sample: sign only */
                    y = k + r + 1;
                }
            }
            for (; y < k + 4 && y < t->h; ++y) {
                if (FL(t, x, y) & (F_SIG | F_PI)) continue;
                if (mq_decode(&t->mq, CTX_ZC + zc_ctx(t, x, y))) become_significant(t, x, y, oph);
            }
        }
    for (uint32_t y = 0; y < t->h; ++y)
        for (uint32_t x = 0; x < t->w; ++x) FL(t, x, y) &= (uint8_t)~F_PI;
}

int32_t orc_t1_decode_block(const uint8_t* coded, uint32_t len, uint32_t numpasses, uint32_t numbps,
                            uint32_t orient, uint32_t w, uint32_t h, int32_t* out)
{
    if (numbps >= 31 - 6) return -1;                                 /* k_max_bit_planes (t1_common.h:70) */
    t1_t t;
    t.w = w; t.h = h; t.fs = w + 2; t.orient = (int)orient; t.vsc = 0; t.raw = 0;
    t.f = (uint8_t*)calloc((size_t)(w + 2) * (h + 2), 1);
    t.v = out;
    memset(out, 0, (size_t)w * h * sizeof(int32_t));
    mq_init(&t.mq, coded, len);
    int bp = (int)numbps, type = 2;                                  /* first pass: cleanup of the top plane */
    for (uint32_t p = 0; p < numpasses && bp >= 1; ++p) {
        if (type == 0) sigpass(&t, bp);
        else if (type == 1) refpass(&t, bp);
        else clnpass(&t, bp);
        if (++type == 3) { type = 0; --bp; }
    }
    free(t.f);
    return 0;
}

/* The general form (T1.cpp:1262-1337): `nsegs` codeword segments of seg_len[i] bytes holding seg_passes[i] passes
 * each, laid end to end in `coded`; cblksty = the COD code-block style bits (LAZY 1, RESET 2, TERMALL 4, VSC 8,
 * PTERM 16, SEGSYM 32).  Returns the number of segmentation symbols that were not 0xA (the reference only warns),
 * or -1 if the block is rejected. */
int32_t orc_t1_decode_block_sty(const uint8_t* coded, uint32_t nsegs, const uint32_t* seg_len, const uint32_t* seg_passes,
                                uint32_t numbps, uint32_t orient, uint32_t cblksty, uint32_t w, uint32_t h, int32_t* out)
{
    if (numbps >= 31 - 6) return -1;
    t1_t t;
    t.w = w; t.h = h; t.fs = w + 2; t.orient = (int)orient; t.vsc = (cblksty & 8u) != 0; t.raw = 0;
    t.f = (uint8_t*)calloc((size_t)(w + 2) * (h + 2), 1);
    t.v = out;
    memset(out, 0, (size_t)w * h * sizeof(int32_t));
    int bp = (int)numbps, type = 2, bad_segsym = 0;
    uint32_t off = 0;
    mq_resetstates(&t.mq);
    for (uint32_t sg = 0; sg < nsegs; ++sg) {
        t.raw = (bp <= (int)numbps - 4) && type < 2 && (cblksty & 1u);            /* decided at the segment start */
        if (t.raw) raw_init_dec(&t.mq, coded + off, seg_len[sg]);
        else mq_init_dec(&t.mq, coded + off, seg_len[sg]);
        off += seg_len[sg];
        for (uint32_t p = 0; p < seg_passes[sg] && bp >= 1; ++p) {
            if (type == 0) sigpass(&t, bp);
            else if (type == 1) refpass(&t, bp);
            else {
                const int was_raw = t.raw;
                t.raw = 0;                                                          /* the cleanup pass is never raw */
                clnpass(&t, bp);
                if (cblksty & 32u) {                                                /* dec_clnpass_check_segsym (:977-993) */
                    uint32_t v = 0;
                    for (int i = 0; i < 4; ++i) v = (v << 1) | mq_decode(&t.mq, CTX_UNI);
                    if (v != 0xA) ++bad_segsym;
                }
                t.raw = was_raw;
            }
            if ((cblksty & 2u) && !t.raw) mq_resetstates(&t.mq);
            if (++type == 3) { type = 0; --bp; }
        }
    }
    free(t.f);
    return bad_segsym;
}

/* ---- dequantisation ------------------------------------------------------------------------------------ */
void orc_t1_dequant_rev(const int32_t* v, uint32_t n, int32_t* out)
{   /* ShiftFilter: C integer division, truncation toward zero */
    for (uint32_t i = 0; i < n; ++i) out[i] = v[i] / 2;
}
void orc_t1_dequant_irrev(const int32_t* v, uint32_t n, float stepsize, float* out)
{   /* ScaleFilter: scale = stepsize / 2 */
    const float scale = stepsize / 2;
    for (uint32_t i = 0; i < n; ++i) out[i] = (float)v[i] * scale;
}
