/* oracle/j2k_decode_oracle.c -- TEST INFRASTRUCTURE ONLY (see j2k_oracle.h).
 *
 * CPU restatement of the decode half of the hot path:
 *   a14  HT cleanup-pass block decoder   t1/t1_ht/coding/ojph_block_decoder.cpp:989-1625 (cleanup only:
 *        Grok always calls it with lengths2 = 0, T1HT.cpp:160-168) with its bit readers :108-545, :780-880
 *   a15  dequantisation                  filters/PostDecompressFilters.h:94-106 (rev), :128-140 (irrev)
 *   a16  inverse 9/7                     transform/WaveletReverse.cpp:938-1074, :1360-1439
 *   a17  inverse RCT/ICT + DC + clamp    point_transform/mct.cpp:109-177, :186-294, :369-465
 *
 * The block decoder is written from the bit-stream definitions (forward MagSgn reader with 0xFF
 * un-stuffing, MSB-first MEL reader, backward VLC reader with 0x8F/0x7F un-stuffing) rather than
 * from the reference's 32-bit-at-a-time readers; on every stream a conforming encoder produces the
 * two read the same bits.  Pinned against the real decoder in tests/test_oracle_decode.py.
 */
#include "j2k_oracle.h"
#include "ht_vlc_tables.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---- bit readers ------------------------------------------------------------------------------ */
typedef struct {            /* MagSgn: forward, LSB first; the byte after 0xFF carries 7 bits */
    const uint8_t* d; int pos, end; uint64_t acc; int n; int unstuff;
} fwd_t;
static void fwd_fill(fwd_t* s)
{
    while (s->n <= 56) {
        uint32_t b = s->pos < s->end ? s->d[s->pos] : 0xFFu;   /* exhausted: feed 0xFF (:823-850) */
        s->pos++;
        s->acc |= (uint64_t)b << s->n;
        s->n += 8 - s->unstuff;
        s->unstuff = b == 0xFF;
    }
}
static uint32_t fwd_get(fwd_t* s, uint32_t nb)
{
    fwd_fill(s);
    uint32_t v = (uint32_t)(s->acc & ((1ull << nb) - 1));
    s->acc >>= nb; s->n -= (int)nb;
    return v;
}

typedef struct {            /* MEL: forward, MSB first, same un-stuffing; scup-1 bytes, last one |= 0x0F */
    const uint8_t* d; int pos, left; uint32_t cur; int nb; int unstuff;
    int k, zeros, one_pending;
} mel_t;
static int mel_bit(mel_t* m)
{
    if (m->nb == 0) {
        uint32_t b = 0xFF;
        if (m->left > 0) { b = m->d[m->pos]; if (m->left == 1) b |= 0x0F; m->pos++; }
        m->left--;
        m->nb = 8 - m->unstuff;
        m->cur = b & (m->unstuff ? 0x7Fu : 0xFFu);
        m->unstuff = b == 0xFF;
    }
    m->nb--;
    return (int)((m->cur >> m->nb) & 1u);
}
static const int MEL_E[13] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5};
/* next MEL event: '1' codeword = 2^E[k] zero events, '0' + E[k] bits r = r zero events then a one (:196-235) */
static int mel_event(mel_t* m)
{
    if (m->zeros == 0 && !m->one_pending) {
        int e = MEL_E[m->k];
        if (mel_bit(m)) { m->zeros = 1 << e; if (m->k < 12) m->k++; }
        else {
            int r = 0;
            for (int i = 0; i < e; ++i) r = (r << 1) | mel_bit(m);
            m->zeros = r; m->one_pending = 1; if (m->k > 0) m->k--;
        }
    }
    if (m->zeros > 0) { m->zeros--; return 0; }
    m->one_pending = 0;
    return 1;
}

typedef struct {            /* VLC: backward, LSB first; after a byte > 0x8F a byte whose 7 LSBs are ones carries 7 bits */
    const uint8_t* d; int pos, left; uint64_t acc; int n; int unstuff;
} rev_t;
static void rev_fill(rev_t* s)
{
    while (s->n <= 56) {
        uint32_t b = (s->left > 0 && s->pos >= 0) ? s->d[s->pos] : 0u;
        s->pos--; s->left--;
        uint32_t w = 8u - ((s->unstuff && (b & 0x7F) == 0x7F) ? 1u : 0u);
        s->acc |= (uint64_t)b << s->n;
        s->n += (int)w;
        s->unstuff = b > 0x8F;
    }
}
static uint32_t rev_peek(rev_t* s) { rev_fill(s); return (uint32_t)s->acc; }
static void rev_skip(rev_t* s, uint32_t nb) { s->acc >>= nb; s->n -= (int)nb; }

/* UVLC prefix: '1' -> u=1, '01' -> u=2, '001' -> 3 + 1-bit suffix, '000' -> 5 + 5-bit suffix (:706-716) */
static void uvlc_prefix(uint32_t bits, uint32_t* plen, uint32_t* slen, uint32_t* base)
{
    if (bits & 1)            { *plen = 1; *slen = 0; *base = 1; }
    else if (bits & 2)       { *plen = 2; *slen = 0; *base = 2; }
    else if (bits & 4)       { *plen = 3; *slen = 1; *base = 3; }
    else                     { *plen = 3; *slen = 5; *base = 5; }
}

int32_t orc_ht_decode_block(const uint8_t* coded, uint32_t len, uint32_t missing_msbs,
                            uint32_t w, uint32_t h, uint32_t* out, uint32_t stride)
{
    if (missing_msbs > 29 || len < 2 || w == 0 || h == 0 || w > 1024) return -1;
    const uint32_t p = 30 - missing_msbs;
    const int lcup = (int)len;
    const int scup = ((int)coded[lcup - 1] << 4) + (coded[lcup - 2] & 0xF);
    if (scup < 2 || scup > lcup || scup > 4079) return -1;

    fwd_t ms = {coded, 0, lcup - scup, 0, 0, 0};
    mel_t mel = {coded, lcup - scup, scup - 1, 0, 0, 0, 0, 0, 0};
    rev_t vlc; memset(&vlc, 0, sizeof vlc);
    {   /* first VLC byte: only its high nibble, 3 bits if they are 111 (:438-447) */
        uint32_t d = coded[lcup - 2];
        vlc.d = coded; vlc.pos = lcup - 3; vlc.left = scup - 2;
        vlc.acc = d >> 4; vlc.n = 4 - (((d >> 4) & 7) == 7);
        vlc.unstuff = (d | 0xF) > 0x8F;
    }

    /* exponent and significance of the bottom sample row of the quad row above, indexed x+1 */
    uint8_t* Ea = (uint8_t*)calloc(2 * (w + 8), 1);
    uint8_t* Sa = (uint8_t*)calloc(2 * (w + 8), 1);
    uint8_t *En = Ea + (w + 8), *Sn = Sa + (w + 8);
    int rc = 0;
    const uint32_t QW = (w + 1) / 2, QH = (h + 1) / 2;

    for (uint32_t qy = 0; qy < QH && rc == 0; ++qy) {
        memset(En, 0, w + 8); memset(Sn, 0, w + 8);
        uint32_t chain = 0;                       /* first row: context; other rows: sigma^W|sigma^SW part */
        for (uint32_t q0 = 0; q0 < QW; q0 += 2) {
            uint32_t qinf[2] = {0, 0}, U[2];
            for (uint32_t j = 0; j < 2; ++j) {
                const uint32_t q = q0 + j;
                if (q >= QW) break;
                uint32_t c = chain;
                if (qy > 0) {
                    const uint32_t x = 2 * q + 1;                 /* index of sample column 2q in Ea/Sa */
                    c |= (uint32_t)(Sa[x - 1] | Sa[x]);           /* sigma^NW | sigma^N  */
                    c |= (uint32_t)(Sa[x + 1] | Sa[x + 2]) << 2;  /* sigma^NE | sigma^NF */
                }
                uint32_t t = (qy == 0 ? HT_VLC_DEC0 : HT_VLC_DEC1)[(c << 7) | (rev_peek(&vlc) & 0x7F)];
                if (c == 0 && !mel_event(&mel)) t = 0;
                rev_skip(&vlc, t & 7);
                qinf[j] = t;
                const uint32_t rho = (t >> 4) & 0xF;
                chain = qy == 0 ? ((rho & 1) | (rho >> 1)) : ((((rho >> 2) | (rho >> 3)) & 1) << 1);
            }
            /* u values of the pair (:668-777) */
            uint32_t mode = ((qinf[0] >> 3) & 1) | (((qinf[1] >> 3) & 1) << 1);
            uint32_t add = 1;                                       /* U = u + kappa, kappa >= 1 */
            if (qy == 0 && mode == 3 && mel_event(&mel)) { mode = 4; add = 3; }
            U[0] = U[1] = 1;
            {
                uint32_t v = rev_peek(&vlc), used = 0, pl, sl, base;
                if (mode == 1 || mode == 2) {
                    uvlc_prefix(v, &pl, &sl, &base); v >>= pl; used = pl + sl;
                    U[mode - 1] = base + (v & ((1u << sl) - 1)) + 1;
                } else if (mode == 3 && qy == 0) {
                    uvlc_prefix(v, &pl, &sl, &base); v >>= pl; used = pl;
                    if (pl > 2) {                                   /* second quad: one bit */
                        U[1] = (v & 1) + 1 + 1; v >>= 1; used += 1 + sl;
                        U[0] = base + (v & ((1u << sl) - 1)) + 1;
                    } else {
                        uint32_t pl2, sl2, base2;
                        uvlc_prefix(v, &pl2, &sl2, &base2); v >>= pl2; used += pl2 + sl + sl2;
                        U[0] = base + (v & ((1u << sl) - 1)) + 1; v >>= sl;
                        U[1] = base2 + (v & ((1u << sl2) - 1)) + 1;
                    }
                } else if (mode >= 3) {                             /* both, plain (or first row with MEL 1) */
                    uint32_t pl2, sl2, base2;
                    uvlc_prefix(v, &pl, &sl, &base); v >>= pl;
                    uvlc_prefix(v, &pl2, &sl2, &base2); v >>= pl2;
                    used = pl + pl2 + sl + sl2;
                    U[0] = base + (v & ((1u << sl) - 1)) + add; v >>= sl;
                    U[1] = base2 + (v & ((1u << sl2) - 1)) + add;
                }
                if (U[0] > missing_msbs || U[1] > missing_msbs) { rc = -1; break; }   /* :1194 */
                rev_skip(&vlc, used);
            }
            for (uint32_t j = 0; j < 2; ++j) {
                const uint32_t q = q0 + j;
                if (q >= QW) break;
                const uint32_t t = qinf[j], rho = (t >> 4) & 0xF, e1 = (t >> 8) & 0xF, ek = (t >> 12) & 0xF;
                uint32_t Uq = U[j];
                if (qy > 0 && (rho & (rho - 1))) {                  /* gamma: kappa = max(1, Emax - 1) (:1383-1398) */
                    const uint32_t x = 2 * q + 1;
                    uint32_t E = Ea[x - 1];
                    if (Ea[x] > E) E = Ea[x];
                    if (Ea[x + 1] > E) E = Ea[x + 1];
                    if (Ea[x + 2] > E) E = Ea[x + 2];
                    Uq += E > 2 ? E - 2 : 0;
                }
                for (uint32_t i = 0; i < 4; ++i) {
                    const uint32_t x = 2 * q + (i >> 1), y = 2 * qy + (i & 1);
                    uint32_t val = 0;
                    if ((rho >> i) & 1) {
                        const uint32_t m = Uq - ((ek >> i) & 1);
                        const uint32_t bits = fwd_get(&ms, m);
                        uint32_t v = bits | (((e1 >> i) & 1) << m) | 1u;
                        val = (bits << 31) | ((v + 2) << (p - 1));
                        if (i & 1) { En[x + 1] = (uint8_t)(32 - __builtin_clz(v)); Sn[x + 1] = 1; }
                    }
                    if (x < w && y < h) out[(size_t)y * stride + x] = val;
                }
            }
        }
        uint8_t* tE = Ea; Ea = En; En = tE;
        uint8_t* tS = Sa; Sa = Sn; Sn = tS;
    }
    free(Ea < En ? Ea : En); free(Sa < Sn ? Sa : Sn);
    return rc;
}

/* ---- a15: dequantisation of the decoder's sign-magnitude words ---------------------------------- */
void orc_ht_dequant_rev(const uint32_t* sm, uint32_t n, uint32_t k_msbs, int32_t* out)
{   /* ShiftHTFilter: shift = 31 - (k_msbs + 1) */
    const uint32_t shift = 31u - (k_msbs + 1u);
    for (uint32_t i = 0; i < n; ++i) {
        const int32_t m = (int32_t)((sm[i] & 0x7FFFFFFFu) >> shift);
        out[i] = (sm[i] & 0x80000000u) ? -m : m;
    }
}
void orc_ht_dequant_irrev(const uint32_t* sm, uint32_t n, float scale, float* out)
{   /* ScaleHTFilter */
    for (uint32_t i = 0; i < n; ++i) {
        const float v = (float)(int32_t)(sm[i] & 0x7FFFFFFFu) * scale;
        out[i] = (sm[i] & 0x80000000u) ? -v : v;
    }
}

/* ---- a16: inverse 9/7, one interleaved line starting on an even coordinate ----------------------- */
static void idwt97_line(float* x, size_t stride, uint32_t n, float* t)
{
    if (n == 1) return;                                     /* WaveletReverse.cpp:1064-1066 */
    const uint32_t sn = (n + 1) >> 1, dn = n - sn;
    const float K = 1.230174105f, twice_invK = 1.625732422f;
    const float c_delta = -0.443506852f, c_gamma = -0.882911075f, c_beta = 0.052980118f, c_alpha = 1.586134342f;
    for (uint32_t i = 0; i < sn; ++i) t[2 * i] = x[(size_t)i * stride] * K;
    for (uint32_t i = 0; i < dn; ++i) t[2 * i + 1] = x[(size_t)(sn + i) * stride] * twice_invK;
    const float cs[4] = {c_delta, c_gamma, c_beta, c_alpha};
    for (int s = 0; s < 4; ++s) {
        const float c = cs[s];
        if ((s & 1) == 0) {                                 /* even samples from their odd neighbours */
            const uint32_t imax = sn < dn ? sn : dn;        /* lenMax = min(sn, dn) */
            for (uint32_t i = 0; i < imax; ++i) {
                const float l = i ? t[2 * i - 1] : t[1];
                t[2 * i] = t[2 * i] + ((l + t[2 * i + 1]) * c);
            }
            if (imax < sn) t[2 * (sn - 1)] = t[2 * (sn - 1)] + t[2 * sn - 3] * (c + c);
        } else {                                            /* odd samples from their even neighbours */
            const uint32_t lm = sn - 1 < dn ? sn - 1 : dn;  /* lenMax = min(dn, sn - 1) */
            for (uint32_t i = 0; i < lm; ++i) t[2 * i + 1] = t[2 * i + 1] + ((t[2 * i] + t[2 * i + 2]) * c);
            if (lm < dn) t[2 * dn - 1] = t[2 * dn - 1] + t[2 * dn - 2] * (c + c);
        }
    }
    for (uint32_t i = 0; i < n; ++i) x[(size_t)i * stride] = t[i];
}
static uint32_t cdiv2n(uint32_t v, uint32_t n) { return (uint32_t)(((uint64_t)v + (1ull << n) - 1) >> n); }
static uint32_t mirror97(int32_t i, uint32_t n)
{
    if (n == 1) return 0;
    const int32_t p = 2 * ((int32_t)n - 1);
    i %= p; if (i < 0) i += p;
    return (uint32_t)(i < (int32_t)n ? i : p - i);
}
/* the same line whose first sample lies on an ODD canonical coordinate (WaveletReverse.cpp:1011-1062 with a = 1, b = 0):
 * whole-sample symmetric extension written as index mirroring -- at an edge (l + l) * c, which is the reference's
 * nbr * (c + c) to the bit (both are the once-rounded product 2 * nbr * c) */
static void idwt97_line_odd(float* x, size_t stride, uint32_t n, float* t)
{
    if (n == 1) return;
    const uint32_t sn = n >> 1, dn = n - sn;                /* lows sit on the odd INDICES now */
    const float K = 1.230174105f, twice_invK = 1.625732422f;
    const float cs[4] = {-0.443506852f, -0.882911075f, 0.052980118f, 1.586134342f};
    for (uint32_t i = 0; i < sn; ++i) t[2 * i + 1] = x[(size_t)i * stride] * K;
    for (uint32_t i = 0; i < dn; ++i) t[2 * i] = x[(size_t)(sn + i) * stride] * twice_invK;
    for (int s = 0; s < 4; ++s) {
        const float c = cs[s];
        for (uint32_t k = (s & 1) ? 0u : 1u; k < n; k += 2) {    /* delta, beta: the low samples; gamma, alpha: the high ones */
            const float l = t[mirror97((int32_t)k - 1, n)], r = t[mirror97((int32_t)k + 1, n)];
            t[k] = t[k] + ((l + r) * c);
        }
    }
    for (uint32_t i = 0; i < n; ++i) x[(size_t)i * stride] = t[i];
}
void orc_dwt97_inv_at(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0)
{
    uint32_t m = (w > h ? w : h) + 4;
    float* t = (float*)malloc(m * sizeof(float));
    for (int32_t l = (int32_t)levels - 1; l >= 0; --l) {
        const uint32_t lx = cdiv2n(x0, (uint32_t)l), ly = cdiv2n(y0, (uint32_t)l);
        const uint32_t cw = cdiv2n(x0 + w, (uint32_t)l) - lx, ch = cdiv2n(y0 + h, (uint32_t)l) - ly;
        for (uint32_t y = 0; y < ch; ++y) { if (lx & 1u) idwt97_line_odd(plane + (size_t)y * stride, 1, cw, t); else idwt97_line(plane + (size_t)y * stride, 1, cw, t); }
        for (uint32_t x = 0; x < cw; ++x) { if (ly & 1u) idwt97_line_odd(plane + x, stride, ch, t); else idwt97_line(plane + x, stride, ch, t); }
    }
    free(t);
}
void orc_dwt97_inv(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels)
{
    orc_dwt97_inv_at(plane, w, h, stride, levels, 0, 0);
}

/* ---- a17: inverse colour transforms + DC shift + clamp ------------------------------------------- */
static int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* float -> int32 as the reference's bulk path does it (_mm256_cvtps_epi32, mct.cpp:248-250): round to
 * nearest even; out of range or NaN gives the "integer indefinite" value 0x80000000 */
static int32_t cvt_rn(float f)
{
    if (!(fabsf(f) < 2147483648.0f)) return INT32_MIN;
    return (int32_t)lrintf(f);
}
void orc_rct_inv_store(int32_t* c0, int32_t* c1, int32_t* c2, size_t n, int32_t shift, int32_t lo, int32_t hi)
{   /* mct.cpp:369-465 */
    for (size_t i = 0; i < n; ++i) {
        const int32_t y = c0[i], u = c1[i], v = c2[i];
        const int32_t g = y - ((u + v) >> 2), r = v + g, b = u + g;
        c0[i] = clampi(r + shift, lo, hi); c1[i] = clampi(g + shift, lo, hi); c2[i] = clampi(b + shift, lo, hi);
    }
}
void orc_ict_inv_store(int32_t* c0, int32_t* c1, int32_t* c2, size_t n, int32_t shift, int32_t lo, int32_t hi)
{   /* mct.cpp:186-294; lrintf = round to nearest even like cvtps_epi32 */
    for (size_t i = 0; i < n; ++i) {
        float y, u, v;
        memcpy(&y, &c0[i], 4); memcpy(&u, &c1[i], 4); memcpy(&v, &c2[i], 4);
        const float r = y + (v * 1.402f);
        const float g = y - (u * 0.34413f) - (v * 0.71414f);
        const float b = y + (u * 1.772f);
        c0[i] = clampi(cvt_rn(r) + shift, lo, hi);
        c1[i] = clampi(cvt_rn(g) + shift, lo, hi);
        c2[i] = clampi(cvt_rn(b) + shift, lo, hi);
    }
}
void orc_dc_store_rev(int32_t* c, size_t n, int32_t shift, int32_t lo, int32_t hi)
{   /* mct.cpp:297-364 */
    for (size_t i = 0; i < n; ++i) c[i] = clampi(c[i] + shift, lo, hi);
}
void orc_dc_store_irrev(int32_t* c, size_t n, int32_t shift, int32_t lo, int32_t hi)
{   /* mct.cpp:109-177 */
    for (size_t i = 0; i < n; ++i) {
        float f; memcpy(&f, &c[i], 4);
        c[i] = clampi(cvt_rn(f) + shift, lo, hi);
    }
}
