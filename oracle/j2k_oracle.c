/* oracle/j2k_oracle.c -- TEST INFRASTRUCTURE ONLY (see j2k_oracle.h).
 *
 * CPU restatement of the reference hot path.  Written from the behavioural description in
 * SURVEY.md Appendix A and checked against the real reference (oracle/_ref) and the Appendix C
 * known answers; it is deliberately organised differently from the reference (per-quad analysis
 * followed by stream emission, mirror-indexed lifting) because it doubles as the specification
 * the HIP kernels are written against.
 */
#include "j2k_oracle.h"
#include "ht_vlc_tables.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------ a1/a2 ingest + DC shift */
void orc_ingest(const void* src, int bps, int32_t* dst, uint32_t w, uint32_t h, uint32_t stride,
                int32_t dc)
{
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            size_t i = (size_t)y * w + x;
            int32_t v;                 /* bps < 0: signed samples, read as int8 / int16 (TileProcessor.cpp:1188-1212) */
            if (bps == 1) v = ((const uint8_t*)src)[i];
            else if (bps == 2) v = ((const uint16_t*)src)[i];
            else if (bps == -1) v = ((const int8_t*)src)[i];
            else v = ((const int16_t*)src)[i];
            dst[(size_t)y * stride + x] = v - dc;
        }
}

/* ------------------------------------------------------------------ a3 RCT (mct.cpp:94-104) */
void orc_rct_fwd(int32_t* c0, int32_t* c1, int32_t* c2, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        int32_t r = c0[i], g = c1[i], b = c2[i];
        c0[i] = (r + 2 * g + b) >> 2;
        c1[i] = b - g;
        c2[i] = r - g;
    }
}
void orc_rct_inv(int32_t* c0, int32_t* c1, int32_t* c2, size_t n)
{   /* mct.cpp:454-464 */
    for (size_t i = 0; i < n; ++i) {
        int32_t y = c0[i], u = c1[i], v = c2[i];
        int32_t g = y - ((u + v) >> 2);
        c0[i] = v + g; c1[i] = g; c2[i] = u + g;
    }
}

/* ------------------------------------------------------------------ a4 ICT (mct.cpp:469-554)
 * y = .299 r + .587 g + .114 b (left-to-right), u = cb*(b-y), v = cr*(r-y); every product and
 * sum individually rounded to fp32 (the file is built with -ffp-contract=off). */
void orc_ict_fwd(int32_t* c0, int32_t* c1, int32_t* c2, size_t n)
{
    const float a_r = 0.299f, a_g = 0.587f, a_b = 0.114f;
    const float cb = 0.5f / (1.0f - a_b), cr = 0.5f / (1.0f - a_r);
    for (size_t i = 0; i < n; ++i) {
        float r = (float)c0[i], g = (float)c1[i], b = (float)c2[i];
        float y = a_r * r;
        y = y + a_g * g;
        y = y + a_b * b;
        float u = cb * (b - y);
        float v = cr * (r - y);
        memcpy(&c0[i], &y, 4); memcpy(&c1[i], &u, 4); memcpy(&c2[i], &v, 4);
    }
}

/* ------------------------------------------------------------------ a6 5/3 lifting
 * One line, first sample on an even coordinate. Whole-sample symmetric extension is expressed by
 * mirroring indices (x[-1]=x[1], x[n]=x[n-2]); this reproduces the reference's edge formulas
 * (WaveletFwd.cpp:866-883) bit for bit because the mirrored operands are the same values. */
static inline uint32_t mirror(int32_t i, uint32_t n)
{
    if (n == 1) return 0;
    int32_t p = 2 * ((int32_t)n - 1);
    i %= p; if (i < 0) i += p;
    return (uint32_t)(i < (int32_t)n ? i : p - i);
}

/* par = parity of the first sample's canonical coordinate (tiles / images on odd origins: WaveletFwd.cpp:884-905 rows,
 * :782-842 columns): the samples on ODD coordinates are predicted, the ones on even coordinates updated, whatever index
 * they have; a single sample is left alone when it is a low one and doubled when it is a high one (:885-887, :812-815) */
static void dwt53_line(const int32_t* in, size_t istride, int32_t* out, size_t ostride, uint32_t n,
                       int32_t* tmp, uint32_t par)
{
    if (n == 1) { out[0] = par ? in[0] * 2 : in[0]; return; }
    uint32_t sn = (n + 1 - par) >> 1, dn = n - sn;
    for (uint32_t k = 0; k < n; ++k) tmp[k] = in[k * istride];
    /* predict: samples on odd coordinates */
    for (uint32_t k = 1 - par; k < n; k += 2)
        tmp[k] -= (tmp[mirror((int32_t)k - 1, n)] + tmp[mirror((int32_t)k + 1, n)]) >> 1;
    /* update: samples on even coordinates */
    for (uint32_t k = par; k < n; k += 2)
        tmp[k] += (tmp[mirror((int32_t)k - 1, n)] + tmp[mirror((int32_t)k + 1, n)] + 2) >> 2;
    for (uint32_t i = 0; i < sn; ++i) out[i * ostride] = tmp[2 * i + par];
    for (uint32_t i = 0; i < dn; ++i) out[(sn + i) * ostride] = tmp[2 * i + 1 - par];
}

void orc_dwt53_fwd_1d(int32_t* x, uint32_t n)
{
    int32_t* tmp = (int32_t*)malloc((n + 1) * sizeof(int32_t));
    dwt53_line(x, 1, x, 1, n, tmp, 0);
    free(tmp);
}
void orc_dwt53_fwd_1d_par(int32_t* x, uint32_t n, uint32_t par)
{
    int32_t* tmp = (int32_t*)malloc((n + 1) * sizeof(int32_t));
    dwt53_line(x, 1, x, 1, n, tmp, par);
    free(tmp);
}

/* a7 9/7 lifting: four sweeps + scaling; (a+b)*c with three separate fp32 roundings
 * (WaveletFwd.cpp:134-214). */
static void dwt97_line(const float* in, size_t istride, float* out, size_t ostride, uint32_t n,
                       float* w, uint32_t par)
{
    static const float alpha = -1.586134342f, beta = -0.052980118f;
    static const float gamma_ = 0.882911075f, delta = 0.443506852f;
    static const float K = 1.230174105f;
    const float invK = (float)(1.0 / 1.230174105);
    if (n == 1) { out[0] = in[0]; return; }      /* either parity (WaveletFwd.cpp:924-926, :985-987) */
    uint32_t sn = (n + 1 - par) >> 1, dn = n - sn;
    for (uint32_t k = 0; k < n; ++k) w[k] = in[k * istride];
    const float c[4] = {alpha, beta, gamma_, delta};
    for (int s = 0; s < 4; ++s) {
        uint32_t first = (s & 1) ? par : 1u - par;      /* alpha,gamma: odd coordinates ; beta,delta: even */
        for (uint32_t k = first; k < n; k += 2) {
            float l = w[mirror((int32_t)k - 1, n)], r = w[mirror((int32_t)k + 1, n)];
            float sum = l + r;
            float prod = sum * c[s];
            w[k] = w[k] + prod;
        }
    }
    for (uint32_t i = 0; i < sn; ++i) out[i * ostride] = w[2 * i + par] * invK;
    for (uint32_t i = 0; i < dn; ++i) out[(sn + i) * ostride] = w[2 * i + 1 - par] * K;
}

void orc_dwt97_fwd_1d(float* x, uint32_t n)
{
    float* tmp = (float*)malloc((n + 1) * sizeof(float));
    dwt97_line(x, 1, x, 1, n, tmp, 0);
    free(tmp);
}
void orc_dwt97_fwd_1d_par(float* x, uint32_t n, uint32_t par)
{
    float* tmp = (float*)malloc((n + 1) * sizeof(float));
    dwt97_line(x, 1, x, 1, n, tmp, par);
    free(tmp);
}

static uint32_t cdivp2(uint32_t v, uint32_t n) { return (uint32_t)(((uint64_t)v + (1ull << n) - 1) >> n); }

/* a5: level loop -- vertical pass over every column, then horizontal over every row
 * (WaveletFwd.cpp:491-602). */
/* (x0, y0): the tile-component's origin on the canonical grid; level l works on [ceil(x0 / 2^l), ceil((x0 + w) / 2^l)) */
void orc_dwt53_fwd_at(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0)
{
    uint32_t m = (w > h ? w : h) + 2;
    int32_t* tmp = (int32_t*)malloc(m * sizeof(int32_t));
    for (uint32_t l = 0; l < levels; ++l) {
        const uint32_t lx = cdivp2(x0, l), ly = cdivp2(y0, l);
        const uint32_t cw = cdivp2(x0 + w, l) - lx, ch = cdivp2(y0 + h, l) - ly;
        for (uint32_t x = 0; x < cw; ++x) dwt53_line(plane + x, stride, plane + x, stride, ch, tmp, ly & 1u);
        for (uint32_t y = 0; y < ch; ++y) dwt53_line(plane + (size_t)y * stride, 1, plane + (size_t)y * stride, 1, cw, tmp, lx & 1u);
    }
    free(tmp);
}
void orc_dwt53_fwd(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels)
{
    orc_dwt53_fwd_at(plane, w, h, stride, levels, 0, 0);
}
void orc_dwt97_fwd_at(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0)
{
    uint32_t m = (w > h ? w : h) + 2;
    float* tmp = (float*)malloc(m * sizeof(float));
    for (uint32_t l = 0; l < levels; ++l) {
        const uint32_t lx = cdivp2(x0, l), ly = cdivp2(y0, l);
        const uint32_t cw = cdivp2(x0 + w, l) - lx, ch = cdivp2(y0 + h, l) - ly;
        for (uint32_t x = 0; x < cw; ++x) dwt97_line(plane + x, stride, plane + x, stride, ch, tmp, ly & 1u);
        for (uint32_t y = 0; y < ch; ++y) dwt97_line(plane + (size_t)y * stride, 1, plane + (size_t)y * stride, 1, cw, tmp, lx & 1u);
    }
    free(tmp);
}
void orc_dwt97_fwd(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels)
{
    orc_dwt97_fwd_at(plane, w, h, stride, levels, 0, 0);
}

/* inverse 5/3 (for round-trip property tests): horizontal then vertical per level, low->high */
static void idwt53_line(int32_t* io, size_t st, uint32_t n, int32_t* tmp, uint32_t par, int vertical)
{
    /* a lone high-pass sample was doubled by the encoder: rows halve it with / 2 (decompress_h_53, WaveletReverse.cpp:598),
     * columns with >> 1 (decompress_v_53, :646) -- the same for what an encoder writes, not for an odd negative value */
    if (n == 1) { if (par) io[0] = vertical ? io[0] >> 1 : io[0] / 2; return; }
    uint32_t sn = (n + 1 - par) >> 1, dn = n - sn;
    for (uint32_t i = 0; i < sn; ++i) tmp[2 * i + par] = io[i * st];
    for (uint32_t i = 0; i < dn; ++i) tmp[2 * i + 1 - par] = io[(sn + i) * st];
    for (uint32_t k = par; k < n; k += 2)
        tmp[k] -= (tmp[mirror((int32_t)k - 1, n)] + tmp[mirror((int32_t)k + 1, n)] + 2) >> 2;
    for (uint32_t k = 1 - par; k < n; k += 2)
        tmp[k] += (tmp[mirror((int32_t)k - 1, n)] + tmp[mirror((int32_t)k + 1, n)]) >> 1;
    for (uint32_t k = 0; k < n; ++k) io[k * st] = tmp[k];
}
void orc_dwt53_inv_at(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0)
{
    uint32_t m = (w > h ? w : h) + 2;
    int32_t* tmp = (int32_t*)malloc(m * sizeof(int32_t));
    for (int32_t l = (int32_t)levels - 1; l >= 0; --l) {
        const uint32_t lx = cdivp2(x0, (uint32_t)l), ly = cdivp2(y0, (uint32_t)l);
        const uint32_t cw = cdivp2(x0 + w, (uint32_t)l) - lx, ch = cdivp2(y0 + h, (uint32_t)l) - ly;
        for (uint32_t y = 0; y < ch; ++y) idwt53_line(plane + (size_t)y * stride, 1, cw, tmp, lx & 1u, 0);
        for (uint32_t x = 0; x < cw; ++x) idwt53_line(plane + x, stride, ch, tmp, ly & 1u, 1);
    }
    free(tmp);
}
void orc_dwt53_inv(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels)
{
    orc_dwt53_inv_at(plane, w, h, stride, levels, 0, 0);
}

/* ------------------------------------------------------------------ a8 exponents / step sizes */
/* BIBO gains of the 5/3 analysis bank per decomposition count (codestream/HTParams.cpp:139-154);
 * values converge after ~15 levels. */
static const float bibo53_l[34] = {1.0000e+00f, 1.5000e+00f, 1.6250e+00f, 1.6875e+00f, 1.6963e+00f,
    1.7067e+00f, 1.7116e+00f, 1.7129e+00f, 1.7141e+00f, 1.7145e+00f, 1.7151e+00f, 1.7152e+00f,
    1.7155e+00f, 1.7155e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f,
    1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f,
    1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f, 1.7156e+00f,
    1.7156e+00f};
static const float bibo53_h[34] = {2.0000e+00f, 2.5000e+00f, 2.7500e+00f, 2.8047e+00f, 2.8198e+00f,
    2.8410e+00f, 2.8558e+00f, 2.8601e+00f, 2.8628e+00f, 2.8656e+00f, 2.8662e+00f, 2.8667e+00f,
    2.8669e+00f, 2.8670e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f,
    2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f,
    2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f, 2.8671e+00f,
    2.8671e+00f};
/* sqrt energy gains of the 9/7 synthesis bank (HTParams.cpp:72-87), first 12 levels */
static const float gain97_l[13] = {1.0000e+00f, 1.4021e+00f, 2.0304e+00f, 2.9012e+00f, 4.1153e+00f,
    5.8245e+00f, 8.2388e+00f, 1.1652e+01f, 1.6479e+01f, 2.3304e+01f, 3.2957e+01f, 4.6609e+01f, 6.5915e+01f};
static const float gain97_h[13] = {1.4425e+00f, 1.9669e+00f, 2.8839e+00f, 4.1475e+00f, 5.8946e+00f,
    8.3472e+00f, 1.1809e+01f, 1.6701e+01f, 2.3620e+01f, 3.3403e+01f, 4.7240e+01f, 6.6807e+01f, 9.4479e+01f};

static int rev_X(float g) { return (int)ceil(log(g * 1.1f) / M_LN2); }

void orc_ht_rev_exponents(uint32_t prec, uint32_t levels, uint8_t* expn)
{
    int B = (int)prec;                               /* D4: no +1 for RCT */
    uint32_t s = 0;
    float bl = bibo53_l[levels];
    expn[s++] = (uint8_t)(B + rev_X(bl * bl));
    for (int d = (int)levels - 1; d >= 0; --d) {
        float l = bibo53_l[d + 1], hh = bibo53_h[d];
        int X = rev_X(hh * l);
        expn[s++] = (uint8_t)(B + X);
        expn[s++] = (uint8_t)(B + X);
        expn[s++] = (uint8_t)(B + rev_X(hh * hh));
    }
}

static uint16_t irrev_code(float delta_b)
{
    uint32_t e = 0;
    while (delta_b < 1.0f) { e++; delta_b *= 2.0f; }
    uint32_t mant = (uint32_t)round(delta_b * (float)(1 << 11)) - (1 << 11);
    if (mant >= (1u << 11)) mant = 0x7FF;
    return (uint16_t)((e << 11) | mant);
}

void orc_ht_irrev_stepsizes(uint32_t prec, uint32_t levels, uint16_t* spqcd, float* delta)
{
    float base = 1.0f / (float)(1u << prec);        /* unsigned components */
    uint32_t s = 0;
    float gl = gain97_l[levels];
    spqcd[s++] = irrev_code(base / (gl * gl));
    for (int d = (int)levels - 1; d >= 0; --d) {
        float l = gain97_l[d + 1], hh = gain97_h[d];
        uint16_t c = irrev_code(base / (l * hh));
        spqcd[s++] = c; spqcd[s++] = c;
        spqcd[s++] = irrev_code(base / (hh * hh));
    }
    /* Quantizer.cpp:41-45, encoder side: numbps = prec + log2gain(0,1,1,2) */
    for (uint32_t b = 0; b < s; ++b) {
        uint32_t orient = b == 0 ? 0 : ((b - 1) % 3) + 1;
        uint32_t gain = orient == 0 ? 0 : (orient == 3 ? 2 : 1);
        int ex = spqcd[b] >> 11, mant = spqcd[b] & 0x7FF;
        delta[b] = (float)((1.0 + mant / 2048.0) * pow(2.0, (int32_t)(prec + gain) - ex));
    }
}

/* ------------------------------------------------------------------ a9 block enumeration */
/* band b of resolution r covers [ceil((v - 2^(n-1) bx) / 2^n)) per coordinate, n = levels - r + 1 (util/util.cpp:49-58,
 * tile/TileComponent.cpp:131-138); its code-blocks are the cells of the 2^cblk_exp grid ANCHORED AT THE ORIGIN OF THE BAND'S
 * COORDINATES that it touches (t1/T1Structs.cpp:118-136), so a band that starts off the grid begins with a partial block */
static uint32_t band_lo(uint32_t v, uint32_t n, uint32_t hi)
{
    if (n == 0) return v;
    const uint64_t off = hi ? (1ull << (n - 1)) : 0;
    return v <= off ? 0u : (uint32_t)(((uint64_t)v - off + (1ull << n) - 1) >> n);
}
/* prc: precinct exponents per resolution (r = 0 coarsest), PPx | PPy << 4 as in the COD marker, or NULL / 0 = 15, 15.  The
 * precincts of a resolution are the cells of the 2^PPx x 2^PPy grid anchored at the origin of ITS coordinates that it touches
 * (t1/T1Structs.cpp:449-493, tile/TileComponent.cpp:100-118); in the bands of a resolution r > 0 a precinct is half as large;
 * a code-block is never larger than that (cblk exponent = min(cblk_exp, precinct exponent of the band)); the blocks are
 * enumerated band -> precinct (raster) -> block (raster) (T1CompressScheduler.cpp:44-85). */
uint32_t orc_enumerate_blocks_prc(uint32_t w, uint32_t h, uint32_t levels, uint32_t cblk_exp, uint32_t x0, uint32_t y0,
                                  const uint8_t* prc, const uint8_t* expn, orc_block* out, uint32_t cap)
{
    uint32_t n = 0;
    for (uint32_t r = 0; r <= levels; ++r) {
        const uint32_t nn = r ? levels - r + 1 : levels;                 /* decomposition the band belongs to */
        const uint32_t lw = r ? cdivp2(x0 + w, nn) - cdivp2(x0, nn) : 0, lh = r ? cdivp2(y0 + h, nn) - cdivp2(y0, nn) : 0;
        const uint32_t rx0 = cdivp2(x0, levels - r), rx1 = cdivp2(x0 + w, levels - r);
        const uint32_t ry0 = cdivp2(y0, levels - r), ry1 = cdivp2(y0 + h, levels - r);
        const uint32_t pe = prc ? prc[r] : 0;
        const uint32_t ppx = pe ? (pe & 15u) : 15u, ppy = pe ? (pe >> 4) : 15u;
        const uint32_t npw = rx1 > rx0 ? (uint32_t)((((uint64_t)rx1 + (1ull << ppx) - 1) >> ppx) - (rx0 >> ppx)) : 0;
        const uint32_t nph = ry1 > ry0 ? (uint32_t)((((uint64_t)ry1 + (1ull << ppy) - 1) >> ppy) - (ry0 >> ppy)) : 0;
        const uint32_t bpx = ppx - (r ? 1u : 0u), bpy = ppy - (r ? 1u : 0u);
        const uint32_t psx = ((rx0 >> ppx) << ppx) >> (r ? 1u : 0u), psy = ((ry0 >> ppy) << ppy) >> (r ? 1u : 0u);
        const uint32_t cxe = cblk_exp < bpx ? cblk_exp : bpx, cye = cblk_exp < bpy ? cblk_exp : bpy;
        uint32_t nb = r ? 3 : 1;
        for (uint32_t bi = 0; bi < nb; ++bi) {
            uint32_t orient = r ? bi + 1 : 0;
            const uint32_t bx0 = band_lo(x0, nn, orient & 1), bx1 = band_lo(x0 + w, nn, orient & 1);
            const uint32_t by0 = band_lo(y0, nn, orient >> 1), by1 = band_lo(y0 + h, nn, orient >> 1);
            uint32_t ox = (orient & 1) ? lw : 0, oy = (orient & 2) ? lh : 0;
            uint32_t qcd_idx = r ? 3 * (r - 1) + 1 + bi : 0;
            for (uint32_t pj = 0; pj < nph; ++pj)
                for (uint32_t pi = 0; pi < npw; ++pi) {
                    const uint64_t qx0 = (uint64_t)psx + ((uint64_t)pi << bpx), qy0 = (uint64_t)psy + ((uint64_t)pj << bpy);
                    const uint64_t cx0 = qx0 > bx0 ? qx0 : bx0, cx1 = qx0 + (1ull << bpx) < bx1 ? qx0 + (1ull << bpx) : bx1;
                    const uint64_t cy0 = qy0 > by0 ? qy0 : by0, cy1 = qy0 + (1ull << bpy) < by1 ? qy0 + (1ull << bpy) : by1;
                    if (cx0 >= cx1 || cy0 >= cy1) continue;
                    const uint64_t gx0 = cx0 >> cxe, gx1 = (cx1 + (1ull << cxe) - 1) >> cxe;
                    const uint64_t gy0 = cy0 >> cye, gy1 = (cy1 + (1ull << cye) - 1) >> cye;
                    for (uint64_t gy = gy0; gy < gy1; ++gy)
                        for (uint64_t gx = gx0; gx < gx1; ++gx) {
                            if (n < cap) {
                                orc_block* b = &out[n];
                                const uint64_t ax0 = (gx << cxe) > cx0 ? (gx << cxe) : cx0, ax1 = ((gx + 1) << cxe) < cx1 ? ((gx + 1) << cxe) : cx1;
                                const uint64_t ay0 = (gy << cye) > cy0 ? (gy << cye) : cy0, ay1 = ((gy + 1) << cye) < cy1 ? ((gy + 1) << cye) : cy1;
                                b->x = ox + (uint32_t)(ax0 - bx0); b->y = oy + (uint32_t)(ay0 - by0);
                                b->w = (uint32_t)(ax1 - ax0); b->h = (uint32_t)(ay1 - ay0);
                                b->res = (uint8_t)r; b->band = (uint8_t)orient;
                                b->kmax = expn ? expn[qcd_idx] : 0; b->pad = 0;
                                b->bx = (uint32_t)(gx - gx0); b->by = (uint32_t)(gy - gy0);
                            }
                            ++n;
                        }
                }
        }
    }
    return n;
}
uint32_t orc_enumerate_blocks_at(uint32_t w, uint32_t h, uint32_t levels, uint32_t cblk_exp, uint32_t x0, uint32_t y0,
                                 const uint8_t* expn, orc_block* out, uint32_t cap)
{
    return orc_enumerate_blocks_prc(w, h, levels, cblk_exp, x0, y0, NULL, expn, out, cap);
}
uint32_t orc_enumerate_blocks(uint32_t w, uint32_t h, uint32_t levels, uint32_t cblk_exp,
                              const uint8_t* expn, orc_block* out, uint32_t cap)
{
    return orc_enumerate_blocks_at(w, h, levels, cblk_exp, 0, 0, expn, out, cap);
}

/* ------------------------------------------------------------------ a10 sign-magnitude */
void orc_ht_signmag_rev(const int32_t* src, uint32_t stride, uint32_t w, uint32_t h, uint32_t kmax,
                        uint32_t* dst)
{
    uint32_t shift = 30 - kmax;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            int32_t v = src[(size_t)y * stride + x];
            uint32_t mag = (uint32_t)(v < 0 ? -(int64_t)v : v);
            dst[y * w + x] = (v < 0 ? 0x80000000u : 0u) | (mag << shift);
        }
}
void orc_ht_signmag_irrev(const float* src, uint32_t stride, uint32_t w, uint32_t h, uint32_t kmax,
                          float inv_delta, uint32_t* dst)
{
    uint32_t shift = 30 - kmax;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            float c = src[(size_t)y * stride + x];
            float a = fabsf(c) * inv_delta;
            uint32_t q = (uint32_t)a;               /* truncation = dead-zone quantiser */
            uint32_t lim = (1u << kmax) - 1; if (q > lim) q = lim;
            dst[y * w + x] = ((c < 0 && q) ? 0x80000000u : 0u) | (q << shift);
        }
}

/* ------------------------------------------------------------------ N3 rate-control hook: distortion decrease of a block's pass */
/* T1::getwmsedec (t1/t1_part1/T1.cpp:394-414) gives every pass of the reference's Part-1 coder (w1 w2 stepsize 2^bpno)^2 nmsedec / 8192,
 * nmsedec a table of squared-error decreases in units of the bit-plane: summed over the passes of a block that is coded to the last
 * plane that is (w1 w2 stepsize)^2 times the energy of the quantised magnitudes.  An HT block is one pass, so its whole decrease is
 * that sum; w2 = T1::getnorm (T1.cpp:224-235, :258-267), w1 = mct::get_norms_* (point_transform/mct.cpp:30-41), level =
 * numresolutions - 1 - resno (t1/t1_part1/T1Part1.cpp:97).  `sm` = the block's sign-magnitude words (orc_ht_signmag_*). */
double orc_ht_block_distortion(const uint32_t* sm, uint32_t n, uint32_t kmax, uint32_t orient, uint32_t level, int reversible,
                               int mct, uint32_t comp, double stepsize)
{
    static const double n53[4][10] = {{1.000, 1.500, 2.750, 5.375, 10.68, 21.34, 42.67, 85.33, 170.7, 341.3},
                                      {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
                                      {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
                                      {.7186, .9218, 1.586, 3.043, 6.019, 12.01, 24.00, 47.97, 95.93, 0}};
    static const double n97[4][10] = {{1.000, 1.965, 4.177, 8.403, 16.90, 33.84, 67.69, 135.3, 270.6, 540.9},
                                      {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
                                      {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
                                      {2.080, 3.865, 8.307, 17.18, 34.71, 69.59, 139.3, 278.6, 557.2, 0}};
    static const double mrev[3] = {1.732, .8292, .8292}, mirr[3] = {1.732, 1.805, 1.573};
    uint64_t e = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t q = (sm[i] & 0x7FFFFFFFu) >> (30 - kmax);
        e += q * q;
    }
    if (orient == 0 && level > 9) level = 9;
    else if (orient > 0 && level > 8) level = 8;
    double w1 = (mct && comp < 3) ? (reversible ? mrev[comp] : mirr[comp]) : 1.0;
    double w2 = reversible ? n53[orient & 3][level] : n97[orient & 3][level];
    double w = w1 * w2 * stepsize;
    return w * w * (double)e;
}

/* ------------------------------------------------------------------ a11 HT cleanup encoder */
/* Optional taps that record the RAW (un-stuffed) MagSgn / VLC bit streams and the MEL state just
 * before termination; used by ht_wave_model.c to validate the wave-parallel phase-B formulation. */
typedef struct { uint32_t* bits; uint32_t n, cap_words; } raw_tap;
static __thread raw_tap* g_tap_ms = NULL;
static __thread raw_tap* g_tap_vlc = NULL;
static __thread int* g_tap_mel = NULL;      /* [pos, acc, left, run] */
static void tap_put(raw_tap* t, uint32_t v, int n)
{
    for (int i = 0; i < n; ++i, ++t->n)
        if (((v >> i) & 1) && (t->n >> 5) < t->cap_words) t->bits[t->n >> 5] |= 1u << (t->n & 31);
}
/* forward LSB-first packer with 0xFF -> next-byte-7-bits stuffing (MagSgn; :415-454) */
typedef struct { uint8_t* buf; uint32_t pos, cap; uint32_t acc; int used, limit; } fwd_bits;
static void fb_init(fwd_bits* b, uint8_t* buf, uint32_t cap) { b->buf = buf; b->pos = 0; b->cap = cap; b->acc = 0; b->used = 0; b->limit = 8; }
static int fb_put(fwd_bits* b, uint32_t v, int n)
{
    if (g_tap_ms) tap_put(g_tap_ms, v, n);
    while (n > 0) {
        int t = b->limit - b->used; if (t > n) t = n;
        b->acc |= (v & ((1u << t) - 1)) << b->used;
        b->used += t; v >>= t; n -= t;
        if (b->used == b->limit) {
            if (b->pos >= b->cap) return -1;
            b->buf[b->pos++] = (uint8_t)b->acc;
            b->limit = (b->acc == 0xFF) ? 7 : 8;
            b->acc = 0; b->used = 0;
        }
    }
    return 0;
}
static void fb_finish(fwd_bits* b)
{
    if (b->used) {
        int t = b->limit - b->used;
        b->acc |= (0xFFu & ((1u << t) - 1)) << b->used;     /* pad with ones */
        if (b->acc != 0xFF) b->buf[b->pos++] = (uint8_t)b->acc;
    } else if (b->limit == 7) {
        b->pos--;                                           /* drop a trailing 0xFF */
    }
}

/* MEL: 13-state adaptive run-length coder, bits MSB-first, same 0xFF stuffing (:217-291) */
typedef struct { uint8_t* buf; uint32_t pos, cap; int left, acc, run, k; } mel_enc;
static const int MEL_E[13] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5};
static void mel_init_(mel_enc* m, uint8_t* buf, uint32_t cap) { m->buf = buf; m->pos = 0; m->cap = cap; m->left = 8; m->acc = 0; m->run = 0; m->k = 0; }
static void mel_bit(mel_enc* m, int v)
{
    m->acc = (m->acc << 1) | v;
    if (--m->left == 0) {
        m->buf[m->pos++] = (uint8_t)m->acc;
        m->left = (m->acc == 0xFF) ? 7 : 8;
        m->acc = 0;
    }
}
static void mel_event(mel_enc* m, int one)
{
    if (!one) {
        if (++m->run >= (1 << MEL_E[m->k])) { mel_bit(m, 1); m->run = 0; if (m->k < 12) m->k++; }
    } else {
        mel_bit(m, 0);
        for (int t = MEL_E[m->k]; t > 0;) mel_bit(m, (m->run >> --t) & 1);
        m->run = 0; if (m->k > 0) m->k--;
    }
}

/* VLC: grows downwards from the end of its buffer; a byte following one that is > 0x8F holds
 * 7 bits when those 7 bits are all ones (:296-351) */
typedef struct { uint8_t* end; uint32_t pos, cap; int used, acc, prev_gt8f; } vlc_enc;
static void vlc_init_(vlc_enc* v, uint8_t* buf, uint32_t cap)
{ v->end = buf + cap - 1; v->pos = 1; v->cap = cap; v->end[0] = 0xFF; v->used = 4; v->acc = 0xF; v->prev_gt8f = 1; }
static void vlc_put(vlc_enc* v, int cw, int n)
{
    if (g_tap_vlc) tap_put(g_tap_vlc, (uint32_t)cw, n);
    while (n > 0) {
        int room = 8 - v->prev_gt8f - v->used;
        int t = room < n ? room : n;
        v->acc |= (cw & ((1 << t) - 1)) << v->used;
        v->used += t; room -= t; n -= t; cw >>= t;
        if (room == 0) {
            if (v->prev_gt8f && v->acc != 0x7F) { v->prev_gt8f = 0; continue; }  /* 8th bit allowed after all */
            *(v->end - v->pos) = (uint8_t)v->acc; v->pos++;
            v->prev_gt8f = v->acc > 0x8F;
            v->acc = 0; v->used = 0;
        }
    }
}

static void uvlc_code(int u, int* pre, int* pre_len, int* suf, int* suf_len)
{   /* :189-210 */
    if (u == 0)      { *pre = 0; *pre_len = 0; *suf = 0; *suf_len = 0; }
    else if (u == 1) { *pre = 1; *pre_len = 1; *suf = 0; *suf_len = 0; }
    else if (u == 2) { *pre = 2; *pre_len = 2; *suf = 0; *suf_len = 0; }
    else if (u <= 4) { *pre = 4; *pre_len = 3; *suf = u - 3; *suf_len = 1; }
    else             { *pre = 0; *pre_len = 3; *suf = u - 5; *suf_len = 5; }
}

typedef struct { uint8_t rho, emax, e[4]; uint32_t v[4]; } quad_info;

int32_t orc_ht_encode_sm(const uint32_t* sm, uint32_t kmax, uint32_t w, uint32_t h, uint8_t* out,
                         uint32_t cap)
{
    const uint32_t p = 30 - kmax;
    const uint32_t QW = (w + 1) >> 1, QH = (h + 1) >> 1;
    const uint32_t MS_CAP = 4 * w * h + 64, MEL_CAP = 192 + (QW * QH) / 4, VLC_CAP = 3072 + 2 * QW * QH;
    uint8_t* ms_buf = (uint8_t*)malloc(MS_CAP);
    uint8_t* mel_buf = (uint8_t*)malloc(MEL_CAP);
    uint8_t* vlc_buf = (uint8_t*)malloc(VLC_CAP);
    quad_info* row = (quad_info*)calloc(QW + 2, sizeof(quad_info));    /* current quad row  */
    uint8_t* eb = (uint8_t*)calloc(2 * QW + 4, 1);     /* exponents of the sample row just above */
    uint8_t* sb = (uint8_t*)calloc(2 * QW + 4, 1);     /* significance of that row               */
    /* eb/sb are indexed by x+1 so that x=-1 is addressable */
    fwd_bits ms; mel_enc mel; vlc_enc vlc;
    fb_init(&ms, ms_buf, MS_CAP); mel_init_(&mel, mel_buf, MEL_CAP); vlc_init_(&vlc, vlc_buf, VLC_CAP);

    for (uint32_t qy = 0; qy < QH; ++qy) {
        /* ---- stage 1: per-quad sample analysis (:513-563) */
        for (uint32_t qx = 0; qx < QW; ++qx) {
            quad_info* q = &row[qx];
            memset(q, 0, sizeof(*q));
            for (int i = 0; i < 4; ++i) {
                uint32_t x = 2 * qx + (uint32_t)(i >> 1), y = 2 * qy + (uint32_t)(i & 1);
                uint32_t t = (x < w && y < h) ? sm[(size_t)y * w + x] : 0;
                uint32_t val = ((t + t) >> p) & ~1u;             /* 2*mu */
                if (val) {
                    q->rho |= (uint8_t)(1 << i);
                    q->e[i] = (uint8_t)(32 - __builtin_clz(val - 1));
                    if (q->e[i] > q->emax) q->emax = q->e[i];
                    q->v[i] = val - 2 + (t >> 31);               /* 2(mu-1) + sign */
                }
            }
        }
        /* ---- stage 2+3: per quad-pair context, VLC/UVLC/MEL/MagSgn emission */
        for (uint32_t qx0 = 0; qx0 < QW; qx0 += 2) {
            int u[2] = {0, 0};
            for (uint32_t j = 0; j < 2 && qx0 + j < QW; ++j) {
                uint32_t qx = qx0 + j;
                const quad_info* q = &row[qx];
                int rho_left = qx ? row[qx - 1].rho : 0;
                int c_q, kappa, U;
                uint16_t tuple;
                if (qy == 0) {
                    c_q = (rho_left >> 1) | (rho_left & 1);                     /* :652, :709 */
                    kappa = 1;
                } else {
                    /* neighbourhood in the sample row above: columns 2qx-1 .. 2qx+2 (index +1) */
                    const uint8_t* E = eb + 2 * qx, *S = sb + 2 * qx;
                    int emx = E[0]; if (E[1] > emx) emx = E[1]; if (E[2] > emx) emx = E[2]; if (E[3] > emx) emx = E[3];
                    int max_e = emx - 1;
                    c_q = (S[0] | S[1]) | (((rho_left >> 2) | (rho_left >> 3)) & 1) << 1 | (S[2] | S[3]) << 2;   /* :723,:799,:872,:912 */
                    kappa = (q->rho & (q->rho - 1)) ? (max_e > 1 ? max_e : 1) : 1;                          /* :783 */
                }
                U = q->emax > kappa ? q->emax : kappa;
                u[j] = U - kappa;
                int eps = 0;
                if (u[j] > 0)
                    for (int i = 0; i < 4; ++i) eps |= (q->e[i] == q->emax) << i;
                tuple = (qy == 0 ? HT_VLC_ENC0 : HT_VLC_ENC1)[(c_q << 8) | (q->rho << 4) | eps];
                vlc_put(&vlc, tuple >> 8, (tuple >> 4) & 7);
                if (c_q == 0) mel_event(&mel, q->rho != 0);
                for (int i = 0; i < 4; ++i) {
                    int m = (q->rho >> i & 1) ? U - ((tuple >> i) & 1) : 0;
                    if (m && fb_put(&ms, q->v[i] & ((m >= 32) ? 0xFFFFFFFFu : ((1u << m) - 1)), m)) goto overflow;
                }
            }
            int pre0, pl0, s0, sl0, pre1, pl1, s1, sl1;
            if (qy == 0) {
                if (u[0] > 0 && u[1] > 0) mel_event(&mel, (u[0] < u[1] ? u[0] : u[1]) > 2);       /* :684 */
                if (u[0] > 2 && u[1] > 2) {
                    uvlc_code(u[0] - 2, &pre0, &pl0, &s0, &sl0); uvlc_code(u[1] - 2, &pre1, &pl1, &s1, &sl1);
                    vlc_put(&vlc, pre0, pl0); vlc_put(&vlc, pre1, pl1); vlc_put(&vlc, s0, sl0); vlc_put(&vlc, s1, sl1);
                } else if (u[0] > 2 && u[1] > 0) {
                    uvlc_code(u[0], &pre0, &pl0, &s0, &sl0);
                    vlc_put(&vlc, pre0, pl0); vlc_put(&vlc, u[1] - 1, 1); vlc_put(&vlc, s0, sl0);
                } else {
                    uvlc_code(u[0], &pre0, &pl0, &s0, &sl0); uvlc_code(u[1], &pre1, &pl1, &s1, &sl1);
                    vlc_put(&vlc, pre0, pl0); vlc_put(&vlc, pre1, pl1); vlc_put(&vlc, s0, sl0); vlc_put(&vlc, s1, sl1);
                }
            } else {
                uvlc_code(u[0], &pre0, &pl0, &s0, &sl0); uvlc_code(u[1], &pre1, &pl1, &s1, &sl1);
                vlc_put(&vlc, pre0, pl0); vlc_put(&vlc, pre1, pl1); vlc_put(&vlc, s0, sl0); vlc_put(&vlc, s1, sl1);
            }
        }
        /* ---- line state for the next quad row: bottom samples (1 and 3) of this row */
        memset(eb, 0, 2 * QW + 4); memset(sb, 0, 2 * QW + 4);
        for (uint32_t qx = 0; qx < QW; ++qx) {
            eb[2 * qx + 1] = row[qx].e[1]; eb[2 * qx + 2] = row[qx].e[3];
            sb[2 * qx + 1] = (row[qx].rho >> 1) & 1; sb[2 * qx + 2] = (row[qx].rho >> 3) & 1;
        }
    }

    /* ---- termination (:357-385, :438-454) */
    if (g_tap_mel) { g_tap_mel[0] = (int)mel.pos; g_tap_mel[1] = mel.acc; g_tap_mel[2] = mel.left; g_tap_mel[3] = mel.run; }
    if (mel.run > 0) mel_bit(&mel, 1);
    {
        int mel_acc = mel.acc << mel.left;
        int mel_mask = (0xFF << mel.left) & 0xFF;
        int vlc_mask = 0xFF >> (8 - vlc.used);
        if ((mel_mask | vlc_mask) != 0) {
            int fuse = mel_acc | vlc.acc;
            if ((((fuse ^ mel_acc) & mel_mask) | ((fuse ^ vlc.acc) & vlc_mask)) == 0 && fuse != 0xFF && vlc.pos > 1) {
                mel.buf[mel.pos++] = (uint8_t)fuse;
            } else {
                mel.buf[mel.pos++] = (uint8_t)mel_acc;
                *(vlc.end - vlc.pos) = (uint8_t)vlc.acc; vlc.pos++;
            }
        }
    }
    fb_finish(&ms);

    int32_t total = (int32_t)(ms.pos + mel.pos + vlc.pos);
    if ((uint32_t)total > cap) goto overflow;
    memcpy(out, ms.buf, ms.pos);
    memcpy(out + ms.pos, mel.buf, mel.pos);
    memcpy(out + ms.pos + mel.pos, vlc.end - vlc.pos + 1, vlc.pos);
    {
        uint32_t scup = mel.pos + vlc.pos;                      /* :930-935 */
        out[total - 1] = (uint8_t)(scup >> 4);
        out[total - 2] = (uint8_t)((out[total - 2] & 0xF0) | (scup & 0xF));
    }
    free(ms_buf); free(mel_buf); free(vlc_buf); free(row); free(eb); free(sb);
    return total;
overflow:
    free(ms_buf); free(mel_buf); free(vlc_buf); free(row); free(eb); free(sb);
    return -1;
}

/* raw streams of one block (test support for the wave model): MagSgn bits, VLC bits (starting
 * with the four initial 1 bits of vlc_init), MEL bytes produced so far + coder state */
int32_t orc_ht_raw_streams(const uint32_t* sm, uint32_t kmax, uint32_t w, uint32_t h,
                           uint32_t* ms_raw, uint32_t ms_words, uint32_t* ms_bits,
                           uint32_t* vlc_raw, uint32_t vlc_words, uint32_t* vlc_bits,
                           uint8_t* mel_bytes, int* mel_state /* [pos,acc,left,run] */)
{
    raw_tap tm = {ms_raw, 0, ms_words}, tv = {vlc_raw, 0, vlc_words};
    memset(ms_raw, 0, ms_words * 4); memset(vlc_raw, 0, vlc_words * 4);
    tap_put(&tv, 0xF, 4);
    uint32_t cap = 4 * w * h + 8192;
    uint8_t* tmp = (uint8_t*)malloc(cap);
    g_tap_ms = &tm; g_tap_vlc = &tv; g_tap_mel = mel_state;
    int32_t n = orc_ht_encode_sm(sm, kmax, w, h, tmp, cap);
    g_tap_ms = NULL; g_tap_vlc = NULL; g_tap_mel = NULL;
    /* MEL bytes are a prefix of the MEL segment of the output: [ms_len, ms_len + pos) -- recover
     * them by re-encoding is unnecessary: Scup tells where MEL starts */
    if (n > 0) {
        uint32_t scup = ((uint32_t)tmp[n - 1] << 4) | (tmp[n - 2] & 0xF);
        memcpy(mel_bytes, tmp + (n - (int32_t)scup), (size_t)mel_state[0]);
    }
    *ms_bits = tm.n; *vlc_bits = tv.n;
    free(tmp);
    return n;
}

int32_t orc_ht_encode_block_rev(const int32_t* src, uint32_t stride, uint32_t w, uint32_t h,
                                uint32_t kmax, uint8_t* out, uint32_t cap)
{
    uint32_t* sm = (uint32_t*)malloc((size_t)w * h * 4);
    orc_ht_signmag_rev(src, stride, w, h, kmax, sm);
    int32_t n = orc_ht_encode_sm(sm, kmax, w, h, out, cap);
    free(sm);
    return n;
}

/* ------------------------------------------------------------------ whole tile, reversible */
static uint32_t stride_for(uint32_t w) { return (w + 31) & ~31u; }     /* util/MemManager.cpp:38-43 */

int32_t orc_encode_tile_rev_at(const void* pixels, int bps, uint32_t ncomp, uint32_t w, uint32_t h,
                               uint32_t prec, uint32_t levels, int mct, uint32_t x0, uint32_t y0, orc_block* blocks_out,
                               uint32_t* lens, uint32_t max_blocks, uint8_t* coded, uint64_t cap,
                               uint64_t* total_bytes);
int32_t orc_encode_tile_rev_prc(const void* pixels, int bps, uint32_t ncomp, uint32_t w, uint32_t h,
                                uint32_t prec, uint32_t levels, int mct, uint32_t x0, uint32_t y0, const uint8_t* prc,
                                orc_block* blocks_out, uint32_t* lens, uint32_t max_blocks, uint8_t* coded, uint64_t cap,
                                uint64_t* total_bytes);
int32_t orc_encode_tile_rev(const void* pixels, int bps, uint32_t ncomp, uint32_t w, uint32_t h,
                            uint32_t prec, uint32_t levels, int mct, orc_block* blocks_out,
                            uint32_t* lens, uint32_t max_blocks, uint8_t* coded, uint64_t cap,
                            uint64_t* total_bytes)
{
    return orc_encode_tile_rev_at(pixels, bps, ncomp, w, h, prec, levels, mct, 0, 0, blocks_out, lens, max_blocks, coded, cap,
                                  total_bytes);
}
int32_t orc_encode_tile_rev_at(const void* pixels, int bps, uint32_t ncomp, uint32_t w, uint32_t h,
                               uint32_t prec, uint32_t levels, int mct, uint32_t x0, uint32_t y0, orc_block* blocks_out,
                               uint32_t* lens, uint32_t max_blocks, uint8_t* coded, uint64_t cap,
                               uint64_t* total_bytes)
{
    return orc_encode_tile_rev_prc(pixels, bps, ncomp, w, h, prec, levels, mct, x0, y0, NULL, blocks_out, lens, max_blocks, coded, cap,
                                   total_bytes);
}
int32_t orc_encode_tile_rev_prc(const void* pixels, int bps, uint32_t ncomp, uint32_t w, uint32_t h,
                                uint32_t prec, uint32_t levels, int mct, uint32_t x0, uint32_t y0, const uint8_t* prc,
                                orc_block* blocks_out, uint32_t* lens, uint32_t max_blocks, uint8_t* coded, uint64_t cap,
                                uint64_t* total_bytes)
{
    uint32_t stride = stride_for(w);
    size_t plane_n = (size_t)stride * h;
    int32_t* planes = (int32_t*)calloc(plane_n * ncomp, sizeof(int32_t));
    uint8_t expn[3 * 32 + 1];
    if (!planes) return -1;
    for (uint32_t c = 0; c < ncomp; ++c)
        orc_ingest((const uint8_t*)pixels + (size_t)c * w * h * bps, bps, planes + c * plane_n, w, h, stride,
                   1 << (prec - 1));
    if (mct && ncomp >= 3) orc_rct_fwd(planes, planes + plane_n, planes + 2 * plane_n, plane_n);
    for (uint32_t c = 0; c < ncomp; ++c) orc_dwt53_fwd_at(planes + c * plane_n, w, h, stride, levels, x0, y0);
    orc_ht_rev_exponents(prec, levels, expn);
    uint32_t nb = orc_enumerate_blocks_prc(w, h, levels, 6, x0, y0, prc, expn, NULL, 0);
    if (nb * ncomp > max_blocks) { free(planes); return -2; }
    uint64_t off = 0; uint32_t k = 0;
    for (uint32_t c = 0; c < ncomp; ++c) {
        orc_enumerate_blocks_prc(w, h, levels, 6, x0, y0, prc, expn, blocks_out + k, nb);
        for (uint32_t i = 0; i < nb; ++i, ++k) {
            orc_block* b = &blocks_out[k];
            b->pad = (uint8_t)c;
            if (cap - off < 20000) { free(planes); return -3; }
            int32_t n = orc_ht_encode_block_rev(planes + c * plane_n + (size_t)b->y * stride + b->x, stride,
                                                b->w, b->h, b->kmax, coded + off, (uint32_t)(cap - off > 0x7fffffff ? 0x7fffffff : cap - off));
            if (n < 0) { free(planes); return -4; }
            lens[k] = (uint32_t)n; off += (uint32_t)n;
        }
    }
    *total_bytes = off;
    free(planes);
    return (int32_t)k;
}
