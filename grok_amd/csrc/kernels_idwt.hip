// grok_amd/csrc/kernels_idwt.hip -- K6: one inverse DWT level (5/3 int32 or 9/7 fp32) and K7: egress
// (inverse RCT/ICT + DC level shift + clamp + narrowing store), gfx950.
//
// K6 replaces one resolution step of decompress_tile_53 / decompress_tile_97
// (transform/WaveletReverse.cpp:852-936, :1360-1439), which synthesises all rows horizontally
// (LL|HL and LH|HH), then all columns vertically, through split windows in memory.  Here both
// passes are fused, mirroring K2:
//  * a workgroup (256 threads) owns 448 output columns (224 coefficient pairs = 7 cache lines of
//    each sub-band row, + 2 halo pairs each side) and a segment of `seg_pairs` output row pairs and
//    streams down the rows;
//  * per row pair it loads one LL, HL, LH and HH row segment (one coefficient per lane and band),
//    exchanges them through a double-buffered LDS line, and every lane synthesises two adjacent
//    output columns of the low row and of the high row with the local inverse lifting stencil;
//  * vertical synthesis runs in registers as a recurrence per column (state 2 values for 5/3,
//    4 for 9/7) and the finished rows are stored 8 bytes per lane (512 B per wave and row).
// Every coefficient is read once and every sample written once: 8 bytes/sample/level.
// The LAST level can write the caller's pixels itself (K7 fused: three MCT components side by side,
// inverse RCT/ICT + DC shift + clamp in registers), for the whole tile or for a window of it; a
// region decode launches only the strips x row segments the window depends on.
//
// Borders: whole-sample symmetric extension by mirroring the interleaved index (lifting preserves
// the symmetry, so the extended synthesis equals the reference's edge formulas, :104-184, :990-1060).
// A level that starts on an odd coordinate (a.px / a.py, see kernels_dwt.hip) is synthesised on the coordinate grid:
// pair J = coordinates (2J, 2J + 1), sample k of the level = coordinate k + parity; a lone high-pass sample is halved by
// the 5/3 (rows: / 2, columns: >> 1 -- WaveletReverse.cpp:598, :646) and passes through the 9/7 (:1064-1066).
// 9/7: low*K, high*(2/K) first, then the four lifting sweeps with coefficients -delta, -gamma,
// -beta, -alpha, each `x + ((l + r) * c)` separately rounded (:1068-1073, :1005-1008).
#include "kernels.h"
#include "pk16.h"
#include <type_traits>

namespace grk_amd {

namespace {

constexpr int kThreads   = 256;
constexpr int kHaloPairs = 2;
// Coefficient pairs per strip.  At most kThreads - 2 * kHaloPairs = 252; 224 pairs = 7 cache lines of each sub-band row
// and 14 of each output row, so loads and stores cover whole, aligned lines.
constexpr int kOutPairs  = 224;
static_assert(kOutPairs <= kThreads - 2 * kHaloPairs, "strip does not fit the staged line");

__device__ __forceinline__ uint32_t mirror_idx(int32_t i, uint32_t n)
{
    if (n == 1) return 0;
    const int32_t p = 2 * ((int32_t)n - 1);
    i %= p;
    if (i < 0) i += p;
    return (uint32_t)(i < (int32_t)n ? i : p - i);
}
// same for indices that leave [0, n) by fewer than 16 samples (the row loop): one reflection, no division
__device__ __forceinline__ uint32_t mirror_row(int32_t i, uint32_t n)
{
    if (n < 16) return mirror_idx(i, n);
    i = i < 0 ? -i : i;
    return (uint32_t)(i < (int32_t)n ? i : 2 * ((int32_t)n - 1) - i);
}

constexpr float kK        = 1.230174105f;
constexpr float kTwoInvK  = 1.625732422f;
constexpr float kIDelta   = -0.443506852f;
constexpr float kIGamma   = -0.882911075f;
constexpr float kIBeta    = 0.052980118f;
constexpr float kIAlpha   = 1.586134342f;

__device__ __forceinline__ float lift(float x, float l, float r, float c)
{
    return __fadd_rn(x, __fmul_rn(__fadd_rn(l, r), c));
}

// ---- horizontal synthesis of one row: s = low half (LL or LH), d = high half (HL or HH) in LDS,
//      lane produces the samples at columns 2j and 2j+1 (j = its local pair index) -----------------
__device__ __forceinline__ void hs53(const int32_t* s, const int32_t* d, int32_t& xe, int32_t& xo)
{
    const int32_t dm = d[-1], d0 = d[0], dp = d[1];
    xe = s[0] - ((dm + d0 + 2) >> 2);
    const int32_t xe2 = s[1] - ((d0 + dp + 2) >> 2);
    xo = d0 + ((xe + xe2) >> 1);
}
__device__ __forceinline__ void hs97(const float* s, const float* d, float& xe, float& xo)
{
    float sk[4], dk[5], e1[4], o1[3], e2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) sk[i] = __fmul_rn(s[i - 1], kK);             // pairs j-1 .. j+2
#pragma unroll
    for (int i = 0; i < 5; ++i) dk[i] = __fmul_rn(d[i - 2], kTwoInvK);       // pairs j-2 .. j+2
#pragma unroll
    for (int i = 0; i < 4; ++i) e1[i] = lift(sk[i], dk[i], dk[i + 1], kIDelta);      // evens j-1 .. j+2
#pragma unroll
    for (int i = 0; i < 3; ++i) o1[i] = lift(dk[i + 1], e1[i], e1[i + 1], kIGamma);  // odds  j-1 .. j+1
#pragma unroll
    for (int i = 0; i < 2; ++i) e2[i] = lift(e1[i + 1], o1[i], o1[i + 1], kIBeta);   // evens j, j+1
    xe = e2[0];
    xo = lift(o1[1], e2[0], e2[1], kIAlpha);
}

// ---- per-column vertical synthesis recurrences; step i consumes (s_i, d_i) ---------------------------
struct IV53 {       // yields rows 2i-1 and 2i
    int32_t dprev, xprev;
    __device__ __forceinline__ void init() { dprev = 0; xprev = 0; }
    __device__ __forceinline__ void step(int32_t s, int32_t d, int32_t& r_odd, int32_t& r_even)
    {
        r_even = s - ((dprev + d + 2) >> 2);
        r_odd = dprev + ((xprev + r_even) >> 1);
        dprev = d; xprev = r_even;
    }
};
struct IV97 {       // yields rows 2(i-1) and 2(i-2)+1
    float dk1, e1p, o1p, e2p;       // dK[i-1], e1[i-1], o1[i-2], e2[i-2]
    __device__ __forceinline__ void init() { dk1 = e1p = o1p = e2p = 0.f; }
    __device__ __forceinline__ void step(float s, float d, float& r_odd, float& r_even)
    {
        const float dk = __fmul_rn(d, kTwoInvK);
        const float e1 = lift(__fmul_rn(s, kK), dk1, dk, kIDelta);     // e1[i]
        const float o1 = lift(dk1, e1p, e1, kIGamma);                  // o1[i-1]
        const float e2 = lift(e1p, o1p, o1, kIBeta);                   // e2[i-1]
        r_odd = lift(o1p, e2p, e2, kIAlpha);                           // row 2(i-2)+1
        r_even = e2;                                                   // row 2(i-1)
        dk1 = dk; e1p = e1; o1p = o1; e2p = e2;
    }
};

// float -> int32 as the reference's bulk path (_mm256_cvtps_epi32, mct.cpp:248-250): round to nearest
// even, out of range / NaN -> 0x80000000 (v_cvt would saturate instead)
__device__ __forceinline__ int32_t cvt_rn(float f)
{
    return fabsf(f) < 2147483648.0f ? __float2int_rn(f) : (int32_t)0x80000000;
}
// K7 for one pixel: inverse RCT/ICT on the MCT triple (mct.cpp:454-464, :278-289), float -> int, DC shift, clamp.
// c[] holds int32 values or float bit patterns (irreversible).
template <int NC>
__device__ __forceinline__ void egress_px(int32_t (&c)[NC], bool irrev, bool mct, int32_t dc, int32_t lo, int32_t hi)
{
    constexpr int K1 = NC >= 3 ? 1 : 0, K2 = NC >= 3 ? 2 : 0;
    if (NC >= 3 && mct) {
        if (!irrev) {
            const int32_t yy = c[0], u = c[K1], v = c[K2];
            const int32_t g = yy - ((u + v) >> 2);
            c[0] = v + g; c[K1] = g; c[K2] = u + g;
        } else {
            const float yy = __int_as_float(c[0]), u = __int_as_float(c[K1]), v = __int_as_float(c[K2]);
            const float r = __fadd_rn(yy, __fmul_rn(v, 1.402f));
            const float g = __fsub_rn(__fsub_rn(yy, __fmul_rn(u, 0.34413f)), __fmul_rn(v, 0.71414f));
            const float b = __fadd_rn(yy, __fmul_rn(u, 1.772f));
            c[0] = cvt_rn(r); c[K1] = cvt_rn(g); c[K2] = cvt_rn(b);
        }
#pragma unroll
        for (int k = 3; k < NC; ++k) if (irrev) c[k] = cvt_rn(__int_as_float(c[k]));
    } else if (irrev) {
#pragma unroll
        for (int k = 0; k < NC; ++k) c[k] = cvt_rn(__int_as_float(c[k]));
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) c[k] = min(max(c[k] + dc, lo), hi);
}

// PXO = 0: the level writes an int32/float plane (every level but the last, and the stage entry point).
// PXO = 1 / 2: the LAST level fused with K7 -- NC (= 3 with MCT) components are synthesised side by side and the
//   finished rows leave as 8- / 16-bit pixels after the inverse colour transform, DC shift and clamp, so the
//   int32 image planes (4 B/sample written + read back by K7) never exist.
// H16 (reversible only): ll / mallat / out hold int16 coefficients -- half the bytes this HBM-bound kernel moves; a synthesised
//   value that does not fit (only a stream no 8-bit image produces) raises bit 3 of *a.status and the decode is reported
//   as out of range instead of returning other pixels.  "Fit" is the packed kernels' input range when the chain runs its
//   larger levels packed (a.pk, set for the whole chain: a level too small or too odd for idwt53_pk_kernel comes here and
//   may feed a packed one -- ADVICE r2).
template <bool F97, int NC, int PXO, bool H16 = false>
__global__ __launch_bounds__(kThreads) void idwt_level_kernel(IdwtLevelArgs a)
{
    static_assert(!(F97 && H16), "16-bit planes are for the reversible transform");
    using T  = typename std::conditional<F97, float, int32_t>::type;
    using PT = typename std::conditional<H16, int16_t, T>::type;             // element type of the planes in memory
    using T2 = typename std::conditional<F97, float2, int2>::type;
    using PIX = typename std::conditional<PXO == 2, uint16_t, uint8_t>::type;
    static_assert(PXO != 0 || NC == 1, "plane output is one component per workgroup");
    // [parity][comp][row: low/high][half: s/d][local pair index]
    __shared__ __attribute__((aligned(16))) T line[2][NC][2][2][kThreads];

    const uint32_t t = threadIdx.x;
    // XCD-aware workgroup order, as in kernels_dwt.hip: the neighbours of a strip share its halo cache lines, so XCD k
    // (= dispatch id mod 8) takes a contiguous run of the linear (strip fastest) order
    uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd) {
        const uint32_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint32_t id = bx + gx * (by + gy * bz);
        const uint32_t q = total >> 3, r = total & 7u, k = id & 7u;
        const uint32_t lid = k * q + min(k, r) + (id >> 3);
        bx = lid % gx; by = (lid / gx) % gy; bz = lid / (gx * gy);
    }
    const uint32_t cw = a.cw, ch = a.ch;
    const uint32_t px = a.px, py = a.py;
    const uint32_t sw = (cw + 1 - px) >> 1, sh = (ch + 1 - py) >> 1;        // low-pass columns / rows
    const uint32_t vpairs = (ch + py + 1) >> 1;                               // row pairs on the coordinate grid

    uint32_t plane0 = bz;
    if constexpr (PXO != 0) plane0 = (bz / a.zdiv) * a.ncomp + a.comp0 + (bz % a.zdiv);
    const PT* ll = reinterpret_cast<const PT*>(a.ll) + (size_t)plane0 * a.ll_pitch;
    const PT* mp = reinterpret_cast<const PT*>(a.mallat) + (size_t)plane0 * a.m_pitch;
    PT* out = reinterpret_cast<PT*>(a.out) + (size_t)plane0 * a.out_pitch;
    uint32_t range = 0;
    // what an LL written as int16 has to stay inside: the 16 bits, or -- when packed levels follow in this chain (a.pk) -- the
    // packed kernels' input range (pk16.h); the bound is a power of two, so OR-ing the biased values finds any that is outside
    const int32_t rbias = a.pk ? kPkDecodeBound + 1 : 32768;
    // pixel output: the window [wx0, wx1) x [wy0, wy1) of the tile (the whole tile unless a region is decoded), tight
    const uint32_t win_w = a.wx1 - a.wx0;
    const size_t comp_px = (size_t)win_w * (a.wy1 - a.wy0);
    PIX* pix = reinterpret_cast<PIX*>(a.pixels) + (size_t)plane0 * comp_px;

    // The pair this lane loads and (lanes [0, kOutPairs)) synthesises: the strip's own pairs in lane order, so that a
    // wave's loads and stores start on cache-line boundaries; the next 2 * kHaloPairs lanes fetch the halo pairs left
    // and right of the strip.  lp = position in the staged line.
    const uint32_t lp = t < (uint32_t)kOutPairs ? t + kHaloPairs : (t < (uint32_t)(kOutPairs + kHaloPairs) ? t - kOutPairs : t);
    // (lanes past the halo pairs ride along: they repeat the last halo lane's loads)
    const int32_t J = (int32_t)((bx + a.strip0) * kOutPairs) - kHaloPairs + (int32_t)min(lp, (uint32_t)(kOutPairs + 2 * kHaloPairs - 1));
    const bool h_lane = t < (uint32_t)kOutPairs;
    const int32_t cE = 2 * J - (int32_t)px, cO = cE + 1;                 // the samples (columns of the level) this lane makes
    // Interior strips of an even-sized level on an even origin, the whole tile wanted, take the FAST instance: no column
    // mirroring, no per-lane store predicates, no window tests -- the last level is as VALU-bound as it is HBM-bound
    // (SQ counters: 286 VALU per row pair and wave against the forward level 0's 184).
    auto body = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    // horizontal mirror in the interleaved domain: low sample 2J, high sample 2J+1
    // (coordinate c holds sample c - px; a mirrored coordinate keeps its parity)
    const uint32_t js = FAST ? (uint32_t)J : (sw ? ((mirror_idx(2 * J - (int32_t)px, cw) + px) >> 1) - px : 0);
    const uint32_t jd = FAST ? (uint32_t)J : (cw > sw ? (mirror_idx(2 * J + 1 - (int32_t)px, cw) + px - 1) >> 1 : 0);
    const bool st_e = h_lane && (FAST || (cE >= 0 && (uint32_t)cE < cw)), st_o = h_lane && (FAST || (cO >= 0 && (uint32_t)cO < cw));

    const int32_t I0 = (int32_t)((by + a.seg0) * a.seg_pairs);
    const int32_t I1 = min((int32_t)vpairs, I0 + (int32_t)a.seg_pairs);
    // rows [2*I0, 2*I1) are this workgroup's; the recurrences lag behind the input by `lag` pairs
    constexpr int lag  = F97 ? 2 : 1;       // rows 2i and 2i+1 are complete after step i + lag
    constexpr int warm = F97 ? 2 : 1;       // steps before I0 whose outputs are discarded

    struct Raw { T ls[NC], ld[NC], hs[NC], hd[NC]; };
    auto fetch = [&](int32_t i, Raw& q) {
        // vertical mirror in the interleaved domain: low row 2i, high row 2i+1
        const uint32_t is = sh ? ((mirror_row(2 * i - (int32_t)py, ch) + py) >> 1) - py : 0;
        const uint32_t id = ch > sh ? (mirror_row(2 * i + 1 - (int32_t)py, ch) + py - 1) >> 1 : 0;
        const bool lc = FAST || sw > 0, hc = FAST || cw > sw, lr = FAST || sh > 0, hr = FAST || ch > sh;    // which halves exist (single row / column)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const PT* llk = ll + (size_t)k * a.ll_pitch;
            const PT* mpk = mp + (size_t)k * a.m_pitch;
            q.ls[k] = (lr && lc) ? (T)llk[(size_t)is * a.ll_stride + js] : T(0);
            q.ld[k] = (lr && hc) ? (T)mpk[(size_t)is * a.m_stride + sw + jd] : T(0);
            q.hs[k] = (hr && lc) ? (T)mpk[(size_t)(sh + id) * a.m_stride + js] : T(0);
            q.hd[k] = (hr && hc) ? (T)mpk[(size_t)(sh + id) * a.m_stride + sw + jd] : T(0);
        }
    };
    // one finished row of this lane's two columns leaves the kernel: as a plane row, or as pixels
    const bool px_vec = FAST || ((win_w | a.wx0 | px) & 1u) == 0;   // tightly packed pixel rows: pairs are aligned only for even widths / origins
    const bool in_e = FAST || (cE >= (int32_t)a.wx0 && cE < (int32_t)a.wx1);
    const bool in_o = FAST || (cO >= (int32_t)a.wx0 && cO < (int32_t)a.wx1);
    auto emit = [&](int32_t r, const T (&vA)[NC], const T (&vB)[NC]) {
        if constexpr (PXO == 0) {
            PT* row = out + (size_t)r * a.out_stride + cE;
            if constexpr (H16) {
                if (st_e && st_o && !px) *reinterpret_cast<uint32_t*>(row) = ((uint32_t)vA[0] & 0xFFFFu) | ((uint32_t)vB[0] << 16);
                else {
                    if (st_e) row[0] = (int16_t)vA[0];
                    if (st_o) row[1] = (int16_t)vB[0];
                }
                range |= (st_e ? (uint32_t)(vA[0] + rbias) : 0u) | (st_o ? (uint32_t)(vB[0] + rbias) : 0u);
            } else if (st_e && st_o && !px) { T2 v; v.x = vA[0]; v.y = vB[0]; *reinterpret_cast<T2*>(row) = v; }
            else {
                if (st_e) row[0] = vA[0];
                if (st_o) row[1] = vB[0];
            }
        } else {
            int32_t cA[NC], cB[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if constexpr (F97) { cA[k] = __float_as_int(vA[k]); cB[k] = __float_as_int(vB[k]); }
                else               { cA[k] = vA[k]; cB[k] = vB[k]; }
            }
            egress_px<NC>(cA, F97, a.mct != 0, a.dc, a.lo, a.hi);
            egress_px<NC>(cB, F97, a.mct != 0, a.dc, a.lo, a.hi);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                if (!FAST && ((uint32_t)r < a.wy0 || (uint32_t)r >= a.wy1)) continue;
                PIX* row = FAST ? pix + (size_t)k * comp_px + (size_t)r * cw + cE
                                : pix + (size_t)k * comp_px + (size_t)((uint32_t)r - a.wy0) * win_w + (cE - (int32_t)a.wx0);
                if (st_o && px_vec && in_e && in_o) {
                    if constexpr (PXO == 1) *reinterpret_cast<uchar2*>(row) = make_uchar2((uint8_t)cA[k], (uint8_t)cB[k]);
                    else                    *reinterpret_cast<ushort2*>(row) = make_ushort2((uint16_t)cA[k], (uint16_t)cB[k]);
                } else {
                    if (st_e && in_e) row[0] = (PIX)cA[k];
                    if (st_o && in_o) row[1] = (PIX)cB[k];
                }
            }
        }
    };

    typename std::conditional<F97, IV97, IV53>::type colA[NC], colB[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { colA[k].init(); colB[k].init(); }
    Raw cur, nxt;
    int32_t i = I0 - warm;
    fetch(i, cur);
    const int32_t i_end = I1 - 1 + lag;
    for (int par = 0; i <= i_end; ++i, par ^= 1) {
        if (i < i_end) fetch(i + 1, nxt);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            line[par][k][0][0][lp] = cur.ls[k]; line[par][k][0][1][lp] = cur.ld[k];
            line[par][k][1][0][lp] = cur.hs[k]; line[par][k][1][1][lp] = cur.hd[k];
        }
        __syncthreads();
        if (h_lane) {
            T oA[NC], eA[NC], oB[NC], eB[NC];                 // finished rows (odd, even) of columns 2J and 2J+1
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                T se, so, de, dodd;                           // low row / high row, even / odd column
                if (!FAST && cw == 1) {
                    if (px) {          // a lone high-pass column: coordinate 1 of pair 0
                        const T v0 = line[par][k][0][1][lp], v1 = line[par][k][1][1][lp];
                        se = 0; de = 0;
                        if constexpr (F97) { so = v0; dodd = v1; } else { so = v0 / 2; dodd = v1 / 2; }
                    } else { se = line[par][k][0][0][lp]; so = 0; de = line[par][k][1][0][lp]; dodd = 0; }
                }
                else if constexpr (F97) {
                    hs97(&line[par][k][0][0][lp], &line[par][k][0][1][lp], se, so);
                    hs97(&line[par][k][1][0][lp], &line[par][k][1][1][lp], de, dodd);
                } else {
                    hs53(&line[par][k][0][0][lp], &line[par][k][0][1][lp], se, so);
                    hs53(&line[par][k][1][0][lp], &line[par][k][1][1][lp], de, dodd);
                }
                if (!FAST && ch == 1) {
                    if (py) {          // a lone high-pass row
                        if constexpr (F97) { eA[k] = de; eB[k] = dodd; } else { eA[k] = de >> 1; eB[k] = dodd >> 1; }
                    } else { eA[k] = se; eB[k] = so; }
                    oA[k] = oB[k] = 0;
                }
                else { colA[k].step(se, de, oA[k], eA[k]); colB[k].step(so, dodd, oB[k], eB[k]); }
            }
            // which output rows these are: coordinates, then rows of the level (coordinate - py)
            const int32_t r_even = (!FAST && ch == 1) ? (int32_t)py : 2 * (i - (lag - 1));
            const int32_t r_odd = r_even - 1;              // 5/3: 2i-1 ; 9/7: 2(i-2)+1
            const int32_t pyi = FAST ? 0 : (int32_t)py;
            const bool ok_e = (!FAST && ch == 1) ? (i == 0) : (r_even >= 2 * I0 && r_even < 2 * I1 && r_even >= pyi && (uint32_t)(r_even - pyi) < ch);
            const bool ok_o = (FAST || ch > 1) && r_odd >= 2 * I0 && r_odd < 2 * I1 && r_odd >= pyi && (uint32_t)(r_odd - pyi) < ch;
            if (ok_e) emit(r_even - pyi, eA, eB);
            if (ok_o) emit(r_odd - pyi, oA, oB);
        }
        cur = nxt;
    }
    };
    {
        // every pair any lane touches lies inside both half-bands (halo and ride-along lanes included)
        const int32_t Jlo = (int32_t)((bx + a.strip0) * kOutPairs) - kHaloPairs, Jhi = Jlo + kOutPairs + 2 * kHaloPairs - 1;
        bool fast = (px | py) == 0 && (cw & 1u) == 0 && ch >= 16 && Jlo >= 0 && (uint32_t)Jhi < (cw >> 1);
        if constexpr (PXO != 0) fast = fast && a.wx0 == 0 && a.wy0 == 0 && a.wx1 == cw && a.wy1 == ch;
        if (fast) body(std::true_type{}); else body(std::false_type{});
    }
    if constexpr (H16 && PXO == 0) {
        if (range > 2u * (uint32_t)rbias - 1u) atomicOr(a.status, 8u);
    }
}

// ---- the inverse 5/3 level on packed int16 pairs, two coefficient pairs (four output columns) per lane ---------------
// The mirror image of dwt53_pk_kernel (kernels_dwt.hip), for the same reasons: dword accesses instead of 2-byte ones (the
// texture addresser takes four lanes a clock whatever the width), a third of the vector instructions, addresses from
// buffer descriptors + scalar row offsets, rows two steps ahead.  For levels the launcher knows to be reversible with
// 16-bit planes, on the origin, whole (no window), of even height >= 16 and a width that is a multiple of 4; PXO = 1: the
// last level, 8-bit unsigned pixels out.
//   per row pair i a lane loads LL / HL (low row) and LH / HH (high row) of its two pairs J, J + 1 -- four dwords per
//   component --, regroups them per pair as S = (LL | LH << 16), D = (HL | HH << 16), so that ONE horizontal synthesis
//   does the low and the high row, regroups the four columns it made as (two columns of the low row), (of the high row)
//   for the vertical recurrence on packed pairs, and stores finished rows four columns at a time.
// Range: every sum stays inside 16 bits as long as the level's inputs lie within +-kPkDecodeBound (pk16.h) -- the block
// decoder's flag vouches for the coefficients, and an intermediate level checks the LL it writes (status bit 3 otherwise;
// the decode is then done again in 32 bits, context.hip decode_impl).
struct IV53pk {       // yields rows 2i-1 and 2i
    pk16 dprev, xprev;
    __device__ __forceinline__ void init() { dprev = (pk16)(0); xprev = (pk16)(0); }
    __device__ __forceinline__ void step(pk16 s, pk16 d, pk16& r_odd, pk16& r_even)
    {
        r_even = s - ((dprev + d + (pk16)(2)) >> 2);
        r_odd = dprev + ((xprev + r_even) >> 1);
        dprev = d; xprev = r_even;
    }
};
__device__ __forceinline__ uint32_t sat_pk_u8(pk16 v)      // two int16 -> two bytes, each clamped to [0, 255]
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(as_u32(v)));
    return r;
}

constexpr int kPkLanePairs = 2;
constexpr int kPkPairs     = kThreads * kPkLanePairs;      // pairs staged per line: 512
constexpr int kPkOutCols   = 960;                          // output columns per strip at most (240 lanes; one halo lane each side)
static_assert(kPkOutCols / 4 + 2 <= kThreads, "strip does not fit the staged line");
// the strips of a level share its width evenly, in multiples of 64 output columns
__host__ __device__ inline uint32_t ipk_strip_cols(uint32_t cw)
{
    const uint32_t n = (cw + kPkOutCols - 1) / kPkOutCols;
    return min((uint32_t)kPkOutCols, ((cw + n - 1) / n + 63u) & ~63u);
}

#define IPK_FENCE __builtin_amdgcn_sched_barrier(0)
template <int NC, int PXO>
__global__ __launch_bounds__(kThreads) void idwt53_pk_kernel(IdwtLevelArgs a)
{
    static_assert(PXO == 0 || PXO == 1, "int16 planes or 8-bit pixels");
    static_assert(PXO != 0 || NC == 1, "plane output is one component per workgroup");
    // [parity][comp][pair][S, D]: S = LL | LH << 16, D = HL | HH << 16
    __shared__ __attribute__((aligned(16))) uint32_t line[2][NC][kPkPairs * 2];

    const uint32_t t = threadIdx.x;
    uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd) {                                           // XCD k takes a contiguous run of the strip-fastest order (see above)
        const uint32_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint32_t id = bx + gx * (by + gy * bz);
        const uint32_t q = total >> 3, r = total & 7u, k = id & 7u;
        const uint32_t lid = k * q + min(k, r) + (id >> 3);
        bx = lid % gx; by = (lid / gx) % gy; bz = lid / (gx * gy);
    }
    const uint32_t cw = a.cw, ch = a.ch;
    const uint32_t sw = cw >> 1, sh = ch >> 1;
    const uint32_t scols = ipk_strip_cols(cw), slanes = scols / 4;

    uint32_t plane0 = bz;
    if constexpr (PXO != 0) plane0 = (bz / a.zdiv) * a.ncomp + a.comp0 + (bz % a.zdiv);
    const int16_t* ll = reinterpret_cast<const int16_t*>(a.ll) + (size_t)plane0 * a.ll_pitch;
    const int16_t* mp = reinterpret_cast<const int16_t*>(a.mallat) + (size_t)plane0 * a.m_pitch;
    int16_t* out = reinterpret_cast<int16_t*>(a.out) + (size_t)plane0 * a.out_pitch;
    const size_t comp_px = (size_t)cw * ch;
    uint8_t* pix = reinterpret_cast<uint8_t*>(a.pixels) + (size_t)plane0 * comp_px;

    // Lane t < slanes carries group t + 1 of the staged line (groups of two pairs; group 0 and group slanes + 1 are the halo,
    // on the two lanes after; the rest repeat lane 0).
    const uint32_t grp = t < slanes ? t + 1 : (t == slanes ? 0u : (t == slanes + 1 ? t : 1u));
    const int32_t J = (int32_t)(bx * (scols / 2)) - kPkLanePairs + (int32_t)(grp * kPkLanePairs);     // first of its two pairs
    // A group outside the level is a mirrored one, in the interleaved domain (low sample 2J, high sample 2J + 1): pairs
    // (-2, -1) hold s[2], s[1] and d[1], d[0]; pairs (sw, sw + 1) hold s[sw-1], s[sw-2] and d[sw-2], d[sw-3] -- two consecutive
    // coefficients read backwards, from different places for the two halves.  (Only the nearest pair of a halo group is
    // used; groups further out load from inside the row.)
    const bool rev = J < 0 || J >= (int32_t)sw;
    int32_t js = J < 0 ? -J - 1 : (J >= (int32_t)sw ? 2 * (int32_t)sw - 2 - J : J);          // lowest s index of the two
    int32_t jd = J < 0 ? -J - 2 : (J >= (int32_t)sw ? 2 * (int32_t)sw - 3 - J : J);          // lowest d index
    js = max(0, min(js, (int32_t)sw - 2)); jd = max(0, min(jd, (int32_t)sw - 2));
    const uint32_t off_s = (uint32_t)js * 2u, off_d = (uint32_t)jd * 2u;                     // bytes into a half-row
    // regrouping selectors (v_perm over hi : lo): pair J takes the low halves of the two loads, pair J + 1 the high halves --
    // the other way round for a mirrored group
    const uint32_t sel0 = rev ? kSelHiHi : kSelLoLo, sel1 = rev ? kSelLoLo : kSelHiHi;

    __amdgpu_buffer_rsrc_t r_ll[NC], r_mp[NC], r_out[NC];
    #pragma unroll
    for (int k = 0; k < NC; ++k) {
        r_ll[k] = buffer_from(ll + (size_t)k * a.ll_pitch);
        r_mp[k] = buffer_from(mp + (size_t)k * a.m_pitch);
        r_out[k] = PXO == 0 ? buffer_from(out) : buffer_from(pix + (size_t)k * comp_px);
    }
    __amdgpu_buffer_rsrc_t r_none = buffer_from(a.mallat, true);

    const int32_t I0 = (int32_t)(by * a.seg_pairs);
    const int32_t I1 = min((int32_t)sh, I0 + (int32_t)a.seg_pairs);

    struct Raw { uint32_t ls[NC], ld[NC], hs[NC], hd[NC]; };
    auto fetch = [&](int32_t i, Raw& q) {                    // row pair i: low row 2i, high row 2i + 1 (mirrored past the ends)
        // (ch >= 16 and the rows leave [0, ch) by a few samples at most: one reflection, no division, no branch)
        auto mirror = [&](int32_t r) { r = r < 0 ? -r : r; return (uint32_t)(r < (int32_t)ch ? r : 2 * ((int32_t)ch - 1) - r); };
        const uint32_t is = mirror(2 * i) >> 1, id = (mirror(2 * i + 1) - 1u) >> 1;
        const uint32_t lrow = is * a.ll_stride * 2u, mlo = is * a.m_stride * 2u + sw * 2u, mhi = (sh + id) * a.m_stride * 2u;
    #pragma unroll
        for (int k = 0; k < NC; ++k) {
            q.ls[k] = __builtin_amdgcn_raw_buffer_load_b32(r_ll[k], off_s, lrow, 0);
            q.ld[k] = __builtin_amdgcn_raw_buffer_load_b32(r_mp[k], off_d, mlo, 0);
            q.hs[k] = __builtin_amdgcn_raw_buffer_load_b32(r_mp[k], off_s, mhi, 0);
            q.hd[k] = __builtin_amdgcn_raw_buffer_load_b32(r_mp[k], off_d, mhi + sw * 2u, 0);
        }
    };

    // horizontal phase: lane t synthesises the four columns of group t + 1; lanes past the strip's last group (halo, idle,
    // beyond the level in the last strip) repeat the last one -- same reads, same values to the same addresses
    const uint32_t nv = min(slanes, (cw - bx * scols + 3u) >> 2);
    const uint32_t tp = min(t, nv - 1u);
    const uint32_t hp = (tp + 1u) * kPkLanePairs;            // its first pair in the staged line
    const uint32_t oc = (bx * scols + tp * 4u) * (PXO == 0 ? 2u : 1u);      // byte offset of its four samples in an output row

    IV53pk colA[NC], colB[NC];
    #pragma unroll
    for (int k = 0; k < NC; ++k) { colA[k].init(); colB[k].init(); }
    uint32_t range = 0;
    struct Rows { pk16 oA[NC], oB[NC], eA[NC], eB[NC]; };    // finished rows 2i - 1 (o) and 2i (e): columns (0, 1) and (2, 3)

    struct Grouped { uint4 w[NC]; };
    auto regroup = [&](const Raw& q, Grouped& g) {           // (the first touch of a fetched row pair: before its registers are refilled)
    #pragma unroll
        for (int k = 0; k < NC; ++k) {
            g.w[k].x = __builtin_amdgcn_perm(q.hs[k], q.ls[k], sel0);     // S of pair J
            g.w[k].y = __builtin_amdgcn_perm(q.hd[k], q.ld[k], sel0);     // D of pair J
            g.w[k].z = __builtin_amdgcn_perm(q.hs[k], q.ls[k], sel1);     // S of pair J + 1
            g.w[k].w = __builtin_amdgcn_perm(q.hd[k], q.ld[k], sel1);     // D
        }
    };
    auto step = [&](int par, const Grouped& g, Rows& f) {
    #pragma unroll
        for (int k = 0; k < NC; ++k) *reinterpret_cast<uint4*>(&line[par][k][grp * kPkLanePairs * 2]) = g.w[k];
        __syncthreads();
    #pragma unroll
        for (int k = 0; k < NC; ++k) {
            const uint32_t* w = static_cast<const uint32_t*>(__builtin_assume_aligned(&line[par][k][hp * 2], 16));
            const pk16 dm = as_pk(w[-1]);
            const uint4 c = *reinterpret_cast<const uint4*>(w);
            const uint2 n = *reinterpret_cast<const uint2*>(w + 4);
            const pk16 s0 = as_pk(c.x), d0 = as_pk(c.y), s1 = as_pk(c.z), d1 = as_pk(c.w), s2 = as_pk(n.x), d2 = as_pk(n.y);
            const pk16 two = (pk16)(2);
            const pk16 e0 = s0 - ((dm + d0 + two) >> 2);
            const pk16 e1 = s1 - ((d0 + d1 + two) >> 2);
            const pk16 e2 = s2 - ((d1 + d2 + two) >> 2);
            const pk16 o0 = d0 + ((e0 + e1) >> 1);
            const pk16 o1 = d1 + ((e1 + e2) >> 1);
            // columns 0..3 = e0, o0, e1, o1, each (low row | high row << 16): regroup as two columns of one row
            const pk16 lA = as_pk(__builtin_amdgcn_perm(as_u32(o0), as_u32(e0), kSelLoLo)), hA = as_pk(__builtin_amdgcn_perm(as_u32(o0), as_u32(e0), kSelHiHi));
            const pk16 lB = as_pk(__builtin_amdgcn_perm(as_u32(o1), as_u32(e1), kSelLoLo)), hB = as_pk(__builtin_amdgcn_perm(as_u32(o1), as_u32(e1), kSelHiHi));
            colA[k].step(lA, hA, f.oA[k], f.eA[k]);
            colB[k].step(lB, hB, f.oB[k], f.eB[k]);
        }
    };
    // one finished row (four columns of it per lane) leaves the kernel; `voff` = oc, or all ones for a row that is not this
    // workgroup's to write (beyond any buffer: dropped)
    auto emit = [&](int32_t r, const pk16 (&vA)[NC], const pk16 (&vB)[NC], uint32_t voff) {
        const uint32_t ru = __builtin_amdgcn_readfirstlane((uint32_t)r);
        if constexpr (PXO == 0) {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{as_u32(vA[0]), as_u32(vB[0])}, r_out[0], voff, ru * a.out_stride * 2u, 0);
            // what the next level reads has to lie inside its range as well
            const pk16 bias = (pk16)((short)(kPkDecodeBound + 1));
            range |= (as_u32(vA[0] + bias) | as_u32(vB[0] + bias)) & (voff == 0xFFFFFFFFu ? 0u : 0xFFFFFFFFu);
        } else {
            pk16 cA[NC], cB[NC];
            if constexpr (NC == 3) {           // inverse RCT (mct.cpp:454-464): G = Y - ((U + V) >> 2), R = V + G, B = U + G
                const pk16 gA = vA[0] - ((vA[1] + vA[2]) >> 2), gB = vB[0] - ((vB[1] + vB[2]) >> 2);
                cA[0] = vA[2] + gA; cA[1] = gA; cA[2] = vA[1] + gA;
                cB[0] = vB[2] + gB; cB[1] = gB; cB[2] = vB[1] + gB;
            } else { cA[0] = vA[0]; cB[0] = vB[0]; }
            const pk16 dc2 = as_pk((uint32_t)a.dc * 0x10001u);
    #pragma unroll
            for (int k = 0; k < NC; ++k) {
                const uint32_t four = __builtin_amdgcn_perm(sat_pk_u8(cB[k] + dc2), sat_pk_u8(cA[k] + dc2), kSelLoLo);
                __builtin_amdgcn_raw_buffer_store_b32(four, r_out[k], voff, ru * cw, 0);
            }
        }
    };
    constexpr int kStores = PXO == 0 ? 2 : 2 * NC;           // memory stores per step
    auto no_stores = [&]() {
    #pragma unroll
        for (int k = 0; k < kStores; ++k) {
            if constexpr (PXO == 0) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, r_none, oc + 64 * k, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(0u, r_none, oc + 64 * k, 0, 0);          // (apart, or they merge into wider ones)
        }
    };
    // Step i consumes row pair i and finishes rows 2i - 1 and 2i; steps I0 - 1 .. I1, the first a warm-up.  Row pairs are fetched
    // two steps ahead into two alternating sets; the rows a step finishes are stored during the next one, after its
    // fetches -- every memory operation of a step in one place.  As in dwt53_pk_kernel the two steps before the loop issue
    // the same sequence as every later pair (fetches, stores, fetches, stores) with their stores dropped, so that the
    // compiler's wait at the loop head allows a full two steps of operations in flight.
    Raw ra, rb;
    fetch(I0 - 1, ra); fetch(I0, rb);
    Rows f;
    const uint32_t none = 0xFFFFFFFFu;
    int32_t i = I0 - 1;
    auto run = [&](Raw& q, int par, bool first) {            // one step: regroup + synthesise row pair i, refill q, store the rows before
        Grouped g;
        regroup(q, g);
        // (the regrouping stays HERE -- the optimiser would sink it to the exchange below, past the refill, and the old rows
        //  would have to be copied out of the refill's way)
    #pragma unroll
        for (int k = 0; k < NC; ++k) asm volatile("" : "+v"(g.w[k].x), "+v"(g.w[k].y), "+v"(g.w[k].z), "+v"(g.w[k].w));
        IPK_FENCE;
        fetch(i + 2, q);
        if (first) no_stores();
        else {
            const int32_t ip = i - 1;                        // the step whose rows are stored now: rows 2ip - 1, 2ip
            emit(2 * ip - 1, f.oA, f.oB, ip > I0 ? oc : none);                  // (row 2 I0 - 1 is the segment above's)
            emit(2 * ip, f.eA, f.eB, ip < I1 ? oc : none);                      // (row 2 I1 the segment below's)
        }
        step(par, g, f);
        ++i;
    };
    run(ra, 0, true);
    IPK_FENCE;
    run(rb, 1, true);
    IPK_FENCE;
    while (i + 1 <= I1) {
        run(ra, 0, false);
        IPK_FENCE;
        run(rb, 1, false);
        IPK_FENCE;
    }
    if (i <= I1) run(ra, 0, false);
    {
        const int32_t ip = i - 1;
        emit(2 * ip - 1, f.oA, f.oB, ip > I0 ? oc : none);
        emit(2 * ip, f.eA, f.eB, ip < I1 ? oc : none);
    }
    if constexpr (PXO == 0) {
        if (__builtin_amdgcn_ballot_w64((range & 0xF000F000u) != 0) != 0 && (t & 63u) == 0) atomicOr(a.status, 8u);
    }
}

// ---- K7 egress, stand-alone (stage entry point; pixel sizes the fused last level does not cover) ----------
template <typename PIX, int NC>
__global__ __launch_bounds__(256) void egress_kernel(EgressArgs a)
{
    const uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 4u;
    const uint32_t y = blockIdx.y;
    const uint32_t tile = blockIdx.z;
    if (x >= a.w) return;
    const uint32_t n = a.w - x < 4 ? a.w - x : 4;
    const size_t comp_px = (size_t)a.w * a.h;
    const int32_t* src = a.planes + (size_t)tile * a.ncomp * a.pitch + (size_t)y * a.stride + x;
    PIX* dst = reinterpret_cast<PIX*>(a.pixels) + (size_t)tile * a.ncomp * comp_px + (size_t)y * a.w + x;
    const bool irrev = a.irreversible != 0;

    int32_t c[NC][4];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int4 v = *reinterpret_cast<const int4*>(src + (size_t)k * a.pitch);     // stride % 32 == 0: in-row padding is readable
        c[k][0] = v.x; c[k][1] = v.y; c[k][2] = v.z; c[k][3] = v.w;
    }
    constexpr int K1 = NC >= 3 ? 1 : 0, K2 = NC >= 3 ? 2 : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (NC >= 3 && a.mct) {
            if (!irrev) {                                   // inverse RCT (mct.cpp:454-464)
                const int32_t yy = c[0][i], u = c[K1][i], v = c[K2][i];
                const int32_t g = yy - ((u + v) >> 2);
                c[0][i] = v + g; c[K1][i] = g; c[K2][i] = u + g;
            } else {                                        // inverse ICT (mct.cpp:278-289), round to nearest even
                const float yy = __int_as_float(c[0][i]), u = __int_as_float(c[K1][i]), v = __int_as_float(c[K2][i]);
                const float r = __fadd_rn(yy, __fmul_rn(v, 1.402f));
                const float g = __fsub_rn(__fsub_rn(yy, __fmul_rn(u, 0.34413f)), __fmul_rn(v, 0.71414f));
                const float b = __fadd_rn(yy, __fmul_rn(u, 1.772f));
                c[0][i] = cvt_rn(r); c[K1][i] = cvt_rn(g); c[K2][i] = cvt_rn(b);
            }
#pragma unroll
            for (int k = 3; k < NC; ++k) if (irrev) c[k][i] = cvt_rn(__int_as_float(c[k][i]));
        } else if (irrev) {
#pragma unroll
            for (int k = 0; k < NC; ++k) c[k][i] = cvt_rn(__int_as_float(c[k][i]));
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) c[k][i] = min(max(c[k][i] + a.dc, a.lo), a.hi);
    }
    const bool vec = (n == 4) && ((a.w & 3u) == 0);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        PIX* d = dst + (size_t)k * comp_px;
        if (vec) {
            if constexpr (sizeof(PIX) == 1) *reinterpret_cast<uchar4*>(d) = make_uchar4((uint8_t)c[k][0], (uint8_t)c[k][1], (uint8_t)c[k][2], (uint8_t)c[k][3]);
            else if constexpr (sizeof(PIX) == 2) *reinterpret_cast<ushort4*>(d) = make_ushort4((uint16_t)c[k][0], (uint16_t)c[k][1], (uint16_t)c[k][2], (uint16_t)c[k][3]);
            else *reinterpret_cast<int4*>(d) = make_int4(c[k][0], c[k][1], c[k][2], c[k][3]);
        } else {
            for (uint32_t i = 0; i < n; ++i) d[i] = (PIX)c[k][i];
        }
    }
}

} // namespace

uint32_t idwt_strip_pairs() { return kOutPairs; }

// the level shape idwt53_pk_kernel takes (a.pk: the launcher's word that the inputs are inside the packed range)
static bool idwt_level_is_pk(const IdwtLevelArgs& a)
{
    // (row offsets are 32-bit byte offsets from a plane's first sample: planes of 2^31 samples and more keep the flat addressing)
    const bool near = (uint64_t)a.m_stride * a.ch < (1ull << 31) && (uint64_t)a.out_stride * a.ch < (1ull << 31);
    return a.h16 && a.pk && !a.irreversible && (a.px | a.py) == 0 && (a.cw & 3u) == 0 && a.cw >= 256u && a.ch >= 16 && (a.ch & 1u) == 0 &&
           a.nstrips == 0 && a.nsegs == 0 && near;
}
uint32_t idwt_level_strip_pairs(const IdwtLevelArgs& a) { return idwt_level_is_pk(a) ? ipk_strip_cols(a.cw) / 2 : (uint32_t)kOutPairs; }

hipError_t launch_idwt_level(const IdwtLevelArgs& a, hipStream_t s)
{
    const uint32_t sw = (a.cw + a.px + 1) >> 1, sh = (a.ch + a.py + 1) >> 1;      // pairs on the coordinate grid
    // strips x row segments: all of them, or the caller's sub-grid (region decode)
    dim3 grid(a.nstrips ? a.nstrips : (sw + kOutPairs - 1) / kOutPairs, a.nsegs ? a.nsegs : (sh + a.seg_pairs - 1) / a.seg_pairs, a.nplanes);
    dim3 block(kThreads);
    if (idwt_level_is_pk(a)) {
        grid.x = (a.cw + ipk_strip_cols(a.cw) - 1) / ipk_strip_cols(a.cw);
        hipLaunchKernelGGL((idwt53_pk_kernel<1, 0>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.irreversible)
        hipLaunchKernelGGL((idwt_level_kernel<true, 1, 0>), grid, block, 0, s, a);
    else if (a.h16)
        hipLaunchKernelGGL((idwt_level_kernel<false, 1, 0, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((idwt_level_kernel<false, 1, 0>), grid, block, 0, s, a);
    return hipGetLastError();
}

// The last level (cw x ch = tile size) straight to pixels (a0.pixels, px_bytes 1 or 2, dc/lo/hi, mct set by the caller);
// a0.nplanes is ignored: the grid covers ntiles x (MCT triple | every component on its own).
hipError_t launch_idwt_level0_fused(const IdwtLevelArgs& a0, uint32_t ntiles, uint32_t ncomp, hipStream_t s)
{
    const uint32_t sw = (a0.cw + a0.px + 1) >> 1, sh = (a0.ch + a0.py + 1) >> 1;
    dim3 block(kThreads);
    auto go = [&](uint32_t comp0, uint32_t zdiv, int nc) {
        IdwtLevelArgs a = a0;
        a.comp0 = comp0; a.zdiv = zdiv; a.ncomp = ncomp;
        dim3 grid(a.nstrips ? a.nstrips : (sw + kOutPairs - 1) / kOutPairs, a.nsegs ? a.nsegs : (sh + a.seg_pairs - 1) / a.seg_pairs, ntiles * zdiv);
#define GRK_I0(F97, NC, PX) hipLaunchKernelGGL((idwt_level_kernel<F97, NC, PX>), grid, block, 0, s, a)
        const int px = a.px_bytes == 1 ? 1 : 2;
        if (a.irreversible) {
            if (nc == 3) { if (px == 1) GRK_I0(true, 3, 1); else GRK_I0(true, 3, 2); }
            else         { if (px == 1) GRK_I0(true, 1, 1); else GRK_I0(true, 1, 2); }
        } else if (px == 1 && idwt_level_is_pk(a) && a.lo == 0 && a.hi == 255 && a.wx0 == 0 && a.wy0 == 0 && a.wx1 == a.cw && a.wy1 == a.ch &&
                   (nc == 1 || a.mct)) {
            grid.x = (a.cw + ipk_strip_cols(a.cw) - 1) / ipk_strip_cols(a.cw);
            if (nc == 3) hipLaunchKernelGGL((idwt53_pk_kernel<3, 1>), grid, block, 0, s, a);
            else         hipLaunchKernelGGL((idwt53_pk_kernel<1, 1>), grid, block, 0, s, a);
        } else if (a.h16) {
            if (nc == 3) { if (px == 1) hipLaunchKernelGGL((idwt_level_kernel<false, 3, 1, true>), grid, block, 0, s, a);
                           else         hipLaunchKernelGGL((idwt_level_kernel<false, 3, 2, true>), grid, block, 0, s, a); }
            else         { if (px == 1) hipLaunchKernelGGL((idwt_level_kernel<false, 1, 1, true>), grid, block, 0, s, a);
                           else         hipLaunchKernelGGL((idwt_level_kernel<false, 1, 2, true>), grid, block, 0, s, a); }
        } else {
            if (nc == 3) { if (px == 1) GRK_I0(false, 3, 1); else GRK_I0(false, 3, 2); }
            else         { if (px == 1) GRK_I0(false, 1, 1); else GRK_I0(false, 1, 2); }
        }
#undef GRK_I0
    };
    if (a0.mct && ncomp >= 3) {
        go(0, 1, 3);
        for (uint32_t k = 3; k < ncomp; ++k) go(k, 1, 1);    // components beyond the triple: no colour transform (NC = 1)
    } else {
        go(0, ncomp, 1);
    }
    return hipGetLastError();
}

hipError_t launch_egress(const EgressArgs& a, hipStream_t s)
{
    dim3 grid((a.w + 1023) / 1024, a.h, a.ntiles), block(256);
#define GRK_EGRESS(PIX)                                                                          \
    switch (a.ncomp) {                                                                          \
    case 1: hipLaunchKernelGGL((egress_kernel<PIX, 1>), grid, block, 0, s, a); break;           \
    case 2: hipLaunchKernelGGL((egress_kernel<PIX, 2>), grid, block, 0, s, a); break;           \
    case 3: hipLaunchKernelGGL((egress_kernel<PIX, 3>), grid, block, 0, s, a); break;           \
    default: hipLaunchKernelGGL((egress_kernel<PIX, 4>), grid, block, 0, s, a); break;          \
    }
    if (a.bytes_per_sample == 1) { GRK_EGRESS(uint8_t) }
    else if (a.bytes_per_sample == 2) { GRK_EGRESS(uint16_t) }
    else { GRK_EGRESS(int32_t) }
#undef GRK_EGRESS
    return hipGetLastError();
}

} // namespace grk_amd
