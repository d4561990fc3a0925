// grok_amd/csrc/kernels_dwt.hip -- K2: one forward DWT level (5/3 int32 or 9/7 fp32), gfx950.
//
// Replaces one iteration of WaveletFwdImpl::encode_procedure (transform/WaveletFwd.cpp:478-604):
// the reference runs a vertical pass over all columns and then a horizontal pass over all rows,
// in place, with a de-interleave copy each (>= 4 full sweeps of the level through memory).
// Here both passes are fused so every sample of the level is read once and written once
// (8 bytes / sample / level = the algorithmic figure of SURVEY.md §8d):
//
//  * a workgroup (256 threads) owns a column strip of 448 output columns (224 pairs = 7 cache
//    lines of every sub-band row, +4 halo columns each side) and a segment of `seg_pairs` output row
//    pairs; it streams down the rows;
//  * vertical lifting runs in registers as a recurrence per column (two columns per lane, the
//    strip's own pairs on lanes 0..223 in order so that row loads start on line boundaries); state
//    is 2 (5/3) or 4 (9/7) values, so there is no vertical halo re-read except the 2..5 warm-up rows
//    at a segment start;
//  * each finished pair of rows (one low, one high) is exchanged through a double-buffered LDS
//    line (4 KiB) and every lane produces one (low,high) output pair per line with the local
//    lifting stencil; outputs go straight to the LL ping-pong plane and to the HL/LH/HH slots of
//    the Mallat plane, whole aligned cache lines per wave;
//  * level 0 can read the caller's pixels itself (K1 fused: DC shift + RCT/ICT in registers, three
//    components side by side) and, for 8-bit reversible content, all planes hold int16 (H16);
//  * interior strips take a single-basic-block FAST instance whose memory operations are all issued
//    in one place per iteration (see the kernel).
//
// Image borders use whole-sample symmetric extension by index mirroring, which reproduces the
// reference's edge formulas exactly (the mirrored operands are the same numbers; A.3/A.4).
// A level whose origin lies on an ODD coordinate of its grid (a.px / a.py; tiles and images off the origin) starts with a
// high-pass sample (WaveletFwd.cpp:884-905, :782-842).  Lifting works on coordinates, so the kernel runs on the
// coordinate grid shifted by the parity: pair i = coordinates (2i, 2i + 1), sample k of the level sits at coordinate
// k + parity, and position 0 of an odd start is a phantom that mirrors into the level like any other outside sample; its
// low-pass output is dropped (low index = pair - parity).  A lone high-pass sample is doubled by the 5/3 (:885-887,
// :812-815) and left alone by the 9/7 (:924-926, :985-987).
// fp32 9/7: (l + r) * c and the accumulate are separately rounded (__fadd_rn/__fmul_rn, no FMA),
// the order of WaveletFwd.cpp:143-160; scaling low*invK, high*K as in :46, :203-213.
#include "kernels.h"
#include "pk16.h"
#include <cstdlib>
#include <type_traits>

namespace grk_amd {

namespace {

constexpr int   kThreads  = 256;
constexpr int   kCols     = 2 * kThreads;      // columns staged per line
constexpr int   kHalo     = 4;                 // columns each side
// Output columns per strip.  At most kCols - 2 * kHalo = 504; 448 makes a strip 224 output pairs = 7 x 128 bytes of
// every sub-band row, so each wave's stores cover whole, aligned cache lines.
constexpr int   kOutCols  = 448;
constexpr int   kOutPairs = kOutCols / 2;
static_assert(kOutCols <= kCols - 2 * kHalo && kOutCols % 2 == 0, "strip does not fit the staged line");

__device__ __forceinline__ uint32_t mirror_idx(int32_t i, uint32_t n)
{
    if (n == 1) return 0;
    const int32_t p = 2 * ((int32_t)n - 1);
    i %= p;
    if (i < 0) i += p;
    return (uint32_t)(i < (int32_t)n ? i : p - i);
}
// same for indices that leave [0, n) by fewer than 16 samples (the row loop): one reflection, no division
// (TALL: the caller knows n >= 16)
template <bool TALL>
__device__ __forceinline__ uint32_t mirror_row(int32_t i, uint32_t n)
{
    if (!TALL && n < 16) return mirror_idx(i, n);
    i = i < 0 ? -i : i;
    return (uint32_t)(i < (int32_t)n ? i : 2 * ((int32_t)n - 1) - i);
}

constexpr float kAlpha = -1.586134342f;
constexpr float kBeta  = -0.052980118f;
constexpr float kGamma = 0.882911075f;
constexpr float kDelta = 0.443506852f;
constexpr float kK     = 1.230174105f;

__device__ __forceinline__ float lift(float x, float l, float r, float c)
{
    return __fadd_rn(x, __fmul_rn(__fadd_rn(l, r), c));
}

// ---- per-column vertical recurrences ---------------------------------------------------------
struct V53 {
    int32_t xe, dp;
    __device__ __forceinline__ void init(int32_t x_even) { xe = x_even; dp = 0; }
    // consumes x[2i+1], x[2i+2]; yields pair i
    __device__ __forceinline__ void step(int32_t x1, int32_t x2, int32_t& s, int32_t& d)
    {
        d = x1 - ((xe + x2) >> 1);
        s = xe + ((dp + d + 2) >> 2);
        xe = x2; dp = d;
    }
};
struct V97 {
    float xe, a1, b2, c3;
    __device__ __forceinline__ void init(float x_even) { xe = x_even; a1 = b2 = c3 = 0.f; }
    // consumes x[2i+1], x[2i+2]; yields pair i-1
    __device__ __forceinline__ void step(float x1, float x2, float& s, float& d, float inv_k)
    {
        float a = lift(x1, xe, x2, kAlpha);
        float b = lift(xe, a1, a, kBeta);
        float c = lift(a1, b2, b, kGamma);
        float e = lift(b2, c3, c, kDelta);
        s = __fmul_rn(e, inv_k);
        d = __fmul_rn(c, kK);
        xe = x2; a1 = a; b2 = b; c3 = c;
    }
};

// ---- the same on PAIRS of int16 in one register (pk16.h); the host vouches for the range level by level
//      (context.hip: pk16_level_ok) ------------------------------------------------------------------------------------
struct V53pk {
    pk16 xe, dp;
    __device__ __forceinline__ void init(pk16 x_even) { xe = x_even; dp = (pk16)(0); }
    __device__ __forceinline__ void step(pk16 x1, pk16 x2, pk16& s, pk16& d)
    {
        d = x1 - ((xe + x2) >> 1);
        s = xe + ((dp + d + (pk16)(2)) >> 2);
        xe = x2; dp = d;
    }
};
// ---- horizontal local stencils on an LDS line; w points at local column 2t ------------------
__device__ __forceinline__ void h53(const int32_t* w, int32_t& s, int32_t& d)
{
    int2 m = *reinterpret_cast<const int2*>(w - 2);   // w[-2], w[-1]
    int2 c = *reinterpret_cast<const int2*>(w);       // w[0],  w[1]
    int32_t p2 = w[2];
    int32_t dm = m.y - ((m.x + c.x) >> 1);
    d = c.y - ((c.x + p2) >> 1);
    s = c.x + ((dm + d + 2) >> 2);
}
__device__ __forceinline__ void h97(const float* w, float& s, float& d, float inv_k)
{
    float2 q0 = *reinterpret_cast<const float2*>(w - 4);  // -4 -3
    float2 q1 = *reinterpret_cast<const float2*>(w - 2);  // -2 -1
    float2 q2 = *reinterpret_cast<const float2*>(w);      //  0  1
    float2 q3 = *reinterpret_cast<const float2*>(w + 2);  //  2  3
    float  p4 = w[4];
    float am3 = lift(q0.y, q0.x, q1.x, kAlpha);
    float am1 = lift(q1.y, q1.x, q2.x, kAlpha);
    float ap1 = lift(q2.y, q2.x, q3.x, kAlpha);
    float ap3 = lift(q3.y, q3.x, p4, kAlpha);
    float bm2 = lift(q1.x, am3, am1, kBeta);
    float b0  = lift(q2.x, am1, ap1, kBeta);
    float bp2 = lift(q3.x, ap1, ap3, kBeta);
    float cm1 = lift(am1, bm2, b0, kGamma);
    float cp1 = lift(ap1, b0, bp2, kGamma);
    float e0  = lift(b0, cm1, cp1, kDelta);
    s = __fmul_rn(e0, inv_k);
    d = __fmul_rn(cp1, kK);
}

// forward colour transform of one pixel (same arithmetic as kernels_ingest.hip: mct.cpp:94-104, :541-553)
__device__ __forceinline__ void color_fwd_px(int32_t& c0, int32_t& c1, int32_t& c2, bool irrev)
{
    if (!irrev) {
        const int32_t r = c0, g = c1, b = c2;
        c0 = (r + 2 * g + b) >> 2;
        c1 = b - g;
        c2 = r - g;
    } else {
        const float a_r = 0.299f, a_g = 0.587f, a_b = 0.114f;
        const float cb = 0.5f / (1.0f - a_b), cr = 0.5f / (1.0f - a_r);
        const float r = (float)c0, g = (float)c1, b = (float)c2;
        float y = __fmul_rn(a_r, r);
        y = __fadd_rn(y, __fmul_rn(a_g, g));
        y = __fadd_rn(y, __fmul_rn(a_b, b));
        c0 = __float_as_int(y);
        c1 = __float_as_int(__fmul_rn(cb, __fsub_rn(b, y)));
        c2 = __float_as_int(__fmul_rn(cr, __fsub_rn(r, y)));
    }
}

// PX = 0: the level reads an int32/float plane (levels >= 1, and level 0 of the stage entry point).
// PX = 1 / 2: level 0 fused with K1 -- the rows come straight from the caller's 8- / 16-bit pixel
//   planes, DC shift and RCT/ICT are applied in registers and NC (= 3 with MCT) components run
//   through the transform side by side, so the int32 ingest planes are never written or read
//   (saves 8 of the 8 + b_in + 4 bytes per sample that K1 + level 0 move separately).
// H16 (reversible only): the planes this level reads (PX = 0) and writes hold int16 coefficients -- half the bytes
//   of the int32 working type; the caller guarantees the range (context.hip: planes16_ok).
// GEN = false: the instance for levels the launcher knows to be even (on the origin, even width >= 4, even height >= 16), whose
//   every strip takes a FAST path -- without the general path in the kernel the three-component 9/7 level 0 needs 83 registers
//   instead of 124 (the 5/3 one 84 instead of 99): five waves per SIMD, and room on a SIMD whose other waves are the block
//   coder's ROOM instance (kernels_ht.hip) -- the pairing that cfg3's pipeline lacked.
template <bool F97, int NC, int PX, bool H16 = false, bool GEN = true>
__global__ __launch_bounds__(kThreads) void dwt_level_kernel(DwtLevelArgs a)
{
    static_assert(!(F97 && H16), "16-bit planes are for the reversible transform");
    // The DWT chain is the critical path of pipelined encodes (its launches run beside K3 of this and of the previous
    // frame, whose waves have slack): its waves take the issue arbiter's top priority.  Measured 0.622 -> 0.603 ms/frame.
    __builtin_amdgcn_s_setprio(3);
    using T  = typename std::conditional<F97, float, int32_t>::type;
    using T2 = typename std::conditional<F97, float2, int2>::type;
    using PIX = typename std::conditional<PX == 2, uint16_t, uint8_t>::type;
    static_assert(PX != 0 || NC == 1, "plane input is one component per workgroup");
    __shared__ __attribute__((aligned(16))) T line[2][NC][2][kCols];   // [parity][comp][low/high][column]

    const uint32_t t = threadIdx.x;
    // Which (strip, row segment, plane) this workgroup takes.  Workgroups go to the 8 XCDs round-robin in dispatch order, and
    // every XCD has its own L2: with the plain mapping the two neighbours of a strip -- whose halo columns lie in cache lines
    // it also reads -- run on OTHER XCDs and those lines are fetched once per XCD.  xcd != 0: XCD k (= dispatch id mod 8)
    // takes a contiguous run of the linear (strip fastest) order instead, so that neighbours share an L2.
    uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if constexpr (PX != 0) { if (a.alloc_reset && (bx | by | bz) == 0) ht_alloc_reset(a.alloc_reset, a.alloc_chunk_units, threadIdx.x, blockDim.x); }
    if (a.xcd) {
        const uint32_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint32_t id = bx + gx * (by + gy * bz);
        const uint32_t q = total >> 3, r = total & 7u, k = id & 7u;
        const uint32_t lid = k * q + min(k, r) + (id >> 3);
        bx = lid % gx; by = (lid / gx) % gy; bz = lid / (gx * gy);
    }
    const uint32_t cw = a.cw, ch = a.ch;
    const uint32_t px = a.px, py = a.py;
    const uint32_t sw = (cw + 1 - px) >> 1, dw = cw - sw;
    const uint32_t sh = (ch + 1 - py) >> 1, dh = ch - sh;
    const uint32_t vpairs = (ch + py + 1) >> 1;            // row pairs on the coordinate grid
    const float inv_k = (float)(1.0 / 1.230174105);

    const int32_t c_first = (int32_t)(bx * kOutCols) - kHalo;    // global column of local 0
    // Which pair of the staged line this lane carries through the vertical pass: lanes [0, kOutPairs) take the strip's
    // own pairs IN ORDER (their row loads start on a cache-line boundary), the next kHalo lanes the halo pairs left and
    // right of it; whatever lanes remain ride along on columns nobody reads.
    constexpr uint32_t HP = kHalo / 2;
    const uint32_t lp = t < (uint32_t)kOutPairs ? t + HP : (t < kOutPairs + HP ? t - kOutPairs : t);
    const int32_t cA = c_first + 2 * (int32_t)lp;        // (coordinate grid: sample index = coordinate - px)
    const uint32_t mA = mirror_idx(cA - (int32_t)px, cw), mB = mirror_idx(cA + 1 - (int32_t)px, cw);
    const bool vec = px == 0 && (cA >= 0) && ((uint32_t)cA + 1 < cw);

    // first plane this workgroup produces
    uint32_t plane0 = bz;
    if constexpr (PX != 0) plane0 = (bz / a.zdiv) * a.ncomp + a.comp0 + (bz % a.zdiv);
    using PT = typename std::conditional<H16, int16_t, T>::type;             // element type of the planes in memory
    const PT* in = reinterpret_cast<const PT*>(a.in) + (size_t)plane0 * a.in_pitch;
    PT* ll = reinterpret_cast<PT*>(a.ll) + (size_t)plane0 * a.ll_pitch;
    PT* mp = reinterpret_cast<PT*>(a.mallat) + (size_t)plane0 * a.m_pitch;
    const size_t comp_px = (size_t)cw * ch;
    const PIX* pix = reinterpret_cast<const PIX*>(a.pixels) + (size_t)plane0 * comp_px;
    const bool pvec = vec && (cw & 1u) == 0;          // tightly packed rows: pairs aligned only for even widths

    const int32_t J0 = (int32_t)(by * a.seg_pairs);
    const int32_t J1 = min((int32_t)vpairs, J0 + (int32_t)a.seg_pairs);
    constexpr int lag  = F97 ? 1 : 0;
    constexpr int warm = F97 ? 2 : 1;

    // Interior strips of a tall level take the FAST instance: no column mirroring, aligned pair loads from a uniform
    // row pointer, unpredicated stores (the generic instance spends more instructions on addresses and predicates
    // than on the transform; this kernel is instruction-issue bound, not bandwidth bound).
    // FAST on an edge strip (EDGE): a mirrored column pair is a real pair read backwards -- (-2, -1) is (x[2], x[1]), and
    // (cw, cw + 1) is (x[cw-2], x[cw-3]) for an even width -- so every lane still does ONE pair load, from the smaller of
    // its two mirrored columns, and swaps the halves if they are mirrored
    const uint32_t ld_edge = min(mA, mB);
    const bool swp_edge = mA > mB;
    auto strip = [&](auto fast_tag, auto edge_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr bool EDGE = decltype(edge_tag)::value;
        const uint32_t lane_col = EDGE ? ld_edge : (uint32_t)cA;
        // raw row fetch (no arithmetic, so that prefetched rows stay in flight) and its conversion
        struct Raw { int32_t a[NC], b[NC]; };
        auto fetch_row = [&](int32_t r, Raw& q) {
            const uint32_t rr = mirror_row<FAST>(FAST ? r : r - (int32_t)py, ch);
            if constexpr (PX == 0) {
                if constexpr (H16) {
                    const int16_t* row = reinterpret_cast<const int16_t*>(in) + (size_t)rr * a.in_stride;
                    if (FAST) q.a[0] = (int32_t)*reinterpret_cast<const uint32_t*>(row + lane_col);   // pair stays packed
                    else { q.a[0] = row[mA]; q.b[0] = row[mB]; }
                } else {
                    const int32_t* row = reinterpret_cast<const int32_t*>(in) + (size_t)rr * a.in_stride;
                    if (FAST || vec) { const int2 v = *reinterpret_cast<const int2*>(row + lane_col); q.a[0] = v.x; q.b[0] = v.y; }
                    else     { q.a[0] = row[mA]; q.b[0] = row[mB]; }
                }
            } else {
    #pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const PIX* row = pix + (size_t)k * comp_px + (size_t)rr * cw;
                    if (FAST) {       // the pair stays packed in one register until convert(): a load whose result is
                                      // unpacked at once is waited for at once, and the prefetch buys nothing
                        if constexpr (PX == 1) q.a[k] = *reinterpret_cast<const uint16_t*>(row + lane_col);
                        else                   q.a[k] = (int32_t)*reinterpret_cast<const uint32_t*>(row + lane_col);
                    } else if (pvec) {
                        if constexpr (PX == 1) { const uchar2 v = *reinterpret_cast<const uchar2*>(row + cA); q.a[k] = v.x; q.b[k] = v.y; }
                        else                   { const ushort2 v = *reinterpret_cast<const ushort2*>(row + cA); q.a[k] = v.x; q.b[k] = v.y; }
                    } else { q.a[k] = row[mA]; q.b[k] = row[mB]; }
                }
            }
        };
        auto convert = [&](const Raw& q, T (&va)[NC], T (&vb)[NC]) {
            if constexpr (PX == 0) {
                if constexpr (F97) { va[0] = __int_as_float(q.a[0]); vb[0] = __int_as_float(q.b[0]); }
                else if constexpr (H16 && FAST) { va[0] = (int32_t)(int16_t)q.a[0]; vb[0] = q.a[0] >> 16; }
                else               { va[0] = q.a[0]; vb[0] = q.b[0]; }
                if constexpr (EDGE) { const T ta = va[0]; va[0] = swp_edge ? vb[0] : ta; vb[0] = swp_edge ? ta : vb[0]; }
            } else {
                int32_t xa[NC], xb[NC];
    #pragma unroll
                for (int k = 0; k < NC; ++k) {          // sign-extend int8/int16 samples, DC shift
                    int32_t pa = q.a[k], pb = q.b[k];
                    if constexpr (FAST) {               // unpack the pair fetch_row left in q.a
                        constexpr int B = PX == 1 ? 8 : 16;
                        pb = (int32_t)((uint32_t)pa >> B); pa &= (1 << B) - 1;
                        if constexpr (EDGE) { const int32_t ta = pa; pa = swp_edge ? pb : ta; pb = swp_edge ? ta : pb; }
                    }
                    xa[k] = ((pa ^ a.sext) - a.sext) - a.dc; xb[k] = ((pb ^ a.sext) - a.sext) - a.dc;
                }
                if constexpr (NC == 3) {            // launched with NC = 3 only for the MCT components
                    color_fwd_px(xa[0], xa[1], xa[2], F97);
                    color_fwd_px(xb[0], xb[1], xb[2], F97);
    #pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        if constexpr (F97) { va[k] = __int_as_float(xa[k]); vb[k] = __int_as_float(xb[k]); }
                        else               { va[k] = xa[k]; vb[k] = xb[k]; }
                    }
                } else {
                    if constexpr (F97) { va[0] = (float)xa[0]; vb[0] = (float)xb[0]; }
                    else               { va[0] = xa[0]; vb[0] = xb[0]; }
                }
            }
        };

        typename std::conditional<F97, V97, V53>::type colA[NC], colB[NC];
        int32_t i = J0 - warm;
        {
            Raw q; T xa[NC], xb[NC];
            fetch_row(2 * i, q);
            convert(q, xa, xb);
    #pragma unroll
            for (int k = 0; k < NC; ++k) { colA[k].init(xa[k]); colB[k].init(xb[k]); }
        }
        Raw n1, n2;                           // prefetched rows of the next step
        fetch_row(2 * i + 1, n1);
        fetch_row(2 * i + 2, n2);

        // the output pair this lane produces in the horizontal phase
        // FAST: the halo lanes repeat the work of the nearest owning lane (same LDS reads, same value to the same address),
        // so that the horizontal phase has no branch at all: with one path through the loop body the compiler's wait for
        // the prefetched rows does not also wait for the stores issued after them
        // lane t owns output pair t of the strip (local columns kHalo + 2t, + 1); lanes past the strip's last pair idle
        // (EDGE: the strip may end before its 224th pair; the lanes past the last real pair repeat it as well)
        const uint32_t nvalid = EDGE ? min((uint32_t)kOutPairs, sw - bx * kOutPairs) : (uint32_t)kOutPairs;
        const uint32_t tp = FAST ? min(t, nvalid - 1u) : t;
        const uint32_t th = tp + kHalo / 2;                                  // its pair index inside the staged line
        const bool h_lane = FAST || t < (uint32_t)kOutPairs;
        const uint32_t Jc = bx * kOutPairs + tp;                     // global pair column
        const bool st_s = h_lane && (FAST || (Jc >= px && Jc - px < sw)), st_d = h_lane && (FAST || Jc < dw);
        const uint32_t Js = Jc - (FAST ? 0u : px);                           // its column in the low-pass bands

        const int32_t i_end = J1 - 1 + lag;
        T sA[NC], dA[NC], sB[NC], dB[NC];
        // one vertical step: consumes the prefetched rows and fetches those of the next step (always, rows past the
        // end mirror back into the plane: the steady-state loop then has a single path, so the wait for a prefetched
        // row never has to cover the stores issued after it)
        auto vstep = [&]() {
            T x1a[NC], x1b[NC], x2a[NC], x2b[NC];
            convert(n1, x1a, x1b);
            convert(n2, x2a, x2b);
            fetch_row(2 * i + 3, n1);
            fetch_row(2 * i + 4, n2);
    #pragma unroll
            for (int k = 0; k < NC; ++k) {
                if constexpr (F97) {
                    colA[k].step(x1a[k], x2a[k], sA[k], dA[k], inv_k);
                    colB[k].step(x1b[k], x2b[k], sB[k], dB[k], inv_k);
                } else {
                    colA[k].step(x1a[k], x2a[k], sA[k], dA[k]);
                    colB[k].step(x1b[k], x2b[k], sB[k], dB[k]);
                }
            }
        };
        for (; i - lag < J0; ++i) vstep();   // warm-up steps produce no output
        // horizontal phase of the row pair the vertical step just finished: exchange through the LDS line, local stencil
        auto hphase = [&](int par, int32_t j, T (&o)[NC][4]) {
            if (!FAST && ch == 1) {          // single-row level: vertical pass is the identity -- or, for a lone HIGH-pass
                Raw q;                       // row (odd start), the doubling of the 5/3
                fetch_row((int32_t)py, q);
                convert(q, sA, sB);
                if (py) {
    #pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        if constexpr (F97) { dA[k] = sA[k]; dB[k] = sB[k]; }
                        else { dA[k] = sA[k] * 2; dB[k] = sB[k] * 2; }
                    }
                }
            }
    #pragma unroll
            for (int k = 0; k < NC; ++k) {
                T2 ql, qh;
                ql.x = sA[k]; ql.y = sB[k]; qh.x = dA[k]; qh.y = dB[k];
                *reinterpret_cast<T2*>(&line[par][k][0][2 * lp]) = ql;
                *reinterpret_cast<T2*>(&line[par][k][1][2 * lp]) = qh;
            }
            __syncthreads();
            if (h_lane) {
                const bool has_h = FAST || (uint32_t)j < dh;      // FAST levels have an even height
    #pragma unroll
                for (int k = 0; k < NC; ++k) {
                    T ls, ld, hs = 0, hd = 0;
                    if (!FAST && cw == 1) {        // single column: a low-pass sample passes, a lone high-pass one (odd start)
                                                   // is doubled by the 5/3; it sits at coordinate px of the staged line
                        const T v0 = line[par][k][0][2 * th + px], v1 = line[par][k][1][2 * th + px];
                        if (px) { ls = 0; hs = 0; ld = F97 ? v0 : v0 * 2; hd = F97 ? v1 : v1 * 2; }
                        else    { ls = v0; hs = v1; ld = 0; hd = 0; }
                    } else if constexpr (F97) {
                        h97(&line[par][k][0][2 * th], ls, ld, inv_k);
                        if (has_h) h97(&line[par][k][1][2 * th], hs, hd, inv_k);
                    } else {
                        h53(&line[par][k][0][2 * th], ls, ld);
                        if (has_h) h53(&line[par][k][1][2 * th], hs, hd);
                    }
                    o[k][0] = ls; o[k][1] = ld; o[k][2] = hs; o[k][3] = hd;
                }
            }
        };
        auto store_out = [&](int32_t j, const T (&o)[NC][4]) {
            if (h_lane) {
                const bool has_h = FAST || (uint32_t)j < dh;
    #pragma unroll
                for (int k = 0; k < NC; ++k) {
                    PT* llk = ll + (size_t)k * a.ll_pitch;
                    PT* mpk = mp + (size_t)k * a.m_pitch;
                    if (FAST) {          // interior strip: every horizontal lane owns a column of all four sub-bands
                        llk[(size_t)j * a.ll_stride + Jc] = (PT)o[k][0];
                        mpk[(size_t)j * a.m_stride + sw + Jc] = (PT)o[k][1];
                        mpk[(size_t)(sh + j) * a.m_stride + Jc] = (PT)o[k][2];
                        mpk[(size_t)(sh + j) * a.m_stride + sw + Jc] = (PT)o[k][3];
                    } else {
                        // pair j of the coordinate grid: its low-pass row is row j - py of LL / HL (none for the phantom
                        // pair of an odd start), its high-pass row is row j of LH / HH
                        const bool has_l = (uint32_t)j >= py && (uint32_t)j - py < sh;
                        const size_t jl = (size_t)((uint32_t)j - py);
                        if (has_l) {
                            if (st_s) llk[jl * a.ll_stride + Js] = (PT)o[k][0];
                            if (st_d) mpk[jl * a.m_stride + sw + Jc] = (PT)o[k][1];
                        }
                        if (has_h) {
                            if (st_s) mpk[(size_t)(sh + j) * a.m_stride + Js] = (PT)o[k][2];
                            if (st_d) mpk[(size_t)(sh + j) * a.m_stride + sw + Jc] = (PT)o[k][3];
                        }
                    }
                }
            }
        };
        T o[NC][4];
        if constexpr (FAST) {
            // The compiler's wait for a prefetched row drains every vector memory operation in flight (gfx9 has one
            // counter for loads and stores, and the wait is merged conservatively over the loop's edges).  So all of them
            // are issued in one place, right after that wait -- the next rows AND the previous pair's results, which stay
            // in registers for one iteration -- and each has a whole iteration to complete before the next wait.
            if (i <= i_end) {
                vstep();
                hphase(0, i - lag, o);
                int32_t jp = i - lag;
                ++i;
                for (int par = 1; i <= i_end; ++i, par ^= 1) {
                    vstep();
                    store_out(jp, o);
                    hphase(par, i - lag, o);
                    jp = i - lag;
                }
                store_out(jp, o);
            }
        } else {
            for (int par = 0; i <= i_end; ++i, par ^= 1) {
                vstep();
                hphase(par, i - lag, o);
                store_out(i - lag, o);
            }
        }
    };
    const bool even = (px | py) == 0 && (cw & 1u) == 0 && cw >= 4 && ch >= 16 && (ch & 1u) == 0;
    const bool interior = c_first >= 0 && (uint32_t)c_first + kCols <= cw && (bx + 1) * kOutPairs <= dw;
    if (even && interior) strip(std::true_type{}, std::false_type{});
    else if (even) strip(std::true_type{}, std::true_type{});
    else if constexpr (GEN) strip(std::false_type{}, std::false_type{});
}

// ---- the 5/3 level on packed int16 pairs, four columns per lane ------------------------------------------------------
// For levels the launcher knows to be reversible with 16-bit planes, on the origin, of even height >= 16 and a width that is a
// multiple of 4, with every intermediate inside 16 bits (DwtLevelArgs::pk).  Same plan as dwt_level_kernel's FAST strips --
// vertical recurrence in registers, one exchange line per row pair, horizontal stencil, four sub-band stores -- recast for
// what the 32-bit form is short of:
//  * memory INSTRUCTIONS.  The texture addresser takes a wave64 access at four lanes a clock whatever its width, so a CU
//    moves 16 x (bytes per lane) / 4 bytes a clock: 2-byte accesses (8-bit pixel pairs in, int16 coefficients out) cap
//    the chip at ~4.9 TB/s of reads + writes together, and the 2-columns-per-lane level 0 sat at 3.9 (measured:
//    tools/mem_width_bench.hip; profiles/r02_mem_width.txt).  Here a lane owns FOUR columns: one dword of pixels (or 8
//    bytes of int16 plane) per row and component in, one dword (two coefficients) per sub-band row out;
//  * vector instructions.  The two columns of a pair ride in one register through unpacking, colour transform and the
//    vertical recurrence (v_pk_add / sub / ashr); the exchange line holds (low row | high row << 16) per column, so the
//    horizontal stencil does both rows of the pair at once, and two output pairs share their middle high-pass term;
//  * addresses.  Buffer descriptors per plane + scalar row offset + a constant 32-bit lane offset: no vector
//    instruction goes into an address;
//  * latency.  Rows are fetched two steps ahead.
constexpr int kPkLaneCols = 4;
// the rows are read once: non-temporal loads (cache policy bit 1 on gfx950) keep them from pushing the block coder's tables and
// streams out of the caches it shares with this kernel in the pipelined encode (measured r03: period 0.49 -> 0.47 ms, alone unchanged;
// non-temporal STORES made this kernel 10 % slower)
constexpr int kPkLoadAux = 2;
// NT lanes per workgroup: 256 (strips of up to 960 columns), or 128 -- half the footprint (two waves, 12 KiB of LDS with three
// components): the better fit for narrow levels (pk_nt below)
constexpr int kPkHalo     = kPkLaneCols;                   // one lane's worth each side (the stencil needs 2 left, 1 right)
// The strips of a level share its width evenly, in multiples of 64 columns (64 bytes of every sub-band row); at most nt - 2
// lanes of four columns (one halo lane each side): 960 columns for 256 lanes, 448 for 128
__host__ __device__ inline uint32_t pk_strip_cols(uint32_t cw, uint32_t nt)
{
    const uint32_t most = ((nt - 2u) * kPkLaneCols) & ~63u;
    const uint32_t n = (cw + most - 1) / most;
    return min(most, ((cw + n - 1) / n + 63u) & ~63u);
}

template <int NC, int PX, int NT>
__global__ __launch_bounds__(NT) void dwt53_pk_kernel(DwtLevelArgs a)
{
    static_assert(PX == 0 || PX == 1, "int16 planes or 8-bit pixels");
    static_assert(PX != 0 || NC == 1, "plane input is one component per workgroup");
    __builtin_amdgcn_s_setprio(3);                         // (as dwt_level_kernel: the DWT chain is the critical path)
    constexpr int kPkCols = NT * kPkLaneCols;                // columns staged per line
    __shared__ __attribute__((aligned(16))) uint32_t line[2][NC][kPkCols];   // [parity][comp][column] = low row | high row << 16

    const uint32_t t = threadIdx.x;
    uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if constexpr (PX != 0) { if (a.alloc_reset && (bx | by | bz) == 0) ht_alloc_reset(a.alloc_reset, a.alloc_chunk_units, threadIdx.x, blockDim.x); }
    if (a.xcd) {                                           // XCD k takes a contiguous run of the strip-fastest order (see above)
        const uint32_t gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        const uint32_t id = bx + gx * (by + gy * bz);
        const uint32_t q = total >> 3, r = total & 7u, k = id & 7u;
        const uint32_t lid = k * q + min(k, r) + (id >> 3);
        bx = lid % gx; by = (lid / gx) % gy; bz = lid / (gx * gy);
    }
    const uint32_t cw = a.cw, ch = a.ch;
    const uint32_t sw = cw >> 1, sh = ch >> 1;
    const uint32_t scols = pk_strip_cols(cw, NT), slanes = scols / kPkLaneCols;     // this level's strip: columns, lanes

    uint32_t plane0 = bz;
    if constexpr (PX != 0) plane0 = (bz / a.zdiv) * a.ncomp + a.comp0 + (bz % a.zdiv);
    const int16_t* in = reinterpret_cast<const int16_t*>(a.in) + (size_t)plane0 * a.in_pitch;
    int16_t* ll = reinterpret_cast<int16_t*>(a.ll) + (size_t)plane0 * a.ll_pitch;
    int16_t* mp = reinterpret_cast<int16_t*>(a.mallat) + (size_t)plane0 * a.m_pitch;
    const size_t comp_px = (size_t)cw * ch;
    const uint8_t* pix = reinterpret_cast<const uint8_t*>(a.pixels) + (size_t)plane0 * comp_px;

    // Lane t < slanes carries group t + 1 of the staged line (groups of four columns; group 0 and group slanes + 1 are the
    // halo, on the two lanes after; the rest repeat lane 0) -- the strip's own groups in lane order, so that row loads and
    // sub-band stores of a wave are whole runs of lines.
    const uint32_t grp = t < slanes ? t + 1 : (t == slanes ? 0u : (t == slanes + 1 ? t : 1u));
    const int32_t c0 = (int32_t)(bx * scols) - kPkHalo + (int32_t)(grp * kPkLaneCols);     // first of its four columns
    // A group outside the image is a mirrored one: (-4 .. -1) is x[4], x[3], x[2], x[1], and (cw + 4m ..) is x[cw-2-4m] downwards:
    // four consecutive samples read backwards (the width is a multiple of 4, so no group straddles an edge).  One load
    // from the lowest of them -- off its natural alignment by one sample -- and a reversal.  (Groups further out than the
    // halo are never read; they load from inside the row all the same.)
    const bool rev = c0 < 0 || c0 >= (int32_t)cw;
    int32_t lo = c0 < 0 ? -c0 - 3 : (c0 >= (int32_t)cw ? 2 * ((int32_t)cw - 1) - c0 - 3 : c0);
    lo = max(0, min(lo, (int32_t)cw - kPkLaneCols));
    const uint32_t lane_off = (uint32_t)lo * (PX == 0 ? 2u : 1u);          // bytes into a row
    // v_perm selectors that unpack a load into the pairs (A: columns 0, 1; B: columns 2, 3), reversed for mirrored groups
    const uint32_t selA = PX == 1 ? (rev ? 0x0c020c03u : 0x0c010c00u) : (rev ? 0x05040706u : 0x03020100u);
    const uint32_t selB = PX == 1 ? (rev ? 0x0c000c01u : 0x0c030c02u) : (rev ? 0x01000302u : 0x07060504u);
    const pk16 dc2 = as_pk((uint32_t)a.dc * 0x10001u);       // (unsigned pixels: the launcher sends signed ones elsewhere)

    __amdgpu_buffer_rsrc_t r_in[NC], r_ll[NC], r_mp[NC];
    #pragma unroll
    for (int k = 0; k < NC; ++k) {
        r_in[k] = PX == 0 ? buffer_from(in) : buffer_from(pix + (size_t)k * comp_px);
        r_ll[k] = buffer_from(ll + (size_t)k * a.ll_pitch);
        r_mp[k] = buffer_from(mp + (size_t)k * a.m_pitch);

    }
    __amdgpu_buffer_rsrc_t r_none = buffer_from(a.mallat, true);

    const int32_t J0 = (int32_t)(by * a.seg_pairs);
    const int32_t J1 = min((int32_t)sh, J0 + (int32_t)a.seg_pairs);

    struct Raw { uint32_t v[NC]; uint2 w[NC]; };
    auto fetch_row = [&](int32_t r, Raw& q) {                // raw row fetch: no arithmetic, so that the rows stay in flight
        const uint32_t rr = mirror_row<true>(r, ch);
        if constexpr (PX == 0) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r_in[0], lane_off, rr * a.in_stride * 2u, kPkLoadAux);
            q.w[0].x = v.x; q.w[0].y = v.y;
        } else {
    #pragma unroll
            for (int k = 0; k < NC; ++k) q.v[k] = __builtin_amdgcn_raw_buffer_load_b32(r_in[k], lane_off, rr * cw, kPkLoadAux);
        }
    };
    auto convert = [&](const Raw& q, pk16 (&va)[NC], pk16 (&vb)[NC]) {
        if constexpr (PX == 0) {
            va[0] = as_pk(__builtin_amdgcn_perm(q.w[0].y, q.w[0].x, selA));
            vb[0] = as_pk(__builtin_amdgcn_perm(q.w[0].y, q.w[0].x, selB));
        } else {
            pk16 xa[NC], xb[NC];
    #pragma unroll
            for (int k = 0; k < NC; ++k) {
                xa[k] = as_pk(__builtin_amdgcn_perm(0u, q.v[k], selA));
                xb[k] = as_pk(__builtin_amdgcn_perm(0u, q.v[k], selB));
            }
            if constexpr (NC == 3) {   // RCT (mct.cpp:94-104) after the DC shift: Cb = B - G and Cr = R - G lose the shift,
                                       // Y = (R + 2G + B) >> 2 = G + ((Cb + Cr) >> 2) exactly, minus the shift
                const pk16 cba = xa[2] - xa[1], cra = xa[0] - xa[1], cbb = xb[2] - xb[1], crb = xb[0] - xb[1];
                va[0] = xa[1] + ((cba + cra) >> 2) - dc2; va[1] = cba; va[2] = cra;
                vb[0] = xb[1] + ((cbb + crb) >> 2) - dc2; vb[1] = cbb; vb[2] = crb;
            } else { va[0] = xa[0] - dc2; vb[0] = xb[0] - dc2; }
        }
    };

    V53pk colA[NC], colB[NC];
    {
        Raw q; pk16 xa[NC], xb[NC];
        fetch_row(2 * (J0 - 1), q);
        convert(q, xa, xb);
    #pragma unroll
        for (int k = 0; k < NC; ++k) { colA[k].init(xa[k]); colB[k].init(xb[k]); }
    }
    Raw ra1, ra2, rb1, rb2;                                  // rows of the next step (a) and of the one after (b)
    fetch_row(2 * J0 - 1, ra1); fetch_row(2 * J0, ra2);
    fetch_row(2 * J0 + 1, rb1); fetch_row(2 * J0 + 2, rb2);

    // horizontal phase: lane t produces the two output pairs of group t + 1; lanes past the strip's last group (the halo
    // lanes, the idle ones, and in the last strip those beyond the image) repeat the last one -- same reads, same values
    // to the same addresses -- so that the loop has no branch
    const uint32_t nv = min(slanes, (sw - bx * (scols / 2) + 1u) >> 1);
    const uint32_t tp = min(t, nv - 1u);
    const uint32_t hc = (tp + 1u) * kPkLaneCols;             // its first column in the staged line
    const uint32_t oc = (bx * (scols / 2) + tp * 2u) * 2u;   // byte offset of its two coefficients in a sub-band row

    pk16 sA[NC], dA[NC], sB[NC], dB[NC];
    auto vstep = [&](int32_t ii, Raw& r1, Raw& r2) {         // step ii: consumes r1, r2 and refills them two steps ahead
        pk16 x1a[NC], x1b[NC], x2a[NC], x2b[NC];
        convert(r1, x1a, x1b); convert(r2, x2a, x2b);
        fetch_row(2 * ii + 5, r1); fetch_row(2 * ii + 6, r2);
    #pragma unroll
        for (int k = 0; k < NC; ++k) { colA[k].step(x1a[k], x2a[k], sA[k], dA[k]); colB[k].step(x1b[k], x2b[k], sB[k], dB[k]); }
    };
    auto hphase = [&](int par, uint32_t (&o)[NC][4]) {
    #pragma unroll
        for (int k = 0; k < NC; ++k) {
            uint4 w;                                         // per column (vertical low | vertical high << 16)
            w.x = __builtin_amdgcn_perm(as_u32(dA[k]), as_u32(sA[k]), kSelLoLo);
            w.y = __builtin_amdgcn_perm(as_u32(dA[k]), as_u32(sA[k]), kSelHiHi);
            w.z = __builtin_amdgcn_perm(as_u32(dB[k]), as_u32(sB[k]), kSelLoLo);
            w.w = __builtin_amdgcn_perm(as_u32(dB[k]), as_u32(sB[k]), kSelHiHi);
            *reinterpret_cast<uint4*>(&line[par][k][grp * kPkLaneCols]) = w;
        }
        __syncthreads();
    #pragma unroll
        for (int k = 0; k < NC; ++k) {
            const uint32_t* w = &line[par][k][hc];
            const uint2 m = *reinterpret_cast<const uint2*>(w - 2);
            const uint4 c = *reinterpret_cast<const uint4*>(w);
            const pk16 m2 = as_pk(m.x), m1 = as_pk(m.y), c0 = as_pk(c.x), c1 = as_pk(c.y), c2 = as_pk(c.z), c3 = as_pk(c.w),
                       p4 = as_pk(w[4]);
            const pk16 dm = m1 - ((m2 + c0) >> 1);
            const pk16 d0 = c1 - ((c0 + c2) >> 1);
            const pk16 d1 = c3 - ((c2 + p4) >> 1);
            const pk16 s0 = c0 + ((dm + d0 + (pk16)(2)) >> 2);
            const pk16 s1 = c2 + ((d0 + d1 + (pk16)(2)) >> 2);
            // (horizontal low | .. of the vertical low row, of the vertical high row): regroup by sub-band row
            o[k][0] = __builtin_amdgcn_perm(as_u32(s1), as_u32(s0), kSelLoLo);       // LL: two columns
            o[k][1] = __builtin_amdgcn_perm(as_u32(s1), as_u32(s0), kSelHiHi);       // LH
            o[k][2] = __builtin_amdgcn_perm(as_u32(d1), as_u32(d0), kSelLoLo);       // HL
            o[k][3] = __builtin_amdgcn_perm(as_u32(d1), as_u32(d0), kSelHiHi);       // HH
        }
    };
    auto store_out = [&](int32_t j, const uint32_t (&o)[NC][4]) {
    #pragma unroll
        for (int k = 0; k < NC; ++k) {
            const uint32_t ju = (uint32_t)j;               // (uniform; said so for the sake of the store after the loop)
            const uint32_t lrow = __builtin_amdgcn_readfirstlane(ju * a.ll_stride * 2u),
                           mlo = __builtin_amdgcn_readfirstlane(ju * a.m_stride * 2u),
                           mhi = __builtin_amdgcn_readfirstlane((sh + ju) * a.m_stride * 2u);
            __builtin_amdgcn_raw_buffer_store_b32(o[k][0], r_ll[k], oc, lrow, 0);
            const __amdgpu_buffer_rsrc_t rm = r_mp[k];
            __builtin_amdgcn_raw_buffer_store_b32(o[k][1], rm, oc, mhi, 0);
            __builtin_amdgcn_raw_buffer_store_b32(o[k][2], rm, oc + 2u * sw, mlo, 0);
            __builtin_amdgcn_raw_buffer_store_b32(o[k][3], rm, oc + 2u * sw, mhi, 0);
        }
    };
    // All memory operations of a step are issued in one place: the rows two steps ahead, then the PREVIOUS pair's results
    // (held in registers for one step).  Steps alternate between the two row sets and the two halves of the exchange line.
    // The compiler's wait for a row allows as many younger operations in flight as the SHORTEST path to that point has
    // issued, and the path into the loop would be the short one -- so the two steps before the loop issue the same
    // sequence as every later pair of steps (rows, 4 NC stores, rows, 4 NC stores), their stores going through the
    // descriptor of zero length.
    auto no_stores = [&]() {
    #pragma unroll
        for (int k = 0; k < 4 * NC; ++k) __builtin_amdgcn_raw_buffer_store_b32(0u, r_none, oc + 64 * k, 0, 0);   // (apart, or they merge into wider ones)
    };
    const int32_t i_end = J1 - 1;
    uint32_t o[NC][4];
    int32_t ii = J0, jp;
    vstep(J0 - 1, ra1, ra2); no_stores();                  // warm-up step: no output
    __builtin_amdgcn_sched_barrier(0);
    vstep(ii, rb1, rb2); no_stores(); hphase(0, o); jp = ii; ++ii;
    __builtin_amdgcn_sched_barrier(0);
    while (ii + 1 <= i_end) {         // (the fences keep the scheduler from pulling a step's first touch of its rows -- and with
                                      //  it the wait for them -- up into the step before)
        vstep(ii, ra1, ra2); store_out(jp, o); hphase(1, o); jp = ii; ++ii;
        __builtin_amdgcn_sched_barrier(0);
        vstep(ii, rb1, rb2); store_out(jp, o); hphase(0, o); jp = ii; ++ii;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (ii <= i_end) { vstep(ii, ra1, ra2); store_out(jp, o); hphase(1, o); jp = ii; }
    store_out(jp, o);
}

} // namespace

uint32_t dwt_strip_cols() { return kOutCols; }

// the level shape dwt53_pk_kernel takes
static bool dwt_level_is_pk(const DwtLevelArgs& a)
{
    // (row offsets are 32-bit byte offsets from a plane's first sample: planes of 2^31 samples and more keep the flat addressing)
    const bool near = (uint64_t)a.m_stride * a.ch < (1ull << 31) && (uint64_t)a.cw * a.ch < (1ull << 31) && (uint64_t)a.in_stride * a.ch < (1ull << 31);
    return a.h16 && a.pk && !a.irreversible && (a.px | a.py) == 0 && (a.cw & 3u) == 0 && a.cw >= 256u &&
           a.ch >= 16 && (a.ch & 1u) == 0 && near;
}
// 256 lanes for wide levels (8K level 0: 127 us against 138 with 128 lanes), 128 for narrow ones, whose strips would leave half of
// 256 lanes idle (64 tiles of 1024^2: levels 0-2 238 -> 203 us)
static uint32_t pk_nt(const DwtLevelArgs& a) { return a.cw <= 2048u ? 128u : 256u; }
uint32_t dwt_level_strip_cols(const DwtLevelArgs& a) { return dwt_level_is_pk(a) ? pk_strip_cols(a.cw, pk_nt(a)) : (uint32_t)kOutCols; }

hipError_t launch_dwt_level(const DwtLevelArgs& a, hipStream_t s)
{
    const uint32_t sh = (a.ch + a.py + 1) >> 1;
    dim3 grid((a.cw + a.px + kOutCols - 1) / kOutCols, (sh + a.seg_pairs - 1) / a.seg_pairs, a.nplanes);
    dim3 block(kThreads);
    if (dwt_level_is_pk(a)) {
        const uint32_t sc = pk_strip_cols(a.cw, pk_nt(a));
        grid.x = (a.cw + sc - 1) / sc;
        if (pk_nt(a) == 128) hipLaunchKernelGGL((dwt53_pk_kernel<1, 0, 128>), grid, dim3(128), 0, s, a);
        else                 hipLaunchKernelGGL((dwt53_pk_kernel<1, 0, 256>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.irreversible)
        hipLaunchKernelGGL((dwt_level_kernel<true, 1, 0>), grid, block, 0, s, a);
    else if (a.h16)
        hipLaunchKernelGGL((dwt_level_kernel<false, 1, 0, true>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((dwt_level_kernel<false, 1, 0>), grid, block, 0, s, a);
    return hipGetLastError();
}

// Level 0 straight from the pixels. `a` describes level 0 (cw x ch = tile size, in/in_stride unused);
// a.nplanes is ignored: the grid covers ntiles x (MCT triple | every component on its own).
hipError_t launch_dwt_level0_fused(const DwtLevelArgs& a0, uint32_t ntiles, uint32_t ncomp, int mct, hipStream_t s)
{
    const uint32_t sh = (a0.ch + a0.py + 1) >> 1;
    dim3 block(kThreads);
    auto go = [&](uint32_t comp0, uint32_t zdiv, int nc) {
        DwtLevelArgs a = a0;
        a.comp0 = comp0; a.zdiv = zdiv; a.ncomp = ncomp;
        if (comp0 != 0) a.alloc_reset = nullptr;             // (the first launch resets the allocator)
        dim3 grid((a.cw + a.px + kOutCols - 1) / kOutCols, (sh + a.seg_pairs - 1) / a.seg_pairs, ntiles * zdiv);
        // (what the kernel calls `even`: every strip of the level takes a FAST path)
        static const bool only_fast_ok = !(getenv("GRK_AMD_DWT_FAST_ONLY") && atoi(getenv("GRK_AMD_DWT_FAST_ONLY")) == 0);   // (=0: A/B runs)
        const bool all_fast = only_fast_ok && (a.px | a.py) == 0 && (a.cw & 1u) == 0 && a.cw >= 4 && a.ch >= 16 && (a.ch & 1u) == 0;
#define GRK_L0(F97, NC, PX) do { if (all_fast) hipLaunchKernelGGL((dwt_level_kernel<F97, NC, PX, false, false>), grid, block, 0, s, a); \
                                 else hipLaunchKernelGGL((dwt_level_kernel<F97, NC, PX>), grid, block, 0, s, a); } while (0)
        const int px = a.px_bytes == 1 ? 1 : 2;
        if (a.irreversible) {
            if (nc == 3) { if (px == 1) GRK_L0(true, 3, 1); else GRK_L0(true, 3, 2); }
            else         { if (px == 1) GRK_L0(true, 1, 1); else GRK_L0(true, 1, 2); }
        } else if (px == 1 && dwt_level_is_pk(a)) {
            const uint32_t sc = pk_strip_cols(a.cw, pk_nt(a));
            grid.x = (a.cw + sc - 1) / sc;
            if (pk_nt(a) == 128) {
                if (nc == 3) hipLaunchKernelGGL((dwt53_pk_kernel<3, 1, 128>), grid, dim3(128), 0, s, a);
                else         hipLaunchKernelGGL((dwt53_pk_kernel<1, 1, 128>), grid, dim3(128), 0, s, a);
            } else {
                if (nc == 3) hipLaunchKernelGGL((dwt53_pk_kernel<3, 1, 256>), grid, block, 0, s, a);
                else         hipLaunchKernelGGL((dwt53_pk_kernel<1, 1, 256>), grid, block, 0, s, a);
            }
        } else if (a.h16 && px == 1) {       // 16-bit planes exist for 8-bit pixels only (context.hip: planes16_ok)
            if (nc == 3) hipLaunchKernelGGL((dwt_level_kernel<false, 3, 1, true>), grid, block, 0, s, a);
            else         hipLaunchKernelGGL((dwt_level_kernel<false, 1, 1, true>), grid, block, 0, s, a);
        } else {
            if (nc == 3) { if (px == 1) GRK_L0(false, 3, 1); else GRK_L0(false, 3, 2); }
            else         { if (px == 1) GRK_L0(false, 1, 1); else GRK_L0(false, 1, 2); }
        }
#undef GRK_L0
    };
    if (mct && ncomp >= 3) {
        go(0, 1, 3);
        for (uint32_t k = 3; k < ncomp; ++k) go(k, 1, 1);
    } else {
        go(0, ncomp, 1);
    }
    return hipGetLastError();
}

} // namespace grk_amd
