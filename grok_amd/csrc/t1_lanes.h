// grok_amd/csrc/t1_lanes.h -- K8L: the Part-1 (EBCOT / MQ) block decoder with ONE CODE-BLOCK PER LANE, the per-lane logic.
//
// Replaces, for the bulk of a frame's blocks, the same reference functions as kernels_t1dec.hip (T1::decompress_cblk,
// t1/t1_part1/T1.cpp:1262-1337; the passes :854-1255; the MQ decoder mqc_dec.cpp:107-177, mqc_dec_inl.h).
//
// Why: EBCOT is one dependent chain per code-block, and a wave alone issues one instruction every ~8.5 cycles
// (profiles/r02_valu_issue_rates.txt), so a chain runs at (instructions per decision) x 8.5 cycles whatever the design.
// One block per WAVE (K8, r01-r03) spends ~80 wave-instructions per decision for ONE block: 41 G instructions per cfg5
// frame, bound by the CU's scalar unit at 45 ms.  Here a wave's 64 lanes run 64 chains at once: every lane is a small state
// machine that makes AT MOST ONE MQ DECISION PER ITERATION of a wave-uniform loop, so the lanes share the one
// decoder body whatever pass / stripe / column each of them is in (the naive "several blocks per wave" form of r01
// serialised the ~25 inlined decoder call sites and lost).  ~3x the instructions per decision, 64 decisions per pass over them.
//
// A lane's state:
//   * MQ registers A (as a << 16), C, CT; the 19 contexts as the Table C.2 ROW of their current state, one dword each in LDS
//     ([context][lane]: conflict-free); the coded bytes through a 64-bit shift register refilled four bytes at a time.
//   * The block's significance / sign / visited / refined bitmaps (T1's sigma, chi, pi, mu; one 64-bit row each per sample
//     row) live in GLOBAL memory (2 KB per block) and only the stripe in work -- its 4 rows + the rows above and below --
//     sits in registers.  Stripe changes are batched: every sixth iteration the lanes that finished a stripe store it and
//     issue the loads of the next one, two iterations later they take delivery -- no lane ever waits on a load it just issued.
//   * Decoded magnitudes are not kept as values at all: per bit-plane the block leaves the significance bitmap at the end of
//     the plane and the refinement bits of the plane's mag-ref pass; t1_recon_kernel (kernels_t1lanes.hip) turns those into
//     coefficients, one wave per block, and dequantises.
// This header is the lane logic only, written so that it also compiles on the host: tests/c/t1_lanes_sim.cpp steps 64 such
// lanes through the same phases on the CPU and compares with the oracle (test infrastructure; the product is the HIP kernel).
#pragma once
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define T1L_FN __device__ __forceinline__
#define T1L_UNROLL _Pragma("unroll")
#else
#define T1L_FN static inline
#define T1L_UNROLL
#endif

namespace t1l {

// ---- lane modes ------------------------------------------------------------------------------------------------------
enum : uint32_t { ST_ZC = 0, ST_SC = 1, ST_MR = 2, ST_AGG = 3, ST_UNI1 = 4, ST_UNI2 = 5,     // one MQ decision this iteration
                  ST_NEEDCOL = 6, ST_NEEDSTRIPE = 7, ST_WAIT = 8, ST_DONE = 9,
                  ST_PASSWAIT = 10 };      // the lane's pass is complete: it waits for the wave's other lanes (pass-synchronous waves)

// ---- LDS tables of a wave (byte offsets into one LDS block) ---------------------------------------------------------------
// ctxrow [19][64] dwords: context cx of lane l at cx * 256 + l * 4
// mqtab  [94] dwords: Table C.2 with the MPS sense folded in (entry = state + 47 * mps):
//                     Qe << 16 | mps << 14 | (entry after LPS) << 7 | (entry after MPS)
// zclut  [4][512] u16: ctxrow byte offset (context * 256) by orientation and the nine neighbourhood bits
// sclut  [256] u16:   (context * 256) | xor bit, by the sign-neighbourhood index
constexpr uint32_t kCtxBytes = 19u * 256u;
constexpr uint32_t kOffMq = kCtxBytes;                  // 4864
constexpr uint32_t kOffZc = kOffMq + 96u * 4u;          // 5248
constexpr uint32_t kOffSc = kOffZc + 4u * 512u * 2u;    // 9344
constexpr uint32_t kLdsBytes = kOffSc + 256u * 2u;      // 9856

// Table C.2: Qe, NMPS, NLPS, SWITCH
struct MqRow { uint16_t qe; uint8_t nmps, nlps, sw; };
constexpr MqRow kMq[47] = {
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
// entry e = state + 47 * mps of the folded table
constexpr uint32_t mq_entry(uint32_t e)
{
    const uint32_t st = e % 47u, mps = e / 47u;
    const uint32_t after_mps = kMq[st].nmps + 47u * mps;
    const uint32_t after_lps = kMq[st].nlps + 47u * (mps ^ kMq[st].sw);
    return ((uint32_t)kMq[st].qe << 16) | (mps << 14) | (after_lps << 7) | after_mps;
}

// Zero-coding context (Table D.1) by orientation and the eight neighbour bits (index: row above (x-1, x, x+1) in bits 0-2, left and
// right in bits 3-4, row below in bits 5-7) -- the rule as in kernels_t1dec.hip
constexpr int zc_context(int orient, uint32_t idx)
{
    const uint32_t w0 = idx & 7u, l = (idx >> 3) & 1u, r = (idx >> 4) & 1u, w2 = idx >> 5;
    int hh = (int)l + (int)r;
    int vv = (int)((w0 >> 1) & 1u) + (int)((w2 >> 1) & 1u);
    const int dd = (int)(w0 & 1u) + (int)((w0 >> 2) & 1u) + (int)(w2 & 1u) + (int)((w2 >> 2) & 1u);
    if (orient == 1) { const int t = hh; hh = vv; vv = t; }
    if (orient == 3) {
        const int hv = hh + vv;
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
// by the NINE bits of a 3 x 3 window (row above in bits 0-2, own row 3-5 -- centre ignored --, row below 6-8)
constexpr uint32_t zc_context9(int orient, uint32_t nine)
{
    const uint32_t w0 = nine & 7u, w1 = (nine >> 3) & 7u, w2 = nine >> 6;
    return (uint32_t)zc_context(orient, w0 | ((w1 & 1u) << 3) | ((w1 & 4u) << 2) | (w2 << 5));
}
// Sign context and XOR bit (Tables D.2 / D.3) by: significant (up, left, right, down) in bits 0, 2, 4, 6, negative in the bit above each
constexpr uint32_t sign_context(uint32_t idx)
{
    int c[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k) c[k] = ((idx >> (2 * k)) & 1u) ? (((idx >> (2 * k + 1)) & 1u) ? -1 : 1) : 0;
    int hc = c[1] + c[2], vc = c[0] + c[3];
    hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc); vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
    uint32_t cxn = 0, xr = 0;
    if (hc == 1)      { cxn = vc == 1 ? 13u : (vc == 0 ? 12u : 11u); xr = 0; }
    else if (hc == 0) { cxn = vc == 0 ? 9u : 10u; xr = vc == -1 ? 1u : 0u; }
    else              { cxn = vc == 1 ? 11u : (vc == 0 ? 12u : 13u); xr = 1; }
    return cxn * 256u | xr;
}

// ---- a block's work area in global memory (uint64 units) ------------------------------------------------------------------
// [0, 256): state, stripe s at s * 16: S rows 0-3, N rows 4-7, P rows 8-11, M rows 12-15
// [256 + 128 i, ...): bit-plane i (0 = the block's top plane): 64 rows of significance at the end of the plane, 64 rows of
//                     refinement bits of the plane's mag-ref pass
constexpr uint32_t kWorkU64 = 2048;                  // 16 KB per block
constexpr uint32_t kPlaneBase = 256, kPlaneU64 = 128;
constexpr uint32_t kMaxPlanes = (kWorkU64 - kPlaneBase) / kPlaneU64;      // 14

struct BlockIn {             // what a lane is told about its block
    const uint8_t* data;     // first coded byte
    uint32_t len;            // coded bytes
    uint32_t numbps, numpasses;
    uint32_t w, h, orient;
    uint64_t* work;          // kWorkU64 words
    const uint8_t* lo; const uint8_t* hi;   // readable range of the coded buffer
};

struct Lane {
    // MQ decoder
    uint32_t A, C, ct;
    uint64_t NB;             // the next bytes of the stream, byte 0 = the one the reference's `bp` points at
    uint32_t nv;             // valid bytes in NB
    uint32_t fpos;           // stream index of the first byte of the next dword to fetch
    uint32_t len;
    const uint8_t* fsrc;     // address of stream byte 0 (the fetches are dword-aligned around it)
    const uint8_t* lo; const uint8_t* hi;
    uint32_t pend;           // a fetched dword waits in `ldw`
    uint32_t ldw;
    // block
    uint32_t w, h, ns, zcbase;
    uint32_t np_left, type, pidx, fresh;
    int32_t  bp;
    uint64_t* work;
    uint64_t wmask;
    // stripe
    uint32_t s, nr;
    uint64_t S[6], N[6], P[4], M[4], R[4];
    uint64_t cm, Q;
    // column
    uint32_t x, nbx, nnx, pv, mv, rf, todo, t, xr, r;
    uint64_t bx;
    uint32_t st;
};

T1L_FN uint32_t ctz32(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
T1L_FN uint32_t ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
T1L_FN uint32_t clz32(uint32_t v) { return (uint32_t)__builtin_clz(v); }
T1L_FN uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// one dword of the stream: the 4 bytes at stream index p (p may be negative for the very first, misaligned fetch: those
// bytes are shifted out by the caller); bytes outside [lo, hi) read as 0xFF
T1L_FN uint32_t fetch_dword(const uint8_t* fsrc, int32_t p, const uint8_t* lo, const uint8_t* hi)
{
    const uint8_t* q = fsrc + p;
    if (q >= lo && q + 4 <= hi) return *reinterpret_cast<const uint32_t*>(q);
    uint32_t v = 0;
    T1L_UNROLL
    for (int b = 0; b < 4; ++b) v |= (uint32_t)((q + b >= lo && q + b < hi) ? q[b] : 0xFFu) << (8 * b);
    return v;
}
// bytes at stream index >= len are the artificial 0xFF terminator (mqc_dec.cpp:113-118)
T1L_FN uint32_t mask_dword(uint32_t v, int32_t p, uint32_t len)
{
    const int32_t valid = (int32_t)len - p;                 // bytes of this dword inside the segment
    if (valid >= 4) return v;
    if (valid <= 0) return 0xFFFFFFFFu;
    return v | (0xFFFFFFFFu << (8 * valid));
}

// ---- block start: everything but the MQ initialisation (which needs the first bytes) ------------------------------------
T1L_FN void lane_init(Lane& L, const BlockIn& b)
{
    L.w = b.w; L.h = b.h; L.ns = (b.h + 3u) >> 2; L.zcbase = kOffZc + (b.orient & 3u) * 1024u;
    L.np_left = b.numpasses; L.type = 2; L.pidx = 0; L.fresh = 1; L.bp = (int32_t)b.numbps;
    L.work = b.work;
    L.wmask = b.w >= 64 ? ~0ull : ((1ull << b.w) - 1ull);
    L.len = b.len; L.lo = b.lo; L.hi = b.hi;
    // the stream: dword-aligned fetches; the first two here
    const uint32_t skew = (uint32_t)(reinterpret_cast<uintptr_t>(b.data) & 3u);
    L.fsrc = b.data;
    // (mask_dword with a negative index: the bytes before the block count as inside; they are shifted out below)
    const uint32_t d0m = mask_dword(fetch_dword(b.data, -(int32_t)skew, b.lo, b.hi), -(int32_t)skew, b.len);
    const uint32_t d1m = mask_dword(fetch_dword(b.data, 4 - (int32_t)skew, b.lo, b.hi), 4 - (int32_t)skew, b.len);
    uint64_t nb = (uint64_t)d0m | ((uint64_t)d1m << 32);
    L.NB = nb >> (8u * skew);
    L.nv = 8u - skew;
    L.fpos = 8u - skew;
    L.pend = 0; L.ldw = 0;
    // mqc_init_dec (mqc_dec.cpp:140-154): c = byte0 << 16; bytein; c <<= 7; ct -= 7; a = 0x8000
    const uint32_t cur = (uint32_t)L.NB & 0xFFu, nxt = (uint32_t)(L.NB >> 8) & 0xFFu;
    uint32_t c = cur << 16, ct;
    if (cur == 0xFFu) {
        if (nxt > 0x8Fu) { c += 0xFF00u; ct = 8; }
        else { c += nxt << 9; ct = 7; L.NB >>= 8; L.nv -= 1; }
    } else { c += nxt << 8; ct = 8; L.NB >>= 8; L.nv -= 1; }
    L.C = c << 7; L.ct = ct - 7u; L.A = 0x80000000u;
    // the first stripe of the first pass: nothing is significant, nothing to load
    L.s = 0;
    T1L_UNROLL
    for (int r = 0; r < 6; ++r) { L.S[r] = 0; L.N[r] = 0; }
    T1L_UNROLL
    for (int r = 0; r < 4; ++r) { L.P[r] = 0; L.M[r] = 0; L.R[r] = 0; }
    L.cm = 0; L.Q = 0; L.x = 0; L.nbx = 0; L.nnx = 0; L.pv = 0; L.mv = 0; L.rf = 0; L.todo = 0; L.t = 0; L.xr = 0; L.r = 0; L.bx = 0;
    L.st = ST_WAIT;              // lane_stripe_enter makes the stripe's masks
}

// The pass type (0 sig-prop, 1 mag-ref, 2 cleanup) is a TEMPLATE parameter of the pass-dependent steps: a wave's lanes run
// their passes in step (a lane that has finished a pass waits for the others, ST_PASSWAIT), so the kernel runs one specialised
// copy of the loop per pass type and the code of the other two types is not there to be executed (T < 0: the lane's own
// L.type at run time -- the simulator's free-running mode).
template <int T> T1L_FN uint32_t pass_type(const Lane& L) { return T < 0 ? L.type : (uint32_t)T; }

// ---- stripe enter: the rows are in the registers (loaded, or zero in the first pass); masks of candidate columns ----
template <int T>
T1L_FN void lane_stripe_enter(Lane& L)
{
    const uint32_t type = pass_type<T>(L);
    if (L.s + 1u == L.ns || L.fresh) { L.S[5] = 0; L.N[5] = 0; }
    L.nr = umin(4u, L.h - 4u * L.s);
    T1L_UNROLL
    for (int r = 0; r < 4; ++r) if ((uint32_t)r >= L.nr) L.P[r] = ~0ull;      // absent rows count as visited
    const uint64_t U = L.S[0] | L.S[1] | L.S[2] | L.S[3] | L.S[4] | L.S[5];
    const uint64_t D = U | (U << 1) | (U >> 1);
    const uint64_t coded = (L.S[1] | L.P[0]) & (L.S[2] | L.P[1]) & (L.S[3] | L.P[2]) & (L.S[4] | L.P[3]);
    uint64_t cm;
    if (type == 0) cm = D & ~coded;                                                         // near something significant, not all coded
    else if (type == 1) cm = (L.S[1] & ~L.P[0]) | (L.S[2] & ~L.P[1]) | (L.S[3] & ~L.P[2]) | (L.S[4] & ~L.P[3]);
    else cm = ~coded;
    L.cm = cm & L.wmask;
    L.Q = (type == 2 && L.nr == 4u) ? ~D & ~(L.P[0] | L.P[1] | L.P[2] | L.P[3]) : 0ull;
    if (type == 1) { L.R[0] = 0; L.R[1] = 0; L.R[2] = 0; L.R[3] = 0; }
    L.st = ST_NEEDCOL;
}

// ---- the next pass of the block (or none): bookkeeping, the first stripe's rows requested --------------------------------------
T1L_FN void lane_next_pass(Lane& L)
{
    L.fresh = 0;
    L.np_left -= 1u;
    if (++L.type == 3u) { L.type = 0; L.bp -= 1; L.pidx += 1u; }
    if (L.np_left == 0u || L.bp < 1) { L.st = ST_DONE; return; }
    L.S[0] = 0; L.N[0] = 0;
    L.s = 0;
    const uint64_t* const np = L.work;
    T1L_UNROLL
    for (int r = 0; r < 4; ++r) { L.S[r + 1] = np[r]; L.N[r + 1] = np[4 + r]; L.P[r] = np[8 + r]; L.M[r] = np[12 + r]; }
    if (1u < L.ns) { L.S[5] = np[16]; L.N[5] = np[20]; }
    L.st = ST_WAIT;
}

// ---- stripe exit: store the stripe; the next stripe's loads, or the end of the pass ----------------------------------------
// Returns with st = ST_WAIT (rows requested: lane_stripe_enter two steps later), ST_PASSWAIT (SYNC: the wave moves on to the
// next pass together, lane_next_pass) or, free-running, whatever lane_next_pass leaves.
template <int T, bool SYNC>
T1L_FN void lane_stripe_exit(Lane& L)
{
    const uint32_t type = pass_type<T>(L);
    uint64_t* const sp = L.work + L.s * 16u;
    if (type == 2) { L.P[0] = 0; L.P[1] = 0; L.P[2] = 0; L.P[3] = 0; }         // the plane is complete (T1.cpp: pi cleared)
    T1L_UNROLL
    for (int r = 0; r < 4; ++r) { sp[r] = L.S[r + 1]; sp[4 + r] = L.N[r + 1]; sp[8 + r] = L.P[r]; sp[12 + r] = L.M[r]; }
    uint64_t* const pl = L.work + kPlaneBase + L.pidx * kPlaneU64 + 4u * L.s;
    if (type == 1) { T1L_UNROLL for (int r = 0; r < 4; ++r) pl[64 + r] = L.R[r]; }
    else           { T1L_UNROLL for (int r = 0; r < 4; ++r) pl[r] = L.S[r + 1]; }
    const uint32_t s = L.s + 1u;
    if (s == L.ns) {
        if (SYNC) L.st = ST_PASSWAIT; else lane_next_pass(L);
        return;
    }
    L.S[0] = L.S[4]; L.N[0] = L.N[4];                     // the row above the next stripe: this stripe's last, as it is now
    L.s = s;
    if (L.fresh) {
        T1L_UNROLL
        for (int r = 1; r < 6; ++r) { L.S[r] = 0; L.N[r] = 0; }
        T1L_UNROLL
        for (int r = 0; r < 4; ++r) { L.P[r] = 0; L.M[r] = 0; }
    } else {
        const uint64_t* const np = L.work + s * 16u;
        T1L_UNROLL
        for (int r = 0; r < 4; ++r) { L.S[r + 1] = np[r]; L.N[r + 1] = np[4 + r]; L.P[r] = np[8 + r]; L.M[r] = np[12 + r]; }
        if (s + 1u < L.ns) { L.S[5] = np[16]; L.N[5] = np[20]; }      // the row below: the next stripe's first, from the pass before
    }
    L.st = ST_WAIT;
}

// ---- the byte stream: request four more bytes / take delivery -----------------------------------------------------------------
T1L_FN bool lane_wants_bytes(const Lane& L) { return L.nv <= 4u && !L.pend && L.st != ST_DONE; }
T1L_FN void lane_fetch_issue(Lane& L)
{
    L.ldw = fetch_dword(L.fsrc, (int32_t)L.fpos, L.lo, L.hi);
    L.pend = 1;
}
T1L_FN void lane_fetch_arrive(Lane& L)
{
    const uint32_t v = mask_dword(L.ldw, (int32_t)L.fpos, L.len);
    const uint64_t keep = L.nv >= 8u ? ~0ull : ((1ull << (8u * L.nv)) - 1ull);
    L.NB = (L.NB & keep) | ((uint64_t)v << (8u * L.nv));
    L.nv += 4u; L.fpos += 4u; L.pend = 0;
}

// 3-bit windows (x-1, x, x+1) of six rows, row r at bits 3r..3r+2
T1L_FN uint32_t extract3(const uint64_t (&rows)[6], uint32_t x)
{
    const uint32_t sh = x ? x - 1u : 0u;
    uint32_t v = 0;
    T1L_UNROLL
    for (int r = 0; r < 6; ++r) v |= ((uint32_t)(rows[r] >> sh) & 7u) << (3 * r);
    return x ? v : ((v << 1) & 0x36DB6u);             // column -1 does not exist
}
// bit x of four rows, row r at bit 3r
T1L_FN uint32_t extract1(const uint64_t (&rows)[4], uint32_t x)
{
    uint32_t v = 0;
    T1L_UNROLL
    for (int r = 0; r < 4; ++r) v |= ((uint32_t)(rows[r] >> x) & 1u) << (3 * r);
    return v;
}
// rows[r + O] |= bx where bit 3r of m is set
template <int O, int NROWS>
T1L_FN void scatter1(uint64_t (&rows)[NROWS], uint32_t m, uint64_t bx)
{
    T1L_UNROLL
    for (int r = 0; r < 4; ++r) rows[r + O] |= ((m >> (3 * r)) & 1u) ? bx : 0ull;
}

// ---- column enter: the next candidate column of the stripe, its neighbourhood words, the first sample to code --------------
template <int T>
T1L_FN void lane_column_enter(Lane& L)
{
    const uint32_t type = pass_type<T>(L);
    if (L.cm == 0) { L.st = ST_NEEDSTRIPE; return; }
    const uint32_t x = ctz64(L.cm);
    const uint64_t bx = 1ull << x;
    L.cm &= ~bx; L.x = x; L.bx = bx;
    if (type == 2 && (L.Q & bx)) {                    // run-length mode (D.3.4): nothing significant around, nothing coded
        L.nbx = 0; L.nnx = 0; L.pv = 0; L.todo = 0; L.st = ST_AGG;
        return;
    }
    const uint32_t nbx = extract3(L.S, x);
    const uint32_t pv = extract1(L.P, x);
    L.nbx = nbx; L.pv = pv;
    const uint32_t sig4 = (nbx >> 4) & 0x249u;
    uint32_t todo;
    if (type == 1) {
        L.mv = extract1(L.M, x); L.rf = 0;
        todo = sig4 & ~pv;
    } else {
        L.nnx = extract3(L.N, x);
        if (type == 0) {
            const uint32_t rowany = nbx | (nbx >> 1) | (nbx >> 2);      // bit 3r: anything in row r's window
            const uint32_t sides = nbx | (nbx >> 2);                    // bit 3r: left or right of row r
            todo = (rowany | (sides >> 3) | (rowany >> 6)) & ~(sig4 | pv) & 0x249u;
        } else todo = ~(sig4 | pv) & 0x249u;
    }
    if (todo == 0) return;                            // (sig-prop: the column mask is a superset) -- next column next iteration
    L.t = ctz32(todo); L.todo = todo & (todo - 1u);
    L.st = type == 1 ? ST_MR : ST_ZC;
}

// the column is finished: its new bits go back into the stripe's rows
template <int T>
T1L_FN void lane_column_exit(Lane& L)
{
    if (T < 0) {
        // (every array addressed on every path: a pointer chosen by the pass type would keep the rows out of registers)
        const bool mr = L.type == 1;
        scatter1<0>(L.M, mr ? L.mv : 0u, L.bx);
        scatter1<0>(L.R, mr ? L.rf : 0u, L.bx);
        scatter1<1>(L.S, mr ? 0u : (L.nbx >> 4) & 0x249u, L.bx);
        scatter1<1>(L.N, mr ? 0u : (L.nnx >> 4) & 0x249u, L.bx);
        scatter1<0>(L.P, L.type == 0 ? L.pv & 0x249u : 0u, L.bx);
    } else if (T == 1) {
        scatter1<0>(L.M, L.mv, L.bx);
        scatter1<0>(L.R, L.rf, L.bx);
    } else {
        scatter1<1>(L.S, (L.nbx >> 4) & 0x249u, L.bx);
        scatter1<1>(L.N, (L.nnx >> 4) & 0x249u, L.bx);
        if (T == 0) scatter1<0>(L.P, L.pv & 0x249u, L.bx);
    }
    L.st = ST_NEEDCOL;
}

// ---- one MQ decision (mqc_dec_inl.h DECODE / RENORMD / BYTEIN).  lds32: the wave's LDS block as dwords ------------------------
T1L_FN uint32_t lane_mq_decode(Lane& L, uint32_t* lds32, uint32_t ctx_dword)
{
    const uint32_t row = lds32[ctx_dword];
    const uint32_t qe = row & 0xFFFF0000u;
    uint32_t A = L.A - qe;
    const bool isL = L.C < qe, X = A < qe;
    const bool flip = isL != X;
    const uint32_t d = ((row >> 14) & 1u) ^ (flip ? 1u : 0u);
    A = isL ? qe : A;
    uint32_t C = isL ? L.C : L.C - qe;
    if (!(A & 0x80000000u)) {                          // RENORMD; the context moves on
        const uint32_t ns = flip ? (row >> 7) & 0x7Fu : row & 0x7Fu;
        lds32[ctx_dword] = lds32[(kOffMq >> 2) + ns];
        uint32_t n = clz32(A);
        A <<= n;
        uint32_t ct = L.ct;
        uint32_t s = umin(n, ct);
        C <<= s; ct -= s; n -= s;
        while (n) {                                    // BYTEIN with ct == 0
            const uint32_t cur = (uint32_t)L.NB & 0xFFu, nxt = (uint32_t)(L.NB >> 8) & 0xFFu;
            const bool ff = cur == 0xFFu, mark = ff && nxt > 0x8Fu;
            C += mark ? 0xFF00u : (nxt << (ff ? 9 : 8));
            ct = (ff && !mark) ? 7u : 8u;
            if (!mark) { L.NB >>= 8; L.nv -= 1u; }
            s = umin(n, ct);
            C <<= s; ct -= s; n -= s;
        }
        L.ct = ct;
    }
    L.A = A; L.C = C;
    return d;
}

// ---- which context the lane's decision uses and what the decision means (T1.cpp:854-1255), without branches: selects on
// the lane's mode -- every lane of a wave runs every path anyway.  lds16: the wave's LDS block as uint16_t[].
template <int T>
T1L_FN uint32_t lane_context(Lane& L, const uint16_t* lds16)
{
    const uint32_t nb = L.nbx >> L.t;
    if (T == 1) return ((L.mv >> L.t) & 1u) ? 16u * 256u : ((nb & 0x1EFu) ? 15u * 256u : 14u * 256u);      // Table D.4
    const uint32_t nn = L.nnx >> L.t;
    const bool isSC = L.st == ST_SC;
    const uint32_t zi = (L.zcbase >> 1) + (nb & 0x1FFu);
    const uint32_t si = (kOffSc >> 1) + (((nb & 0xAAu) >> 1) | (nn & 0xAAu));
    const uint32_t e = lds16[isSC ? si : zi];                         // one look-up for both kinds (others: a harmless read)
    L.xr = isSC ? (e & 1u) : L.xr;
    if (T == 0) return e & ~1u;                                       // sig-prop: zero coding or sign, nothing else
    const uint32_t mr = ((L.mv >> L.t) & 1u) ? 16u * 256u : ((nb & 0x1EFu) ? 15u * 256u : 14u * 256u);
    uint32_t off = L.st == ST_AGG ? 17u * 256u : 18u * 256u;
    if (T < 0) off = L.st == ST_MR ? mr : off;
    off = L.st <= ST_SC ? (e & ~1u) : off;
    return off;
}

template <int T>
T1L_FN void lane_apply(Lane& L, uint32_t d)
{
    const uint32_t st = L.st, t = L.t;
    const uint32_t bit = 1u << t;
    if (T == 1) {                                                     // mag-ref: the bit, then the column's next sample
        L.rf |= d << t; L.mv |= bit;
        const uint32_t todo = L.todo;
        if (todo) { L.t = ctz32(todo); L.todo = todo & (todo - 1u); }
        else lane_column_exit<T>(L);
        return;
    }
    const bool isZC = st == ST_ZC, isSC = st == ST_SC, isMR = T < 0 && st == ST_MR;
    const bool isAGG = T != 0 && st == ST_AGG, isU1 = T != 0 && st == ST_UNI1, isU2 = T != 0 && st == ST_UNI2;
    const bool sp = pass_type<T>(L) == 0, mrp = T < 0 && L.type == 1;
    const bool one = d != 0;
    // visited (sig-prop: every sample the pass looks at)
    L.pv |= (sp && (isSC || (isZC && !one))) ? bit : 0u;
    // a sample turns significant: sign decoded
    const uint32_t neg = d ^ L.xr;
    L.nbx |= isSC ? bit << 4 : 0u;
    L.nnx |= (isSC && neg) ? bit << 4 : 0u;
    uint32_t todo = L.todo | ((isSC && sp) ? ((bit << 3) & ~((L.nbx >> 4) | L.pv) & 0x249u) : 0u);
    const uint64_t bx1 = L.bx << 1;
    if (T <= 0) L.cm |= (isSC && sp) ? (bx1 & L.wmask) : 0ull;
    if (T != 0) L.Q &= (isSC && !sp) ? ~bx1 : ~0ull;
    // refinement (free-running mode only: the specialised mag-ref pass is above)
    if (T < 0) { L.rf |= isMR ? d << t : 0u; L.mv |= isMR ? bit : 0u; }
    // run-length position
    const uint32_t r2 = L.r * 2u + d, t2 = r2 * 3u;
    L.r = isU1 ? d : L.r;
    // where next
    const bool adv = isSC || isMR || (isZC && !one);
    const bool more = todo != 0u;
    const uint32_t tn = ctz32(todo | 0x80000000u), todon = todo & (todo - 1u);
    uint32_t nst = (mrp ? (uint32_t)ST_MR : (uint32_t)ST_ZC);             // adv && more
    nst = (adv && !more) ? (uint32_t)ST_NEEDCOL : nst;
    nst = (isZC && one) ? (uint32_t)ST_SC : nst;
    nst = isAGG ? (one ? (uint32_t)ST_UNI1 : (uint32_t)ST_NEEDCOL) : nst;
    nst = isU1 ? (uint32_t)ST_UNI2 : nst;
    nst = isU2 ? (uint32_t)ST_SC : nst;
    L.t = (adv && more) ? tn : (isU2 ? t2 : t);
    L.todo = (adv && more) ? todon : (isU2 ? (0x249u & ~((2u << t2) - 1u)) : todo);
    L.st = nst;
    if (adv && !more) lane_column_exit<T>(L);
}

// ---- reconstruction of one sample from the planes a block left (t1_recon_kernel; the values T1 keeps in its data array) --------
// snap(i) / ref(i): the sample's bit in plane i's significance / refinement bitmap; numbps, numpasses as decoded.
// Returns the magnitude in T1's fixed point (one fraction bit: the block leaves v / 2 or v * stepsize / 2).
template <class FS, class FR>
T1L_FN uint32_t recon_magnitude(uint32_t numbps, uint32_t numpasses, FS snap, FR ref)
{
    // passes: cleanup of plane numbps, then (sig-prop, mag-ref, cleanup) per plane, while bp >= 1
    uint32_t mag = 0; bool prev = false;
    for (uint32_t i = 0; i < numbps && 3u * i < numpasses + 2u; ++i) {        // plane i has a pass: 1 + 3 (i - 1) < numpasses
        if (i && 1u + 3u * (i - 1u) >= numpasses) break;
        const uint32_t p = numbps - i;                                        // one = 1 << p
        const bool cur = snap(i);
        if (cur && !prev) mag = 3u << (p - 1u);                               // oneplushalf
        else if (prev && 3u * i <= numpasses) mag = ref(i) ? mag + (1u << (p - 1u)) : mag - (1u << (p - 1u));   // mag-ref ran: pass 3 i - 1
        prev = cur;
    }
    return mag;
}

} // namespace t1l
