// grok_amd/csrc/kernels_t1lanes.hip -- K8L: Part-1 (EBCOT / MQ) block decoding with one code-block per LANE, gfx950.
//
// The bulk of a frame's blocks (default code-block style, one codeword segment, 9..64 rows, at most 14 bit-planes) is
// decoded 64 blocks to a wave by t1_lanes_kernel; the lane logic and the reasoning are in t1_lanes.h.  The few blocks that
// are much longer than the rest (a frame's LL band: 17 bit-planes of 4096 mag-ref decisions each in BASELINE configs[4]) are one
// dependent chain of 5-7 times the decisions, and a chain is faster alone in a wave: those, and everything the lane form does
// not take (code-block styles, segments, tiny blocks), stay with kernels_t1dec.hip (K8), which runs beside this kernel.
//   t1_lanes_kernel  -- lanes = blocks of `list` (sorted by coded length, longest first: a wave's lanes finish together).
//                       Wave-uniform loop, at most one MQ decision per lane and iteration; a block's state between stripes
//                       and its per-plane result bitmaps live in its 16 KB of `work`.
//   t1_recon_kernel  -- one wave per block of `list`: rows of the bitmaps loaded lane-per-row, then lanes = columns: significance / refinement bitmaps -> coefficients,
//                       dequantised (ShiftFilter / ScaleFilter, filters/PostDecompressFilters.h:26-35, :60-71) into the Mallat plane.
#include "kernels.h"
#include "t1_lanes.h"
#include <type_traits>
// the wave-per-block decoder's body (t1_dec_block) for the fused launch below
#define GRK_T1_FUSED_INCLUDE
#include "kernels_t1dec.hip"
#undef GRK_T1_FUSED_INCLUDE

#ifndef T1L_ROUND_STEPS
#define T1L_ROUND_STEPS 6        // steps between two stripe hand-overs (4, 6 or 8: fewer hand-over passes, lanes wait longer for theirs)
#endif

namespace grk_amd {

namespace {

using namespace t1l;

struct LaneTables {
    uint32_t mq[96];
    uint16_t zc[4][512];
    uint16_t sc[256];
    constexpr LaneTables() : mq{}, zc{}, sc{}
    {
        for (uint32_t e = 0; e < 94; ++e) mq[e] = t1l::mq_entry(e);
        for (int o = 0; o < 4; ++o)
            for (uint32_t i = 0; i < 512; ++i) zc[o][i] = (uint16_t)(t1l::zc_context9(o, i) * 256u);
        for (uint32_t i = 0; i < 256; ++i) sc[i] = (uint16_t)t1l::sign_context(i);
    }
};
static_assert(sizeof(LaneTables) == kLdsBytes - kCtxBytes, "the tables follow the context rows in LDS");
__device__ const LaneTables g_lane_tables{};

// SYNC: the wave's lanes run their passes in step, one specialised copy of the loop per pass type (t1_lanes.h) -- the host puts
// blocks with the same number of bit-planes and passes into a wave.  !SYNC: every lane at its own pace, one general loop.
// (188 vector registers, two waves per SIMD.  A version with packed lane state -- 124 / 154 registers -- was measured and lost
//  15 % to the extra instructions: profiles/r04_hw_queues.txt; the limit of a decode sequence was the hardware queues.)
template <bool SYNC>
__device__ __forceinline__ void t1_lanes_wave(const T1LaneArgs& a, const uint32_t wave)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds32[kLdsBytes / 4];
    uint16_t* const lds16 = reinterpret_cast<uint16_t*>(lds32);
    const uint32_t lane = threadIdx.x;
    {   // the look-up tables behind the context rows
        const uint32_t* const src = reinterpret_cast<const uint32_t*>(&g_lane_tables);
        for (uint32_t i = lane; i < (kLdsBytes - kCtxBytes) / 4; i += 64) lds32[kCtxBytes / 4 + i] = src[i];
    }
    // mqc_resetstates (mqc_dec.cpp:168-175): every context in state 0 but UNI (46), AGG (3), ZC 0 (4)
#pragma unroll
    for (uint32_t cx = 0; cx < 19; ++cx) lds32[cx * 64 + lane] = mq_entry(cx == 18 ? 46u : cx == 17 ? 3u : cx == 0 ? 4u : 0u);
    Lane L;
    const uint32_t idx = wave * 64u + lane;
    const uint32_t blk = idx < a.count ? a.list[idx] : kT1NoBlock;
    if (blk != kT1NoBlock) {
        const HtDecBlock in = a.table[blk];
        const HtBlockDesc bd = a.blocks[blk % a.blocks_per_tile];
        BlockIn b;
        b.data = a.coded + in.offset; b.len = in.length;
        b.numbps = in.missing_msbs & 0xFFu; b.numpasses = in.missing_msbs >> 8;
        b.w = bd.w; b.h = bd.h; b.orient = bd.pad;
        b.work = a.work + (size_t)blk * kWorkU64;
        b.lo = a.coded; b.hi = a.coded + a.coded_bytes;
        lane_init(L, b);
    } else {                                       // (a wave's spare lanes: the host fills waves per group of equal blocks)
        L = Lane{};
        L.st = ST_DONE; L.nv = 8; L.pend = 0;
    }
    __syncthreads();
    // one step: the lanes that need one pick their next column; every lane with a pending decision makes it.
    // A round is T1L_ROUND_STEPS (six) steps: stripes stored / the next ones and coded bytes requested before the first, delivery
    // before the fourth (six against four: 1-3 % per frame in a decode sequence, the same for one frame alone; r04_hw_queues.txt).
    auto round = [&](auto TT) {
        constexpr int T = decltype(TT)::value;
        auto step = [&]() {
            if (L.st == ST_NEEDCOL) lane_column_enter<T>(L);
            if (L.st <= ST_UNI2 && L.nv >= 3u) {
                const uint32_t off = lane_context<T>(L, lds16);
                const uint32_t d = lane_mq_decode(L, lds32, (off >> 2) + lane);
                lane_apply<T>(L, d);
            }
        };
        if (L.st == ST_NEEDSTRIPE) lane_stripe_exit<T, SYNC>(L);
        if (lane_wants_bytes(L)) lane_fetch_issue(L);
        step(); step();
#if T1L_ROUND_STEPS >= 6
        step();
#endif
        if (L.st == ST_WAIT) lane_stripe_enter<T>(L);
        if (L.pend) lane_fetch_arrive(L);
        step(); step();
#if T1L_ROUND_STEPS >= 6
        step();
#endif
#if T1L_ROUND_STEPS >= 8
        step(); step();
#endif
    };
    if constexpr (SYNC) {
        uint32_t T = 2;                            // every block starts with the cleanup pass of its top plane
        for (;;) {
            if (T == 0) { do round(std::integral_constant<int, 0>{}); while (__builtin_amdgcn_ballot_w64(L.st < ST_DONE) != 0); }
            else if (T == 1) { do round(std::integral_constant<int, 1>{}); while (__builtin_amdgcn_ballot_w64(L.st < ST_DONE) != 0); }
            else { do round(std::integral_constant<int, 2>{}); while (__builtin_amdgcn_ballot_w64(L.st < ST_DONE) != 0); }
            if (__builtin_amdgcn_ballot_w64(L.st == ST_PASSWAIT) == 0) break;
            if (L.st == ST_PASSWAIT) lane_next_pass(L);
            T = T == 2 ? 0u : T + 1u;
        }
    } else {
        do round(std::integral_constant<int, -1>{}); while (__builtin_amdgcn_ballot_w64(L.st < ST_DONE) != 0);
    }
}

template <bool SYNC>
__global__ __launch_bounds__(64) void t1_lanes_kernel(T1LaneArgs a) { t1_lanes_wave<SYNC>(a, blockIdx.x); }

// ONE launch for a frame's block decoding (r06): workgroups [0, d.count) are the wave-per-block decoder's blocks -- the long chains, so
// they start first --, the rest the lane decoder's waves.  Two launches on two streams (r04-r05) cost a frame in a decode SEQUENCE two of
// the runtime's four hardware queues per priority level; streams that share a queue run their kernels in turn.
template <bool IRREV, bool SYNC>
__global__ __launch_bounds__(64) void t1_fused_kernel(T1DecArgs d, T1LaneArgs l)
{
    if (blockIdx.x < d.count) t1_dec_block<IRREV>(d, blockIdx.x);
    else t1_lanes_wave<SYNC>(l, blockIdx.x - d.count);
}

// One wave per block.  Lane y first loads ROW y of every bitmap the block left (coalesced 512-byte loads, all in flight at once:
// the sign rows and, per bit-plane, the significance rows at the end of the plane and the mag-ref bits); then the wave goes through
// the rows, lane x taking bit x of each row's words as they are broadcast from lane y (v_readlane) -- no memory access on the chain.
template <bool IRREV>
__global__ __launch_bounds__(64) void t1_recon_kernel(T1LaneArgs a)
{
    const uint32_t blk = a.list[blockIdx.x];
    if (blk == kT1NoBlock) return;                 // (a spare lane of a lane-decoder wave)
    const HtDecBlock in = a.table[blk];
    const HtBlockDesc bd = a.blocks[blk % a.blocks_per_tile];
    const uint32_t numbps = in.missing_msbs & 0xFFu, numpasses = in.missing_msbs >> 8;
    const uint64_t* const wk = a.work + (size_t)blk * kWorkU64;
    const uint32_t x = threadIdx.x;
    const uint32_t tile = blk / a.blocks_per_tile;
    int32_t* const dst = a.mallat + ((size_t)tile * a.ncomp + bd.comp) * a.pitch + (size_t)bd.py * a.stride + bd.px;
    const float scale = bd.inv_step / 2;
    // planes with a pass: plane i has one iff 1 + 3 (i - 1) < numpasses (i = 0: the first cleanup); at most kMaxPlanes
    uint32_t nplanes = 0;
    for (uint32_t i = 0; i < numbps && i < kMaxPlanes && (i == 0 || 3u * i - 2u < numpasses); ++i) nplanes = i + 1u;
    uint64_t snap[kMaxPlanes], ref[kMaxPlanes];
    const uint64_t neg = wk[(x >> 2) * 16u + 4u + (x & 3u)];
#pragma unroll
    for (uint32_t i = 0; i < kMaxPlanes; ++i) {
        snap[i] = i < nplanes ? wk[kPlaneBase + i * kPlaneU64 + x] : 0ull;
        ref[i] = i < nplanes ? wk[kPlaneBase + i * kPlaneU64 + 64u + x] : 0ull;
    }
    auto row_bit = [&](uint64_t v, uint32_t y) -> bool {           // bit x of lane y's word
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)y);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)y);
        return (((x < 32u ? lo : hi) >> (x & 31u)) & 1u) != 0;
    };
    for (uint32_t y = 0; y < bd.h; ++y) {
        uint32_t mag = 0; bool prev = false;
#pragma unroll
        for (uint32_t i = 0; i < kMaxPlanes; ++i) {
            if (i < nplanes) {                                       // (uniform)
                const uint32_t p = numbps - i;
                const bool cur = row_bit(snap[i], y);
                const bool rb = row_bit(ref[i], y);
                const bool mr_ran = 3u * i <= numpasses && i > 0;
                const uint32_t half = 1u << (p - 1u);
                mag = (cur && !prev) ? 3u << (p - 1u) : ((prev && mr_ran) ? (rb ? mag + half : mag - half) : mag);
                prev = cur;
            }
        }
        const bool ng = row_bit(neg, y);
        const int32_t v = ng ? -(int32_t)mag : (int32_t)mag;
        int32_t o;
        if constexpr (IRREV) o = __float_as_int(__fmul_rn((float)v, scale));
        else o = v / 2;
        if (x < bd.w) dst[(size_t)y * a.stride + x] = o;
    }
}

} // namespace

hipError_t launch_t1_fused(const T1DecArgs& d, const T1LaneArgs& a, hipStream_t s)
{
    if (!a.count || !d.count || !d.list) return hipErrorInvalidValue;
    const uint32_t grid = d.count + (a.count + 63u) / 64u;
    if (d.irreversible) {
        if (a.pass_sync) hipLaunchKernelGGL((t1_fused_kernel<true, true>), dim3(grid), dim3(64), 0, s, d, a);
        else hipLaunchKernelGGL((t1_fused_kernel<true, false>), dim3(grid), dim3(64), 0, s, d, a);
    } else {
        if (a.pass_sync) hipLaunchKernelGGL((t1_fused_kernel<false, true>), dim3(grid), dim3(64), 0, s, d, a);
        else hipLaunchKernelGGL((t1_fused_kernel<false, false>), dim3(grid), dim3(64), 0, s, d, a);
    }
    if (a.irreversible) hipLaunchKernelGGL(t1_recon_kernel<true>, dim3(a.count), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(t1_recon_kernel<false>, dim3(a.count), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_t1_lanes(const T1LaneArgs& a, hipStream_t s)
{
    if (!a.count) return hipSuccess;
    if (a.pass_sync) hipLaunchKernelGGL(t1_lanes_kernel<true>, dim3((a.count + 63u) / 64u), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(t1_lanes_kernel<false>, dim3((a.count + 63u) / 64u), dim3(64), 0, s, a);
    if (a.irreversible) hipLaunchKernelGGL(t1_recon_kernel<true>, dim3(a.count), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(t1_recon_kernel<false>, dim3(a.count), dim3(64), 0, s, a);
    return hipGetLastError();
}

} // namespace grk_amd
