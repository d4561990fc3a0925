// tests/c/t1_lanes_sim.cpp -- TEST INFRASTRUCTURE: the lane logic of the one-block-per-lane Part-1 decoder
// (grok_amd/csrc/t1_lanes.h, compiled for the host) stepped through the kernel's phases on the CPU, 64 lanes to a "wave",
// so that the state machine can be compared with the oracle without a GPU.  Mirrors the loop of kernels_t1lanes.hip.
#include "../../grok_amd/csrc/t1_lanes.h"
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdio>
#include <type_traits>

using namespace t1l;

extern "C" {

struct SimBlock { uint64_t offset; uint32_t length, numbps, numpasses, w, h, orient; };

// out: [nblocks][64 * 64] decoded values (T1's data array: sign * magnitude with one fraction bit), row stride 64
// stats: [0] iterations summed over waves, [1] decisions, [2] lane-slots without a decision, [3] max iterations of a wave, [4] decision slots
int t1l_sim_decode(const uint8_t* coded, uint64_t coded_bytes, uint32_t nblocks, const SimBlock* blk, int32_t* out, uint64_t* stats)
{
    std::vector<uint8_t> lds(kLdsBytes);
    uint32_t* const lds32 = reinterpret_cast<uint32_t*>(lds.data());
    uint16_t* const lds16 = reinterpret_cast<uint16_t*>(lds.data());
    for (uint32_t e = 0; e < 94; ++e) lds32[(kOffMq >> 2) + e] = mq_entry(e);
    for (int o = 0; o < 4; ++o)
        for (uint32_t i = 0; i < 512; ++i) lds16[(kOffZc >> 1) + o * 512 + i] = (uint16_t)(zc_context9(o, i) * 256u);
    for (uint32_t i = 0; i < 256; ++i) lds16[(kOffSc >> 1) + i] = (uint16_t)sign_context(i);
    std::vector<uint64_t> work((size_t)64 * kWorkU64);
    FILE* trace = std::getenv("T1L_SIM_TRACE") ? std::fopen(std::getenv("T1L_SIM_TRACE"), "w") : nullptr;
    const uint32_t kslots_max = std::getenv("T1L_KMAX") ? (uint32_t)std::atoi(std::getenv("T1L_KMAX")) : 1u;
    const uint32_t kslots_min = std::getenv("T1L_KMIN") ? (uint32_t)std::atoi(std::getenv("T1L_KMIN")) : 1u;
    const uint32_t frac_pct = std::getenv("T1L_FRAC") ? (uint32_t)std::atoi(std::getenv("T1L_FRAC")) : 0u;
    const bool free_running = std::getenv("T1L_FREE") != nullptr;      // lanes at their own pace instead of pass by pass
    uint64_t slots_total = 0;
    uint64_t it_total = 0, dec_total = 0, idle_total = 0, it_max = 0;
    for (uint32_t base = 0; base < nblocks; base += 64) {
        const uint32_t nl = nblocks - base < 64 ? nblocks - base : 64;
        Lane L[64];
        // (the work areas are NOT cleared: the kernel must not depend on their contents)
        for (size_t i = 0; i < work.size(); ++i) work[i] = 0xDEADBEEFCAFEF00Dull;
        for (uint32_t l = 0; l < 64; ++l) {
            if (l >= nl) { L[l].st = ST_DONE; L[l].nv = 8; L[l].pend = 0; continue; }
            const SimBlock& b = blk[base + l];
            BlockIn in;
            in.data = coded + b.offset; in.len = b.length; in.numbps = b.numbps; in.numpasses = b.numpasses;
            in.w = b.w; in.h = b.h; in.orient = b.orient; in.work = work.data() + (size_t)l * kWorkU64;
            in.lo = coded; in.hi = coded + coded_bytes;
            lane_init(L[l], in);
            // mqc_resetstates: every context in state 0 but UNI (46), AGG (3), ZC 0 (4)
            for (uint32_t cx = 0; cx < 19; ++cx) lds32[cx * 64 + l] = mq_entry(cx == 18 ? 46u : cx == 17 ? 3u : cx == 0 ? 4u : 0u);
        }
        uint64_t it = 0;
        // one step of every lane: column enter where needed, then the decision slots of this step (policy: experiments)
        auto steps = [&](auto TT) {
            constexpr int T = decltype(TT)::value;
            for (uint32_t l = 0; l < 64; ++l) if (L[l].st == ST_NEEDCOL) lane_column_enter<T>(L[l]);
            for (uint32_t slot = 0;; ++slot) {
                uint32_t can = 0, live = 0;
                for (uint32_t l = 0; l < 64; ++l) { if (L[l].st < ST_DONE) ++live; if (L[l].st <= ST_UNI2 && L[l].nv >= 3u) ++can; }
                if (slot >= kslots_max) break;
                if (slot >= kslots_min && can * 100u < live * frac_pct) break;
                if (can == 0) break;
                ++slots_total;
                for (uint32_t l = 0; l < 64; ++l) {
                    if (L[l].st <= ST_UNI2 && L[l].nv >= 3u) {
                        const uint32_t off = lane_context<T>(L[l], lds16);
                        const uint32_t d = lane_mq_decode(L[l], lds32, (off >> 2) + l);
                        if (trace && l == 0) std::fprintf(trace, "%u %u\n", off >> 8, d);
                        lane_apply<T>(L[l], d);
                        ++dec_total;
                    } else if (L[l].st < ST_DONE) ++idle_total;
                }
            }
            ++it;
        };
        // a round = four steps, stripes stored / requested before the first, delivered before the third (kernels_t1lanes.hip)
        auto round = [&](auto TT, auto SS) {
            constexpr int T = decltype(TT)::value;
            constexpr bool SYNC = decltype(SS)::value;
            for (uint32_t l = 0; l < 64; ++l) {
                if (L[l].st == ST_NEEDSTRIPE) lane_stripe_exit<T, SYNC>(L[l]);
                if (lane_wants_bytes(L[l])) lane_fetch_issue(L[l]);
            }
            steps(TT); steps(TT);
            for (uint32_t l = 0; l < 64; ++l) {
                if (L[l].st == ST_WAIT) lane_stripe_enter<T>(L[l]);
                if (L[l].pend) lane_fetch_arrive(L[l]);
            }
            steps(TT); steps(TT);
        };
        auto active = [&]() { for (uint32_t l = 0; l < 64; ++l) if (L[l].st < ST_DONE) return true; return false; };
        if (free_running) {                       // every lane at its own pace through its passes (the lane's own L.type)
            while (active()) { round(std::integral_constant<int, -1>{}, std::false_type{}); if (it > 4000000) return -1; }
        } else {                                  // the kernel's form: the wave's lanes go from pass to pass together
            uint32_t T = 2;
            for (;;) {
                while (active()) {
                    if (T == 0) round(std::integral_constant<int, 0>{}, std::true_type{});
                    else if (T == 1) round(std::integral_constant<int, 1>{}, std::true_type{});
                    else round(std::integral_constant<int, 2>{}, std::true_type{});
                    if (it > 4000000) return -1;
                }
                bool any = false;
                for (uint32_t l = 0; l < 64; ++l) if (L[l].st == ST_PASSWAIT) { lane_next_pass(L[l]); any = true; }
                if (!any) break;
                T = (T + 1u) % 3u;
            }
        }
        it_total += it; if (it > it_max) it_max = it;
        // reconstruction
        for (uint32_t l = 0; l < nl; ++l) {
            const SimBlock& b = blk[base + l];
            const uint64_t* wk = work.data() + (size_t)l * kWorkU64;
            int32_t* o = out + (size_t)(base + l) * 4096;
            for (uint32_t y = 0; y < b.h; ++y)
                for (uint32_t x = 0; x < b.w; ++x) {
                    auto snap = [&](uint32_t i) { return ((wk[kPlaneBase + i * kPlaneU64 + y] >> x) & 1u) != 0; };
                    auto ref = [&](uint32_t i) { return ((wk[kPlaneBase + i * kPlaneU64 + 64 + y] >> x) & 1u) != 0; };
                    const uint32_t mag = recon_magnitude(b.numbps, b.numpasses, snap, ref);
                    const bool neg = (wk[(y >> 2) * 16 + 4 + (y & 3)] >> x) & 1u;
                    o[y * 64 + x] = neg ? -(int32_t)mag : (int32_t)mag;
                }
        }
    }
    if (trace) std::fclose(trace);
    if (stats) { stats[0] = it_total; stats[1] = dec_total; stats[2] = idle_total; stats[3] = it_max; stats[4] = slots_total; }
    return 0;
}

}
