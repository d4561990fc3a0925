// grok_amd/csrc/context.hip -- implementation of the C-ABI in include/grok_amd.h.
//
// One grk_amd_ctx per process per GPU.  All device memory is owned by the context and grows
// monotonically (288 GB of HBM3E: an 8K x 8K x 3 tile needs ~3.6 GB of working planes, a batch of
// 256 1024^2 tiles ~12 GB), so steady-state encode calls perform no allocation and enqueue
// nothing but kernels on one HIP stream.
#include "../../include/grok_amd.h"
#include "geometry.h"
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <cmath>
#include <atomic>
#include <thread>

using namespace grk_amd;

namespace {

// (Streams and hardware queues: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 unless
//  the variable says otherwise, and kernels of two streams that share a queue run one after the other.  A decode sequence with
//  three or more frames in flight -- two streams of long kernels each -- gains nothing over two frames on 4 queues; on 8 the
//  Part-1 sequence goes from 9.1 to 6.7 ms per frame.  It is the HOST's setting, process-wide and read when the runtime starts:
//  the library does not touch it, because the encode pipeline beside an RCCL exchange was measured 25 % slower on anything but
//  4 (profiles/r04_hw_queues.txt).)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = n + (n >> 3) + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Timer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double total_ms = 0; uint32_t launches = 0;
};

// Pinned staging for the host-pointer entry points (pixels in, coded bytes / pixels out).  A buffer that IS pinned
// (grk_amd_host_alloc, hipHostMalloc, a pinned torch tensor) goes over the link as it lies -- one DMA at the link's rate.
// Pageable memory is moved through context-owned pinned chunks by kLanes copy threads, each double-buffered on its own
// stream (a memcpy into / out of one chunk while the other chunk's DMA runs): the threads' memcpy rate adds up, where one
// thread -- what a plain hipMemcpy of pageable memory amounts to -- is the limit otherwise.
struct HostStage {
    static constexpr size_t kChunk = 8u << 20;
    static constexpr int kLanes = 4;
    void* buf[kLanes][2] = {};
    hipEvent_t ev[kLanes][2] = {};
    hipEvent_t ev_in = nullptr, ev_out[kLanes] = {};
    hipStream_t st[kLanes] = {};
    bool ready = false;
    hipError_t ensure()
    {
        if (ready) return hipSuccess;
        hipError_t e = hipEventCreateWithFlags(&ev_in, hipEventDisableTiming);
        for (int t = 0; t < kLanes && e == hipSuccess; ++t) {
            e = hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_out[t], hipEventDisableTiming);
            for (int k = 0; k < 2 && e == hipSuccess; ++k) {
                e = hipHostMalloc(&buf[t][k], kChunk, hipHostMallocDefault);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[t][k], hipEventDisableTiming);
            }
        }
        ready = e == hipSuccess;
        return e;
    }
    void release()
    {
        for (int t = 0; t < kLanes; ++t) {
            if (st[t]) { (void)hipStreamSynchronize(st[t]); (void)hipStreamDestroy(st[t]); st[t] = nullptr; }
            if (ev_out[t]) { (void)hipEventDestroy(ev_out[t]); ev_out[t] = nullptr; }
            for (int k = 0; k < 2; ++k) {
                if (buf[t][k]) { (void)hipHostFree(buf[t][k]); buf[t][k] = nullptr; }
                if (ev[t][k]) { (void)hipEventDestroy(ev[t][k]); ev[t][k] = nullptr; }
            }
        }
        if (ev_in) { (void)hipEventDestroy(ev_in); ev_in = nullptr; }
        ready = false;
    }
};

} // namespace

struct grk_amd_ctx {
    int device = 0;
    int verbose = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // working set
    DevBuf pixels, p0, p1, llA, llB, blockdesc, lengths, offsets, arena, flag;
    DevBuf dec_desc, dec_table, dec_quads, dec_mslen, dec_coded, dec_pixels, dec_work;
    // geometry cache
    grk_amd_tile_params gp{};
    bool have_geom = false;
    TileGeom geom;
    std::vector<HtBlockDesc> h_desc, h_desc_dec;
    std::vector<uint16_t> dec_qcd;                          // decode: QCD words of a foreign stream (optional)
    std::vector<float> dec_steps;                           // decode: band step sizes as the host holds them (optional), [comp][band]
    std::vector<uint32_t> dec_seg_first;                    // Part-1 decode: codeword segments (optional), [nblocks + 1]
    std::vector<grk_amd_segment> dec_segs;
    DevBuf dec_seg_dev;
    HtClass ht_classes[kHtMaxClasses]; uint32_t ht_num_classes = 0;   // block classes of K3: {top resolution, rest} x {LDS small, large}
    uint8_t ht_class_top[kHtMaxClasses] = {}, ht_class_big[kHtMaxClasses] = {};
    int seq_index = -1;               // >= 0: one of a decode sequence's internal contexts (grk_amd_set_decode_pipelining)
    int seq_flavour = 0;              // ... whose two streams are made for 0: HT frames, 1: Part-1 frames (sequence_streams)
    hipStream_t side2 = nullptr; hipEvent_t ev_side2 = nullptr;      // the large-LDS classes run beside the small-LDS ones
    hipStream_t side = nullptr;                             // K3 of the top resolution runs here beside DWT levels >= 1
    hipEvent_t ev_level0 = nullptr, ev_side = nullptr;
    bool overlap = false;
    // Pipelining of consecutive encodes (grk_amd_set_pipelining): a second set of per-encode buffers, so that the next
    // encode's DWT can start while the side streams still code the blocks of this one
    static constexpr int kMaxAltSets = 7;
    struct AltSet { DevBuf p1, arena, lengths, offsets, flag, ovf, llA, llB; hipEvent_t ev_side = nullptr, ev_side2 = nullptr; } alts[kMaxAltSets];
    int alt_head = 0;                // the OLDEST of the sets not in use (a ring: the set a call retires becomes the newest)
    int pipe_depth = 2;              // buffer sets in rotation when pipelining: grk_amd_set_pipelining(ctx, n) -> n + 1 of them (2 ..
                                     // 8): the results of a call then stay valid until the (n + 1)-th next call
    DevBuf ovf;                      // K3: blocks handed to the fallback launch (kernels.h: HtArgs::ovf_list)
    bool lds_cap = true;             // K3 with capped LDS buffers + fallback launch (GRK_AMD_LDS_CAP=0: worst-case buffers)
    bool pipelining = false;
    hipEvent_t ev_main = nullptr;
    bool side_pending = false;       // side-stream work of the latest encode has not been joined on the main stream yet
    bool dec_planes16 = true;                               // 16-bit planes between K5b and K6 for 8-bit reversible HT tiles
                                                            // (GRK_AMD_DEC_PLANES16=0 / grk_amd_set_decode_planes16: int32)
    int dwt_pk = 1;                                         // packed int16 pairs in K2 / K6 where the range allows (GRK_AMD_DWT_PK=0: 32-bit)
    int dwt_xcd = 1;                                        // XCD-aware workgroup order in K2 / K6 (GRK_AMD_DWT_XCD=0: plain)
    bool fuse_egress = true;                                // K7 inside the last inverse DWT level (GRK_AMD_FUSE_EGRESS=0: separate)
    bool planes16 = true;                                   // int16 planes between K2 and K3 where the range allows (GRK_AMD_PLANES16=0: never)
    DevBuf ht_sel;
    DevBuf energy;                   // grk_amd_block_distortion: sum of q^2 per block
    // Tier-2 on the device (grk_amd_assemble_device): the packets of the geometry in the order last asked for, scratch, and the
    // finished tile-parts
    struct T2State {
        bool valid = false; grk_amd_tile_params p{}; uint32_t order = 0;
        T2Plan plan; uint32_t max_blocks = 0;
        DevBuf packets, pob;
    } t2;
    DevBuf t2_u, t2_h, t2_rel, t2_pkhdr, t2_pkbody, t2_pkdst, t2_lit, t2_litlen, t2_index;      // scratch: ONE stream at a time uses it
    // the finished tile-parts, their places and lengths ([tile]: uint64 / uint32) and {bytes assembled by the call, end of the output};
    // the asynchronous form rotates as many of these as the encoder rotates buffer sets, so that a frame's tile-parts stay where they
    // are while an exchange sends them
    struct T2Out { DevBuf out, tile_dst, part_len, total; };
    T2Out t2_outs[kMaxAltSets + 1];
    int t2_cur = 0;
    uint64_t t2_out_used = 0;        // (the synchronous form) bytes of t2_outs[t2_cur].out that hold tile-parts
    std::vector<uint64_t> h_off;
    std::vector<uint32_t> h_len;
    uint32_t last_ntiles = 0;
    uint64_t last_nblocks = 0;
    bool last_h16 = false;           // the latest encode left int16 coefficients in the Mallat planes
    HostStage stage;                 // pinned chunks for pageable host buffers (copy_h2d)
    void* d2h_pin = nullptr; size_t d2h_cap = 0; std::vector<hipEvent_t> d2h_ev;   // copy_d2h: a staging area of the transfer's size, an event per piece
    // A decode call's tables -- the code-block rows (a window's skipped blocks marked), behind them K5's scratch index and the list
    // of blocks with data -- are put together in pinned memory the context owns and fetched by a kernel of the call's stream
    // (launch_dec_upload); two sets in turn: the kernel of one call may still be queued when the next call fills its tables
    // Full decode with overlap on: K5b of the top resolution's blocks (3/4 of them) runs on the side stream beside K5b of the
    // other blocks and the inverse levels that need only those; the last inverse level waits for it
    hipEvent_t ev_dec_front = nullptr, ev_dec_top = nullptr;
    bool dec_top_pending = false;
    // Pipelined encodes of SMALL frames (up to kFrameStreamSamples samples per call): a frame's whole chain on ONE of the two side streams,
    // taken in turn -- no event inside a frame (12 instead of 18 runtime calls), consecutive frames overlap through the streams.  A call
    // is bound by the host's launches below ~2048^2 x 3: 512^2 x 3 0.058 -> 0.045 ms, 2048^2 x 3 0.073 -> 0.058; at 4096^2 it makes no
    // difference, at 8192^2 it loses 19 % (no top-resolution K3 beside the remaining levels, no stream priorities).
    // GRK_AMD_FRAME_STREAMS = 0: never, 1 (default): by size, 2: always
    static constexpr uint64_t kFrameStreamSamples = 16ull << 20;
    int frame_streams = 1; int fs_parity = 0;
    // ... the caller's pixels are then read on the frame's stream, not on the context's: level 0 -- their only reader -- is followed by
    // this event, and the context's stream waits for it, so that whatever the caller queues behind the call in stream order (the
    // next frame's pixels into the same buffer, a stream-ordered free) still comes after the read, as it does on the other paths
    // (grk_amd_set_pixel_hold(ctx, 1): the caller keeps a call's pixels untouched until grk_amd_stream_wait_pixels / a synchronisation;
    //  the wait -- two queue hand-overs between consecutive small frames, 0.038 -> 0.057 ms per 512^2 call -- is then left out)
    hipEvent_t ev_px = nullptr; bool want_px_event = false; bool px_hold = false; bool px_event_valid = false;
    // The encoder's three streams have to DISPATCH side by side.  Hardware queues are served by a few dispatch pipes; two queues on one
    // pipe take turns while one of them has a large grid in flight, and which queue a stream gets depends on how many streams the process
    // made before (profiles/r06_hw_queues.txt: 0.37 -> 0.55 ms per 8K frame with 4, 5 or 8 earlier streams).  Before the first
    // overlapped encode on a given main stream the three are probed pairwise (a grid that stays in dispatch for ~150 us on one, a
    // one-workgroup kernel on the other) and a side stream that has to wait is replaced (GRK_AMD_STREAM_PROBE=0: never)
    int stream_probe = 1; hipStream_t probed_main = nullptr; int side_priority = 0;
    hipStream_t probed_before[4] = {};   // main streams probed earlier: a host that alternates between a few streams is not probed at every switch
    int probe_replaced = 0;           // side streams replaced by the probe so far (grk_amd_stream_probe_result)
    bool probe_warm = false;          // the probe's kernels have been launched once (their first launch loads their code: not to be measured)
    bool seq_vetted = false;          // (a sequence's internal context) its streams have been vetted against its neighbours' (vet_sequence_streams)
    unsigned long long* pend_alloc = nullptr; uint32_t pend_alloc_units = 0;   // K3's allocator reset handed to the fused level 0 (run_dwt)
    bool alloc_in_level0 = true;      // GRK_AMD_ALLOC_IN_LEVEL0=0: a launch of its own, as before r06
    int k3_room = 3;                  // pipelined encodes: K3 launches that leave registers for the next frame's level 0 -- bit 0 the top class, bit 1 the rest (GRK_AMD_K3_ROOM)
    // Part-1 decode: blocks of the default style go 64 to a wave (K8L, kernels_t1lanes.hip) unless much longer than the rest
    // (GRK_AMD_T1_LANES=0: every block its own wave, K8 as in r01-r03; 2: lanes wherever they can be used; GRK_AMD_T1_TAIL_RATIO: see run_t1_decode)
    // Decode of a SEQUENCE of frames (grk_amd_set_decode_pipelining): consecutive grk_amd_decode_tiles calls with device buffers
    // go in turn to this context and to `dec_kids` -- contexts of their own on the same device: own streams, tables, planes --,
    // each behind an event on the caller's stream.  A frame's serial block-decoding chains (K5a / K8) leave most of the machine
    // idle; the next frame's kernels take what is free.  grk_amd_synchronize / grk_amd_decode_status cover them all.
    std::vector<grk_amd_ctx*> dec_kids;
    uint32_t dec_seq = 0;
    hipEvent_t ev_seq = nullptr;
    hipEvent_t ev_frame_done = nullptr;   // (an internal context of a sequence) behind the last frame it was given: grk_amd_decode_stream_wait_slot
    int t1_lanes = 1;                    // 0: never, 1: where the cost model below says they are faster, 2: wherever they can (tests)
    float t1_tail_ratio = 0.25f;
    float t1_tail_share = 0.0f;          // ... and at least this share of the blocks (the longest ones) to K8 as well
    bool t1_pass_sync = true;            // K8L's waves hold blocks of equal bit-plane / pass counts and run pass by pass (GRK_AMD_T1_SYNC=0: free-running lanes)
    struct DecUpload { char* p = nullptr; char* dp = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; } dec_up[2];
    uint32_t dec_turn = 0;
    // timing
    bool timing = false;
    Timer timers[10];
};

namespace {

bool create_alt_events(grk_amd_ctx* c)
{
    for (auto& as : c->alts)
        if (hipEventCreateWithFlags(&as.ev_side, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&as.ev_side2, hipEventDisableTiming) != hipSuccess) return false;
    return true;
}

int fail(grk_amd_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
    if (c) {
        c->err = what;
        if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
        if (c->verbose) fprintf(stderr, "[grok_amd] %s\n", c->err.c_str());
    }
    return code;
}

#define HIP_TRY(c, call, what)                                                      \
    do { hipError_t _e = (call); if (_e != hipSuccess) return fail(c, GRK_AMD_ERR_NO_DEVICE, what, _e); } while (0)

#ifndef GRK_AMD_OVERLAP_DEFAULT
#define GRK_AMD_OVERLAP_DEFAULT 1
#endif

bool host_is_pinned(const void* p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (plain malloc'ed memory: "invalid value")
    return a.type == hipMemoryTypeHost;
}

// Host -> device, ordered after what the context's stream holds so far; on return the source may be reused (pageable) or the
// copy is queued on the stream (pinned).  Device -> host likewise; the caller synchronises the stream before reading pinned
// memory, pageable memory is complete on return.
int copy_h2d(grk_amd_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (bytes < 2 * HostStage::kChunk || host_is_pinned(src)) {
        HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream), "upload");
        return GRK_AMD_OK;
    }
    HostStage& hs = c->stage;
    HIP_TRY(c, hs.ensure(), "alloc pinned staging");
    HIP_TRY(c, hipEventRecord(hs.ev_in, c->stream), "record");
    const size_t nchunks = (bytes + HostStage::kChunk - 1) / HostStage::kChunk;
    hipError_t errs[HostStage::kLanes] = {};
    std::thread th[HostStage::kLanes];
    for (int t = 0; t < HostStage::kLanes; ++t)
        th[t] = std::thread([&, t]() {
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess) e = hipStreamWaitEvent(hs.st[t], hs.ev_in, 0);
            int k = 0;
            for (size_t i = (size_t)t; i < nchunks && e == hipSuccess; i += HostStage::kLanes, k ^= 1) {
                const size_t off = i * HostStage::kChunk, n = std::min(HostStage::kChunk, bytes - off);
                e = hipEventSynchronize(hs.ev[t][k]);                  // the DMA that last read this chunk (never recorded: returns at once)
                if (e != hipSuccess) break;
                std::memcpy(hs.buf[t][k], (const char*)src + off, n);
                e = hipMemcpyAsync((char*)dst + off, hs.buf[t][k], n, hipMemcpyHostToDevice, hs.st[t]);
                if (e == hipSuccess) e = hipEventRecord(hs.ev[t][k], hs.st[t]);
            }
            if (e == hipSuccess) e = hipEventRecord(hs.ev_out[t], hs.st[t]);
            errs[t] = e;
        });
    for (auto& x : th) x.join();
    for (int t = 0; t < HostStage::kLanes; ++t) {
        HIP_TRY(c, errs[t], "staged upload");
        HIP_TRY(c, hipStreamWaitEvent(c->stream, hs.ev_out[t], 0), "join the copy streams");
    }
    return GRK_AMD_OK;
}

// Device -> pageable host memory: ONE DMA stream (the context's) brings the bytes into a pinned staging area of the transfer's size,
// piece by piece with an event behind each, while a few host threads copy the pieces that have arrived to where they belong -- the link
// runs near its rate (99 MB in 2.17 ms; four lanes with a DMA and a memcpy each in turn: 2.42; into pinned memory 1.78).
int copy_d2h(grk_amd_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (bytes < 2 * HostStage::kChunk || host_is_pinned(dst)) {
        HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream), "download");
        return GRK_AMD_OK;
    }
    // (measured on 99 MB, ms: pieces of 4 / 8 / 16 / 32 MB with four threads 2.37 / 2.17 / 2.25 / 2.45, eight threads 2.5 / 2.3 / 2.3 / 2.5 --
    //  a DMA of a few MB costs its set-up, a thread its creation; pinned memory: 1.78)
    constexpr size_t kPiece = 8u << 20;
    if (c->d2h_cap < bytes) {
        if (c->d2h_pin) { (void)hipHostFree(c->d2h_pin); c->d2h_pin = nullptr; c->d2h_cap = 0; }
        const size_t want = bytes + (bytes >> 3);
        HIP_TRY(c, hipHostMalloc(&c->d2h_pin, want, hipHostMallocDefault), "alloc pinned staging");
        c->d2h_cap = want;
    }
    struct Piece { size_t off, n; };
    std::vector<Piece> pieces;
    for (size_t off = 0; off < bytes; off += kPiece) pieces.push_back(Piece{off, std::min(kPiece, bytes - off)});
    while (c->d2h_ev.size() < pieces.size()) {
        hipEvent_t ev = nullptr;
        HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "create event");
        c->d2h_ev.push_back(ev);
    }
    std::atomic<size_t> issued{0};
    std::atomic<int> failed{0};
    constexpr int nthr = 4;                       // every thread copies its quarter of every piece: what is left behind the last DMA is 2 MB each
    std::thread th[nthr];
    for (int t = 0; t < nthr; ++t)
        th[t] = std::thread([&, t]() {
            (void)hipSetDevice(c->device);
            for (size_t i = 0; i < pieces.size(); ++i) {
                while (issued.load(std::memory_order_acquire) <= i) { if (failed.load()) return; std::this_thread::yield(); }   // (its event has been recorded)
                if (hipEventSynchronize(c->d2h_ev[i]) != hipSuccess) { failed.store(1); return; }
                const size_t a0 = pieces[i].n * (size_t)t / nthr, a1 = pieces[i].n * (size_t)(t + 1) / nthr;
                std::memcpy((char*)dst + pieces[i].off + a0, (const char*)c->d2h_pin + pieces[i].off + a0, a1 - a0);
            }
        });
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < pieces.size() && e == hipSuccess; ++i) {
        e = hipMemcpyAsync((char*)c->d2h_pin + pieces[i].off, (const char*)src + pieces[i].off, pieces[i].n, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->d2h_ev[i], c->stream);
        if (e == hipSuccess) issued.store(i + 1, std::memory_order_release);
    }
    if (e != hipSuccess) failed.store(1);
    for (int t = 0; t < nthr; ++t) th[t].join();
    HIP_TRY(c, e, "staged download");
    if (failed.load()) return fail(c, GRK_AMD_ERR_NO_DEVICE, "staged download");
    return GRK_AMD_OK;
}

bool same_params(const grk_amd_tile_params& a, const grk_amd_tile_params& b)
{
    return a.tile_w == b.tile_w && a.tile_h == b.tile_h && a.num_comps == b.num_comps && a.prec == b.prec &&
           a.sgnd == b.sgnd && a.irreversible == b.irreversible && a.mct == b.mct &&
           a.num_levels == b.num_levels && a.cblk_w_exp == b.cblk_w_exp && a.cblk_h_exp == b.cblk_h_exp &&
           a.reserved[0] == b.reserved[0] && a.reserved[1] == b.reserved[1] && a.tile_x0 == b.tile_x0 && a.tile_y0 == b.tile_y0 &&
           std::memcmp(a.precinct_exp, b.precinct_exp, sizeof a.precinct_exp) == 0;
}

int ensure_geom(grk_amd_ctx* c, const grk_amd_tile_params* p)
{
    if (c->have_geom && same_params(c->gp, *p)) return GRK_AMD_OK;
    // the tables rebuilt below may still be read by K3 launches of a pipelined predecessor on the side streams
    if (c->side) HIP_TRY(c, hipStreamSynchronize(c->side), "sync side stream");
    if (c->side2) HIP_TRY(c, hipStreamSynchronize(c->side2), "sync side stream 2");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    c->side_pending = false;
    c->have_geom = false;                      // valid again only once every table below is on the device
    int rc = build_tile_geom(*p, c->geom);
    if (rc != GRK_AMD_OK) return fail(c, rc, "unsupported tile parameters");
    c->gp = *p;
    const TileGeom& g = c->geom;
    c->h_desc.clear();
    std::vector<uint8_t> h_res;                 // resolution of each block (0 = coarsest)
    for (uint32_t k = 0; k < p->num_comps; ++k)
        for (const auto& b : g.blocks_comp0) {
            h_res.push_back(b.res);
            HtBlockDesc d;
            d.px = b.px; d.py = b.py;
            d.w = (uint16_t)(b.x1 - b.x0); d.h = (uint16_t)(b.y1 - b.y0);
            d.comp = (uint16_t)k; d.kmax = b.kmax; d.pad = b.band;
            d.inv_step = 1.0f / b.stepsize;
            c->h_desc.push_back(d);
        }
    // K3 block classes, each its own launch: the top resolution's sub-bands (3/4 of all blocks) are final after DWT
    // level 0, so they are coded beside the remaining levels (run_dwt, overlap), the rest after the last level; a third
    // class holds ALL blocks, for when nothing overlaps (one launch, one tail).  The LDS buffers of a launch are sized
    // for what the class's typical block needs (capped, kernels_ht.hip) -- the LDS per wave is what fixes the occupancy
    // -- and the blocks that outgrow them, e.g. the few high-Kmax blocks of the low resolutions, go through the fallback
    // launch.  Only with GRK_AMD_LDS_CAP=0 (worst-case buffers, no fallback) are the large-LDS blocks classes of their
    // own, so that they do not cost every block a wave per SIMD.  Few classes on purpose: a launch ends with a tail of
    // long-running waves, and launches on one stream do not overlap (measured: one class per resolution costs 0.15 ms at 8K).
    {
        constexpr size_t kLdsFor16Waves = 10240;
        std::vector<uint32_t> sel;
        std::vector<HtClass> cls;
        std::vector<size_t> first;
        std::vector<uint8_t> ctop, cbig;
        auto add_class = [&](int top, int big) {       // top: 1 top resolution, 0 the rest, 2 every block;  big: -1 any, 0 / 1 by LDS need
            HtClass cl{nullptr, 0, 0, 0, 0, 0, 0};
            const size_t at = sel.size();
            uint32_t hist[64] = {0};
            for (uint32_t i = 0; i < c->h_desc.size(); ++i) {
                if (top != 2 && (int)(h_res[i] == g.p.num_levels && g.p.num_levels >= 1) != top) continue;
                const HtBlockDesc& d = c->h_desc[i];
                const uint32_t samples = (uint32_t)d.w * d.h, quads = ((d.w + 1u) / 2u) * ((d.h + 1u) / 2u);
                if (big >= 0 && (int)(ht_lds_bytes(samples, quads, d.kmax) > kLdsFor16Waves) != big) continue;
                cl.count++;
                cl.max_kmax = std::max<uint32_t>(cl.max_kmax, d.kmax);
                cl.max_samples = std::max<uint32_t>(cl.max_samples, samples);
                cl.max_quads = std::max<uint32_t>(cl.max_quads, quads);
                hist[d.kmax & 63u] += samples;
                sel.push_back(i);
            }
            for (uint32_t k = 0; k < 64; ++k) if (hist[k] > hist[cl.cap_kmax]) cl.cap_kmax = k;     // where most of the samples are
            if (cl.count) { cls.push_back(cl); first.push_back(at); ctop.push_back((uint8_t)top); cbig.push_back((uint8_t)(big > 0)); }
        };
        if (c->lds_cap) { add_class(1, -1); add_class(0, -1); add_class(2, -1); }
        else { add_class(1, 0); add_class(1, 1); add_class(0, 0); add_class(0, 1); }
        HIP_TRY(c, c->ht_sel.ensure(sel.size() * 4 + 16), "alloc class index");
        HIP_TRY(c, hipMemcpyAsync(c->ht_sel.p, sel.data(), sel.size() * 4, hipMemcpyHostToDevice, c->stream), "upload class index");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "sync class index");
        for (size_t k = 0; k < cls.size(); ++k) {
            cls[k].sel = (const uint32_t*)c->ht_sel.p + first[k];
            c->ht_classes[k] = cls[k];
            c->ht_class_top[k] = ctop[k]; c->ht_class_big[k] = cbig[k];
        }
        c->ht_num_classes = (uint32_t)cls.size();
    }
    // decode-side descriptors: inv_step carries the dequantisation scale of the band
    // (codestream/Quantizer.cpp:41-63 with compress = false: log2_gain 0, then / 2^(31 - numbps))
    c->h_desc_dec = c->h_desc;
    {
        size_t i = 0;
        for (uint32_t k = 0; k < p->num_comps; ++k)
            for (const auto& b : g.blocks_comp0) {
                float scale = 1.0f;
                const uint32_t bi = b.res == 0 ? 0u : 3u * b.res - 2u + (b.band - 1u);
                if (p->irreversible && c->dec_steps.size() == (size_t)p->num_comps * g.num_bands_total) {
                    scale = c->dec_steps[(size_t)k * g.num_bands_total + bi];      // the host's TileBand::stepsize, fix-ups included
                } else if (p->irreversible) {
                    const uint16_t wq = (c->dec_qcd.size() == g.num_bands_total) ? c->dec_qcd[bi] : g.qcd_words[bi];
                    const double step = (1.0 + (wq & 0x7FF) / 2048.0) * std::pow(2.0, (int)p->prec - (int)(wq >> 11));
                    scale = (float)step;
                    if (!p->reserved[0]) scale /= (float)(1u << (31 - b.kmax));     // HT only (Quantizer.cpp:54-63)
                }
                c->h_desc_dec[i++].inv_step = scale;
            }
    }
    HIP_TRY(c, c->dec_desc.ensure(c->h_desc_dec.size() * sizeof(HtBlockDesc)), "alloc decode block table");
    HIP_TRY(c, hipMemcpyAsync(c->dec_desc.p, c->h_desc_dec.data(), c->h_desc_dec.size() * sizeof(HtBlockDesc),
                              hipMemcpyHostToDevice, c->stream), "upload decode block table");
    HIP_TRY(c, c->blockdesc.ensure(c->h_desc.size() * sizeof(HtBlockDesc)), "alloc block table");
    HIP_TRY(c, hipMemcpyAsync(c->blockdesc.p, c->h_desc.data(), c->h_desc.size() * sizeof(HtBlockDesc),
                              hipMemcpyHostToDevice, c->stream), "upload block table");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync block table");
    c->have_geom = true;
    return GRK_AMD_OK;
}

struct ScopedTimer {
    grk_amd_ctx* c; int which; hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    ScopedTimer(grk_amd_ctx* c_, int w, hipStream_t s = nullptr) : c(c_), which(w), st(s ? s : c_->stream)
    {
        if (!c->timing) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, st);
    }
    void cancel() { if (a) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); a = b = nullptr; } }
    ~ScopedTimer()
    {
        if (!a) return;
        (void)hipEventRecord(b, st);
        c->timers[which].ev.emplace_back(a, b);
    }
};

void drain_timers(grk_amd_ctx* c)
{
    for (auto& t : c->timers) {
        for (auto& pr : t.ev) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                t.total_ms += ms; t.launches++;
            }
            (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
        }
        t.ev.clear();
    }
}

uint32_t ll_stride_for(uint32_t w) { return (((w + 1) >> 1) + 31u) & ~31u; }

int run_ingest(grk_amd_ctx* c, uint32_t ntiles, const void* d_pixels, void* d_planes)
{
    const TileGeom& g = c->geom;
    IngestArgs a{};
    a.pixels = d_pixels; a.planes = (int32_t*)d_planes;
    a.w = g.p.tile_w; a.h = g.p.tile_h; a.stride = g.stride; a.pitch = g.plane_elems;
    a.ncomp = g.p.num_comps; a.ntiles = ntiles;
    a.bytes_per_sample = (g.p.prec + 7) / 8;
    a.dc = g.p.sgnd ? 0 : (1 << (g.p.prec - 1));
    a.sext = g.p.sgnd ? (1 << (8 * a.bytes_per_sample - 1)) : 0;
    a.mct = g.p.mct; a.irreversible = g.p.irreversible;
    ScopedTimer t(c, 0);
    HIP_TRY(c, launch_ingest(a, c->stream), "launch ingest");
    return GRK_AMD_OK;
}

// d_pixels != nullptr: level 0 reads the caller's pixels directly (K1 fused into K2), d_in is unused
HtArgs make_ht_args(grk_amd_ctx* c, uint32_t ntiles, const void* d_mallat, int* rc, bool h16 = false);

// everything that reads or overwrites the results of the latest encode on the main stream comes after its side streams
// ---- stream probe -----------------------------------------------------------------------------------------------------------------
// 16 384 workgroups that hold 40 KB of LDS (four to a CU) for ~10 us each: ~160 us during which the grid is still being dispatched
__global__ __launch_bounds__(64) void probe_spin_kernel(unsigned int ticks, unsigned int* sink)
{
    extern __shared__ unsigned int pad[];              // 40 KB asked for at the launch
    pad[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && pad[threadIdx.x] == 0xFFFFFFFFu) *sink = 1;
}
__global__ void probe_tick_kernel(unsigned int* sink) { if (sink && threadIdx.x == 1024) *sink = 1; }

// does a kernel launched on `b` while a large grid of `a` is in dispatch run at once?  (both streams idle on entry and on return)
int streams_side_by_side(grk_amd_ctx* c, hipStream_t a, hipStream_t b, bool* yes)
{
    hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&ea);
    if (e == hipSuccess) e = hipEventCreate(&eb);
    if (e == hipSuccess) e = hipEventRecord(e0, a);
    if (e == hipSuccess) { hipLaunchKernelGGL(probe_spin_kernel, dim3(16384), dim3(64), 40960, a, 1000u, (unsigned int*)nullptr); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipEventRecord(ea, a);
    // (the small kernel is launched once the grid has started: e0 has passed)
    if (e == hipSuccess) { while ((e = hipEventQuery(e0)) == hipErrorNotReady) {} }
    if (e == hipSuccess) { hipLaunchKernelGGL(probe_tick_kernel, dim3(1), dim3(64), 0, b, (unsigned int*)nullptr); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipEventRecord(eb, b);
    if (e == hipSuccess) e = hipStreamSynchronize(a);
    if (e == hipSuccess) e = hipStreamSynchronize(b);
    float ta = 0, tb = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ta, e0, ea);
    if (e == hipSuccess) e = hipEventElapsedTime(&tb, e0, eb);
    if (e0) (void)hipEventDestroy(e0);
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    if (e != hipSuccess) return fail(c, GRK_AMD_ERR_NO_DEVICE, "stream probe", e);
    *yes = tb < 0.6f * ta;
    if (c->verbose) fprintf(stderr, "[grok_amd] stream probe: grid %.3f ms, small kernel done after %.3f ms -> %s\n", ta, tb, *yes ? "side by side" : "in turn");
    return GRK_AMD_OK;
}

int probe_warmup(grk_amd_ctx* c, hipStream_t st)
{
    if (c->probe_warm) return GRK_AMD_OK;
    hipLaunchKernelGGL(probe_spin_kernel, dim3(256), dim3(64), 40960, st, 10u, (unsigned int*)nullptr);
    hipLaunchKernelGGL(probe_tick_kernel, dim3(1), dim3(64), 0, st, (unsigned int*)nullptr);
    HIP_TRY(c, hipGetLastError(), "stream probe");
    HIP_TRY(c, hipStreamSynchronize(st), "sync");
    c->probe_warm = true;
    return GRK_AMD_OK;
}

// *cur, or a stream made now with *cur's priority, whose kernels are dispatched side by side with every stream of `against` (both
// directions); *cur is replaced (and destroyed) when a better one is found within eight tries, else kept.  All streams idle on entry.
int vetted_stream(grk_amd_ctx* c, hipStream_t* cur, const std::vector<hipStream_t>& against, int* replaced)
{
    int rc = probe_warmup(c, *cur); if (rc) return rc;
    int prio = 0;
    if (hipStreamGetPriority(*cur, &prio) != hipSuccess) { (void)hipGetLastError(); prio = 0; }
    std::vector<hipStream_t> rejects;
    hipStream_t cand = *cur;
    for (int tries = 0; tries < 9; ++tries) {
        bool ok = true;
        for (hipStream_t a : against) {
            if (!a || a == cand) continue;
            rc = streams_side_by_side(c, a, cand, &ok);
            if (rc == GRK_AMD_OK && ok) rc = streams_side_by_side(c, cand, a, &ok);
            if (rc || !ok) break;
        }
        if (rc) break;
        if (ok) { if (cand != *cur) { rejects.push_back(*cur); *cur = cand; if (replaced) ++*replaced; } cand = nullptr; break; }
        if (cand != *cur) rejects.push_back(cand);
        cand = nullptr;
        if (tries == 8 || hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, prio) != hipSuccess) { (void)hipGetLastError(); cand = nullptr; break; }
    }
    if (cand && cand != *cur) rejects.push_back(cand);
    for (hipStream_t r : rejects) (void)hipStreamDestroy(r);
    return rc;
}

int probe_streams(grk_amd_ctx* c)
{
    if (!c->stream_probe || c->probed_main == c->stream || !c->side) return GRK_AMD_OK;
    c->probed_main = c->stream;
    for (hipStream_t seen : c->probed_before) if (seen == c->stream) return GRK_AMD_OK;
    for (int i = 3; i > 0; --i) c->probed_before[i] = c->probed_before[i - 1];
    c->probed_before[0] = c->stream;
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    HIP_TRY(c, hipStreamSynchronize(c->side), "sync");
    if (c->side2) HIP_TRY(c, hipStreamSynchronize(c->side2), "sync");
    c->side_pending = false;
    { const int wr = probe_warmup(c, c->stream); if (wr) return wr; }
    std::vector<hipStream_t> rejects;                  // kept alive until the end: a stream made now gets another queue than these
    auto good = [&](hipStream_t cand, hipStream_t other, bool* ok) -> int {
        bool y = false;
        int rc = streams_side_by_side(c, c->stream, cand, &y); if (rc) return rc;
        if (y) { rc = streams_side_by_side(c, cand, c->stream, &y); if (rc) return rc; }
        if (y && other) { rc = streams_side_by_side(c, other, cand, &y); if (rc) return rc; }
        if (y && other) { rc = streams_side_by_side(c, cand, other, &y); if (rc) return rc; }
        *ok = y;
        return GRK_AMD_OK;
    };
    int rc = GRK_AMD_OK;
    for (int which = 0; which < 2 && rc == GRK_AMD_OK; ++which) {
        hipStream_t& mine = which ? c->side2 : c->side;
        if (!mine) continue;
        hipStream_t other = which ? c->side : nullptr;
        bool ok = false;
        rc = good(mine, other, &ok);
        for (int tries = 0; rc == GRK_AMD_OK && !ok && tries < 8; ++tries) {
            hipStream_t cand = nullptr;
            if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, c->side_priority) != hipSuccess) { (void)hipGetLastError(); break; }
            rc = good(cand, other, &ok);
            if (rc == GRK_AMD_OK && ok) { rejects.push_back(mine); mine = cand; ++c->probe_replaced; }
            else rejects.push_back(cand);
        }
        // (none found: the stream stays as it was)
    }
    for (hipStream_t r : rejects) (void)hipStreamDestroy(r);
    return rc;
}

int join_side(grk_amd_ctx* c)
{
    if (!c->side_pending) return GRK_AMD_OK;
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_side, 0), "join side stream");
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_side2, 0), "join side stream 2");
    c->side_pending = false;
    return GRK_AMD_OK;
}

// 16-bit planes are safe when no coefficient of any level can leave int16.  Bound (5/3, L1 norms of the analysis
// filters: low-pass 1.5, high-pass 2 per dimension; RCT chroma is one bit wider than the pixels): the LL of level l is
// below M * 2.25^l, a detail band of level l below 4 * M * 2.25^(l-1), with M = 2^prec the largest input magnitude.
bool planes16_ok(const grk_amd_tile_params& p)
{
    if (p.irreversible || p.prec > 8 || p.num_levels == 0) return false;
    double bound = (double)(1u << p.prec) * 4.0;
    for (uint32_t l = 1; l < p.num_levels; ++l) bound *= 2.25;
    return bound + 8.0 * p.num_levels < 32767.0;
}

// Level l of such a tile on PACKED int16 pairs (kernels_dwt.hip, strip_pk): every intermediate of the 2-D lifting step has to
// stay inside 16 bits as well.  With M the largest magnitude entering the level (2^prec after DC shift and RCT, times the
// low-pass gain 1.5 x 1.5 per level before, plus rounding), the largest is the horizontal update's sum of two high-pass
// values of a vertically high-pass row: 2 x 2 x 2M each, 8M + 2 in all.
bool pk16_level_ok(const grk_amd_tile_params& p, uint32_t l)
{
    if (p.sgnd) return false;                        // (the packed unpacking is written for unsigned pixels)
    double m = (double)(1u << p.prec);
    for (uint32_t i = 0; i < l; ++i) m = m * 2.25 + 4.0;
    return 8.0 * m + 16.0 < 32767.0;
}

int run_dwt(grk_amd_ctx* c, uint32_t nplanes, void* d_in, void* d_out, const void* d_pixels = nullptr, uint32_t ntiles = 0,
            bool overlap_ht = false, bool h16 = false)
{
    const TileGeom& g = c->geom;
    const uint32_t L = g.p.num_levels;
    const uint32_t W = g.p.tile_w, H = g.p.tile_h;
    if (L == 0) {
        HIP_TRY(c, hipMemcpyAsync(d_out, d_in, (size_t)nplanes * g.plane_elems * 4, hipMemcpyDeviceToDevice, c->stream), "copy planes");
        return GRK_AMD_OK;
    }
    // LL ping-pong storage: A holds LL1, LL3, ...; B holds LL2, LL4, ...
    const uint32_t sA = ll_stride_for(W), hA = (H + 1) >> 1;
    const uint32_t sB = ll_stride_for((W + 1) >> 1), hB = (hA + 1) >> 1;
    const uint64_t pitchA = (uint64_t)sA * hA, pitchB = (uint64_t)sB * hB;
    HIP_TRY(c, c->llA.ensure((size_t)nplanes * pitchA * 4 + 256), "alloc LL ping");
    HIP_TRY(c, c->llB.ensure((size_t)nplanes * pitchB * 4 + 256), "alloc LL pong");
    ScopedTimer t(c, 1);
    for (uint32_t l = 0; l < L; ++l) {
        DwtLevelArgs a{};
        a.cw = level_geom(g, l).w; a.ch = level_geom(g, l).h;
        a.px = level_geom(g, l).x0 & 1u; a.py = level_geom(g, l).y0 & 1u;
        if (l == 0) { a.in = (const int32_t*)d_in; a.in_stride = g.stride; a.in_pitch = g.plane_elems; }
        else if (l & 1) { a.in = (const int32_t*)c->llA.p; a.in_stride = sA; a.in_pitch = pitchA; }
        else { a.in = (const int32_t*)c->llB.p; a.in_stride = sB; a.in_pitch = pitchB; }
        a.mallat = (int32_t*)d_out; a.m_stride = g.stride; a.m_pitch = g.plane_elems;
        if (l + 1 == L) { a.ll = (int32_t*)d_out; a.ll_stride = g.stride; a.ll_pitch = g.plane_elems; }
        else if ((l + 1) & 1) { a.ll = (int32_t*)c->llA.p; a.ll_stride = sA; a.ll_pitch = pitchA; }
        else { a.ll = (int32_t*)c->llB.p; a.ll_stride = sB; a.ll_pitch = pitchB; }
        a.nplanes = nplanes;
        a.irreversible = g.p.irreversible;
        a.h16 = h16 ? 1 : 0;
        a.pk = h16 && c->dwt_pk && pk16_level_ok(g.p, l);
        a.xcd = c->dwt_xcd;
        // enough workgroups to cover the chip several times, few enough to amortise warm-up rows (profiles/r06_dwt_reads.txt: at 4096
        // the 8K level 0 ran 16-row segments and read 1.55 x its pixels; 2048 -> 32-row segments, 1.35 x, the DWT 1 % faster)
        const uint32_t sh = (a.ch + a.py + 1) >> 1;           // row pairs on the coordinate grid
        uint32_t seg = 64;
        const uint64_t strips = (a.cw + a.px + dwt_level_strip_cols(a) - 1) / dwt_level_strip_cols(a);
        // workgroups along z: planes, or for the fused level 0 tiles (x components when there is no MCT triple)
        const uint32_t zslots = (l == 0 && d_pixels) ? ntiles * ((g.p.mct && g.p.num_comps >= 3) ? 1u : g.p.num_comps) : nplanes;
        // (... for the packed 5/3 kernel; the 32-bit kernels -- 448-column strips, twice the workgroups per row -- are better off with
        //  the finer cut: cfg3's 9/7 family 0.361 ms at 4096, 0.394 at 2048)
        static const int kMinWgsEnv = getenv("GRK_AMD_DWT_MIN_WGS") ? std::max(1, atoi(getenv("GRK_AMD_DWT_MIN_WGS"))) : 0;
        const uint32_t kMinWgs = kMinWgsEnv ? (uint32_t)kMinWgsEnv : (a.pk ? 2048u : 4096u);
        while (seg > 8 && strips * ((sh + seg - 1) / seg) * zslots < kMinWgs) seg >>= 1;
        a.seg_pairs = seg;
        if (a.cw == 0 || a.ch == 0) {
            // a level without samples (a narrow tile off the origin: [ceil(x0 / 2^l), ceil((x0 + w) / 2^l)) can be empty):
            // nothing to transform, and nothing deeper either
        } else if (l == 0 && d_pixels) {
            a.pixels = d_pixels; a.px_bytes = (g.p.prec + 7) / 8;
            a.alloc_reset = c->pend_alloc; a.alloc_chunk_units = c->pend_alloc_units; c->pend_alloc = nullptr;
            a.dc = g.p.sgnd ? 0 : (1 << (g.p.prec - 1));
            a.sext = g.p.sgnd ? (1 << (8 * a.px_bytes - 1)) : 0;
            HIP_TRY(c, launch_dwt_level0_fused(a, ntiles, g.p.num_comps, g.p.mct, c->stream), "launch fused dwt level 0");
            if (c->want_px_event) HIP_TRY(c, hipEventRecord(c->ev_px, c->stream), "record the pixels' last read");
        } else {
            HIP_TRY(c, launch_dwt_level(a, c->stream), "launch dwt level");
        }
        if (overlap_ht && (l == 0 || l + 1 == L)) {
            // After level 0 the top resolution's sub-bands are final: its code-blocks (3/4 of all) are coded on
            // low-priority side streams while the remaining levels -- short, latency-bound launches that are the
            // critical path -- run here.  After the last level the rest follows: small-LDS class on this stream (run_ht),
            // large-LDS class on the second side stream, so that the launches' tails overlap.
            int rc = GRK_AMD_OK;
            const HtArgs h = make_ht_args(c, ntiles, d_out, &rc, h16);
            if (rc) return rc;
            HIP_TRY(c, hipEventRecord(c->ev_level0, c->stream), "record level");
            for (uint32_t k = 0; k < h.num_classes; ++k) {
                if (c->ht_class_top[k] == 2) continue;               // (the all-blocks class is for the non-overlapped path)
                const bool top = c->ht_class_top[k] != 0, big = c->ht_class_big[k] != 0;
                hipStream_t st = nullptr;
                if (l == 0 && top) st = big ? c->side2 : c->side;
                if (l + 1 == L && !top && big) st = c->side2;
                // the rest: on the main stream (run_ht) beside the tail of the top resolution -- unless consecutive encodes
                // are pipelined: then the main stream carries nothing but the DWT chain, so that the next encode's level 0
                // starts as early as possible, and every K3 launch queues on a side stream
                if (l + 1 == L && !top && !big && c->pipelining) st = c->side2;    // (its tail then overlaps the top class's)
                if (!st) continue;
                HIP_TRY(c, hipStreamWaitEvent(st, c->ev_level0, 0), "side stream waits for the level");
                ScopedTimer tt(c, st == c->side ? 4 : 8, st);
                // (consecutive encodes pipelined: the top class is still running when the next encode's level 0 arrives)
                HtArgs hs = h;
                hs.room = (c->pipelining && (c->k3_room & (top ? 1 : 2))) ? 1 : 0;
                HIP_TRY(c, launch_ht_classes(hs, k, k + 1, st), "launch ht encode (side stream)");
            }
            if (l + 1 == L) {
                HIP_TRY(c, hipEventRecord(c->ev_side, c->side), "record side stream");
                HIP_TRY(c, hipEventRecord(c->ev_side2, c->side2), "record side stream 2");
            }
        }
    }
    return GRK_AMD_OK;
}

// Region decode (the reference: grk_decompress_set_window -> WaveletReverse.cpp:1466-2213 partial
// synthesis over a sparse buffer).  need[l] = the part of LL_l (l = 0: the image) that has to be right so that the
// window is; level l is synthesised from the coefficient pairs pairs[l] of LL_{l+1} and of resolution L - l's bands.
// A synthesised sample depends on the pairs within 1 (5/3) or 2 (9/7) of its own, the kernel's strip halo and the
// recurrence warm-up reach 2 pairs further: the margins below are conservative on purpose.
struct Rect { uint32_t x0, y0, x1, y1; };
struct RegionPlan { std::vector<Rect> need, pairs; std::vector<uint32_t> px, py; };
// (a level that starts on an odd coordinate works on the coordinate grid shifted by the parity, kernels_idwt.hip: sample c
//  of the level belongs to pair (c + parity) / 2, pair J's low-pass sample has index J - parity, its high-pass sample J)
RegionPlan plan_region(const TileGeom& g, Rect win)
{
    RegionPlan r;
    const uint32_t L = g.p.num_levels, M = g.p.irreversible ? 4u : 2u;
    r.need.resize(L + 1); r.pairs.resize(L); r.px.resize(L); r.py.resize(L);
    r.need[0] = win;
    auto sat = [](uint32_t a, uint32_t b) { return a > b ? a - b : 0u; };
    for (uint32_t l = 0; l < L; ++l) {
        const ResGeom& R = level_geom(g, l);
        const uint32_t px = R.x0 & 1u, py = R.y0 & 1u;
        const uint32_t npx = (R.w + px + 1) >> 1, npy = (R.h + py + 1) >> 1;       // pairs on the coordinate grid
        const uint32_t sw = (R.w + 1 - px) >> 1, sh = (R.h + 1 - py) >> 1;         // low-pass samples
        r.px[l] = px; r.py[l] = py;
        const Rect n = r.need[l];
        Rect q;
        q.x0 = sat((n.x0 + px) / 2, M); q.y0 = sat((n.y0 + py) / 2, M);
        q.x1 = std::min(npx, (n.x1 - 1 + px) / 2 + M + 1); q.y1 = std::min(npy, (n.y1 - 1 + py) / 2 + M + 1);
        r.pairs[l] = q;
        Rect lo;                                             // what of LL_{l+1} those pairs read
        lo.x0 = std::min(sat(q.x0, px), sw); lo.y0 = std::min(sat(q.y0, py), sh);
        lo.x1 = std::min(std::max(sat(q.x1, px), lo.x0 + 1), sw); lo.y1 = std::min(std::max(sat(q.y1, py), lo.y0 + 1), sh);
        r.need[l + 1] = lo;
    }
    return r;
}

// d_pixels != nullptr: the last level writes the pixels itself (K7 fused, out_bytes 1 or 2) and d_out is not touched;
// plan != nullptr: only what the window needs is synthesised, and d_pixels is the window (K7 fused required)
int run_idwt(grk_amd_ctx* c, uint32_t nplanes, const void* d_mallat, void* d_out, void* d_pixels = nullptr,
             uint32_t ntiles = 0, uint32_t out_bytes = 0, const RegionPlan* plan = nullptr, bool h16 = false)
{
    const TileGeom& g = c->geom;
    const uint32_t L = g.p.num_levels;
    const uint32_t W = g.p.tile_w, H = g.p.tile_h;
    if (L == 0) {
        HIP_TRY(c, hipMemcpyAsync(d_out, d_mallat, (size_t)nplanes * g.plane_elems * 4, hipMemcpyDeviceToDevice, c->stream), "copy planes");
        return GRK_AMD_OK;
    }
    // same ping-pong storage as the forward transform: A holds LL1, LL3, ...; B holds LL2, LL4, ...
    const uint32_t sA = ll_stride_for(W), hA = (H + 1) >> 1;
    const uint32_t sB = ll_stride_for((W + 1) >> 1), hB = (hA + 1) >> 1;
    const uint64_t pitchA = (uint64_t)sA * hA, pitchB = (uint64_t)sB * hB;
    HIP_TRY(c, c->llA.ensure((size_t)nplanes * pitchA * 4 + 256), "alloc LL ping");
    HIP_TRY(c, c->llB.ensure((size_t)nplanes * pitchB * 4 + 256), "alloc LL pong");
    ScopedTimer t(c, 6);
    for (int32_t l = (int32_t)L - 1; l >= 0; --l) {
        IdwtLevelArgs a{};
        a.cw = level_geom(g, (uint32_t)l).w; a.ch = level_geom(g, (uint32_t)l).h;
        a.px = level_geom(g, (uint32_t)l).x0 & 1u; a.py = level_geom(g, (uint32_t)l).y0 & 1u;
        if ((uint32_t)l + 1 == L) { a.ll = (const int32_t*)d_mallat; a.ll_stride = g.stride; a.ll_pitch = g.plane_elems; }
        else if ((l + 1) & 1) { a.ll = (const int32_t*)c->llA.p; a.ll_stride = sA; a.ll_pitch = pitchA; }
        else { a.ll = (const int32_t*)c->llB.p; a.ll_stride = sB; a.ll_pitch = pitchB; }
        a.mallat = (const int32_t*)d_mallat; a.m_stride = g.stride; a.m_pitch = g.plane_elems;
        if (l == 0) { a.out = (int32_t*)d_out; a.out_stride = g.stride; a.out_pitch = g.plane_elems; }
        else if (l & 1) { a.out = (int32_t*)c->llA.p; a.out_stride = sA; a.out_pitch = pitchA; }
        else { a.out = (int32_t*)c->llB.p; a.out_stride = sB; a.out_pitch = pitchB; }
        a.nplanes = nplanes;
        a.irreversible = g.p.irreversible;
        a.xcd = c->dwt_xcd;
        a.h16 = h16 ? 1 : 0; a.status = (unsigned int*)c->flag.p;
        a.pk = h16 && c->dwt_pk && !plan;            // (the block decoder flagged every coefficient outside the packed range)
        const uint32_t sh = (a.ch + a.py + 1) >> 1;
        uint32_t seg = 64;
        const uint64_t strips = (((a.cw + a.px + 1) >> 1) + idwt_level_strip_pairs(a) - 1) / idwt_level_strip_pairs(a);
        const uint32_t zslots = (l == 0 && d_pixels) ? ntiles * ((g.p.mct && g.p.num_comps >= 3) ? 1u : g.p.num_comps) : nplanes;
        while (seg > 8 && strips * ((sh + seg - 1) / seg) * zslots < 4096) seg >>= 1;
        a.seg_pairs = seg;
        a.wx0 = 0; a.wy0 = 0; a.wx1 = a.cw; a.wy1 = a.ch;
        if (plan) {       // the strips and row segments that produce need[l]
            const Rect n = plan->need[(uint32_t)l];
            const uint32_t op = idwt_strip_pairs();
            seg = 16;
            a.seg_pairs = seg;
            a.strip0 = ((n.x0 + a.px) / 2) / op; a.nstrips = ((n.x1 - 1 + a.px) / 2) / op - a.strip0 + 1;
            a.seg0 = ((n.y0 + a.py) / 2) / seg; a.nsegs = ((n.y1 - 1 + a.py) / 2) / seg - a.seg0 + 1;
            if (l == 0) { a.wx0 = n.x0; a.wy0 = n.y0; a.wx1 = n.x1; a.wy1 = n.y1; }
        }
        if (a.cw == 0 || a.ch == 0) continue;       // (a level without samples, see run_dwt)
        if (l == 0 && c->dec_top_pending) {           // the top resolution's blocks are decoded on the side stream
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_dec_top, 0), "wait for the top resolution's blocks");
            c->dec_top_pending = false;
        }
        if (l == 0 && d_pixels) {
            a.pixels = d_pixels; a.px_bytes = out_bytes;
            a.dc = g.p.sgnd ? 0 : (1 << (g.p.prec - 1));
            a.lo = g.p.sgnd ? -(1 << (g.p.prec - 1)) : 0;
            a.hi = g.p.sgnd ? (1 << (g.p.prec - 1)) - 1 : (1 << g.p.prec) - 1;
            a.mct = g.p.mct;
            HIP_TRY(c, launch_idwt_level0_fused(a, ntiles, g.p.num_comps, c->stream), "launch fused idwt level 0");
        } else {
            HIP_TRY(c, launch_idwt_level(a, c->stream), "launch idwt level");
        }
    }
    return GRK_AMD_OK;
}

// The pinned tables of this call with the caller's rows in them (room for K5's index behind the rows); the set's last upload
// has been waited for (two calls ago: long done)
int stage_table(grk_amd_ctx* c, const grk_amd_coded_block* table, uint64_t nblocks, grk_amd_ctx::DecUpload** out)
{
    grk_amd_ctx::DecUpload* u = &c->dec_up[c->dec_turn++ & 1u];
    if (!u->ev) HIP_TRY(c, hipEventCreateWithFlags(&u->ev, hipEventDisableTiming), "create event");
    HIP_TRY(c, hipEventSynchronize(u->ev), "wait for the tables' last upload");
    const size_t need = (size_t)nblocks * (sizeof(grk_amd_coded_block) + 16) + 64;        // rows + the launch lists behind them
    if (u->cap < need) {
        if (u->p) (void)hipHostFree(u->p);
        u->p = u->dp = nullptr; u->cap = 0;
        HIP_TRY(c, hipHostMalloc((void**)&u->p, need, hipHostMallocDefault), "alloc pinned tables");
        HIP_TRY(c, hipHostGetDevicePointer((void**)&u->dp, u->p, 0), "map pinned tables");
        u->cap = need;
    }
    std::memcpy(u->p, table, (size_t)nblocks * sizeof(grk_amd_coded_block));
    *out = u;
    return GRK_AMD_OK;
}

// rows (+ `extra` bytes behind them) -> dec_table on the call's stream, the status block cleared
int upload_table(grk_amd_ctx* c, grk_amd_ctx::DecUpload* u, size_t bytes)
{
    HIP_TRY(c, c->dec_table.ensure(bytes + 64), "alloc decode table");
    HIP_TRY(c, c->flag.ensure(kHtAllocBytes), "alloc status");
    HIP_TRY(c, launch_dec_upload(u->dp, c->dec_table.p, bytes, c->flag.p, c->stream), "upload decode tables");
    HIP_TRY(c, hipEventRecord(u->ev, c->stream), "record the tables' upload");
    return GRK_AMD_OK;
}

int run_ht_decode(grk_amd_ctx* c, uint32_t ntiles, grk_amd_ctx::DecUpload* up, const void* d_coded, uint64_t coded_bytes, void* d_mallat,
                  bool h16 = false, bool split = false)
{
    const TileGeom& g = c->geom;
    const uint32_t bpt = g.blocks_per_comp * g.p.num_comps;
    const uint64_t nblocks = (uint64_t)bpt * ntiles;
    const grk_amd_coded_block* const table = (const grk_amd_coded_block*)up->p;
    uint32_t max_len = 0;
    // behind the rows: the blocks that have data at all -- K5a's lanes (a window's skipped blocks and absent blocks do not cost a
    // lane of a serial chain)
    uint32_t* const h_active = (uint32_t*)(up->p + nblocks * sizeof(grk_amd_coded_block));
    uint32_t nactive = 0;
    for (uint64_t i = 0; i < nblocks; ++i) {
        max_len = std::max(max_len, table[i].length);
        if (table[i].offset > coded_bytes || table[i].length > coded_bytes - table[i].offset)
            return fail(c, GRK_AMD_ERR_INVALID, "block table row points outside the coded buffer");
        if (table[i].length) h_active[nactive++] = (uint32_t)i;
    }
    if (max_len > (48u << 10)) return fail(c, GRK_AMD_ERR_UNSUPPORTED, "code-block longer than 48 KiB");
    static_assert(sizeof(HtDecBlock) == sizeof(grk_amd_coded_block), "decode table rows are grk_amd_coded_block");
    HIP_TRY(c, c->dec_quads.ensure(nblocks * 1024 * 2 + 64), "alloc quad info");
    HIP_TRY(c, c->dec_mslen.ensure(nblocks * 4), "alloc ms lengths");
    { const int rc = upload_table(c, up, nblocks * sizeof(HtDecBlock) + (size_t)nactive * 4); if (rc) return rc; }
    const uint32_t* const d_active = (const uint32_t*)((const char*)c->dec_table.p + nblocks * sizeof(HtDecBlock));
    HtDecArgs a{};
    a.table = (const HtDecBlock*)c->dec_table.p;
    a.blocks = (const HtBlockDesc*)c->dec_desc.p; a.blocks_per_tile = bpt; a.nblocks = (uint32_t)nblocks; a.ncomp = g.p.num_comps;
    a.coded = (const uint8_t*)d_coded; a.coded_bytes = coded_bytes;
    a.quads = (uint32_t*)c->dec_quads.p; a.ms_len = (uint32_t*)c->dec_mslen.p; a.status = (unsigned int*)c->flag.p;
    a.active = nactive == nblocks ? nullptr : d_active; a.nactive = nactive;
    a.mallat = (int32_t*)d_mallat; a.stride = g.stride; a.pitch = g.plane_elems;
    a.irreversible = g.p.irreversible;
    a.h16 = h16 ? 1 : 0;
    a.h16_bias = (h16 && c->dwt_pk) ? 2048 : 32768;        // (pk16.h kPkDecodeBound + 1: the inverse transform runs on packed pairs)
    if (!c->dec_seg_first.empty()) {
        // HT blocks with refinement passes: segment 0 = the cleanup pass, segment 1 = SigProp (+ MagRef), end to end
        if (c->dec_seg_first.size() != nblocks + 1 || c->dec_seg_first.back() != c->dec_segs.size())
            return fail(c, GRK_AMD_ERR_INVALID, "segment list does not match the number of blocks");
        std::vector<uint2> ref(nblocks, make_uint2(0u, 1u));
        for (uint64_t i = 0; i < nblocks; ++i) {
            const uint32_t s0 = c->dec_seg_first[i], ns = c->dec_seg_first[i + 1] - s0;
            if (ns > 2) return fail(c, GRK_AMD_ERR_INVALID, "an HT code-block has at most two codeword segments");
            uint64_t sum = 0;
            for (uint32_t k = 0; k < ns; ++k) sum += c->dec_segs[s0 + k].length;
            if (ns && sum != table[i].length) return fail(c, GRK_AMD_ERR_INVALID, "segment lengths do not add up to the block's length");
            if (ns == 2 && c->dec_segs[s0 + 1].length) {
                const uint32_t passes = 1u + std::min<uint32_t>(c->dec_segs[s0 + 1].numpasses, 2u);
                ref[i] = make_uint2(c->dec_segs[s0 + 1].length, passes);
                a.max_refine_bytes = std::max(a.max_refine_bytes, ref[i].x);
            }
        }
        if (a.max_refine_bytes > (16u << 10)) return fail(c, GRK_AMD_ERR_UNSUPPORTED, "refinement segment longer than 16 KiB");
        HIP_TRY(c, c->dec_seg_dev.ensure(nblocks * sizeof(uint2) + 16), "alloc refinement table");
        HIP_TRY(c, hipMemcpyAsync(c->dec_seg_dev.p, ref.data(), nblocks * sizeof(uint2), hipMemcpyHostToDevice, c->stream), "upload refinement table");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "sync refinement table");       // (uploaded from a local)
        a.refine = (const uint2*)c->dec_seg_dev.p;
    }
    // K5b in two parts when the call goes on with the inverse transform (decode_impl): the levels below the last one need the
    // blocks of the lower resolutions only -- a quarter of them --, and those short, latency-bound launches hide beside the
    // top resolution's K5b on the low-priority side stream
    const uint32_t L = g.p.num_levels;
    const uint32_t first_top = L >= 1 ? g.res[L].band[0].first_block : 0;
    if (split && c->overlap && c->side && L >= 2 && !a.refine && first_top > 0 && first_top < g.blocks_per_comp) {
        if (!c->ev_dec_front) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_dec_front, hipEventDisableTiming), "create event");
        if (!c->ev_dec_top) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_dec_top, hipEventDisableTiming), "create event");
        HIP_TRY(c, launch_ht_decode_front(a, c->stream), "launch ht decode");
        HIP_TRY(c, hipEventRecord(c->ev_dec_front, c->stream), "record K5a");
        HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_dec_front, 0), "side stream waits for K5a");
        a.ms_bpc = g.blocks_per_comp;
        a.ms_first = first_top; a.ms_count = g.blocks_per_comp - first_top;
        HIP_TRY(c, launch_ht_decode_ms(a, max_len, c->side), "launch K5b, top resolution");
        HIP_TRY(c, hipEventRecord(c->ev_dec_top, c->side), "record K5b");
        c->dec_top_pending = true;
        a.ms_first = 0; a.ms_count = first_top;
        HIP_TRY(c, launch_ht_decode_ms(a, max_len, c->stream), "launch K5b, lower resolutions");
        return GRK_AMD_OK;
    }
    ScopedTimer t(c, 5);
    HIP_TRY(c, launch_ht_decode(a, max_len, c->stream), "launch ht decode");
    return GRK_AMD_OK;
}

int run_t1_decode(grk_amd_ctx* c, uint32_t ntiles, grk_amd_ctx::DecUpload* up, const void* d_coded, uint64_t coded_bytes, void* d_mallat)
{
    const TileGeom& g = c->geom;
    const uint32_t bpt = g.blocks_per_comp * g.p.num_comps;
    const uint64_t nblocks = (uint64_t)bpt * ntiles;
    const grk_amd_coded_block* const table = (const grk_amd_coded_block*)up->p;
    for (uint64_t i = 0; i < nblocks; ++i)
        if (table[i].offset > coded_bytes || table[i].length > coded_bytes - table[i].offset)
            return fail(c, GRK_AMD_ERR_INVALID, "block table row points outside the coded buffer");
    static_assert(kT1WorkBytes == 4096 * 4, "K8 and K8L share a block's part of the workspace");
    HIP_TRY(c, c->dec_work.ensure(nblocks * kT1WorkBytes), "alloc Part-1 workspace");
    // Which decoder takes which block.  A block is one dependent chain of MQ decisions (about ten per coded byte); 64 chains
    // to a wave (K8L) make the throughput, but a chain alone in a wave (K8) advances ~2.5 times faster, and a frame's time
    // is its longest chain's: the blocks longer than t1_tail_ratio x the longest one -- a handful: the LL band -- and
    // whatever the lane form does not take go to K8, longest first; the rest to K8L, sorted by length so that the lanes of a
    // wave finish together.  Both lists behind the rows in the pinned tables (stage_table leaves 8 bytes per block).
    uint32_t* const h_lane = (uint32_t*)(up->p + nblocks * sizeof(grk_amd_coded_block));      // (room for 2 nblocks entries: padding)
    uint32_t* const h_tail = h_lane + 2 * nblocks;
    uint32_t n_lane = 0, n_tail = 0;
    const bool lanes_on = c->t1_lanes && g.p.reserved[1] == 0 && c->dec_seg_first.empty() && nblocks <= 0xFFFFFFFFull;
    if (lanes_on) {
        auto eligible = [&](uint64_t i) {
            const uint32_t bps = table[i].missing_msbs & 0xFFu, np = table[i].missing_msbs >> 8;
            // (a row with more passes than its bit-planes can have -- a malformed packet header -- would alias into another group of
            //  the pass-synchronous waves: K8 takes it and stops where the data does)
            return table[i].length != 0 && table[i].missing_msbs != kSkipBlock && np != 0 && bps != 0 && bps <= kT1LaneMaxPlanes &&
                   np <= 3u * bps - 2u && c->h_desc_dec[i % bpt].h >= kT1LaneMinRows;
        };
        uint32_t max_len = 0;
        for (uint64_t i = 0; i < nblocks; ++i) max_len = std::max(max_len, table[i].length);
        const uint32_t thr = (uint32_t)std::min<double>((double)max_len, std::max(64.0, (double)c->t1_tail_ratio * max_len));
        // counting sort by length (4-byte buckets), longest first.  The bucket index is clamped: a code-block of 64 x 64 samples
        // cannot need more than 64 KiB, and a row that CLAIMS hundreds of megabytes (a malformed packet header: the length is
        // bounded by the coded buffer only) must not cost a table of that size -- such rows share the top bucket, i.e. sort first
        // and go to K8's list like every long block
        constexpr uint32_t kMaxBucketLen = 64u << 10;
        const uint32_t nb = (std::min(max_len, kMaxBucketLen) >> 2) + 2u;
        auto bucket = [&](uint64_t i) { return nb - 1u - (std::min(table[i].length, kMaxBucketLen) >> 2); };
        std::vector<uint32_t> cnt, order;
        try { cnt.assign(nb + 1, 0u); order.resize(nblocks); }
        catch (const std::bad_alloc&) { return fail(c, GRK_AMD_ERR_NOMEM, "host memory for the Part-1 launch lists"); }
        for (uint64_t i = 0; i < nblocks; ++i) cnt[bucket(i)]++;
        uint32_t run = 0;
        for (uint32_t k = 0; k <= nb; ++k) { const uint32_t v = cnt[k]; cnt[k] = run; run += v; }
        for (uint64_t i = 0; i < nblocks; ++i) order[cnt[bucket(i)]++] = (uint32_t)i;
        const uint64_t share = (uint64_t)((double)c->t1_tail_share * (double)nblocks);
        for (uint64_t k = 0; k < nblocks; ++k) {
            const uint32_t i = order[k];
            if (k >= share && table[i].length <= thr && eligible(i)) h_lane[n_lane++] = i; else h_tail[n_tail++] = i;
        }
        if (n_lane >= 64u && c->t1_pass_sync) {
            // pass-synchronous waves: a wave's lanes go from pass to pass together, so a wave holds blocks with the SAME number of
            // bit-planes and passes (table word missing_msbs), longest first within the group; a group fills whole waves (spare
            // lanes: kT1NoBlock); groups too small for a wave go to K8
            std::vector<uint32_t> lane(h_lane, h_lane + n_lane);
            auto key = [&](uint32_t i) { return (((table[i].missing_msbs >> 8) & 0xFFu) << 4) | (table[i].missing_msbs & 0xFu); };   // passes, planes (<= 14)
            constexpr uint32_t kKeys = 256u << 4;
            std::vector<uint32_t> cnt(kKeys, 0u), at(kKeys, 0u);
            for (uint32_t i : lane) cnt[key(i)]++;
            uint32_t out = 0;
            // (a group that would fill only a few waves runs them from pass to pass half empty, and with more passes than the
            //  bulk it is the kernel's last wave to finish: groups below 0.5 % of the lane blocks go to K8 as well)
            const uint32_t min_group = std::max<uint32_t>(64u, n_lane / 200u);
            for (uint32_t k = kKeys; k-- > 0;) {                              // (more passes first: the longest-running waves start first)
                if (cnt[k] < min_group) { at[k] = kT1NoBlock; continue; }
                at[k] = out;
                out += (cnt[k] + 63u) & ~63u;
            }
            for (uint32_t j = 0; j < out; ++j) h_lane[j] = kT1NoBlock;
            for (uint32_t i : lane) {                                         // (the groups keep the longest-first order)
                const uint32_t k = key(i);
                if (at[k] == kT1NoBlock) h_tail[n_tail++] = i; else h_lane[at[k]++] = i;
            }
            n_lane = out;
        }
        if (n_lane >= 64u) {
            // Is the lane form the faster one for THIS call?  A lane's chain advances at ~10 ns per coded byte (0.85 us per step, ~10
            // decisions per byte, ~30 % of the steps idle), a wave's at ~2.5 ns per byte, and K8's throughput with every SIMD full is
            // ~0.9 ns per byte (r03: 55 MB in 48 ms): a small image -- fewer blocks than K8 has wave slots -- is done sooner by K8
            // alone, in the time of its longest block.
            uint64_t bytes_all = 0, bytes_tail = 0;
            uint32_t max_lane = 0, max_tail = 0;
            for (uint64_t i = 0; i < nblocks; ++i) bytes_all += table[i].length;
            for (uint32_t j = 0; j < n_lane; ++j) if (h_lane[j] != kT1NoBlock) max_lane = std::max(max_lane, table[h_lane[j]].length);
            for (uint32_t j = 0; j < n_tail; ++j) { bytes_tail += table[h_tail[j]].length; max_tail = std::max(max_tail, table[h_tail[j]].length); }
            const double t_k8 = std::max(2.5e-9 * max_len, 0.9e-9 * (double)bytes_all);
            const double t_mix = std::max(std::max(10.0e-9 * max_lane, 2.5e-9 * max_tail), 0.9e-9 * (double)bytes_tail);
            if (t_k8 <= t_mix && c->t1_lanes != 2) n_lane = 0;
        }
        if (n_lane < 64u) { n_lane = 0; n_tail = 0; }                   // not worth a second launch: K8 in table order
    }
    { const int rc = upload_table(c, up, nblocks * sizeof(HtDecBlock) + nblocks * 12); if (rc) return rc; }
    const uint32_t* const d_lane = (const uint32_t*)((const char*)c->dec_table.p + nblocks * sizeof(HtDecBlock));
    T1DecArgs a{};
    a.table = (const HtDecBlock*)c->dec_table.p;
    a.blocks = (const HtBlockDesc*)c->dec_desc.p; a.blocks_per_tile = bpt; a.nblocks = (uint32_t)nblocks; a.ncomp = g.p.num_comps;
    a.coded = (const uint8_t*)d_coded; a.coded_bytes = coded_bytes;
    a.work = (int32_t*)c->dec_work.p; a.status = (unsigned int*)c->flag.p;
    a.mallat = (int32_t*)d_mallat; a.stride = g.stride; a.pitch = g.plane_elems;
    a.irreversible = g.p.irreversible;
    a.cblksty = g.p.reserved[1];
    if (!c->dec_seg_first.empty()) {
        if (c->dec_seg_first.size() != nblocks + 1 || c->dec_seg_first.back() != c->dec_segs.size())
            return fail(c, GRK_AMD_ERR_INVALID, "segment list does not match the number of blocks");
        static_assert(sizeof(grk_amd_segment) == sizeof(uint2), "segments are {bytes, passes}");
        const size_t nf = c->dec_seg_first.size() * 4, ns = c->dec_segs.size() * sizeof(grk_amd_segment);
        const size_t ns_off = (nf + 15) & ~(size_t)15;
        HIP_TRY(c, c->dec_seg_dev.ensure(ns_off + ns + 16), "alloc segment list");
        HIP_TRY(c, hipMemcpyAsync(c->dec_seg_dev.p, c->dec_seg_first.data(), nf, hipMemcpyHostToDevice, c->stream), "upload segment index");
        if (ns) HIP_TRY(c, hipMemcpyAsync((char*)c->dec_seg_dev.p + ns_off, c->dec_segs.data(), ns, hipMemcpyHostToDevice, c->stream), "upload segments");
        a.seg_first = (const uint32_t*)c->dec_seg_dev.p;
        a.segs = (const uint2*)((const char*)c->dec_seg_dev.p + ns_off);
    }
    ScopedTimer t(c, 5);
    if (n_lane) {
        T1LaneArgs la{};
        la.table = a.table; la.blocks = a.blocks; la.blocks_per_tile = bpt; la.ncomp = a.ncomp;
        la.list = d_lane; la.count = n_lane;
        la.coded = a.coded; la.coded_bytes = coded_bytes;
        la.work = (uint64_t*)c->dec_work.p;
        la.mallat = a.mallat; la.stride = a.stride; la.pitch = a.pitch; la.irreversible = a.irreversible;
        la.pass_sync = c->t1_pass_sync ? 1 : 0;
        a.list = d_lane + 2 * nblocks; a.count = n_tail;
        // one launch, one stream (r06): the long chains are the launch's first workgroups, the lane waves follow; a decode SEQUENCE then
        // needs one hardware queue per frame in flight instead of two (GRK_AMD_T1_FUSED=0: two launches on two streams, as r04-r05)
        static const bool fused_t1 = !(getenv("GRK_AMD_T1_FUSED") && atoi(getenv("GRK_AMD_T1_FUSED")) == 0);
        if (fused_t1 && n_tail) {
            HIP_TRY(c, launch_t1_fused(a, la, c->stream), "launch Part-1 decode (both decoders)");
            return GRK_AMD_OK;
        }
        if (c->overlap && c->side) {
            // the long chains on the call's stream, the lanes beside them on the side stream
            if (!c->ev_dec_front) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_dec_front, hipEventDisableTiming), "create event");
            if (!c->ev_dec_top) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_dec_top, hipEventDisableTiming), "create event");
            HIP_TRY(c, hipEventRecord(c->ev_dec_front, c->stream), "record the tables");
            HIP_TRY(c, hipStreamWaitEvent(c->side, c->ev_dec_front, 0), "side stream waits for the tables");
            HIP_TRY(c, launch_t1_decode(a, c->stream), "launch Part-1 decode (long blocks)");
            HIP_TRY(c, launch_t1_lanes(la, c->side), "launch Part-1 decode (lanes)");
            HIP_TRY(c, hipEventRecord(c->ev_dec_top, c->side), "record the lanes");
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_dec_top, 0), "join the lanes");
        } else {
            HIP_TRY(c, launch_t1_decode(a, c->stream), "launch Part-1 decode (long blocks)");
            HIP_TRY(c, launch_t1_lanes(la, c->stream), "launch Part-1 decode (lanes)");
        }
        return GRK_AMD_OK;
    }
    HIP_TRY(c, launch_t1_decode(a, c->stream), "launch Part-1 decode");
    return GRK_AMD_OK;
}

int check_decode_status(grk_amd_ctx* c)
{
    uint32_t st = 0;
    if (!c->flag.p) return GRK_AMD_OK;                 // nothing was decoded on this context (a sequence's frames are on its children)
    HIP_TRY(c, hipMemcpyAsync(&st, c->flag.p, 4, hipMemcpyDeviceToHost, c->stream), "fetch status");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    if (st & 4u) return fail(c, GRK_AMD_ERR_INVALID, "corrupt HT code-block (bad Scup or U_q > missing_msbs)");
    if (st & 8u) return fail(c, GRK_AMD_ERR_RANGE, "a coefficient left the 16-bit planes: decode again after grk_amd_set_decode_planes16(ctx, 0)");
    return GRK_AMD_OK;
}

int run_egress(grk_amd_ctx* c, uint32_t ntiles, const void* d_planes, void* d_pixels, uint32_t out_bytes)
{
    const TileGeom& g = c->geom;
    EgressArgs a{};
    a.planes = (const int32_t*)d_planes; a.pixels = d_pixels;
    a.w = g.p.tile_w; a.h = g.p.tile_h; a.stride = g.stride; a.pitch = g.plane_elems;
    a.ncomp = g.p.num_comps; a.ntiles = ntiles;
    a.bytes_per_sample = out_bytes;
    a.dc = g.p.sgnd ? 0 : (1 << (g.p.prec - 1));
    a.lo = g.p.sgnd ? -(1 << (g.p.prec - 1)) : 0;
    a.hi = g.p.sgnd ? (1 << (g.p.prec - 1)) - 1 : (1 << g.p.prec) - 1;
    a.mct = g.p.mct; a.irreversible = g.p.irreversible;
    ScopedTimer t(c, 7);
    HIP_TRY(c, launch_egress(a, c->stream), "launch egress");
    return GRK_AMD_OK;
}

HtArgs make_ht_args(grk_amd_ctx* c, uint32_t ntiles, const void* d_mallat, int* rc, bool h16)
{
    HtArgs a{};
    *rc = GRK_AMD_OK;
    auto try_ = [&](hipError_t e, const char* what) { if (e != hipSuccess && *rc == GRK_AMD_OK) *rc = fail(c, GRK_AMD_ERR_NO_DEVICE, what, e); };
    const TileGeom& g = c->geom;
    const uint32_t bpt = g.blocks_per_comp * g.p.num_comps;
    const uint64_t nblocks = (uint64_t)bpt * ntiles;
    try_(c->lengths.ensure(nblocks * 4), "alloc lengths");
    try_(c->offsets.ensure((nblocks + 1) * 8), "alloc offsets");
    try_(c->flag.ensure(kHtAllocBytes), "alloc allocator state");
    // arena: worst case of the HT cleanup pass is ~ (kmax+1)/8 * 8/7 bytes per sample + VLC/MEL;
    // twice the raw input size plus per-block slack covers every lossless case we accept
    const uint64_t raw = (uint64_t)ntiles * g.p.num_comps * g.p.tile_w * g.p.tile_h * ((g.p.prec + 7) / 8);
    // Allocation regions: every block reserves its bytes with an atomic on its region's word, and the blocks of a launch that fits
    // the machine in one round (up to ~6 000) all arrive there within microseconds of each other -- atomics on ONE address are
    // served one after the other, and a chunk refill makes the region's other waves wait.  At least one region per 64 blocks (r04:
    // with one per 256, K3 of a 2048^2 frame took 0.151 ms, with this 0.048; 1024^2 0.079 -> 0.038, 3072^2 0.177 -> 0.069; from
    // 4096^2 on all 64 regions were in use before: tools/k3_sizes.py); small jobs take smaller chunks, so that the slack of the
    // regions' half-used chunks stays small against their coded bytes.
    static const uint32_t kBlocksPerRegion = getenv("GRK_AMD_BLOCKS_PER_REGION") ? (uint32_t)std::max(1, atoi(getenv("GRK_AMD_BLOCKS_PER_REGION"))) : 64u;
    uint32_t regions = 1;
    while (regions < kHtAllocRegions && nblocks / (regions * 2) >= kBlocksPerRegion) regions *= 2;
    // (a chunk holds at least two of the largest blocks the geometry can produce: worst case (Kmax + 2) bits per sample and 15 VLC
    //  bits per quad, stuffing 1 bit in 15, 256 MEL bytes -- ~20 KiB for a 64 x 64 block at Kmax 31)
    size_t worst_block = 0;
    for (uint32_t k = 0; k < c->ht_num_classes; ++k) {
        const HtClass& hc = c->ht_classes[k];
        worst_block = std::max(worst_block, ((size_t)hc.max_samples * (hc.max_kmax + 2u) + (size_t)hc.max_quads * 15u) * 16u / 15u / 8u + 280u);
    }
    const uint32_t chunk = (nblocks < 16384 && 2 * worst_block <= kHtAllocChunkSmall) ? kHtAllocChunkSmall : kHtAllocChunk;
    try_(c->arena.ensure(raw * 2 + nblocks * 64 + (size_t)(regions + 1) * kHtAllocChunk + (1u << 20)), "alloc coded arena");
    a.mallat = (const int32_t*)d_mallat; a.stride = g.stride; a.pitch = g.plane_elems; a.h16 = h16 ? 1 : 0;
    a.blocks = (const HtBlockDesc*)c->blockdesc.p; a.blocks_per_tile = bpt; a.ncomp = g.p.num_comps; a.ntiles = ntiles;
    a.arena = (uint8_t*)c->arena.p; a.arena_bytes = c->arena.cap;
    a.alloc = (unsigned long long*)c->flag.p;        // [0] status flags, [1] bytes used (launch_ht_alloc_init resets them)
    a.lengths = (uint32_t*)c->lengths.p; a.offsets = (unsigned long long*)c->offsets.p;
    try_(c->ovf.ensure(2 * nblocks * 4 + 16), "alloc fallback list");      // (every block is in two classes)
    a.ovf_list = c->lds_cap ? (uint32_t*)c->ovf.p : nullptr;
    a.region_mask = regions - 1;
    a.chunk_units = chunk / 16u;
    a.irreversible = g.p.irreversible;
    a.num_classes = c->ht_num_classes;
    uint32_t ovf_base = 0;
    for (uint32_t k = 0; k < c->ht_num_classes; ++k) {
        a.classes[k] = c->ht_classes[k];
        a.classes[k].ovf_base = ovf_base;
        ovf_base += c->ht_classes[k].count * ntiles;
    }
    return a;
}

// overlapped: the top resolution and the large-LDS classes are already running on the side streams (run_dwt)
int run_ht(grk_amd_ctx* c, uint32_t ntiles, const void* d_mallat, bool overlapped = false, bool h16 = false, bool room = false)
{
    int rc = GRK_AMD_OK;
    HtArgs a = make_ht_args(c, ntiles, d_mallat, &rc, h16);
    if (rc) return rc;
    a.room = room ? 1 : 0;
    {
        ScopedTimer t(c, 2);
        if (!overlapped) {         // one launch of every block where there is such a class, else class by class
            HIP_TRY(c, launch_ht_alloc_init(a, c->stream), "reset arena allocator");
            bool all = false;
            for (uint32_t k = 0; k < a.num_classes; ++k) all = all || c->ht_class_top[k] == 2;
            for (uint32_t k = 0; k < a.num_classes; ++k)
                if ((c->ht_class_top[k] == 2) == all) HIP_TRY(c, launch_ht_classes(a, k, k + 1, c->stream), "launch ht encode");
        } else {
            for (uint32_t k = 0; k < a.num_classes && !c->pipelining; ++k)
                if (!c->ht_class_top[k] && !c->ht_class_big[k]) HIP_TRY(c, launch_ht_classes(a, k, k + 1, c->stream), "launch ht encode");
        }
    }
    if (overlapped) {
        c->side_pending = true;
        if (!c->pipelining) { const int jr = join_side(c); if (jr) return jr; }    // pipelining: the next consumer joins
    }
    c->last_ntiles = ntiles;
    c->last_nblocks = (uint64_t)c->geom.blocks_per_comp * c->geom.p.num_comps * ntiles;
    return GRK_AMD_OK;
}

} // namespace

extern "C" {

const char* grk_amd_version(void) { return "grok_amd 0.1 (gfx950)"; }

namespace {
// decode_only: one of a decode sequence's internal contexts (grk_amd_set_decode_pipelining) -- the call's stream and ONE side
// stream, nothing else: the HIP runtime deals its (default 4) hardware queues to streams in the order they are made, and two
// frames in flight then sit on four queues of their own whatever else the process has made before (a third stream per context
// that decoding never uses made frame 2's lane kernel share a queue with frame 1's long chains: 14.8 instead of 9.1 ms per frame)
int create_context(int device_id, int verbose, bool decode_only, grk_amd_ctx** out);
}

int grk_amd_create(int device_id, int verbose, grk_amd_ctx** out) { return create_context(device_id, verbose, false, out); }

namespace {
int create_context(int device_id, int verbose, bool decode_only, grk_amd_ctx** out)
{
    if (!out) return GRK_AMD_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GRK_AMD_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= n) return GRK_AMD_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return GRK_AMD_ERR_NO_DEVICE;
    auto* c = new grk_amd_ctx();
    c->device = device_id; c->verbose = verbose;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return GRK_AMD_ERR_NO_DEVICE; }
    c->own_stream = true;
    {   // side stream for K3 of the top resolution (lowest priority: the DWT chain on the main stream is the critical path)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (const char* e16 = getenv("GRK_AMD_PLANES16")) c->planes16 = atoi(e16) != 0;
        if (const char* ef = getenv("GRK_AMD_FUSE_EGRESS")) c->fuse_egress = atoi(ef) != 0;
        if (const char* ex = getenv("GRK_AMD_DWT_XCD")) c->dwt_xcd = atoi(ex) != 0;
        if (const char* ex = getenv("GRK_AMD_DWT_PK")) c->dwt_pk = atoi(ex) != 0;
        if (const char* ed = getenv("GRK_AMD_DEC_PLANES16")) c->dec_planes16 = atoi(ed) != 0;
        if (const char* el = getenv("GRK_AMD_LDS_CAP")) c->lds_cap = atoi(el) != 0;
        if (const char* ek = getenv("GRK_AMD_K3_ROOM")) c->k3_room = atoi(ek) & 3;
        if (const char* ef2 = getenv("GRK_AMD_FRAME_STREAMS")) c->frame_streams = atoi(ef2);
        if (const char* et = getenv("GRK_AMD_T1_LANES")) c->t1_lanes = atoi(et);
        if (const char* er = getenv("GRK_AMD_T1_TAIL_RATIO")) c->t1_tail_ratio = (float)atof(er);
        if (const char* es = getenv("GRK_AMD_T1_TAIL_SHARE")) c->t1_tail_share = (float)atof(es);
        if (const char* ey = getenv("GRK_AMD_T1_SYNC")) c->t1_pass_sync = atoi(ey) != 0;
        if (const char* ep = getenv("GRK_AMD_STREAM_PROBE")) c->stream_probe = atoi(ep);
        if (const char* ea = getenv("GRK_AMD_ALLOC_IN_LEVEL0")) c->alloc_in_level0 = atoi(ea) != 0;
        c->side_priority = least;
        const char* e = getenv("GRK_AMD_OVERLAP");
        c->overlap = e ? atoi(e) != 0 : GRK_AMD_OVERLAP_DEFAULT;
        if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, least) != hipSuccess ||
            (!decode_only && hipStreamCreateWithPriority(&c->side2, hipStreamNonBlocking, least) != hipSuccess) ||
            hipEventCreateWithFlags(&c->ev_side2, hipEventDisableTiming) != hipSuccess ||
            !create_alt_events(c) ||
            hipEventCreateWithFlags(&c->ev_level0, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_side, hipEventDisableTiming) != hipSuccess) {
            c->side = nullptr; c->overlap = false;
        }
    }
    *out = c;
    return GRK_AMD_OK;
}
} // namespace

void grk_amd_destroy(grk_amd_ctx* c)
{
    if (!c) return;
    for (grk_amd_ctx* k : c->dec_kids) grk_amd_destroy(k);
    c->dec_kids.clear();
    (void)hipSetDevice(c->device);
    if (c->ev_seq) (void)hipEventDestroy(c->ev_seq);
    if (c->ev_frame_done) (void)hipEventDestroy(c->ev_frame_done);
    (void)hipStreamSynchronize(c->stream);
    drain_timers(c);
    for (DevBuf* b : {&c->pixels, &c->p0, &c->p1, &c->llA, &c->llB, &c->blockdesc, &c->lengths,
                      &c->offsets, &c->arena, &c->flag, &c->dec_desc, &c->dec_table, &c->dec_quads, &c->dec_mslen,
                      &c->dec_coded, &c->dec_pixels, &c->dec_work, &c->ht_sel, &c->energy})
        b->release();
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->side2) { (void)hipStreamSynchronize(c->side2); (void)hipStreamDestroy(c->side2); }
    if (c->ev_side2) (void)hipEventDestroy(c->ev_side2);
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    if (c->ev_px) (void)hipEventDestroy(c->ev_px);
    for (auto& alt_set : c->alts) {
        auto* as = &alt_set;
        if (as->ev_side) (void)hipEventDestroy(as->ev_side);
        if (as->ev_side2) (void)hipEventDestroy(as->ev_side2);
        for (DevBuf* b : {&as->p1, &as->arena, &as->lengths, &as->offsets, &as->flag, &as->ovf, &as->llA, &as->llB}) b->release();
    }
    c->ovf.release();
    for (DevBuf* b : {&c->t2.packets, &c->t2.pob, &c->t2_u, &c->t2_h, &c->t2_rel, &c->t2_pkhdr, &c->t2_pkbody, &c->t2_pkdst, &c->t2_lit,
                      &c->t2_litlen, &c->t2_index})
        b->release();
    for (auto& o : c->t2_outs) for (DevBuf* b : {&o.out, &o.tile_dst, &o.part_len, &o.total}) b->release();
    if (c->ev_level0) (void)hipEventDestroy(c->ev_level0);
    if (c->ev_side) (void)hipEventDestroy(c->ev_side);
    for (DevBuf* b : {&c->dec_seg_dev}) b->release();
    c->stage.release();
    if (c->d2h_pin) (void)hipHostFree(c->d2h_pin);
    for (hipEvent_t ev : c->d2h_ev) (void)hipEventDestroy(ev);
    if (c->ev_dec_front) (void)hipEventDestroy(c->ev_dec_front);
    if (c->ev_dec_top) (void)hipEventDestroy(c->ev_dec_top);
    for (auto& u : c->dec_up) {
        if (u.p) (void)hipHostFree(u.p);
        if (u.ev) (void)hipEventDestroy(u.ev);
    }
    delete c;
}

const char* grk_amd_last_error(grk_amd_ctx* c) { return c ? c->err.c_str() : "null context"; }

int grk_amd_set_stream(grk_amd_ctx* c, void* s)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    if (s) c->stream = (hipStream_t)s;
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return GRK_AMD_ERR_NO_DEVICE;
        c->own_stream = true;
    }
    return GRK_AMD_OK;
}

int64_t grk_amd_tile_num_blocks(const grk_amd_tile_params* p)
{
    if (!p) return GRK_AMD_ERR_INVALID;
    TileGeom g;
    int rc = build_tile_geom(*p, g);
    if (rc != GRK_AMD_OK) return rc;
    return (int64_t)g.blocks_per_comp * p->num_comps;
}

int64_t grk_amd_tile_layout(const grk_amd_tile_params* p, grk_amd_block* blocks, uint64_t cap, uint16_t* qcd)
{
    if (!p) return GRK_AMD_ERR_INVALID;
    TileGeom g;
    int rc = build_tile_geom(*p, g);
    if (rc != GRK_AMD_OK) return rc;
    const uint64_t n = (uint64_t)g.blocks_per_comp * p->num_comps;
    if (blocks) {
        if (cap < n) return GRK_AMD_ERR_INVALID;
        uint64_t i = 0;
        for (uint32_t k = 0; k < p->num_comps; ++k)
            for (auto b : g.blocks_comp0) { b.comp = (uint16_t)k; blocks[i++] = b; }
    }
    if (qcd) std::memcpy(qcd, g.qcd_words, sizeof(uint16_t) * g.num_bands_total);
    return (int64_t)n;
}

int grk_amd_tile_precincts(const grk_amd_tile_params* p, uint32_t* counts)
{
    if (!p || !counts) return GRK_AMD_ERR_INVALID;
    TileGeom g;
    const int rc = build_tile_geom(*p, g);
    if (rc != GRK_AMD_OK) return rc;
    for (uint32_t r = 0; r <= p->num_levels; ++r) counts[r] = g.res[r].npw * g.res[r].nph;
    return GRK_AMD_OK;
}

uint32_t grk_amd_plane_stride(const grk_amd_tile_params* p) { return p ? ((p->tile_w + 31u) & ~31u) : 0; }
uint64_t grk_amd_plane_elems(const grk_amd_tile_params* p) { return p ? (uint64_t)grk_amd_plane_stride(p) * p->tile_h : 0; }

int grk_amd_stage_ingest_mct(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const void* d_pixels, void* d_planes)
{
    if (!c || !p || !d_pixels || !d_planes) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = ensure_geom(c, p); if (rc) return rc;
    return run_ingest(c, ntiles, d_pixels, d_planes);
}

int grk_amd_stage_dwt_fwd(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t nplanes, void* d_in, void* d_out)
{
    if (c) { const int jr = join_side(c); if (jr) return jr; }
    if (!c || !p || !d_in || !d_out) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = ensure_geom(c, p); if (rc) return rc;
    return run_dwt(c, nplanes, d_in, d_out);
}

int grk_amd_stage_ht_encode(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const void* d_mallat)
{
    if (c) { const int jr = join_side(c); if (jr) return jr; }
    if (!c || !p || !d_mallat) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = ensure_geom(c, p); if (rc) return rc;
    return run_ht(c, ntiles, d_mallat);
}

int grk_amd_stage_dwt_inv(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t nplanes, const void* d_mallat, void* d_out)
{
    if (c) { const int jr = join_side(c); if (jr) return jr; }
    if (!c || !p || !d_mallat || !d_out) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = ensure_geom(c, p); if (rc) return rc;
    return run_idwt(c, nplanes, d_mallat, d_out);
}

int grk_amd_stage_ht_decode(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles,
                            const grk_amd_coded_block* table, const void* d_coded, uint64_t coded_bytes, void* d_mallat)
{
    if (c) { const int jr = join_side(c); if (jr) return jr; }
    if (!c || !p || !table || !d_coded || !d_mallat || ntiles == 0) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = ensure_geom(c, p); if (rc) return rc;
    grk_amd_ctx::DecUpload* up = nullptr;
    rc = stage_table(c, table, (uint64_t)c->geom.blocks_per_comp * c->geom.p.num_comps * ntiles, &up); if (rc) return rc;
    rc = p->reserved[0] ? run_t1_decode(c, ntiles, up, d_coded, coded_bytes, d_mallat)
                        : run_ht_decode(c, ntiles, up, d_coded, coded_bytes, d_mallat);
    if (rc) return rc;
    return check_decode_status(c);
}

static int decode_impl(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles,
                       const grk_amd_coded_block* table, const void* coded, uint64_t coded_bytes, int coded_on_device,
                       void* pixels, int pixels_on_device, const Rect* win, bool force32 = false)
{
    if (!c || !p || !table || !coded || !pixels || ntiles == 0) return GRK_AMD_ERR_INVALID;
    const grk_amd_coded_block* table_in = table;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = join_side(c); if (rc) return rc;        // (the Mallat planes and the status word are shared with the encoder)
    // (the block decoders of the top resolution run on the side stream beside the rest: the two have to be dispatched side by side)
    if (c->overlap && c->side && c->seq_index < 0 && c->stream_probe && c->probed_main != c->stream && probe_streams(c) != GRK_AMD_OK) {
        c->stream_probe = 0; (void)hipGetLastError();
    }
    rc = ensure_geom(c, p); if (rc) return rc;
    const TileGeom& g = c->geom;
    const uint32_t nplanes = ntiles * g.p.num_comps;
    const uint32_t bps = (g.p.prec + 7u) / 8u;
    const bool fuse_out = g.p.num_levels >= 1 && bps <= 2 && c->fuse_egress;
    // region decode: the blocks no sample of the window depends on are not decoded, the synthesis covers what is needed
    RegionPlan plan;
    grk_amd_ctx::DecUpload* up = nullptr;
    rc = stage_table(c, table, (uint64_t)g.blocks_per_comp * g.p.num_comps * ntiles, &up); if (rc) return rc;
    if (win) {
        if (ntiles != 1 || win->x0 >= win->x1 || win->y0 >= win->y1 || win->x1 > g.p.tile_w || win->y1 > g.p.tile_h)
            return fail(c, GRK_AMD_ERR_INVALID, "window outside the tile");
        if (!fuse_out) return fail(c, GRK_AMD_ERR_UNSUPPORTED, "region decode needs at least one DWT level and 8-/16-bit pixels");
        plan = plan_region(g, *win);
        const uint32_t L = g.p.num_levels;
        grk_amd_coded_block* const wtable = (grk_amd_coded_block*)up->p;
        size_t i = 0;
        auto sat = [](uint32_t a, uint32_t b) { return a > b ? a - b : 0u; };
        for (uint32_t k = 0; k < g.p.num_comps; ++k)
            for (const auto& b : g.blocks_comp0) {
                // the block in its band's own index space against what the synthesis reads of that band: low-pass indices
                // are pair - parity, high-pass indices the pair itself
                const BandGeom& B = g.res[b.res].band[b.res ? b.band - 1 : 0];
                Rect need;
                if (b.res == 0) need = plan.need[L];
                else {
                    const uint32_t l = L - b.res;
                    const Rect& q = plan.pairs[l];
                    const uint32_t px = plan.px[l], py = plan.py[l];
                    need.x0 = (b.band & 1) ? q.x0 : sat(q.x0, px); need.x1 = (b.band & 1) ? q.x1 : sat(q.x1, px);
                    need.y0 = (b.band & 2) ? q.y0 : sat(q.y0, py); need.y1 = (b.band & 2) ? q.y1 : sat(q.y1, py);
                }
                const uint32_t bx0 = b.x0 - B.x0, bx1 = b.x1 - B.x0, by0 = b.y0 - B.y0, by1 = b.y1 - B.y0;
                if (bx0 >= need.x1 || bx1 <= need.x0 || by0 >= need.y1 || by1 <= need.y0) {
                    wtable[i].offset = 0; wtable[i].length = 0; wtable[i].missing_msbs = kSkipBlock;
                }
                ++i;
            }
    }
    const void* d_coded = coded;
    if (!coded_on_device) {
        HIP_TRY(c, c->dec_coded.ensure(coded_bytes + 64), "alloc coded staging");
        rc = copy_h2d(c, c->dec_coded.p, coded, coded_bytes); if (rc) return rc;
        d_coded = c->dec_coded.p;
    }
    const size_t px_bytes = win ? (size_t)g.p.num_comps * (win->x1 - win->x0) * (win->y1 - win->y0) * bps
                                : (size_t)nplanes * g.p.tile_w * g.p.tile_h * bps;
    void* d_px = pixels;
    if (!pixels_on_device) {
        HIP_TRY(c, c->dec_pixels.ensure(px_bytes), "alloc pixel staging");
        d_px = c->dec_pixels.p;
    }
    if (!fuse_out) HIP_TRY(c, c->p0.ensure((size_t)nplanes * g.plane_elems * 4 + 256), "alloc planes");
    HIP_TRY(c, c->p1.ensure((size_t)nplanes * g.plane_elems * 4 + 256), "alloc Mallat planes");
    // 8-bit reversible HT tiles: int16 planes between K5b and K6 (both HBM-side halves of the decode move half the bytes).
    // Every coefficient and every synthesised LL sample of a stream that an 8-bit image produced fits (the encoder's
    // planes16_ok bound); a stream whose values do not is reported by decode_status (GRK_AMD_ERR_RANGE), and a synchronous
    // call decodes it again with int32 planes right here -- never other pixels.
    const bool h16 = c->dec_planes16 && !force32 && fuse_out && !p->reserved[0] && !g.p.irreversible && g.p.prec <= 8 &&
                     c->dec_seg_first.empty();
    {
        ScopedTimer t(c, 3);
        rc = p->reserved[0] ? run_t1_decode(c, ntiles, up, d_coded, coded_bytes, c->p1.p)
                            : run_ht_decode(c, ntiles, up, d_coded, coded_bytes, c->p1.p, h16, true);
        if (rc) return rc;
        // with at least one DWT level and 8-/16-bit pixels the last level writes the pixels itself (K7 fused): the
        // int32 image planes (4 bytes per sample written and read back) never exist
        if (fuse_out) {
            rc = run_idwt(c, nplanes, c->p1.p, nullptr, d_px, ntiles, bps, win ? &plan : nullptr, h16); if (rc) return rc;
        } else {
            rc = run_idwt(c, nplanes, c->p1.p, c->p0.p); if (rc) return rc;
            rc = run_egress(c, ntiles, c->p0.p, d_px, bps); if (rc) return rc;
        }
    }
    if (!pixels_on_device) {
        rc = copy_d2h(c, pixels, d_px, px_bytes); if (rc) return rc;
        rc = check_decode_status(c);
        if (rc == GRK_AMD_ERR_RANGE && h16)           // (synchronous call: the exact path, at once)
            return decode_impl(c, p, ntiles, table_in, coded, coded_bytes, coded_on_device, pixels, pixels_on_device, win, true);
        return rc;
    }
    return GRK_AMD_OK;                 // (a window's adapted table lives in the context's pinned memory: nothing to wait for)
}

namespace {
// The two streams of one of a sequence's contexts, made for the kind of frames the sequence carries.  The HIP runtime keeps a pool
// of (by default 4) hardware queues PER PRIORITY LEVEL, and kernels of streams that share a queue run one after the other.
// Part-1 frames are two long kernels each (the long chains on the call's stream, the lane kernel on the side stream, ~13 ms
// both): their streams go over the priority levels in turn, so that n frames in flight use the queues of every pool -- three
// frames in flight 10.4 -> 7.4 ms per frame, six 6.8, eight 6.4, without GPU_MAX_HW_QUEUES (profiles/r04_hw_queues.txt).  The HT
// decoder's kernels are short and lose with streams of mixed priority (0.74 -> 0.86 ms per frame): plain streams for those.
// A sequence that changes its kind of frames pays one synchronisation per context.
int sequence_streams(grk_amd_ctx* k, bool part1)
{
    const int flavour = part1 ? 1 : 0;
    if (k->seq_index < 0 || k->seq_flavour == flavour || !k->own_stream) return GRK_AMD_OK;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int levels = least - greatest + 1;
    const char* const epr = getenv("GRK_AMD_SEQ_PRIORITIES");
    if (levels < 2 || (epr && atoi(epr) == 0)) { k->seq_flavour = flavour; return GRK_AMD_OK; }
    const int rc = grk_amd_synchronize(k); if (rc) return rc;
    hipStream_t ns = nullptr, nside = nullptr;
    const int p0 = part1 ? greatest + k->seq_index % levels : least, p1 = part1 ? greatest + (k->seq_index + 1) % levels : least;
    hipError_t e = part1 ? hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, p0) : hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
    if (e == hipSuccess && k->side) e = hipStreamCreateWithPriority(&nside, hipStreamNonBlocking, p1);
    if (e != hipSuccess) { if (ns) (void)hipStreamDestroy(ns); return fail(k, GRK_AMD_ERR_NO_DEVICE, "streams of a decode sequence", e); }
    (void)hipStreamDestroy(k->stream); k->stream = ns;
    if (k->side) { (void)hipStreamDestroy(k->side); k->side = nside; }
    k->seq_flavour = flavour; k->seq_vetted = false;
    return GRK_AMD_OK;
}
} // namespace

int grk_amd_decode_tiles(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles,
                         const grk_amd_coded_block* table, const void* coded, uint64_t coded_bytes, int coded_on_device,
                         void* pixels, int pixels_on_device)
{
    if (c && !c->dec_kids.empty() && coded_on_device && pixels_on_device) {
        {
            // (every frame of the sequence on one of the internal contexts, none on this one: the event below must stand for what the
            //  CALLER queued on this context's stream, not for an earlier frame of the sequence)
            grk_amd_ctx* k = c->dec_kids[c->dec_seq++ % (uint32_t)c->dec_kids.size()];
            // what the caller set on the context applies to the frame wherever it is decoded
            if (k->dec_qcd != c->dec_qcd || k->dec_steps != c->dec_steps) { k->dec_qcd = c->dec_qcd; k->dec_steps = c->dec_steps; k->have_geom = false; }
            if (k->dec_seg_first != c->dec_seg_first) k->dec_seg_first = c->dec_seg_first;
            if (k->dec_segs.size() != c->dec_segs.size() ||
                (!c->dec_segs.empty() && std::memcmp(k->dec_segs.data(), c->dec_segs.data(), c->dec_segs.size() * sizeof(c->dec_segs[0])) != 0))
                k->dec_segs = c->dec_segs;
            k->dec_planes16 = c->dec_planes16; k->fuse_egress = c->fuse_egress; k->dwt_pk = c->dwt_pk; k->dwt_xcd = c->dwt_xcd;
            k->overlap = c->overlap && k->side != nullptr; k->t1_lanes = c->t1_lanes; k->t1_tail_ratio = c->t1_tail_ratio;
            HIP_TRY(c, hipSetDevice(c->device), "set device");
            { const int sr = sequence_streams(k, p && p->reserved[0] != 0); if (sr) { c->err = k->err; return sr; } }
            // The contexts' streams, vetted in the contexts' order: a context's two streams against each other and against the (up to
            // three) streams accepted just before -- four dispatch pipes: two frames in flight can have a pipe per stream (HT frames:
            // 0.66 instead of 0.75-0.81 ms per frame when the runtime's choice collides, tools/hwq_alias_dec.py), more cannot
            if (c->stream_probe && !k->seq_vetted) {
                int vr = grk_amd_synchronize(k);
                const int nk = (int)c->dec_kids.size();
                for (int which = 0; which < 2 && vr == GRK_AMD_OK; ++which) {
                    hipStream_t* st = which ? &k->side : &k->stream;
                    if (!*st) continue;
                    // (the streams as they are NOW: a context that changed its kind of frames has re-made its own)
                    std::vector<hipStream_t> against;
                    if (which) against.push_back(k->stream);
                    for (int back = 1; back < nk && against.size() < 3; ++back) {
                        grk_amd_ctx* o = c->dec_kids[(size_t)((k->seq_index - back + nk) % nk)];
                        if (!o->seq_vetted) continue;
                        if (o->side && against.size() < 3) against.push_back(o->side);
                        if (against.size() < 3) against.push_back(o->stream);
                    }
                    vr = vetted_stream(k, st, against, &c->probe_replaced);
                }
                if (vr) { c->stream_probe = 0; (void)hipGetLastError(); }
                k->seq_vetted = true;
            }
            // ... behind whatever the caller queued on this context's stream (its uploads of the coded bytes)
            HIP_TRY(c, hipEventRecord(c->ev_seq, c->stream), "record the caller's stream");
            HIP_TRY(c, hipStreamWaitEvent(k->stream, c->ev_seq, 0), "order the frame behind the caller's stream");
            const int rc = decode_impl(k, p, ntiles, table, coded, coded_bytes, 1, pixels, 1, nullptr);
            // (the frame's last kernels -- the final inverse level, behind its join with the side stream -- are on k's stream; a call
            //  that failed half-way may have queued kernels that still read the coded bytes or write the pixels: the set's event covers
            //  those too, its side stream joined first)
            if (!k->ev_frame_done) HIP_TRY(c, hipEventCreateWithFlags(&k->ev_frame_done, hipEventDisableTiming), "create event");
            if (rc && k->side) {
                if (!k->ev_dec_top) HIP_TRY(c, hipEventCreateWithFlags(&k->ev_dec_top, hipEventDisableTiming), "create event");
                HIP_TRY(c, hipEventRecord(k->ev_dec_top, k->side), "record the side stream");
                HIP_TRY(c, hipStreamWaitEvent(k->stream, k->ev_dec_top, 0), "join the side stream");
                k->dec_top_pending = false;
            }
            HIP_TRY(c, hipEventRecord(k->ev_frame_done, k->stream), "record the frame's end");
            if (rc) c->err = k->err;
            return rc;
        }
    }
    return decode_impl(c, p, ntiles, table, coded, coded_bytes, coded_on_device, pixels, pixels_on_device, nullptr);
}

int grk_amd_decode_stream_wait_slot(grk_amd_ctx* c, void* hip_stream)
{
    if (!c || !hip_stream) return GRK_AMD_ERR_INVALID;
    if (c->dec_kids.empty()) return grk_amd_stream_wait_results(c, hip_stream);       // no sequence: the context's own streams
    grk_amd_ctx* k = c->dec_kids[c->dec_seq % (uint32_t)c->dec_kids.size()];          // the set the NEXT call uses
    if (k->ev_frame_done) HIP_TRY(c, hipStreamWaitEvent((hipStream_t)hip_stream, k->ev_frame_done, 0), "wait for the set's last frame");
    return GRK_AMD_OK;
}

int grk_amd_set_decode_pipelining(grk_amd_ctx* c, int frames_in_flight)
{
    if (!c || frames_in_flight < 0 || frames_in_flight > 8) return GRK_AMD_ERR_INVALID;
    int rc = grk_amd_synchronize(c);
    for (grk_amd_ctx* k : c->dec_kids) grk_amd_destroy(k);
    c->dec_kids.clear();
    c->dec_seq = 0;
    if (rc) return rc;
    if (frames_in_flight >= 2 && !c->ev_seq) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_seq, hipEventDisableTiming), "create event");
    for (int i = 0; i < frames_in_flight && frames_in_flight >= 2; ++i) {
        grk_amd_ctx* k = nullptr;
        rc = create_context(c->device, c->verbose, true, &k);
        if (rc) return fail(c, rc, "a further decode context could not be made");
        k->seq_index = i;
        c->dec_kids.push_back(k);
    }
    return GRK_AMD_OK;
}

int grk_amd_decode_region(grk_amd_ctx* c, const grk_amd_tile_params* p,
                          const grk_amd_coded_block* table, const void* coded, uint64_t coded_bytes, int coded_on_device,
                          uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, void* pixels, int pixels_on_device)
{
    const Rect win{x0, y0, x1, y1};
    return decode_impl(c, p, 1, table, coded, coded_bytes, coded_on_device, pixels, pixels_on_device, &win);
}

int grk_amd_set_decode_qcd(grk_amd_ctx* c, const uint16_t* words, uint32_t count)
{
    if (!c || (count && !words)) return GRK_AMD_ERR_INVALID;
    c->dec_qcd.assign(words, words + count);
    c->have_geom = false;                  // the per-block dequantisation scales are rebuilt on the next call
    return GRK_AMD_OK;
}

int grk_amd_set_decode_steps(grk_amd_ctx* c, const float* steps, uint32_t count)
{
    if (!c || (count && !steps)) return GRK_AMD_ERR_INVALID;
    c->dec_steps.assign(steps, steps + count);
    c->have_geom = false;                  // the per-block dequantisation scales are rebuilt on the next call
    return GRK_AMD_OK;
}

int grk_amd_set_decode_segments(grk_amd_ctx* c, const uint32_t* first_segment, const grk_amd_segment* segments, uint32_t nblocks)
{
    if (!c || (nblocks && (!first_segment || (first_segment[nblocks] && !segments)))) return GRK_AMD_ERR_INVALID;
    c->dec_seg_first.clear(); c->dec_segs.clear();
    if (nblocks) {
        c->dec_seg_first.assign(first_segment, first_segment + nblocks + 1);
        c->dec_segs.assign(segments, segments + first_segment[nblocks]);
    }
    return GRK_AMD_OK;
}

int grk_amd_decode_status(grk_amd_ctx* c)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = check_decode_status(c);
    for (grk_amd_ctx* k : c->dec_kids) {            // (a sequence: the frames decoded on the other contexts as well)
        const int kr = check_decode_status(k);
        if (kr && !rc) { rc = kr; c->err = k->err; }
    }
    return rc;
}

int grk_amd_stage_egress(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const void* d_planes, void* d_pixels)
{
    if (!c || !p || !d_planes || !d_pixels) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = ensure_geom(c, p); if (rc) return rc;
    return run_egress(c, ntiles, d_planes, d_pixels, (p->prec + 7u) / 8u);
}

int grk_amd_fetch_table(grk_amd_ctx* c, grk_amd_coded_block* table, uint64_t* total)
{
    if (!c || !c->last_nblocks) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    { const int jr = join_side(c); if (jr) return jr; }
    const uint64_t n = c->last_nblocks;
    uint64_t flagwords[2] = {0, 0};       // [0] low 32 bits: overflow flag, [1]: arena cursor
    HIP_TRY(c, hipMemcpyAsync(flagwords, c->flag.p, 16, hipMemcpyDeviceToHost, c->stream), "fetch flag");
    if (table) {
        c->h_off.resize(n); c->h_len.resize(n);
        HIP_TRY(c, hipMemcpyAsync(c->h_off.data(), c->offsets.p, n * 8, hipMemcpyDeviceToHost, c->stream), "fetch offsets");
        HIP_TRY(c, hipMemcpyAsync(c->h_len.data(), c->lengths.p, n * 4, hipMemcpyDeviceToHost, c->stream), "fetch lengths");
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    if (flagwords[0] & 1u) return fail(c, GRK_AMD_ERR_OVERFLOW, "coded arena overflow");
    if (flagwords[0] & 2u) return fail(c, GRK_AMD_ERR_UNSUPPORTED, "coefficient magnitude exceeds Kmax+1 bits");
    if (table) {
        const uint32_t bpt = (uint32_t)c->h_desc.size();
        for (uint64_t i = 0; i < n; ++i) {
            table[i].offset = c->h_off[i]; table[i].length = c->h_len[i];
            table[i].missing_msbs = c->h_desc[i % bpt].kmax - 1u;      // numbps = 1 is signalled (T1HT.cpp:123)
        }
    }
    if (total) *total = flagwords[1];
    return GRK_AMD_OK;
}

int grk_amd_fetch_coded(grk_amd_ctx* c, uint8_t* dst, uint64_t nbytes)
{
    if (!c || !dst) return GRK_AMD_ERR_INVALID;
    if (nbytes > c->arena.cap) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    { const int jr = join_side(c); if (jr) return jr; }
    { const int rc = copy_d2h(c, dst, c->arena.p, nbytes); if (rc) return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    return GRK_AMD_OK;
}

int grk_amd_fetch_coded_async(grk_amd_ctx* c, uint8_t* dst, uint64_t nbytes)
{
    if (!c || !dst) return GRK_AMD_ERR_INVALID;
    if (nbytes > c->arena.cap) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    if (!host_is_pinned(dst)) return fail(c, GRK_AMD_ERR_INVALID, "grk_amd_fetch_coded_async needs pinned memory (grk_amd_host_alloc)");
    { const int jr = join_side(c); if (jr) return jr; }
    if (nbytes) HIP_TRY(c, hipMemcpyAsync(dst, c->arena.p, nbytes, hipMemcpyDeviceToHost, c->stream), "download");
    return GRK_AMD_OK;
}

int grk_amd_fetch_coefficients(grk_amd_ctx* c, uint32_t comp, int32_t* dst, uint32_t dst_stride)
{
    if (!c || !dst || !c->have_geom || !c->last_nblocks || comp >= c->geom.p.num_comps || dst_stride < c->geom.p.tile_w)
        return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    { const int jr = join_side(c); if (jr) return jr; }
    const TileGeom& g = c->geom;
    const uint32_t W = g.p.tile_w, H = g.p.tile_h;
    if (c->last_h16) {              // 16-bit planes between K2 and K3 (8-bit reversible content): widened here
        std::vector<int16_t> tmp((size_t)g.stride * H);
        HIP_TRY(c, hipMemcpyAsync(tmp.data(), (const int16_t*)c->p1.p + (size_t)comp * g.plane_elems, tmp.size() * 2,
                                  hipMemcpyDeviceToHost, c->stream), "fetch coefficients");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) dst[(size_t)y * dst_stride + x] = tmp[(size_t)y * g.stride + x];
    } else {
        HIP_TRY(c, hipMemcpy2DAsync(dst, (size_t)dst_stride * 4, (const int32_t*)c->p1.p + (size_t)comp * g.plane_elems,
                                    (size_t)g.stride * 4, (size_t)W * 4, H, hipMemcpyDeviceToHost, c->stream), "fetch coefficients");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    }
    return GRK_AMD_OK;
}

// Weights of T1::getwmsedec (t1/t1_part1/T1.cpp:394-414): L2 norms of the synthesis basis functions by orientation and decomposition
// level (dwt_norms / dwt_norms_real, T1.cpp:224-235; T1::getnorm clamps the level, :258-267) and of the inverse colour transform's
// columns (mct_norms_rev / _irrev, point_transform/mct.cpp:30-35)
static double band_norm(uint32_t orient, uint32_t level, bool reversible)
{
    static const double n53[4][10] = {{1.000, 1.500, 2.750, 5.375, 10.68, 21.34, 42.67, 85.33, 170.7, 341.3},
                                      {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
                                      {1.038, 1.592, 2.919, 5.703, 11.33, 22.64, 45.25, 90.48, 180.9, 0},
                                      {.7186, .9218, 1.586, 3.043, 6.019, 12.01, 24.00, 47.97, 95.93, 0}};
    static const double n97[4][10] = {{1.000, 1.965, 4.177, 8.403, 16.90, 33.84, 67.69, 135.3, 270.6, 540.9},
                                      {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
                                      {2.022, 3.989, 8.355, 17.04, 34.27, 68.63, 137.3, 274.6, 549.0, 0},
                                      {2.080, 3.865, 8.307, 17.18, 34.71, 69.59, 139.3, 278.6, 557.2, 0}};
    if (orient == 0 && level > 9) level = 9;
    else if (orient > 0 && level > 8) level = 8;
    return reversible ? n53[orient & 3u][level] : n97[orient & 3u][level];
}

int grk_amd_block_distortion(grk_amd_ctx* c, double* out, uint64_t cap)
{
    if (!c || !out || !c->have_geom || !c->last_nblocks || cap < c->last_nblocks) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    { const int jr = join_side(c); if (jr) return jr; }
    const TileGeom& g = c->geom;
    const uint64_t n = c->last_nblocks;
    const uint32_t bpt = (uint32_t)c->h_desc.size();
    HIP_TRY(c, c->energy.ensure(n * 8), "alloc block energies");
    HIP_TRY(c, launch_block_energy(c->p1.p, c->last_h16 ? 1 : 0, g.p.irreversible, g.stride, g.plane_elems, (const HtBlockDesc*)c->blockdesc.p,
                                   bpt, g.p.num_comps, n, (unsigned long long*)c->energy.p, c->stream), "launch block energy");
    std::vector<unsigned long long> e(n);
    HIP_TRY(c, hipMemcpyAsync(e.data(), c->energy.p, n * 8, hipMemcpyDeviceToHost, c->stream), "fetch block energies");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    static const double mct_rev[3] = {1.732, .8292, .8292}, mct_irrev[3] = {1.732, 1.805, 1.573};
    for (uint64_t i = 0; i < n; ++i) {
        const grk_amd_block& b = g.blocks_comp0[(i % bpt) % g.blocks_per_comp];
        const uint32_t comp = (uint32_t)((i % bpt) / g.blocks_per_comp);
        const double w1 = (g.p.mct && g.p.num_comps >= 3 && comp < 3) ? (g.p.irreversible ? mct_irrev[comp] : mct_rev[comp]) : 1.0;
        const double w2 = band_norm(b.band, g.p.num_levels - b.res, !g.p.irreversible);
        const double w = w1 * w2 * (double)b.stepsize;
        out[i] = w * w * (double)e[i];
    }
    return GRK_AMD_OK;
}

// ---- Tier-2 on the device ----------------------------------------------------------------------------------------------------
// The finished tile-parts of the LATEST grk_amd_encode_tiles call, made where the coded bytes are (kernels_t2.hip): KT1 writes every
// packet's header, KT1b frames the tile-parts (SOT, PLT, SOD) and says where every packet goes, KT2 gathers headers, code-block bytes
// and frames into the output.  Nothing has to come to the host in between.
namespace {
// the kernels queued on `st`, which must already be ordered behind the encode's results; `o` takes the tile-parts from dst_offset on
// (what lies below is kept when the buffer has to grow).  The scratch is the context's: one stream at a time.
int assemble_enqueue(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const uint32_t* tile_index, uint32_t flags,
                     uint64_t dst_offset, hipStream_t st, grk_amd_ctx::T2Out& o)
{
    const TileGeom& g = c->geom;
    const uint32_t order = (flags >> GRK_AMD_CS_PROG_SHIFT) & 7u;
    auto& T = c->t2;
    if (!T.valid || !same_params(T.p, *p) || T.order != order) {
        T.valid = false;
        const int rc = t2_device_plan(g, flags, T.plan);
        if (rc) return fail(c, rc, "tile layout beyond the device writer's tables");
        T.max_blocks = 0;
        for (const T2Packet& k : T.plan.packets) T.max_blocks = std::max(T.max_blocks, k.nblocks);
        // (the tables of the plan before may still be read by a gather that is queued: drained first)
        HIP_TRY(c, hipStreamSynchronize(st), "sync");
        HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
        HIP_TRY(c, T.packets.ensure(T.plan.packets.size() * sizeof(T2Packet)), "alloc packet table");
        HIP_TRY(c, T.pob.ensure(T.plan.packet_of_block.size() * 4), "alloc packet-of-block table");
        HIP_TRY(c, hipMemcpyAsync(T.packets.p, T.plan.packets.data(), T.plan.packets.size() * sizeof(T2Packet), hipMemcpyHostToDevice, st), "upload");
        HIP_TRY(c, hipMemcpyAsync(T.pob.p, T.plan.packet_of_block.data(), T.plan.packet_of_block.size() * 4, hipMemcpyHostToDevice, st), "upload");
        HIP_TRY(c, hipStreamSynchronize(st), "sync");
        T.p = *p; T.order = order; T.valid = true;
    }
    const size_t npk = T.plan.packets.size();
    const uint32_t bpt = (uint32_t)(c->last_nblocks / ntiles);
    const size_t nq = npk * ntiles;
    // a frame: SOT 12, SOD 2, PLT: at most 6 bytes per packet and 5 per marker segment of 65 532
    const uint32_t lit_stride = (uint32_t)((14 + 6 * npk + 5 * (6 * npk / 65000 + 2) + 15) & ~(size_t)15);
    // what the call can write at most: every byte of the arena, every header, every frame
    const uint64_t bound = (uint64_t)c->arena.cap + (uint64_t)ntiles * (T.plan.h_bytes + lit_stride + 8ull * npk);
    auto need = [&](DevBuf& b, size_t n, const char* what) -> int {
        if (n <= b.cap) return GRK_AMD_OK;
        // (a scratch buffer about to be replaced may be in use by kernels queued earlier on the stream)
        HIP_TRY(c, hipStreamSynchronize(st), "sync");
        HIP_TRY(c, b.ensure(n), what);
        return GRK_AMD_OK;
    };
    int rc;
    if ((rc = need(c->t2_u, (size_t)T.plan.u_words * 4 * ntiles + 16, "alloc header bits"))) return rc;
    if ((rc = need(c->t2_h, (size_t)T.plan.h_bytes * ntiles + 16, "alloc headers"))) return rc;
    if ((rc = need(c->t2_rel, c->last_nblocks * 4, "alloc block places"))) return rc;
    if ((rc = need(c->t2_pkhdr, nq * 4, "alloc packet lengths"))) return rc;
    if ((rc = need(c->t2_pkbody, nq * 8, "alloc packet lengths"))) return rc;
    if ((rc = need(c->t2_pkdst, nq * 8, "alloc packet places"))) return rc;
    if ((rc = need(c->t2_lit, (size_t)lit_stride * ntiles, "alloc frames"))) return rc;
    if ((rc = need(c->t2_litlen, (size_t)ntiles * 4, "alloc frames"))) return rc;
    if ((rc = need(c->t2_index, (size_t)ntiles * 4, "alloc tile numbers"))) return rc;
    if ((rc = need(o.tile_dst, (size_t)ntiles * 8, "alloc tile-part places"))) return rc;
    if ((rc = need(o.part_len, (size_t)ntiles * 4, "alloc tile-part lengths"))) return rc;
    if ((rc = need(o.total, 16, "alloc tile-part total"))) return rc;
    if (o.out.cap < dst_offset + bound) {
        DevBuf bigger;
        HIP_TRY(c, hipStreamSynchronize(st), "sync");
        HIP_TRY(c, bigger.ensure(dst_offset + bound), "alloc tile-parts");
        if (dst_offset) HIP_TRY(c, hipMemcpyAsync(bigger.p, o.out.p, dst_offset, hipMemcpyDeviceToDevice, st), "keep tile-parts");
        HIP_TRY(c, hipStreamSynchronize(st), "sync");
        o.out.release();
        o.out = bigger;
    }
    HIP_TRY(c, hipMemsetAsync(c->t2_u.p, 0, (size_t)T.plan.u_words * 4 * ntiles, st), "clear header bits");
    // (the tile numbers: pageable memory of the caller's -- the runtime has staged them when the call returns)
    HIP_TRY(c, hipMemcpyAsync(c->t2_index.p, tile_index, (size_t)ntiles * 4, hipMemcpyHostToDevice, st), "upload");
    T2HeaderArgs ha{};
    ha.packets = (const T2Packet*)T.packets.p; ha.npackets = (uint32_t)npk;
    ha.lengths = (const uint32_t*)c->lengths.p; ha.bpt = bpt; ha.ntiles = ntiles;
    ha.ubits = (uint32_t*)c->t2_u.p; ha.u_words = T.plan.u_words;
    ha.hdr = (uint8_t*)c->t2_h.p; ha.h_bytes = T.plan.h_bytes;
    ha.rel = (uint32_t*)c->t2_rel.p; ha.pk_hdr = (uint32_t*)c->t2_pkhdr.p; ha.pk_body = (uint64_t*)c->t2_pkbody.p;
    ha.status = (unsigned int*)c->flag.p;
    HIP_TRY(c, launch_t2_header(ha, T.max_blocks, st), "launch Tier-2 headers");
    const uint32_t sop = (flags & GRK_AMD_CS_SOP) ? 6u : 0u, eph = (flags & GRK_AMD_CS_EPH) ? 2u : 0u;
    T2FrameArgs fa{};
    fa.npackets = (uint32_t)npk; fa.ntiles = ntiles; fa.pk_hdr = ha.pk_hdr; fa.pk_body = ha.pk_body;
    fa.tile_index = (const uint32_t*)c->t2_index.p; fa.extra = sop + eph; fa.plt = (flags & GRK_AMD_CS_PLT) ? 1u : 0u;
    fa.dst_offset = dst_offset;
    fa.lit = (uint8_t*)c->t2_lit.p; fa.lit_stride = lit_stride; fa.lit_len = (uint32_t*)c->t2_litlen.p;
    fa.pk_dst = (uint64_t*)c->t2_pkdst.p;
    fa.part_len = (uint32_t*)o.part_len.p; fa.tile_dst = (unsigned long long*)o.tile_dst.p; fa.total = (unsigned long long*)o.total.p;
    fa.status = (unsigned int*)c->flag.p;
    HIP_TRY(c, launch_t2_frame(fa, st), "launch Tier-2 frames");
    T2GatherArgs ga{};
    ga.packets = (const T2Packet*)T.packets.p; ga.npackets = (uint32_t)npk; ga.packet_of_block = (const uint32_t*)T.pob.p;
    ga.lengths = (const uint32_t*)c->lengths.p; ga.offsets = (const uint64_t*)c->offsets.p; ga.arena = (const uint8_t*)c->arena.p;
    ga.bpt = bpt; ga.ntiles = ntiles;
    ga.hdr = (const uint8_t*)c->t2_h.p; ga.h_bytes = T.plan.h_bytes;
    ga.rel = (const uint32_t*)c->t2_rel.p; ga.pk_hdr = (const uint32_t*)c->t2_pkhdr.p; ga.pk_dst = (const uint64_t*)c->t2_pkdst.p;
    ga.lit = (const uint8_t*)c->t2_lit.p; ga.lit_stride = lit_stride; ga.lit_len = (const uint32_t*)c->t2_litlen.p;
    ga.tile_dst = (const unsigned long long*)o.tile_dst.p;
    ga.out = (uint8_t*)o.out.p;
    ga.sop = sop; ga.eph = eph;
    HIP_TRY(c, launch_t2_gather(ga, st), "launch Tier-2 gather");
    return GRK_AMD_OK;
}

int assemble_check(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const uint32_t* tile_index)
{
    if (!c || !p || !ntiles || !tile_index) return GRK_AMD_ERR_INVALID;
    if (!c->have_geom || !same_params(c->gp, *p) || ntiles != c->last_ntiles || !c->last_nblocks)
        return fail(c, GRK_AMD_ERR_INVALID, "grk_amd_assemble_device assembles the grk_amd_encode_tiles call before it: same tiles, same parameters");
    return GRK_AMD_OK;
}
} // namespace

int64_t grk_amd_assemble_device(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const uint32_t* tile_index, uint32_t flags,
                                uint64_t dst_offset, uint32_t* part_bytes)
{
    { const int rc = assemble_check(c, p, ntiles, tile_index); if (rc) return rc; }
    if (dst_offset > c->t2_out_used) return fail(c, GRK_AMD_ERR_INVALID, "grk_amd_assemble_device: dst_offset lies behind what has been assembled");
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    { const int jr = join_side(c); if (jr) return jr; }
    auto& o = c->t2_outs[c->t2_cur];
    { const int rc = assemble_enqueue(c, p, ntiles, tile_index, flags, dst_offset, c->stream, o); if (rc) return rc; }
    uint64_t flagwords[2] = {0, 0}, total[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(flagwords, c->flag.p, 16, hipMemcpyDeviceToHost, c->stream), "fetch flag");
    HIP_TRY(c, hipMemcpyAsync(total, o.total.p, 16, hipMemcpyDeviceToHost, c->stream), "fetch total");
    if (part_bytes) HIP_TRY(c, hipMemcpyAsync(part_bytes, o.part_len.p, (size_t)ntiles * 4, hipMemcpyDeviceToHost, c->stream), "fetch tile-part lengths");
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    if (flagwords[0] & 1u) return fail(c, GRK_AMD_ERR_OVERFLOW, "coded arena overflow");
    if (flagwords[0] & 2u) return fail(c, GRK_AMD_ERR_UNSUPPORTED, "coefficient magnitude exceeds Kmax+1 bits");
    if (flagwords[0] & 4u) return fail(c, GRK_AMD_ERR_UNSUPPORTED, "a code-block longer than the device writer takes");
    if (flagwords[0] & 8u) return fail(c, GRK_AMD_ERR_UNSUPPORTED, "tile-part beyond 4 GB or packet lengths beyond what PLT can carry");
    c->t2_out_used = total[1];
    return (int64_t)total[0];
}

// The same without the host: queued on `hip_stream` (made to wait for the encode's results first, as grk_amd_stream_wait_results does),
// nothing is waited for.  Pipelined encodes rotate as many outputs as buffer sets: a frame's tile-parts, their places / lengths and the
// total (grk_amd_assembled_device_ptr, grk_amd_assembled_table_ptr) stay untouched until that many further calls -- time for an
// exchange to send them.  Every asynchronous call of a context has to use the SAME stream (the scratch is shared, stream order keeps
// the calls apart); errors (bits 0-3 of the encode's status word) are the consumer's to find: the bytes will not parse.
int grk_amd_assemble_device_async(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const uint32_t* tile_index, uint32_t flags,
                                  void* hip_stream)
{
    { const int rc = assemble_check(c, p, ntiles, tile_index); if (rc) return rc; }
    if (!hip_stream) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    { const int rc = grk_amd_stream_wait_results(c, hip_stream); if (rc) return rc; }
    const int nsets = c->pipelining ? std::max(1, std::min(c->pipe_depth, grk_amd_ctx::kMaxAltSets + 1)) : 1;
    c->t2_cur = (c->t2_cur + 1) % nsets;
    c->t2_out_used = 0;
    return assemble_enqueue(c, p, ntiles, tile_index, flags, 0, (hipStream_t)hip_stream, c->t2_outs[c->t2_cur]);
}

void* grk_amd_assembled_device_ptr(grk_amd_ctx* c) { return c ? c->t2_outs[c->t2_cur].out.p : nullptr; }
// of the latest assemble call -- 0: uint64[tiles] where each tile-part starts, 1: uint32[tiles] its length, 2: uint64[2] {bytes the call
// assembled, end of the output}
void* grk_amd_assembled_table_ptr(grk_amd_ctx* c, int which)
{
    if (!c) return nullptr;
    auto& o = c->t2_outs[c->t2_cur];
    return which == 0 ? o.tile_dst.p : which == 1 ? o.part_len.p : which == 2 ? o.total.p : nullptr;
}

// bytes [offset, offset + nbytes) of the assembled tile-parts to host memory: pinned memory in one DMA, pageable memory through the
// context's pinned chunks on several copy threads (copy_d2h); complete on return
int grk_amd_fetch_assembled(grk_amd_ctx* c, uint64_t offset, uint64_t nbytes, uint8_t* dst)
{
    if (!c || !dst || offset + nbytes > c->t2_out_used) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    if (!nbytes) return GRK_AMD_OK;
    { const int rc = copy_d2h(c, dst, (const uint8_t*)c->t2_outs[c->t2_cur].out.p + offset, nbytes); if (rc) return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    return GRK_AMD_OK;
}

// the same queued on the context's stream, for pinned memory only; complete after grk_amd_synchronize
int grk_amd_fetch_assembled_async(grk_amd_ctx* c, uint64_t offset, uint64_t nbytes, uint8_t* dst)
{
    if (!c || !dst || offset + nbytes > c->t2_out_used) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    if (!host_is_pinned(dst)) return fail(c, GRK_AMD_ERR_INVALID, "grk_amd_fetch_assembled_async needs pinned memory (grk_amd_host_alloc)");
    if (nbytes) HIP_TRY(c, hipMemcpyAsync(dst, (const uint8_t*)c->t2_outs[c->t2_cur].out.p + offset, nbytes, hipMemcpyDeviceToHost, c->stream), "download");
    return GRK_AMD_OK;
}

// The probe for a host's own streams (an exchange's stream that waits for the encoder's results holds its dispatch pipe while it waits:
// it must not share the main stream's): 1 when kernels of `a` and `b` are dispatched side by side, in both directions; 0 when one
// waits for the other's grid.  Both streams are synchronised.
int grk_amd_streams_side_by_side(grk_amd_ctx* c, void* a, void* b)
{
    if (!c || !a || !b || a == b) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)a), "sync");
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)b), "sync");
    { const int wr = probe_warmup(c, (hipStream_t)a); if (wr) return wr; }
    bool y = false;
    int rc = streams_side_by_side(c, (hipStream_t)a, (hipStream_t)b, &y); if (rc) return rc;
    if (y) { rc = streams_side_by_side(c, (hipStream_t)b, (hipStream_t)a, &y); if (rc) return rc; }
    return y ? 1 : 0;
}
// the context's streams as they are now: 0 the main stream (grk_amd_set_stream's, or its own), 1 / 2 the side streams
void* grk_amd_internal_stream(grk_amd_ctx* c, int which) { return !c ? nullptr : which == 0 ? (void*)c->stream : which == 1 ? (void*)c->side : which == 2 ? (void*)c->side2 : nullptr; }
// the context's own probe now (it runs by itself before the first pipelined encode on a main stream)
int grk_amd_probe_streams(grk_amd_ctx* c)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    { const int jr = join_side(c); if (jr) return jr; }
    return probe_streams(c);
}

// side streams the probe has replaced so far (-1: the probe is switched off)
int grk_amd_stream_probe_result(grk_amd_ctx* c) { return !c ? GRK_AMD_ERR_INVALID : c->stream_probe ? c->probe_replaced : -1; }

void* grk_amd_coded_device_ptr(grk_amd_ctx* c) { return c ? c->arena.p : nullptr; }
void* grk_amd_table_device_ptr(grk_amd_ctx* c, int which)
{
    if (!c) return nullptr;
    switch (which) {
    case 0: return c->offsets.p;                                    // uint64[nblocks]
    case 1: return c->lengths.p;                                    // uint32[nblocks]
    case 2: return c->flag.p ? (uint8_t*)c->flag.p + 8 : nullptr;   // uint64: bytes used in the arena
    case 3: return c->flag.p ? (uint8_t*)c->flag.p + 16 : nullptr;  // uint64[24]: blocks each K3 class handed to its fallback launch
    default: return nullptr;
    }
}
void* grk_amd_plane_device_ptr(grk_amd_ctx* c, int which) { return c ? (which ? c->p1.p : c->p0.p) : nullptr; }

int grk_amd_synchronize(grk_amd_ctx* c)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    { const int jr = join_side(c); if (jr) return jr; }
    HIP_TRY(c, hipStreamSynchronize(c->stream), "sync");
    if (c->side) HIP_TRY(c, hipStreamSynchronize(c->side), "sync side stream");        // (a pipelined predecessor)
    if (c->side2) HIP_TRY(c, hipStreamSynchronize(c->side2), "sync side stream 2");
    for (grk_amd_ctx* k : c->dec_kids) { const int kr = grk_amd_synchronize(k); if (kr) { c->err = k->err; return kr; } }
    return GRK_AMD_OK;
}

int grk_amd_encode_tiles(grk_amd_ctx* c, const grk_amd_tile_params* p, uint32_t ntiles, const void* pixels,
                         int on_device, grk_amd_coded_block* table, uint64_t* total)
{
    if (!c || !p || !pixels || ntiles == 0) return GRK_AMD_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device), "set device");
    int rc = ensure_geom(c, p); if (rc) return rc;
    const TileGeom& g = c->geom;
    const size_t tile_bytes = (size_t)g.p.num_comps * g.p.tile_w * g.p.tile_h * ((g.p.prec + 7) / 8);
    const void* d_px = pixels;
    if (!on_device) {
        HIP_TRY(c, c->pixels.ensure(tile_bytes * ntiles), "alloc pixel staging");
        rc = copy_h2d(c, c->pixels.p, pixels, tile_bytes * ntiles); if (rc) return rc;
        d_px = c->pixels.p;
    }
    const uint32_t nplanes = ntiles * g.p.num_comps;
    // with at least one DWT level, level 0 consumes the pixels itself and the int32 ingest planes
    // (4 bytes per sample written and read back) never exist
    const bool fused = g.p.num_levels >= 1 && ((uintptr_t)d_px & 3u) == 0;
    if (!fused) HIP_TRY(c, c->p0.ensure((size_t)nplanes * g.plane_elems * 4 + 256), "alloc planes");
    HIP_TRY(c, c->p1.ensure((size_t)nplanes * g.plane_elems * 4 + 256), "alloc Mallat planes");
    {
        ScopedTimer t(c, 3);
        const bool ov = c->overlap && g.p.num_levels >= 1 && c->side != nullptr;
        if (ov && c->pipelining && c->seq_index < 0 && probe_streams(c) != GRK_AMD_OK) {
            // (the probe is a convenience: when it cannot run, the streams stay as they are and it is not tried again)
            c->stream_probe = 0; (void)hipGetLastError();
        }
        // (device-resident pixels only: the staging buffer of host pixels is filled on the main stream, which must then carry level 0)
        // (... and the fused level 0: the stand-alone ingest writes planes that are not part of a buffer set)
        const bool fs = ov && c->pipelining && c->side2 != nullptr && on_device && fused &&
                        (c->frame_streams == 2 || (c->frame_streams == 1 && (uint64_t)nplanes * g.plane_elems <= grk_amd_ctx::kFrameStreamSamples));
        hipStream_t fs_st = nullptr;
        if (ov && c->pipelining) {
            // take the other buffer set: the blocks of the previous encode may still be being coded from the set used
            // last; the set taken now was last used two encodes ago, and its side-stream work is waited for here
            // (hipStreamWaitEvent on an event never recorded is a no-op)
            auto swap_with = [&](grk_amd_ctx::AltSet& as) {
                std::swap(c->p1, as.p1); std::swap(c->arena, as.arena); std::swap(c->lengths, as.lengths);
                std::swap(c->offsets, as.offsets); std::swap(c->flag, as.flag); std::swap(c->ovf, as.ovf);
                std::swap(c->ev_side, as.ev_side); std::swap(c->ev_side2, as.ev_side2);
            };
            // the oldest of the pipe_depth - 1 other sets becomes current; the set retired here takes its slot as the newest
            swap_with(c->alts[c->alt_head]);
            // (the LL ping-pong buffers belong to the set as well: frames on different streams transform at the same time, and a call
            //  of the other form -- the same geometry, more tiles -- must not take a running frame's)
            std::swap(c->llA, c->alts[c->alt_head].llA); std::swap(c->llB, c->alts[c->alt_head].llB);
            c->alt_head = (c->alt_head + 1) % (c->pipe_depth - 1);
            c->side_pending = false;
            if (fs) {
                fs_st = c->fs_parity ? c->side2 : c->side; c->fs_parity ^= 1;
                if (!c->ev_main) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming), "create event");
                HIP_TRY(c, hipEventRecord(c->ev_main, c->stream), "record the caller's stream");
                HIP_TRY(c, hipStreamWaitEvent(fs_st, c->ev_main, 0), "the frame's stream waits for the pixels");
            }
            HIP_TRY(c, hipStreamWaitEvent(fs ? fs_st : c->stream, c->ev_side, 0), "wait for the buffer set");
            HIP_TRY(c, hipStreamWaitEvent(fs ? fs_st : c->stream, c->ev_side2, 0), "wait for the buffer set");
            HIP_TRY(c, c->p1.ensure((size_t)nplanes * g.plane_elems * 4 + 256), "alloc Mallat planes");
        } else {
            rc = join_side(c); if (rc) return rc;
        }
        // 8-bit reversible content: int16 LL / Mallat planes between K2 and K3 (half the bytes written and read back);
        // needs the fused level 0 (the stand-alone ingest kernel writes int32 planes)
        const bool h16 = c->planes16 && fused && planes16_ok(g.p);
        c->last_h16 = h16;
        if (fs) {
            t.cancel();
            // the whole frame on its stream, as the non-overlapped path lays it out (one K3 launch of every block, the ROOM instance)
            struct StreamSwap { grk_amd_ctx* c; hipStream_t keep; StreamSwap(grk_amd_ctx* c_, hipStream_t s) : c(c_), keep(c_->stream) { c->stream = s; }
                                ~StreamSwap() { c->stream = keep; } } on_frame_stream(c, fs_st);
            if (!c->ev_px) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_px, hipEventDisableTiming), "create event");
            ScopedTimer tf(c, 3);              // (the call's timer on the stream that carries the call)
            c->want_px_event = true;
            rc = run_dwt(c, nplanes, nullptr, c->p1.p, d_px, ntiles, false, h16);
            c->want_px_event = false;
            if (rc) return rc;
            // the pixel-lifetime contract of every other path: work queued on the context's stream after this call comes after the read
            c->px_event_valid = true;
            if (!c->px_hold)
                HIP_TRY(c, hipStreamWaitEvent(on_frame_stream.keep, c->ev_px, 0), "the context's stream waits for the pixels' last read");
            rc = run_ht(c, ntiles, c->p1.p, false, h16, (c->k3_room & 1) != 0); if (rc) return rc;
            HIP_TRY(c, hipEventRecord(c->ev_side, fs_st), "record the frame's stream");
            HIP_TRY(c, hipEventRecord(c->ev_side2, fs_st), "record the frame's stream");
            c->side_pending = true;
            if (table || total) return grk_amd_fetch_table(c, table, total);
            return GRK_AMD_OK;
        }
        c->px_event_valid = false;          // (the pixels are read on the context's stream itself from here on)
        if (ov) {       // the allocator must be reset before the first K3 launch of either stream
            int rc2 = GRK_AMD_OK;
            const HtArgs h = make_ht_args(c, ntiles, c->p1.p, &rc2, h16);
            if (rc2) return rc2;
            // (with the fused level 0 its first workgroup does it: one launch less on the main stream's chain)
            if (fused && c->alloc_in_level0) { c->pend_alloc = h.alloc; c->pend_alloc_units = h.chunk_units; }
            else HIP_TRY(c, launch_ht_alloc_init(h, c->stream), "reset arena allocator");
        }
        if (fused) {
            rc = run_dwt(c, nplanes, nullptr, c->p1.p, d_px, ntiles, ov, h16); if (rc) return rc;
        } else {
            rc = run_ingest(c, ntiles, d_px, c->p0.p); if (rc) return rc;
            rc = run_dwt(c, nplanes, c->p0.p, c->p1.p, nullptr, ntiles, ov); if (rc) return rc;
        }
        rc = run_ht(c, ntiles, c->p1.p, ov, h16); if (rc) return rc;
    }
    if (table || total) return grk_amd_fetch_table(c, table, total);
    return GRK_AMD_OK;
}

int grk_amd_stream_wait_results(grk_amd_ctx* c, void* hip_stream)
{
    if (!c || !hip_stream) return GRK_AMD_ERR_INVALID;
    hipStream_t s = (hipStream_t)hip_stream;
    if (!c->ev_main) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming), "create event");
    HIP_TRY(c, hipEventRecord(c->ev_main, c->stream), "record main stream");
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_main, 0), "wait for the main stream");
    if (c->side_pending) {
        HIP_TRY(c, hipStreamWaitEvent(s, c->ev_side, 0), "wait for the side stream");
        HIP_TRY(c, hipStreamWaitEvent(s, c->ev_side2, 0), "wait for the side stream 2");
    }
    return GRK_AMD_OK;
}

int grk_amd_set_pixel_hold(grk_amd_ctx* c, int on)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    c->px_hold = on != 0;
    return GRK_AMD_OK;
}

int grk_amd_stream_wait_pixels(grk_amd_ctx* c, void* hip_stream)
{
    if (!c || !hip_stream) return GRK_AMD_ERR_INVALID;
    hipStream_t s = (hipStream_t)hip_stream;
    if (c->px_event_valid) { HIP_TRY(c, hipStreamWaitEvent(s, c->ev_px, 0), "wait for the pixels' last read"); return GRK_AMD_OK; }
    if (s == c->stream) return GRK_AMD_OK;        // (stream order)
    if (!c->ev_main) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming), "create event");
    HIP_TRY(c, hipEventRecord(c->ev_main, c->stream), "record main stream");
    HIP_TRY(c, hipStreamWaitEvent(s, c->ev_main, 0), "wait for the main stream");
    return GRK_AMD_OK;
}

int grk_amd_get_pipelining(grk_amd_ctx* c)
{
    return c && c->pipelining && c->overlap && c->side ? c->pipe_depth - 1 : 0;
}

int grk_amd_set_pipelining(grk_amd_ctx* c, int on)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    const int rc = grk_amd_synchronize(c);
    c->pipelining = on != 0 && c->side != nullptr && c->side2 != nullptr;
    c->pipe_depth = std::min(std::max(on, 1) + 1, grk_amd_ctx::kMaxAltSets + 1);
    c->alt_head = 0;
    return rc;
}

void* grk_amd_host_alloc(grk_amd_ctx* c, uint64_t bytes)
{
    if (!c || !bytes) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void grk_amd_host_free(grk_amd_ctx* c, void* p)
{
    if (p) { if (c) (void)hipSetDevice(c->device); (void)hipHostFree(p); }
}

int grk_amd_plane_sample_bytes(grk_amd_ctx* c, const grk_amd_tile_params* p, int decode, uint32_t* packed_levels)
{
    if (!c || !p) return GRK_AMD_ERR_INVALID;
    const uint32_t bps = (p->prec + 7u) / 8u;
    bool h16;
    if (decode)
        h16 = c->dec_planes16 && p->num_levels >= 1 && bps <= 2 && c->fuse_egress && !p->reserved[0] && !p->irreversible &&
              p->prec <= 8 && c->dec_seg_first.empty();
    else
        h16 = c->planes16 && p->num_levels >= 1 && planes16_ok(*p);
    if (packed_levels) {
        *packed_levels = 0;
        if (h16 && c->dwt_pk && !decode)
            for (uint32_t l = 0; l < p->num_levels; ++l) *packed_levels += pk16_level_ok(*p, l) ? 1u : 0u;
    }
    return h16 ? 2 : 4;
}

int grk_amd_set_decode_planes16(grk_amd_ctx* c, int on)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    c->dec_planes16 = on != 0;
    return GRK_AMD_OK;
}

int grk_amd_set_overlap(grk_amd_ctx* c, int on)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    (void)grk_amd_synchronize(c);
    c->overlap = on != 0 && c->side != nullptr && c->side2 != nullptr;
    return GRK_AMD_OK;
}

int grk_amd_enable_timing(grk_amd_ctx* c, int on)
{
    if (!c) return GRK_AMD_ERR_INVALID;
    (void)hipStreamSynchronize(c->stream);
    drain_timers(c);
    for (auto& t : c->timers) { t.total_ms = 0; t.launches = 0; }
    c->timing = on != 0;
    return GRK_AMD_OK;
}

double grk_amd_kernel_ms(grk_amd_ctx* c, int which, uint32_t* launches)
{
    if (!c || which < 0 || which > 9) return -1.0;
    (void)hipStreamSynchronize(c->stream);
    drain_timers(c);
    if (launches) *launches = c->timers[which].launches;
    return c->timers[which].launches ? c->timers[which].total_ms / c->timers[which].launches : 0.0;
}

} // extern "C"
