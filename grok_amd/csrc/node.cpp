// grok_amd/csrc/node.cpp -- one image over the GPUs of a node, natively: one grk_amd_ctx and one host thread per device,
// tiles t -> device t mod R, the coded tile-parts brought together into ONE codestream.
//
// The reference's analogue is its tile-level task pool (codestream/CodeStreamCompress.cpp:535-603: tiles are independent
// tasks, their tile-parts are written in index order); SURVEY.md §8(e): the path shards by tile with no data-path collective,
// the one real exchange is making one file of the devices' tile-parts.  Two forms of that exchange (grk_amd_node_encode_image):
//   * parallel writers (default): every worker makes its tiles' finished tile-parts in its own HBM (Tier-2 on the device:
//     grk_amd_assemble_device, kernels_t2.hip) and brings them to the host over its own PCIe link; the caller's thread writes the main
//     header (TLM from the sizes) and says where they go.  GRK_AMD_NODE_T2=host: the r06 route -- loose coded bytes to the host, Tier-2
//     there as a plan (grk_amd_plan_tile_part) on the workers' threads, 49 152 segments per 8K frame placed by the host's threads;
//   * gather (GRK_AMD_NODE_GATHER): every worker copies its coded bytes device-to-device (hipMemcpyPeer: xGMI between two
//     GPUs) into the frame's WRITER device, which rotates with the frame number so that consecutive frames spread over all
//     GPUs' links; the writer brings everything to the host in one piece and runs Tier-2 for all tiles -- north_star's
//     "gather of coded tile-parts over xGMI" in a single process, without RCCL.
// A device may appear more than once in the list (two contexts on one GPU): every code path runs on a one-GPU box.
#include "../../include/grok_amd.h"
#include "geometry.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace grk_amd;

struct grk_amd_node {
    struct Worker {
        int device = 0;
        grk_amd_ctx* ctx = nullptr;
        uint8_t* pin_px = nullptr; size_t pin_px_cap = 0;        // tile pixels of one geometry group, pinned
        void* dev_px = nullptr; size_t dev_px_cap = 0;           // the same on the device (device-resident image: the tiles are cut out by 2-D copies)
        std::vector<hipEvent_t> copied;                          // gather: group k's coded bytes have left this worker's arena
        uint8_t* pin_coded = nullptr; size_t pin_coded_cap = 0;  // coded bytes on the host, pinned
        void* gather = nullptr; size_t gather_cap = 0;           // device memory: where the other workers' bytes land when this one is the writer
        hipStream_t copy = nullptr;                              // this worker's device-to-device / download stream
    };
    std::vector<Worker> w;
    uint64_t frame = 0;
    std::string err;
    std::mutex mu;                                               // one grk_amd_node_encode_image at a time
};

namespace {

bool pin_ensure(grk_amd_ctx* ctx, uint8_t*& p, size_t& cap, size_t n)
{
    if (n <= cap) return true;
    if (p) grk_amd_host_free(ctx, p);
    cap = 0;
    p = (uint8_t*)grk_amd_host_alloc(ctx, n + (n >> 3) + 4096);
    if (!p) return false;
    cap = n + (n >> 3) + 4096;
    return true;
}

constexpr int kRing = 3;       // a worker's encodes rotate kRing + 1 buffer sets (a regular tiling has at most four geometry groups)

// rows of `w` bytes, `src_pitch` apart -> tight; a few threads when there is enough to copy (one core moves ~10 GB/s, a worker of
// BASELINE configs[3] stages 100 MB per image)
void copy_rows(uint8_t* dst, const uint8_t* src, size_t w, size_t src_pitch, size_t rows)
{
    for (size_t y = 0; y < rows; ++y) std::memcpy(dst + y * w, src + y * src_pitch, w);
}

// the same, keeping the first `keep` bytes when the buffer has to grow
bool pin_grow(grk_amd_ctx* ctx, uint8_t*& p, size_t& cap, size_t n, size_t keep)
{
    if (n <= cap) return true;
    const size_t want = n + (n >> 2) + 4096;
    uint8_t* q = (uint8_t*)grk_amd_host_alloc(ctx, want);
    if (!q) return false;
    if (p && keep) std::memcpy(q, p, keep);
    if (p) grk_amd_host_free(ctx, p);
    p = q; cap = want;
    return true;
}

struct TileJob {               // what the workers leave per tile
    std::vector<grk_amd_coded_block> rows;     // offsets into the coded bytes on the host (the owning worker's, or the writer's)
    uint64_t part_len = 0;                     // the tile-part's size
    bool planned = false;                      // ... its plan was made by the tile's worker (beside the download of its bytes)
    std::vector<uint8_t> lit;                  // its plan (plan_tile_part): marker segments + packet headers ...
    std::vector<grk_amd_tp_segment> segs;      // ... and the segments it is made of
    uint64_t dev_at = 0;                       // Tier-2 on the device: where the finished tile-part lies in its worker's assembled bytes
};

// fn(t) for t in [0, n) on up to `threads` host threads; the first error wins
template <class F> int parallel_tiles(uint32_t n, uint32_t threads, F fn)
{
    threads = std::max(1u, std::min(threads, n));
    std::atomic<uint32_t> next{0};
    std::atomic<int> rc{GRK_AMD_OK};
    auto work = [&]() {
        for (;;) {
            const uint32_t t = next.fetch_add(1);
            if (t >= n || rc.load() != GRK_AMD_OK) return;
            const int r = fn(t);
            if (r != GRK_AMD_OK) { int ok = GRK_AMD_OK; (void)rc.compare_exchange_strong(ok, r); }
        }
    };
    std::vector<std::thread> th;
    for (uint32_t i = 1; i < threads; ++i) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return rc.load();
}

bool is_pinned(const void* p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// GRK_AMD_NODE_TRACE=1: where an image's time goes (stderr, one line per phase of worker 0 and of the caller's thread)
struct Trace {
    bool on; std::chrono::steady_clock::time_point t0;
    Trace() : on(std::getenv("GRK_AMD_NODE_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[node] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Tier-2 where the coded bytes are (grk_amd_assemble_device) unless GRK_AMD_NODE_T2=host asks for the host writer's plan
bool device_t2()
{
    const char* e = std::getenv("GRK_AMD_NODE_T2");                  // (read per image: the tests take both routes in one process)
    return !(e && std::strcmp(e, "host") == 0);
}

} // namespace

extern "C" int grk_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int grk_amd_node_create(const int* devices, uint32_t n, int verbose, grk_amd_node** out)
{
    if (!out) return GRK_AMD_ERR_INVALID;
    *out = nullptr;
    const int have = grk_amd_device_count();
    if (have <= 0) return GRK_AMD_ERR_NO_DEVICE;
    std::vector<int> devs;
    if (devices && n) devs.assign(devices, devices + n);
    else for (int d = 0; d < have; ++d) devs.push_back(d);           // all GPUs of the node
    auto* nd = new grk_amd_node();
    nd->w.resize(devs.size());
    for (size_t i = 0; i < devs.size(); ++i) {
        nd->w[i].device = devs[i];
        const int rc = grk_amd_create(devs[i], verbose, &nd->w[i].ctx);
        if (rc != GRK_AMD_OK) { grk_amd_node_destroy(nd); return rc; }
        // (the rotation of kRing + 1 buffer sets -- each a set of Mallat planes, arena and tables: several GB per 8K-tile worker --
        //  is switched on by the first GATHER encode only: parallel writers never use it, and a context handed out through
        //  grk_amd_node_ctx keeps the result lifetime and memory behaviour of a context the caller made itself)
        if (hipSetDevice(devs[i]) != hipSuccess || hipStreamCreateWithFlags(&nd->w[i].copy, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            grk_amd_node_destroy(nd);
            return GRK_AMD_ERR_NO_DEVICE;
        }
    }
    // device-to-device copies between distinct GPUs go over xGMI once peer access is on (without it they pass through the host)
    for (size_t i = 0; i < devs.size(); ++i)
        for (size_t j = 0; j < devs.size(); ++j) {
            if (devs[i] == devs[j]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[i], devs[j]) == hipSuccess && can && hipSetDevice(devs[i]) == hipSuccess)
                (void)hipDeviceEnablePeerAccess(devs[j], 0);         // ("already enabled" is fine)
            (void)hipGetLastError();
        }
    *out = nd;
    return GRK_AMD_OK;
}

extern "C" void grk_amd_node_destroy(grk_amd_node* nd)
{
    if (!nd) return;
    for (auto& w : nd->w) {
        if (w.pin_px) grk_amd_host_free(w.ctx, w.pin_px);
        if (w.pin_coded) grk_amd_host_free(w.ctx, w.pin_coded);
        if (w.dev_px) { (void)hipSetDevice(w.device); (void)hipFree(w.dev_px); }
        for (hipEvent_t e : w.copied) { (void)hipSetDevice(w.device); (void)hipEventDestroy(e); }
        if (w.gather) { (void)hipSetDevice(w.device); (void)hipFree(w.gather); }
        if (w.copy) { (void)hipSetDevice(w.device); (void)hipStreamDestroy(w.copy); }
        if (w.ctx) grk_amd_destroy(w.ctx);
    }
    delete nd;
}

extern "C" uint32_t grk_amd_node_size(const grk_amd_node* nd) { return nd ? (uint32_t)nd->w.size() : 0u; }
extern "C" grk_amd_ctx* grk_amd_node_ctx(grk_amd_node* nd, uint32_t i) { return nd && i < nd->w.size() ? nd->w[i].ctx : nullptr; }
extern "C" const char* grk_amd_node_last_error(grk_amd_node* nd) { return nd ? nd->err.c_str() : "null node"; }

static int64_t node_encode_image(grk_amd_node* nd, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                 const void* pixels, int pixels_device, uint32_t flags, uint8_t* out, uint64_t cap, bool host_t2);

static int64_t node_encode_locked(grk_amd_node* nd, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                  const void* pixels, int pixels_device, uint32_t flags, uint8_t* out, uint64_t cap)
{
    if (!nd) return GRK_AMD_ERR_INVALID;
    std::lock_guard<std::mutex> lk(nd->mu);
    nd->err.clear();
    int64_t rc = node_encode_image(nd, im, base, pixels, pixels_device, flags, out, cap, false);
    // (a layout beyond the device writer's tables -- header bits of one packet past 2^31, a code-block of 512 MB: the host writer takes it)
    if (rc == GRK_AMD_ERR_UNSUPPORTED && !(flags & GRK_AMD_NODE_GATHER) && device_t2()) {
        nd->err.clear();
        rc = node_encode_image(nd, im, base, pixels, pixels_device, flags, out, cap, true);
    }
    if (rc < 0 && nd->err.empty()) {
        nd->err = rc == GRK_AMD_ERR_INVALID ? "invalid argument" : rc == GRK_AMD_ERR_UNSUPPORTED ? "unsupported layout (TLM with more than 255 tiles?)"
                : rc == GRK_AMD_ERR_NOMEM ? "out of (pinned or device) memory" : rc == GRK_AMD_ERR_NO_DEVICE ? "a HIP call failed (device lost or peer copy refused)"
                : rc == GRK_AMD_ERR_OVERFLOW ? "output or gather buffer too small" : "encode failed";
        nd->err += " (code " + std::to_string((long long)rc) + ")";
    }
    return rc;
}

// One image at a time per node: the call owns the workers' contexts, their pinned buffers and the gather buffers for its duration
// (callers from several threads queue on the node's mutex).  Every error return leaves its reason in grk_amd_node_last_error.
extern "C" int64_t grk_amd_node_encode_image(grk_amd_node* nd, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                             const void* pixels, uint32_t flags, uint8_t* out, uint64_t cap)
{
    return node_encode_locked(nd, im, base, pixels, -1, flags, out, cap);
}

// The image resident in the memory of HIP device `pixels_device` (component-major planar, tight): every worker cuts its tiles out
// with 2-D device-to-device copies (over xGMI from another GPU's memory) -- no pixel crosses PCIe.
extern "C" int64_t grk_amd_node_encode_image_device(grk_amd_node* nd, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                                    const void* pixels, int pixels_device, uint32_t flags, uint8_t* out, uint64_t cap)
{
    if (pixels_device < 0 || pixels_device >= grk_amd_device_count()) {
        if (nd) { std::lock_guard<std::mutex> lk(nd->mu); nd->err = "invalid argument: pixels_device"; }
        return GRK_AMD_ERR_INVALID;
    }
    return node_encode_locked(nd, im, base, pixels, pixels_device, flags, out, cap);
}

static int64_t node_encode_image(grk_amd_node* nd, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                 const void* pixels, int pixels_device, uint32_t flags, uint8_t* out, uint64_t cap, bool host_t2)
{
    if (nd->w.empty() || !im || !base || !pixels || !out) return GRK_AMD_ERR_INVALID;
    const int64_t nt = grk_amd_layout_num_tiles(im);
    if (nt < 0) return nt;
    const uint32_t ntiles = (uint32_t)nt, R = (uint32_t)nd->w.size();
    const bool gather = (flags & GRK_AMD_NODE_GATHER) != 0;
    const uint32_t cs_flags = flags & ~GRK_AMD_NODE_GATHER;
    if ((cs_flags & GRK_AMD_CS_TLM) && ntiles > 255) return GRK_AMD_ERR_UNSUPPORTED;
    const uint32_t W = im->x1 - im->x0, H = im->y1 - im->y0;
    const uint32_t bps = (base->prec + 7u) / 8u, nc = base->num_comps;
    // the tiles, grouped by geometry (image.cpp): a batch of grk_amd_encode_tiles shares one
    std::vector<grk_amd_tile_params> tp(ntiles);
    std::vector<TileGeom> geoms;
    std::vector<uint32_t> group_of(ntiles);
    for (uint32_t t = 0; t < ntiles; ++t) {
        int rc = grk_amd_layout_tile(im, base, t, &tp[t]);
        if (rc) return rc;
        TileGeom g;
        rc = build_tile_geom(tp[t], g);
        if (rc) return rc;
        size_t k = 0;
        for (; k < geoms.size(); ++k) if (same_geometry(geoms[k], g)) break;
        if (k == geoms.size()) geoms.push_back(std::move(g));
        group_of[t] = (uint32_t)k;
    }
    std::vector<TileJob> jobs(ntiles);
    const uint32_t writer = (uint32_t)(nd->frame++ % R);
    // gather: worker r's bytes land at gather_at[r] of the writer's device buffer; an upper bound of what a worker can produce
    // (raw size x 2 + slack per block, what grk_amd_encode_tiles sizes its own arena with) keeps the offsets independent of
    // the coding, so that nobody waits for anybody's byte count
    std::vector<uint64_t> gather_at(R + 1, 0), used(R, 0);
    if (gather) {
        for (uint32_t r = 0; r < R; ++r) {
            uint64_t ub = 0;
            for (uint32_t t = r; t < ntiles; t += R)
                ub += (uint64_t)tp[t].tile_w * tp[t].tile_h * nc * bps * 2u + (uint64_t)geoms[group_of[t]].blocks_per_comp * nc * 64u + (1u << 20);
            gather_at[r + 1] = gather_at[r] + ((ub + 255u) & ~255ull);
        }
        auto& ww = nd->w[writer];
        if (ww.gather_cap < gather_at[R]) {
            if (hipSetDevice(ww.device) != hipSuccess) return GRK_AMD_ERR_NO_DEVICE;
            if (ww.gather) (void)hipFree(ww.gather);
            ww.gather = nullptr; ww.gather_cap = 0;
            if (hipMalloc(&ww.gather, gather_at[R]) != hipSuccess) { (void)hipGetLastError(); return GRK_AMD_ERR_NOMEM; }
            ww.gather_cap = gather_at[R];
        }
    }
    // Parallel writers, Tier-2 on the device (the default): a worker's encode is followed by grk_amd_assemble_device -- packet headers
    // and the gather of the code-blocks' bytes into finished tile-parts in its HBM --, no table and no loose coded bytes come to the
    // host; the tile-parts are brought to their places in the file once the main header's length is known.
    const bool dev_t2 = !gather && !host_t2 && device_t2();
    std::vector<uint64_t> asm_used(R, 0);
    std::vector<int> rcs(R, GRK_AMD_OK);
    Trace trace;
    std::vector<std::thread> th;
    auto worker = [&](uint32_t r) {
            auto& w = nd->w[r];
            int rc = GRK_AMD_OK;
            Trace wtrace;
            uint64_t coded_used = 0;                       // this worker's coded bytes so far (all its groups, one after the other)
            size_t ngroup = 0;                             // gather: groups whose bytes are on their way
            // gather: kRing + 1 buffer sets in rotation -- the coded bytes of a geometry group stay where they are while the next
            // groups are coded, so that their copy to the writer's device runs beside those encodes instead of being waited for
            // (memory: x (kRing + 1) of the worker's planes / arena / tables); parallel writers: one set
            if (gather && geoms.size() > 1 && grk_amd_get_pipelining(w.ctx) < kRing) (void)grk_amd_set_pipelining(w.ctx, kRing);
            for (size_t k = 0; k < geoms.size() && rc == GRK_AMD_OK; ++k) {
                std::vector<uint32_t> mine;
                for (uint32_t t = r; t < ntiles; t += R) if (group_of[t] == k) mine.push_back(t);
                if (mine.empty()) continue;
                const grk_amd_tile_params& p = tp[mine[0]];
                const size_t tile_bytes = (size_t)p.tile_w * p.tile_h * nc * bps;
                const void* enc_px = nullptr;
                if (pixels_device < 0 && mine.size() == 1 && p.tile_w == W && p.tile_h == H) {
                    // (the tile IS the image: uploaded from where it lies -- grk_amd_encode_tiles moves pageable memory through pinned chunks
                    //  with the copies and the DMAs overlapped, pinned memory in one DMA -- instead of being staged whole first)
                    enc_px = pixels;
                } else if (pixels_device < 0) {
                    if (!pin_ensure(w.ctx, w.pin_px, w.pin_px_cap, tile_bytes * mine.size())) { rc = GRK_AMD_ERR_NOMEM; break; }
                    // the tiles' rows out of the caller's image into pinned memory; several threads when it is much, each taking
                    // its share of the rows of every tile component (one tile of 8192 x 8192 x 3 is 200 MB for one worker)
                    const size_t nthr = tile_bytes * mine.size() >= (32u << 20) ? 4 : 1;
                    auto stage = [&](size_t j) {
                        for (size_t i = 0; i < mine.size(); ++i) {
                            const grk_amd_tile_params& q = tp[mine[i]];
                            const size_t ox = q.tile_x0 - im->x0, oy = q.tile_y0 - im->y0;
                            const size_t y0 = (size_t)q.tile_h * j / nthr, y1 = (size_t)q.tile_h * (j + 1) / nthr;
                            for (uint32_t c = 0; c < nc; ++c)
                                copy_rows(w.pin_px + i * tile_bytes + ((size_t)c * q.tile_h + y0) * q.tile_w * bps,
                                          (const uint8_t*)pixels + (((size_t)c * H + oy + y0) * W + ox) * bps, (size_t)q.tile_w * bps, (size_t)W * bps, y1 - y0);
                        }
                    };
                    if (nthr > 1) {
                        std::vector<std::thread> st;
                        for (size_t j = 1; j < nthr; ++j) st.emplace_back(stage, j);
                        stage(0);
                        for (auto& t : st) t.join();
                    } else stage(0);
                    enc_px = w.pin_px;
                } else {
                    // device-resident image: 2-D copies (rows of the tile out of rows of the image) on this worker's copy stream
                    if (hipSetDevice(w.device) != hipSuccess) { rc = GRK_AMD_ERR_NO_DEVICE; break; }
                    // (the tile IS the image, in this worker's own memory: coded where it lies -- 200 MB less to read and write per 8K frame)
                    if (mine.size() == 1 && p.tile_w == W && p.tile_h == H && pixels_device == w.device) { enc_px = pixels; goto staged; }
                    if (w.dev_px_cap < tile_bytes * mine.size()) {
                        // (the encode that read the old buffer has returned: grk_amd_encode_tiles below is synchronous)
                        if (w.dev_px) (void)hipFree(w.dev_px);
                        w.dev_px = nullptr; w.dev_px_cap = 0;
                        if (hipMalloc(&w.dev_px, tile_bytes * mine.size() + 256) != hipSuccess) { (void)hipGetLastError(); rc = GRK_AMD_ERR_NOMEM; break; }
                        w.dev_px_cap = tile_bytes * mine.size();
                    }
                    hipError_t e = hipSuccess;
                    for (size_t i = 0; i < mine.size() && e == hipSuccess; ++i) {
                        const grk_amd_tile_params& q = tp[mine[i]];
                        const size_t ox = q.tile_x0 - im->x0, oy = q.tile_y0 - im->y0;
                        for (uint32_t c = 0; c < nc && e == hipSuccess; ++c)
                            e = hipMemcpy2DAsync((uint8_t*)w.dev_px + i * tile_bytes + (size_t)c * q.tile_h * q.tile_w * bps, (size_t)q.tile_w * bps,
                                                 (const uint8_t*)pixels + (((size_t)c * H + oy) * W + ox) * bps, (size_t)W * bps,
                                                 (size_t)q.tile_w * bps, q.tile_h, hipMemcpyDeviceToDevice, w.copy);
                    }
                    if (e == hipSuccess) e = hipStreamSynchronize(w.copy);
                    if (e != hipSuccess) { (void)hipGetLastError(); rc = GRK_AMD_ERR_NO_DEVICE; break; }
                    enc_px = w.dev_px;
                }
                staged:
                const uint64_t bpt = (uint64_t)geoms[k].blocks_per_comp * nc;
                std::vector<grk_amd_coded_block> table(dev_t2 ? 0 : bpt * mine.size());
                uint64_t total = 0;
                // (the buffer set this encode takes was used kRing + 1 encodes ago: its bytes must have left by now; without the
                //  rotation -- no DWT level, overlap switched off -- every encode writes the one arena: wait for the last copy)
                const bool ring = p.num_levels >= 1 && grk_amd_get_pipelining(w.ctx) >= kRing;
                if (gather && !ring && ngroup && hipStreamSynchronize(w.copy) != hipSuccess) { (void)hipGetLastError(); rc = GRK_AMD_ERR_NO_DEVICE; break; }
                if (gather && ring && ngroup > (size_t)kRing) {
                    if (hipEventSynchronize(w.copied[(ngroup - 1 - kRing) % w.copied.size()]) != hipSuccess) { (void)hipGetLastError(); rc = GRK_AMD_ERR_NO_DEVICE; break; }
                }
                if (dev_t2) {
                    if (r == 0) wtrace.mark("worker 0: pixels staged");
                    rc = grk_amd_encode_tiles(w.ctx, &p, (uint32_t)mine.size(), enc_px, pixels_device >= 0, nullptr, nullptr);
                    if (rc) break;
                    if (r == 0) wtrace.mark("worker 0: encode queued");
                    std::vector<uint32_t> lens(mine.size());
                    const int64_t n = grk_amd_assemble_device(w.ctx, &p, (uint32_t)mine.size(), mine.data(), cs_flags, asm_used[r], lens.data());
                    if (r == 0) wtrace.mark("worker 0: assembled (synced)");
                    if (n < 0) { rc = (int)n; break; }
                    uint64_t at = asm_used[r];
                    for (size_t i = 0; i < mine.size(); ++i) { TileJob& j = jobs[mine[i]]; j.part_len = lens[i]; j.dev_at = at; at += lens[i]; }
                    asm_used[r] += (uint64_t)n;
                    continue;
                }
                rc = grk_amd_encode_tiles(w.ctx, &p, (uint32_t)mine.size(), enc_px, pixels_device >= 0, table.data(), &total);
                if (rc) break;
                if (gather) {
                    // device to device into the writer's buffer.  Ordering contract: grk_amd_encode_tiles was given a table pointer, so it
                    // returned through grk_amd_fetch_table -- side streams joined and the context's stream synchronised --: the arena
                    // is complete on the device before this copy is queued on w.copy (an asynchronous encode_tiles would need an
                    // event from the context's stream here instead).  The copy is NOT waited for: the context rotates kRing + 1
                    // buffer sets, the next groups are coded into other arenas while this one's bytes travel, and an event per group
                    // says when its set may be written again (above) -- the worker waits once, behind its last group.
                    auto& ww = nd->w[writer];
                    if (gather_at[r] + coded_used + total > gather_at[r + 1]) { rc = GRK_AMD_ERR_OVERFLOW; break; }
                    if (hipSetDevice(w.device) != hipSuccess) { rc = GRK_AMD_ERR_NO_DEVICE; break; }
                    while (w.copied.size() < (size_t)kRing + 1) {
                        hipEvent_t ev = nullptr;
                        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); rc = GRK_AMD_ERR_NO_DEVICE; break; }
                        w.copied.push_back(ev);
                    }
                    if (rc) break;
                    if ((total && hipMemcpyPeerAsync((char*)ww.gather + gather_at[r] + coded_used, ww.device, grk_amd_coded_device_ptr(w.ctx),
                                                     w.device, total, w.copy) != hipSuccess) ||
                        hipEventRecord(w.copied[ngroup % w.copied.size()], w.copy) != hipSuccess) { (void)hipGetLastError(); rc = GRK_AMD_ERR_NO_DEVICE; break; }
                    ++ngroup;
                } else {
                    // (the worker's groups one after the other: Tier-2 runs over all of them once every worker is done)
                    if (!pin_grow(w.ctx, w.pin_coded, w.pin_coded_cap, coded_used + total + 16, coded_used)) { rc = GRK_AMD_ERR_NOMEM; break; }
                    // (queued, not waited for: Tier-2 of these tiles needs their table only and runs below while the bytes cross the link)
                    rc = grk_amd_fetch_coded_async(w.ctx, w.pin_coded + coded_used, total);
                    if (rc) break;
                }
                for (size_t i = 0; i < mine.size(); ++i) {
                    TileJob& j = jobs[mine[i]];
                    j.rows.assign(table.begin() + i * bpt, table.begin() + (i + 1) * bpt);
                    for (auto& row : j.rows) row.offset += coded_used;
                }
                // Tier-2 of these tiles here, on the worker's thread, while their bytes travel (over PCIe to the pinned buffer, or
                // device to device to the frame's writer): a plan needs the table only
                for (size_t i = 0; i < mine.size() && rc == GRK_AMD_OK; ++i) {
                    TileJob& j = jobs[mine[i]];
                    const int64_t need = plan_tile_part(tp[mine[i]], mine[i], cs_flags, j.rows.data(), j.lit, j.segs);
                    if (need < 0) rc = (int)need; else { j.part_len = (uint64_t)need; j.planned = true; }
                }
                if (!gather) {
                    // the bytes are in the pinned buffer before the next group's encode writes the arena again
                    const int sr = grk_amd_synchronize(w.ctx);
                    if (rc == GRK_AMD_OK) rc = sr;
                }
                if (rc) break;
                coded_used += total;
            }
            if (gather && (hipSetDevice(w.device) != hipSuccess || hipStreamSynchronize(w.copy) != hipSuccess)) {      // the last groups' bytes
                (void)hipGetLastError();
                if (!rc) rc = GRK_AMD_ERR_NO_DEVICE;
            }
            used[r] = coded_used;
            rcs[r] = rc;
        };
    // (one worker: on the caller's thread -- making and joining a thread costs an 8K frame 0.05-0.08 ms)
    if (R == 1) worker(0);
    else for (uint32_t r = 0; r < R; ++r) th.emplace_back(worker, r);
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < R; ++r)
        if (rcs[r]) { nd->err = std::string("worker ") + std::to_string(r) + ": " + grk_amd_last_error(nd->w[r].ctx); return rcs[r]; }

    trace.mark("workers done");
    if (dev_t2) {
        std::vector<uint32_t> sizes(ntiles);
        for (uint32_t t = 0; t < ntiles; ++t) sizes[t] = (uint32_t)jobs[t].part_len;
        const int64_t hdr = grk_amd_write_main_header_layout(im, base, cs_flags, sizes.data(), out, cap);
        if (hdr < 0) return hdr;
        std::vector<uint64_t> at(ntiles + 1, (uint64_t)hdr);
        for (uint32_t t = 0; t < ntiles; ++t) at[t + 1] = at[t] + jobs[t].part_len;
        if (at[ntiles] + 2 > cap) return GRK_AMD_ERR_OVERFLOW;
        int rc = GRK_AMD_OK;
        if (R == 1 && geoms.size() == 1) {
            // one worker, one batch: its assembled bytes ARE the file behind the main header -- one DMA into pinned memory, pinned
            // chunks on several copy threads into pageable memory
            rc = grk_amd_fetch_assembled(nd->w[0].ctx, 0, asm_used[0], out + hdr);
        } else if (is_pinned(out)) {
            for (uint32_t t = 0; t < ntiles && rc == GRK_AMD_OK; ++t)
                rc = grk_amd_fetch_assembled_async(nd->w[t % R].ctx, jobs[t].dev_at, jobs[t].part_len, out + at[t]);
            for (uint32_t r = 0; r < R; ++r) { const int sr = grk_amd_synchronize(nd->w[r].ctx); if (rc == GRK_AMD_OK) rc = sr; }
        } else {
            // every worker's assembled bytes over its own link into its pinned buffer, then the tile-parts -- whole, not 49 152
            // code-blocks each -- to their places on the host's threads
            for (uint32_t r = 0; r < R && rc == GRK_AMD_OK; ++r) {
                auto& w = nd->w[r];
                if (!pin_ensure(w.ctx, w.pin_coded, w.pin_coded_cap, asm_used[r] + 16)) { rc = GRK_AMD_ERR_NOMEM; break; }
                rc = grk_amd_fetch_assembled_async(w.ctx, 0, asm_used[r], w.pin_coded);
            }
            for (uint32_t r = 0; r < R; ++r) { const int sr = grk_amd_synchronize(nd->w[r].ctx); if (rc == GRK_AMD_OK) rc = sr; }
            if (rc == GRK_AMD_OK) {
                struct Piece { uint32_t t; uint64_t o, n; };
                std::vector<Piece> pieces;
                for (uint32_t t = 0; t < ntiles; ++t)
                    for (uint64_t o = 0; o < jobs[t].part_len; o += 2u << 20) pieces.push_back(Piece{t, o, std::min<uint64_t>(2u << 20, jobs[t].part_len - o)});
                std::atomic<size_t> next{0};
                auto work = [&]() {
                    for (;;) {
                        const size_t k = next.fetch_add(1);
                        if (k >= pieces.size()) break;
                        const Piece& pc = pieces[k];
                        std::memcpy(out + at[pc.t] + pc.o, nd->w[pc.t % R].pin_coded + jobs[pc.t].dev_at + pc.o, pc.n);
                    }
                };
                const uint32_t nthr = (uint32_t)std::min<size_t>(std::min<uint32_t>(16u, std::max(1u, std::thread::hardware_concurrency() / 4u)), pieces.size());
                std::vector<std::thread> pool;
                for (uint32_t i = 1; i < nthr; ++i) pool.emplace_back(work);
                work();
                for (auto& th2 : pool) th2.join();
            }
        }
        if (rc) {
            for (uint32_t r = 0; r < R; ++r) if (*grk_amd_last_error(nd->w[r].ctx)) { nd->err = std::string("worker ") + std::to_string(r) + ": " + grk_amd_last_error(nd->w[r].ctx); break; }
            return rc;
        }
        trace.mark("header + tile-parts fetched");
        uint64_t end = at[ntiles];
        out[end++] = 0xFF; out[end++] = 0xD9;
        return (int64_t)end;
    }
    // Where every tile's coded bytes are on the host: with parallel writers in its own worker's pinned buffer (fetched over
    // that worker's PCIe link), in the gather form in the writer's (one piece per worker, brought over by the writer's device).
    std::vector<const uint8_t*> src(ntiles, nullptr);
    if (gather) {
        auto& ww = nd->w[writer];
        uint64_t host_total = 0;
        std::vector<uint64_t> host_at(R, 0);
        for (uint32_t r = 0; r < R; ++r) { host_at[r] = host_total; host_total += used[r]; }
        if (!pin_ensure(ww.ctx, ww.pin_coded, ww.pin_coded_cap, host_total + 16)) return GRK_AMD_ERR_NOMEM;
        if (hipSetDevice(ww.device) != hipSuccess) return GRK_AMD_ERR_NO_DEVICE;
        for (uint32_t r = 0; r < R; ++r)
            if (used[r] && hipMemcpyAsync(ww.pin_coded + host_at[r], (const char*)ww.gather + gather_at[r], used[r], hipMemcpyDeviceToHost,
                                          ww.copy) != hipSuccess) {
                (void)hipGetLastError();
                return GRK_AMD_ERR_NO_DEVICE;
            }
        if (hipStreamSynchronize(ww.copy) != hipSuccess) { (void)hipGetLastError(); return GRK_AMD_ERR_NO_DEVICE; }
        for (uint32_t t = 0; t < ntiles; ++t) src[t] = ww.pin_coded + host_at[t % R];
    } else {
        for (uint32_t t = 0; t < ntiles; ++t) src[t] = nd->w[t % R].pin_coded;
    }
    // Tier-2 and the codestream.  Tier-2 ONCE per tile, as a plan (plan_tile_part: the marker segments and packet headers as literal
    // bytes + the list of segments the tile-part is made of; ~1 ms for the 49 152 blocks of an 8K tile), tiles on several host
    // threads; the plans give the tile-parts' sizes, those the main header (TLM) and every tile-part's place; then the segments --
    // ~100 MB of coded bytes per 8K frame -- are copied to where they belong by all the threads, whatever the number of tiles (r05
    // ran Tier-2 twice, once to size and once to write, and one tile's bytes were one thread's work: 12.8 ms per 8K frame as one tile).
    const uint32_t host_threads = std::min<uint32_t>(16u, std::max(1u, std::thread::hardware_concurrency() / 4u));
    int rc = parallel_tiles(ntiles, host_threads, [&](uint32_t t) -> int {
        if (jobs[t].planned) return GRK_AMD_OK;
        const int64_t need = plan_tile_part(tp[t], t, cs_flags, jobs[t].rows.data(), jobs[t].lit, jobs[t].segs);
        if (need < 0) return (int)need;
        jobs[t].part_len = (uint64_t)need;
        return GRK_AMD_OK;
    });
    if (rc) return rc;
    std::vector<uint32_t> sizes(ntiles);
    for (uint32_t t = 0; t < ntiles; ++t) sizes[t] = (uint32_t)jobs[t].part_len;
    const int64_t hdr = grk_amd_write_main_header_layout(im, base, cs_flags, sizes.data(), out, cap);
    if (hdr < 0) return hdr;
    std::vector<uint64_t> at(ntiles + 1, (uint64_t)hdr);
    for (uint32_t t = 0; t < ntiles; ++t) at[t + 1] = at[t] + jobs[t].part_len;
    if (at[ntiles] + 2 > cap) return GRK_AMD_ERR_OVERFLOW;
    {
        // pieces of ~2 MB of output: (tile, first segment, segment count), handed out by a counter
        struct Piece { uint32_t t; size_t s0, s1; };
        std::vector<Piece> pieces;
        for (uint32_t t = 0; t < ntiles; ++t) {
            const auto& sg = jobs[t].segs;
            size_t s0 = 0; uint64_t bytes = 0;
            for (size_t i = 0; i < sg.size(); ++i) {
                bytes += sg[i].len;
                if (bytes >= (2u << 20) || i + 1 == sg.size()) { pieces.push_back(Piece{t, s0, i + 1}); s0 = i + 1; bytes = 0; }
            }
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= pieces.size()) break;
                const Piece& pc = pieces[k];
                const TileJob& j = jobs[pc.t];
                uint8_t* const dst = out + at[pc.t];
                for (size_t i = pc.s0; i < pc.s1; ++i) {
                    const grk_amd_tp_segment& sgm = j.segs[i];
                    std::memcpy(dst + sgm.dst, (sgm.kind ? src[pc.t] : j.lit.data()) + sgm.src, sgm.len);
                }
            }
        };
        std::vector<std::thread> pool;
        const uint32_t nthr = (uint32_t)std::min<size_t>(host_threads, pieces.size());
        for (uint32_t i = 1; i < nthr; ++i) pool.emplace_back(work);
        work();
        for (auto& th2 : pool) th2.join();
    }
    uint64_t end = at[ntiles];
    out[end++] = 0xFF; out[end++] = 0xD9;
    return (int64_t)end;
}
