// grok_amd/csrc/node.cpp -- one image over the GPUs of a node, natively: one grk_amd_ctx and one host thread per device,
// tiles t -> device t mod R, the coded tile-parts brought together into ONE codestream.
//
// The reference's analogue is its tile-level task pool (codestream/CodeStreamCompress.cpp:535-603: tiles are independent
// tasks, their tile-parts are written in index order); SURVEY.md §8(e): the path shards by tile with no data-path collective,
// the one real exchange is making one file of the devices' tile-parts.  Two forms of that exchange (grk_amd_node_encode_image):
//   * parallel writers (default): every worker brings its own coded bytes to the host over its own PCIe link, runs Tier-2 for
//     its own tiles (grk_amd_write_tile_part) -- R host threads write tile-parts at once -- and the caller's thread puts the main
//     header (TLM from the sizes) and the tile-parts together;
//   * gather (GRK_AMD_NODE_GATHER): every worker copies its coded bytes device-to-device (hipMemcpyPeer: xGMI between two
//     GPUs) into the frame's WRITER device, which rotates with the frame number so that consecutive frames spread over all
//     GPUs' links; the writer brings everything to the host in one piece and runs Tier-2 for all tiles -- north_star's
//     "gather of coded tile-parts over xGMI" in a single process, without RCCL.
// A device may appear more than once in the list (two contexts on one GPU): every code path runs on a one-GPU box.
#include "../../include/grok_amd.h"
#include "geometry.h"
#include <hip/hip_runtime.h>
#include <atomic>
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
        uint8_t* pin_coded = nullptr; size_t pin_coded_cap = 0;  // coded bytes on the host, pinned
        void* gather = nullptr; size_t gather_cap = 0;           // device memory: where the other workers' bytes land when this one is the writer
        hipStream_t copy = nullptr;                              // this worker's device-to-device / download stream
        std::vector<uint8_t> parts;                              // this worker's tile-parts, one after the other
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

struct TileJob {               // what the workers leave per tile
    std::vector<grk_amd_coded_block> rows;     // offsets into the owning worker's coded bytes
    uint64_t part_at = 0, part_len = 0;        // parallel writers: the tile-part inside the owner's `parts`
};

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
                                 const void* pixels, uint32_t flags, uint8_t* out, uint64_t cap);

// One image at a time per node: the call owns the workers' contexts, their pinned buffers and the gather buffers for its duration
// (callers from several threads queue on the node's mutex).  Every error return leaves its reason in grk_amd_node_last_error.
extern "C" int64_t grk_amd_node_encode_image(grk_amd_node* nd, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                             const void* pixels, uint32_t flags, uint8_t* out, uint64_t cap)
{
    if (!nd) return GRK_AMD_ERR_INVALID;
    std::lock_guard<std::mutex> lk(nd->mu);
    nd->err.clear();
    const int64_t rc = node_encode_image(nd, im, base, pixels, flags, out, cap);
    if (rc < 0 && nd->err.empty()) {
        nd->err = rc == GRK_AMD_ERR_INVALID ? "invalid argument" : rc == GRK_AMD_ERR_UNSUPPORTED ? "unsupported layout (TLM with more than 255 tiles?)"
                : rc == GRK_AMD_ERR_NOMEM ? "out of (pinned or device) memory" : rc == GRK_AMD_ERR_NO_DEVICE ? "a HIP call failed (device lost or peer copy refused)"
                : rc == GRK_AMD_ERR_OVERFLOW ? "output or gather buffer too small" : "encode failed";
        nd->err += " (code " + std::to_string((long long)rc) + ")";
    }
    return rc;
}

static int64_t node_encode_image(grk_amd_node* nd, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                 const void* pixels, uint32_t flags, uint8_t* out, uint64_t cap)
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
    std::vector<int> rcs(R, GRK_AMD_OK);
    std::vector<std::thread> th;
    for (uint32_t r = 0; r < R; ++r)
        th.emplace_back([&, r]() {
            auto& w = nd->w[r];
            int rc = GRK_AMD_OK;
            uint64_t coded_used = 0;                       // this worker's coded bytes so far (all its groups, one after the other)
            w.parts.clear();
            for (size_t k = 0; k < geoms.size() && rc == GRK_AMD_OK; ++k) {
                std::vector<uint32_t> mine;
                for (uint32_t t = r; t < ntiles; t += R) if (group_of[t] == k) mine.push_back(t);
                if (mine.empty()) continue;
                const grk_amd_tile_params& p = tp[mine[0]];
                const size_t tile_bytes = (size_t)p.tile_w * p.tile_h * nc * bps;
                if (!pin_ensure(w.ctx, w.pin_px, w.pin_px_cap, tile_bytes * mine.size())) { rc = GRK_AMD_ERR_NOMEM; break; }
                for (size_t i = 0; i < mine.size(); ++i) {
                    const grk_amd_tile_params& q = tp[mine[i]];
                    const size_t ox = q.tile_x0 - im->x0, oy = q.tile_y0 - im->y0;
                    for (uint32_t c = 0; c < nc; ++c)
                        for (uint32_t y = 0; y < q.tile_h; ++y)
                            std::memcpy(w.pin_px + i * tile_bytes + ((size_t)c * q.tile_h + y) * q.tile_w * bps,
                                        (const uint8_t*)pixels + (((size_t)c * H + oy + y) * W + ox) * bps, (size_t)q.tile_w * bps);
                }
                const uint64_t bpt = (uint64_t)geoms[k].blocks_per_comp * nc;
                std::vector<grk_amd_coded_block> table(bpt * mine.size());
                uint64_t total = 0;
                rc = grk_amd_encode_tiles(w.ctx, &p, (uint32_t)mine.size(), w.pin_px, 0, table.data(), &total);
                if (rc) break;
                if (gather) {
                    // device to device into the writer's buffer.  Ordering contract: grk_amd_encode_tiles was given a table pointer, so it
                    // returned through grk_amd_fetch_table -- side streams joined and the context's stream synchronised --: the arena
                    // is complete on the device before this copy is queued on w.copy (an asynchronous encode_tiles would need an
                    // event from the context's stream here instead)
                    auto& ww = nd->w[writer];
                    if (gather_at[r] + coded_used + total > gather_at[r + 1]) { rc = GRK_AMD_ERR_OVERFLOW; break; }
                    // (a device-to-device copy returns before it is done: waited for here, the next group's encode writes the same arena)
                    if (total && (hipSetDevice(w.device) != hipSuccess ||
                                  hipMemcpyPeerAsync((char*)ww.gather + gather_at[r] + coded_used, ww.device, grk_amd_coded_device_ptr(w.ctx),
                                                     w.device, total, w.copy) != hipSuccess ||
                                  hipStreamSynchronize(w.copy) != hipSuccess)) { (void)hipGetLastError(); rc = GRK_AMD_ERR_NO_DEVICE; break; }
                } else {
                    if (!pin_ensure(w.ctx, w.pin_coded, w.pin_coded_cap, coded_used + total)) {
                        // (growing: what is there has been consumed by the tile-parts already written)
                        rc = GRK_AMD_ERR_NOMEM; break;
                    }
                    rc = grk_amd_fetch_coded(w.ctx, w.pin_coded + coded_used, total);
                    if (rc) break;
                }
                for (size_t i = 0; i < mine.size(); ++i) {
                    TileJob& j = jobs[mine[i]];
                    j.rows.assign(table.begin() + i * bpt, table.begin() + (i + 1) * bpt);
                    for (auto& row : j.rows) row.offset += coded_used;
                    if (!gather) {         // parallel writers: this tile's tile-part, now, by this thread
                        const int64_t need = grk_amd_write_tile_part(&tp[mine[i]], mine[i], cs_flags, j.rows.data(), w.pin_coded, nullptr, 0);
                        if (need < 0) { rc = (int)need; break; }
                        j.part_at = w.parts.size(); j.part_len = (uint64_t)need;
                        w.parts.resize(w.parts.size() + (size_t)need);
                        const int64_t got = grk_amd_write_tile_part(&tp[mine[i]], mine[i], cs_flags, j.rows.data(), w.pin_coded,
                                                                    w.parts.data() + j.part_at, (uint64_t)need);
                        if (got != need) { rc = got < 0 ? (int)got : GRK_AMD_ERR_INVALID; break; }
                    }
                }
                if (!gather && rc == GRK_AMD_OK) coded_used = 0;      // the group's bytes are in its tile-parts: the buffer is free again
                else coded_used += total;
            }
            used[r] = coded_used;
            rcs[r] = rc;
        });
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < R; ++r)
        if (rcs[r]) { nd->err = std::string("worker ") + std::to_string(r) + ": " + grk_amd_last_error(nd->w[r].ctx); return rcs[r]; }

    if (gather) {
        // the writer: everything to the host in one piece per worker, Tier-2 for all tiles, one codestream
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
        std::vector<grk_amd_coded_block> all;
        for (uint32_t t = 0; t < ntiles; ++t) {
            const uint64_t at = host_at[t % R];
            for (auto row : jobs[t].rows) { row.offset += at; all.push_back(row); }
        }
        return grk_amd_write_codestream_layout(im, base, all.data(), ww.pin_coded, cs_flags, out, cap);
    }
    // parallel writers: main header (TLM from the tile-parts' sizes), the tile-parts in index order, EOC
    std::vector<uint32_t> sizes(ntiles);
    for (uint32_t t = 0; t < ntiles; ++t) sizes[t] = (uint32_t)jobs[t].part_len;
    const int64_t hdr = grk_amd_write_main_header_layout(im, base, cs_flags, sizes.data(), out, cap);
    if (hdr < 0) return hdr;
    uint64_t at = (uint64_t)hdr;
    for (uint32_t t = 0; t < ntiles; ++t) {
        if (at + jobs[t].part_len + 2 > cap) return GRK_AMD_ERR_OVERFLOW;
        std::memcpy(out + at, nd->w[t % R].parts.data() + jobs[t].part_at, (size_t)jobs[t].part_len);
        at += jobs[t].part_len;
    }
    out[at++] = 0xFF; out[at++] = 0xD9;
    return (int64_t)at;
}
