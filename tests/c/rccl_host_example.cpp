// tests/c/rccl_host_example.cpp -- INTEGRATION.md 5c as real code: a native host, ONE PROCESS PER GPU, that drives a sequence of
// tile-sharded frames through libgrok_amd.so's C ABI and moves the coded tile-parts with RCCL.
//
//   rank r owns tile r of every frame (a frame = R tiles of T x T side by side);  per frame f:
//     grk_amd_encode_tiles (asynchronous, pipelined with DEPTH + LAG + 1 buffer sets) -> the rank's blocks in its coded arena
//     ncclAllGather of the bytes used in the arena (8 bytes per rank, device word -> pinned host words, no host sync)
//     LAG frames later: exact-size ncclSend / ncclRecv of arena + block table to the frame's writer, rank f mod R -- DEPTH of
//       these gathers in flight, frame f on communicator and stream f mod DEPTH (ncclCommSplit of the first one; one
//       ncclGroupStart / End per frame), the same algorithm as grok_amd/dist.py's FramePipeline(depth, lag) that bench.py times
//     the encoder's stream waits for the gather of frame f - (DEPTH + LAG + 1) before it takes that frame's buffer set again
//     the writer: tables merged tile by tile -> grk_amd_write_codestream -> the frame's codestream (frames 0 and the last)
//   rank 0 compares frame 0's codestream with the same image coded by ONE context (grk_amd_encode_image): identical bytes.
//
// Rendezvous without MPI: RANK, WORLD_SIZE in the environment, the ncclUniqueId through the file named by NCCL_ID_FILE
// (rank 0 writes it, the others wait for it).  With WORLD_SIZE unset: one rank (what a one-GPU box can run; RCCL refuses two
// ranks on one device).  build: hipcc -std=c++17 rccl_host_example.cpp -I include -L grok_amd/lib -lgrok_amd -lrccl
#include "grok_amd.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { std::fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); return 1; } } while (0)

int main()
{
    const int world = std::getenv("WORLD_SIZE") ? std::atoi(std::getenv("WORLD_SIZE")) : 1;
    const int rank = std::getenv("RANK") ? std::atoi(std::getenv("RANK")) : 0;
    const int dev = std::getenv("LOCAL_RANK") ? std::atoi(std::getenv("LOCAL_RANK")) : rank;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= dev) { std::printf("no device\n"); return 3; }
    CK(hipSetDevice(dev));
    // ---- communicator
    ncclUniqueId id;
    const char* idfile = std::getenv("NCCL_ID_FILE");
    if (rank == 0) {
        CK(ncclGetUniqueId(&id));
        if (world > 1) {
            if (!idfile) { std::fprintf(stderr, "NCCL_ID_FILE not set\n"); return 1; }
            FILE* f = std::fopen((std::string(idfile) + ".tmp").c_str(), "wb");
            if (!f || std::fwrite(&id, sizeof id, 1, f) != 1) return 1;
            std::fclose(f);
            std::rename((std::string(idfile) + ".tmp").c_str(), idfile);
        }
    } else {
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = std::fopen(idfile ? idfile : "", "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (!f || std::fread(&id, sizeof id, 1, f) != 1) { std::fprintf(stderr, "no id file\n"); return 1; }
        std::fclose(f);
    }
    ncclComm_t comm;
    CK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t cs;                                            // the stream the counts travel on (and the waits for the encoder)
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    // one communicator and stream per gather in flight (GATHER_DEPTH, default 2; 1: everything on the first communicator)
    constexpr int kMaxDepth = 4, kLag = 2;
    const int depth = std::max(1, std::min(kMaxDepth, std::getenv("GATHER_DEPTH") ? std::atoi(std::getenv("GATHER_DEPTH")) : 2));
    ncclComm_t gcomm[kMaxDepth]; hipStream_t gs[kMaxDepth];
    for (int g = 0; g < depth; ++g) {
        if (depth == 1) { gcomm[g] = comm; gs[g] = cs; continue; }
        CK(ncclCommSplit(comm, 0, rank, &gcomm[g], nullptr));
        CK(hipStreamCreateWithFlags(&gs[g], hipStreamNonBlocking));
    }

    // ---- the job: frames of R tiles of T x T x 3 8-bit, RCT + 5/3, 4 levels
    const uint32_t T = 1024, R = (uint32_t)world, frames = 11;      // (a multiple of 2^levels x the code-block size: every tile has one geometry)
    grk_amd_tile_params tp;
    std::memset(&tp, 0, sizeof tp);
    tp.tile_w = T; tp.tile_h = T; tp.num_comps = 3; tp.prec = 8; tp.mct = 1; tp.num_levels = 4; tp.cblk_w_exp = 6; tp.cblk_h_exp = 6;
    grk_amd_tile_params mine = tp;
    mine.tile_x0 = (uint32_t)rank * T;                          // tile r lies r tiles to the right of the origin
    grk_amd_ctx* ctx = nullptr;
    CK(grk_amd_create(dev, 0, &ctx));
    hipStream_t es;                                            // the encoder's stream: ours, so that it can wait for a gather's event
    CK(hipStreamCreateWithFlags(&es, hipStreamNonBlocking));
    CK(grk_amd_set_stream(ctx, es));
    const int nsets = depth + kLag + 1;                        // buffer sets in rotation: frame f's set is taken again by frame f + nsets
    CK(grk_amd_set_pipelining(ctx, nsets - 1));
    const int64_t nb = grk_amd_tile_num_blocks(&mine);
    if (nb <= 0) return 1;
    // the image of a frame (every rank makes all of it: the check on rank 0 needs it, the others take their tile out of it)
    const uint32_t W = T * R, H = T;
    std::vector<uint8_t> img((size_t)3 * W * H);
    auto fill = [&](uint32_t f) {
        uint32_t s = 12345u + f;
        for (size_t i = 0; i < img.size(); ++i) { s = s * 1664525u + 1013904223u; img[i] = (uint8_t)(((i % W) + (i / W) % H) / 6 + ((s >> 24) & 7)); }
    };
    const size_t tile_bytes = (size_t)3 * T * T;
    uint8_t* h_tile = (uint8_t*)grk_amd_host_alloc(ctx, tile_bytes);
    constexpr int kTiles = kMaxDepth + kLag + 1;
    void* d_tile[kTiles];
    for (auto& p : d_tile) CK(hipMalloc(&p, tile_bytes));
    // exchange buffers: the counts of the frames whose gather has not been issued yet (kLag + 2 slots)
    constexpr int kCounts = kLag + 2;
    uint64_t* d_counts[kCounts]; uint64_t* h_counts[kCounts]; hipEvent_t ev_counts[kCounts], ev_enc[kCounts];
    for (int k = 0; k < kCounts; ++k) {
        CK(hipMalloc((void**)&d_counts[k], R * 8));
        h_counts[k] = (uint64_t*)grk_amd_host_alloc(ctx, R * 8);
        CK(hipEventCreateWithFlags(&ev_counts[k], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ev_enc[k], hipEventDisableTiming));
    }
    const size_t arena_cap = tile_bytes * 2 + (1u << 20);
    // a writer's receive storage, one set per gather slot (every rank is a writer in turn), and per slot the event behind its last gather
    std::vector<uint8_t*> r_bytes((size_t)R * depth, nullptr); std::vector<uint64_t*> r_off((size_t)R * depth, nullptr); std::vector<uint32_t*> r_len((size_t)R * depth, nullptr);
    for (size_t r = 0; r < (size_t)R * depth; ++r) {
        CK(hipMalloc((void**)&r_bytes[r], arena_cap)); CK(hipMalloc((void**)&r_off[r], (size_t)nb * 8)); CK(hipMalloc((void**)&r_len[r], (size_t)nb * 4));
    }
    std::vector<hipEvent_t> ev_done(frames);                    // frame f's gather has read its source tensors (and filled its storage)
    for (auto& e : ev_done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    struct Pending { uint32_t frame = 0; void* arena = nullptr; void* off = nullptr; void* len = nullptr; };
    std::vector<Pending> pend;
    std::vector<uint8_t> cs0, cs_last;                          // codestreams of frame 0 (writer: rank 0) and of the last frame

    auto gather = [&](const Pending& p) -> int {               // exact sizes to the frame's writer
        const int k = (int)(p.frame % kCounts), g = (int)(p.frame % (uint32_t)depth);
        CK(hipEventSynchronize(ev_counts[k]));                  // the counts of that frame are on the host (queued kLag frames ago)
        const int writer = (int)(p.frame % R);
        // the slot's stream: behind the frame's encode (long finished) and behind the gather that used this receive storage last
        CK(hipStreamWaitEvent(gs[g], ev_enc[k], 0));
        if (p.frame >= (uint32_t)depth) CK(hipStreamWaitEvent(gs[g], ev_done[p.frame - depth], 0));
        uint8_t** rb = &r_bytes[(size_t)g * R]; uint64_t** ro = &r_off[(size_t)g * R]; uint32_t** rl = &r_len[(size_t)g * R];
        CK(ncclGroupStart());
        if (rank == writer) {
            for (uint32_t r = 0; r < R; ++r) {
                if ((int)r == rank) continue;
                CK(ncclRecv(rb[r], h_counts[k][r], ncclUint8, (int)r, gcomm[g], gs[g]));
                CK(ncclRecv(ro[r], (size_t)nb, ncclUint64, (int)r, gcomm[g], gs[g]));
                CK(ncclRecv(rl[r], (size_t)nb, ncclUint32, (int)r, gcomm[g], gs[g]));
            }
        } else {
            CK(ncclSend(p.arena, h_counts[k][rank], ncclUint8, writer, gcomm[g], gs[g]));
            CK(ncclSend(p.off, (size_t)nb, ncclUint64, writer, gcomm[g], gs[g]));
            CK(ncclSend(p.len, (size_t)nb, ncclUint32, writer, gcomm[g], gs[g]));
        }
        CK(ncclGroupEnd());
        CK(hipEventRecord(ev_done[p.frame], gs[g]));
        uint8_t** r_bytes_s = rb; uint64_t** r_off_s = ro; uint32_t** r_len_s = rl;
        hipStream_t cs_s = gs[g];
        std::vector<uint8_t>& cs_out = p.frame == 0 ? cs0 : cs_last;
        if (rank == writer && (p.frame == 0 || p.frame == frames - 1)) {      // Tier-2 over the gathered tables: tile r's rows, then its bytes
            CK(hipStreamSynchronize(cs_s));
            std::vector<grk_amd_coded_block> table((size_t)nb * R);
            std::vector<uint8_t> coded;
            std::vector<uint64_t> off((size_t)nb); std::vector<uint32_t> len((size_t)nb);
            for (uint32_t r = 0; r < R; ++r) {
                const void* so = (int)r == rank ? p.off : (void*)r_off_s[r];
                const void* sl = (int)r == rank ? p.len : (void*)r_len_s[r];
                const void* sb = (int)r == rank ? p.arena : (void*)r_bytes_s[r];
                CK(hipMemcpy(off.data(), so, (size_t)nb * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(len.data(), sl, (size_t)nb * 4, hipMemcpyDeviceToHost));
                const size_t base = coded.size();
                coded.resize(base + h_counts[k][r]);
                CK(hipMemcpy(coded.data() + base, sb, h_counts[k][r], hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < nb; ++i) {
                    table[(size_t)r * nb + i].offset = off[(size_t)i] + base;
                    table[(size_t)r * nb + i].length = len[(size_t)i];
                    table[(size_t)r * nb + i].missing_msbs = 0;
                }
            }
            cs_out.resize(coded.size() + (1u << 20));
            const int64_t n = grk_amd_write_codestream(&tp, W, H, table.data(), coded.data(), cs_out.data(), cs_out.size());
            if (n <= 0) { std::fprintf(stderr, "write_codestream: %lld\n", (long long)n); return 1; }
            cs_out.resize((size_t)n);
        }
        return 0;
    };

    for (uint32_t f = 0; f < frames; ++f) {
        fill(f);
        for (uint32_t c = 0; c < 3; ++c)
            for (uint32_t y = 0; y < T; ++y)
                std::memcpy(h_tile + ((size_t)c * T + y) * T, img.data() + ((size_t)c * H + y) * W + (size_t)rank * T, T);
        // this frame takes the buffer set (and the pixel buffer) of frame f - nsets: its gather must have read them
        if (f >= (uint32_t)nsets) CK(hipStreamWaitEvent(es, ev_done[f - nsets], 0));
        CK(hipMemcpyAsync(d_tile[f % nsets], h_tile, tile_bytes, hipMemcpyHostToDevice, es));
        CK(hipStreamSynchronize(es));                           // (h_tile is filled again for the next frame)
        CK(grk_amd_encode_tiles(ctx, &mine, 1, d_tile[f % nsets], /*on_device*/ 1, nullptr, nullptr));      // returns at once
        const int k = (int)(f % kCounts);
        CK(grk_amd_stream_wait_results(ctx, cs));               // ONE stream waits for the encode as it is queued; the encoder goes on
        CK(hipEventRecord(ev_enc[k], cs));                      // (the gather streams wait for this event, kLag frames later)
        CK(ncclAllGather(grk_amd_table_device_ptr(ctx, 2), d_counts[k], 1, ncclUint64, comm, cs));
        CK(hipMemcpyAsync(h_counts[k], d_counts[k], R * 8, hipMemcpyDeviceToHost, cs));
        CK(hipEventRecord(ev_counts[k], cs));
        while (pend.size() >= (size_t)kLag) {                   // the frame kLag before: its sizes are there by now
            if (gather(pend.front())) return 1;
            pend.erase(pend.begin());
        }
        Pending p; p.frame = f;
        p.arena = grk_amd_coded_device_ptr(ctx); p.off = grk_amd_table_device_ptr(ctx, 0); p.len = grk_amd_table_device_ptr(ctx, 1);
        pend.push_back(p);
    }
    for (const Pending& p : pend) if (gather(p)) return 1;
    for (int g = 0; g < depth; ++g) CK(hipStreamSynchronize(gs[g]));
    CK(hipStreamSynchronize(cs));
    CK(grk_amd_synchronize(ctx));

    int ok = 1;
    if (rank == 0) {                                            // frame 0 again, as ONE context codes the whole image
        fill(0);
        grk_amd_image_layout im = {0, 0, W, H, 0, 0, T, T};
        std::vector<uint8_t> one(img.size() * 2 + (1u << 20));
        grk_amd_ctx* c1 = nullptr;
        CK(grk_amd_create(dev, 0, &c1));
        const int64_t n1 = grk_amd_encode_image(c1, &im, &tp, img.data(), 0, one.data(), one.size());
        grk_amd_destroy(c1);
        ok = n1 > 0 && (size_t)n1 == cs0.size() && std::memcmp(one.data(), cs0.data(), cs0.size()) == 0;
        std::printf("ranks %d  frames %u  gathers in flight %d (lag %d, %d buffer sets)  frame 0: gathered codestream %zu bytes, one context %lld bytes: %s\n",
                    world, frames, depth, kLag, nsets, cs0.size(), (long long)n1, ok ? "identical" : "DIFFERENT");
    }
    if ((int)((frames - 1) % R) == rank) {                      // the LAST frame on its writer: the rotation's late frames against one context too
        fill(frames - 1);
        grk_amd_image_layout im = {0, 0, W, H, 0, 0, T, T};
        std::vector<uint8_t> one(img.size() * 2 + (1u << 20));
        grk_amd_ctx* c1 = nullptr;
        CK(grk_amd_create(dev, 0, &c1));
        const int64_t n1 = grk_amd_encode_image(c1, &im, &tp, img.data(), 0, one.data(), one.size());
        grk_amd_destroy(c1);
        const int okl = n1 > 0 && (size_t)n1 == cs_last.size() && std::memcmp(one.data(), cs_last.data(), cs_last.size()) == 0;
        std::printf("rank %d: frame %u (the last): gathered codestream %zu bytes: %s\n", rank, frames - 1, cs_last.size(), okl ? "identical" : "DIFFERENT");
        ok = ok && okl;
    }
    for (auto& p : d_tile) (void)hipFree(p);
    grk_amd_host_free(ctx, h_tile);
    grk_amd_destroy(ctx);
    (void)ncclCommDestroy(comm);
    return ok ? 0 : 2;
}
