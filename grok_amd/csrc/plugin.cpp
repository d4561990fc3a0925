// grok_amd/csrc/plugin.cpp -- libgrokj2k_plugin.so: the Grok plugin entry points on top of the
// C-ABI of libgrok_amd.so.  Replaces the in-tree do-nothing stub (src/lib/jp2_plugin/Plugin.cpp)
// and follows the protocol of SURVEY.md §3.3 / §8(b):
//
//   host: grk_initialize(dir) -> dlopen -> minpf_post_load_plugin()      registration
//         grk_plugin_init({deviceId, verbose}) -> plugin_init()          one context per process
//         grk_plugin_compress(params, cb) -> plugin_encode()             read image, run the hot
//             path on the MI355X, build the grk_plugin_tile tree, call the host's callback, which
//             runs grk_compress_with_plugin(codec, tile) (rate control 1, single tile: D2, D3)
//   any request outside the hot path (non-HT, multi-tile, unsupported file type, ...) returns a
//   non-zero code, which the host treats as "not handled" and takes its CPU path
//   (src/bin/jp2/grk_compress.cpp:2125-2168).
//
// The plugin owns the coded bytes the tile tree points at; they stay alive until the callback
// returns (the host aliases them: plugin_bridge.cpp:198-201).
#include "../../include/grk_plugin_abi.h"
#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <dlfcn.h>

namespace {

grk_amd_ctx* g_ctx = nullptr;          // the device Grok named (grk_plugin_init_info.deviceId): the single-file entry points
bool g_verbose = false;
// Every GPU the plugin may use: g_ctx first, then the node's other GPUs (GRK_AMD_PLUGIN_DEVICES=0,2,5 names them instead; an
// entry may repeat -- two contexts on one GPU).  Grok's batch protocol (grk_compress.cpp:2024-2050: a directory of images,
// each its own codestream) is where a node's GPUs run side by side with no exchange at all: file i goes to whichever device
// is free.
struct Dev { grk_amd_ctx* ctx = nullptr; std::mutex* mu = nullptr; std::mutex own; };
std::vector<std::unique_ptr<Dev>> g_devs;
// Self-check mode (grok.h:1719-1739, GRK_PLUGIN_STATE_DEBUG; set by the environment, GRK_AMD_PLUGIN_DEBUG=1, at
// plugin_init): the host skips its DC shift, MCT and DWT, runs its own Tier-1 over the coefficients the plugin hands it
// as "image" data, and compares every code-block (bytes, rates, pass counts, bounding boxes, step sizes) with the plugin's
// (TileProcessor.cpp:667-690, plugin_bridge.cpp:138-252).  A free parity check by the reference itself.
uint32_t g_debug_state = GRA_PLUGIN_STATE_NO_DEBUG;
std::mutex g_mu;

// ---- the tile tree ------------------------------------------------------------------------------
// 49 152 blocks of an 8K frame are 82 MB of gra_plugin_code_block: allocating, zeroing and filling that per frame (and as much
// again for the coded bytes) cost more than the transfers.  A tree is therefore built once per geometry and kept (a few per
// geometry: the batch pipeline holds up to three tiles at a time); a frame patches the three per-block fields that change.
// The coded bytes live in pinned memory (grk_amd_host_alloc): the download is one DMA at the link's rate.
struct TileOwner {
    gra_plugin_tile tile{};
    grk_amd_tile_params params{};
    std::vector<gra_plugin_tile_component> comps;   std::vector<gra_plugin_tile_component*> comp_ptr;
    std::vector<gra_plugin_resolution> ress;        std::vector<gra_plugin_resolution*> res_ptr;
    std::vector<gra_plugin_band> bands;             std::vector<gra_plugin_band*> band_ptr;
    std::vector<gra_plugin_precinct> precs;         std::vector<gra_plugin_precinct*> prec_ptr;
    std::vector<gra_plugin_code_block> blocks;      std::vector<gra_plugin_code_block*> block_ptr;
    std::vector<grk_amd_coded_block> table;
    std::vector<float> band_steps;                   // the bands' step sizes as make_owner set them (a decode lets the host overwrite them)
    bool served_decode = false;                      // the coded buffer was sized for a decode (every block's worst case + the file)
    bool no_cache = false;                           // per-component geometry (sub-sampled components): not the tree its parameters name
    uint8_t* coded = nullptr; size_t coded_cap = 0; bool coded_pinned = false;
    ~TileOwner() { free_coded(); }
    void free_coded()
    {
        if (coded) { if (coded_pinned) grk_amd_host_free(nullptr, coded); else std::free(coded); }
        coded = nullptr; coded_cap = 0;
    }
    bool ensure_coded(grk_amd_ctx* ctx, size_t n)
    {
        if (n <= coded_cap) return true;
        free_coded();
        const size_t want = n + (n >> 3) + 4096;
        coded = static_cast<uint8_t*>(grk_amd_host_alloc(ctx, want));
        coded_pinned = coded != nullptr;
        if (!coded) coded = static_cast<uint8_t*>(std::malloc(want));
        if (!coded) return false;
        coded_cap = want;
        return true;
    }
};
static_assert(offsetof(TileOwner, tile) == 0, "tile must be the first member: destroy() casts back");

std::mutex g_cache_mu;
std::vector<TileOwner*> g_tile_cache;           // trees not in use, any geometry
constexpr size_t kTileCacheMax = 8;
constexpr size_t kTileCacheBytes = 512u << 20;   // pinned coded buffers the kept trees may hold together

// the geometry-dependent part of the tree: everything but compressedData / compressedDataLength / passes[0] of the blocks
// comp_params: sub-sampled components -- the rectangle of every component of the tile (grk_amd_layout_tile_comp, num_comps = 1), each
// with its own block layout and precinct counts; nullptr: every component has p's
TileOwner* make_owner(const grk_amd_tile_params& p, const std::vector<grk_amd_tile_params>* comp_params = nullptr)
{
    std::vector<grk_amd_block> layout;
    const uint32_t nres = p.num_levels + 1u;
    std::vector<std::vector<uint32_t>> nprec_c(p.num_comps, std::vector<uint32_t>((size_t)nres, 1u));
    if (!comp_params) {
        const int64_t nbl = grk_amd_tile_num_blocks(&p);
        if (nbl <= 0) return nullptr;
        layout.resize((size_t)nbl);
        if (grk_amd_tile_layout(&p, layout.data(), (uint64_t)nbl, nullptr) != nbl) return nullptr;
        for (auto& v : nprec_c) (void)grk_amd_tile_precincts(&p, v.data());
    } else {
        if (comp_params->size() != p.num_comps) return nullptr;
        for (uint32_t c = 0; c < p.num_comps; ++c) {
            const grk_amd_tile_params& pc = (*comp_params)[c];
            const int64_t nbl = grk_amd_tile_num_blocks(&pc);
            if (nbl <= 0) return nullptr;
            const size_t at = layout.size();
            layout.resize(at + (size_t)nbl);
            if (grk_amd_tile_layout(&pc, layout.data() + at, (uint64_t)nbl, nullptr) != nbl) return nullptr;
            for (size_t i = at; i < layout.size(); ++i) layout[i].comp = (uint16_t)c;
            (void)grk_amd_tile_precincts(&pc, nprec_c[c].data());
        }
    }
    auto* o = new TileOwner();
    o->params = p;
    const size_t nb = layout.size();
    o->table.resize(nb);
    const size_t nbands_c = 3 * p.num_levels + 1;
    o->comps.resize(p.num_comps); o->comp_ptr.resize(p.num_comps);
    o->ress.resize((size_t)p.num_comps * nres); o->res_ptr.resize(o->ress.size());
    o->bands.resize((size_t)p.num_comps * nbands_c); o->band_ptr.resize(o->bands.size());
    // precincts per band of every resolution (the same for its three bands)
    size_t total_prec = 0;
    for (auto& v : nprec_c)
        for (uint32_t r = 0; r < nres; ++r) {
            v[r] = std::max(v[r], 1u);                  // (a resolution without samples: the host's tree has none either,
                                                        //  one empty entry keeps the arrays well-formed)
            total_prec += (size_t)v[r] * (r ? 3 : 1);
        }
    o->precs.resize(total_prec); o->prec_ptr.resize(o->precs.size());
    o->blocks.resize(nb); o->block_ptr.resize(nb);     // (value-initialised: zeros)
    for (size_t i = 0; i < nb; ++i) {
        const grk_amd_block& b = layout[i];
        gra_plugin_code_block& cb = o->blocks[i];
        cb.x0 = b.x0; cb.y0 = b.y0; cb.x1 = b.x1; cb.y1 = b.y1;
        cb.numPix = (b.x1 - b.x0) * (b.y1 - b.y0);
        cb.numBitPlanes = 1;                     // T1HT::compress sets cblk->numbps = 1 (T1HT.cpp:123)
        cb.numPasses = 1;
        cb.passes[0].distortionDecrease = 0.0;
        o->block_ptr[i] = &cb;
    }
    size_t bi = 0, blk = 0, pk = 0;
    for (uint32_t c = 0; c < p.num_comps; ++c) {
        const std::vector<uint32_t>& nprec = nprec_c[c];
        gra_plugin_tile_component& tc = o->comps[c];
        tc.numResolutions = nres;
        tc.resolutions = &o->res_ptr[(size_t)c * nres];
        o->comp_ptr[c] = &tc;
        for (uint32_t r = 0; r < nres; ++r) {
            gra_plugin_resolution& R = o->ress[(size_t)c * nres + r];
            o->res_ptr[(size_t)c * nres + r] = &R;
            R.level = r;
            R.numBands = r ? 3 : 1;
            R.band = &o->band_ptr[bi];
            for (size_t k = 0; k < R.numBands; ++k, ++bi) {
                gra_plugin_band& B = o->bands[bi];
                o->band_ptr[bi] = &B;
                const uint8_t orient = (uint8_t)(r ? k + 1 : 0);
                B.orientation = orient;
                B.numPrecincts = nprec[r];
                B.precincts = &o->prec_ptr[pk];
                // the blocks of a band are contiguous in enumeration order, precinct by precinct
                const size_t band_first = blk;
                for (uint32_t q = 0; q < nprec[r]; ++q, ++pk) {
                    o->prec_ptr[pk] = &o->precs[pk];
                    const size_t first = blk;
                    while (blk < nb && layout[blk].comp == c && layout[blk].res == r && layout[blk].band == orient && layout[blk].precinct == q) ++blk;
                    o->precs[pk].numBlocks = blk - first;
                    o->precs[pk].blocks = first < nb ? &o->block_ptr[first] : nullptr;
                }
                B.stepsize = blk > band_first ? layout[band_first].stepsize : 1.0f;
            }
        }
    }
    o->tile.decompress_flags = 0;
    o->tile.numComponents = p.num_comps;
    o->tile.tileComponents = o->comp_ptr.data();
    o->band_steps.resize(o->bands.size());
    for (size_t i = 0; i < o->bands.size(); ++i) o->band_steps[i] = o->bands[i].stepsize;
    return o;
}

// what a frame changes: where each block's bytes are and how many
void patch_owner(TileOwner* o)
{
    const size_t nb = o->blocks.size();
    // (a tree that served a decode comes back with the step sizes the host wrote, plugin_bridge.cpp:40)
    for (size_t i = 0; i < o->bands.size() && i < o->band_steps.size(); ++i) o->bands[i].stepsize = o->band_steps[i];
    o->tile.decompress_flags = 0;
    for (size_t i = 0; i < nb; ++i) {
        gra_plugin_code_block& cb = o->blocks[i];
        const uint32_t len = o->table[i].length;
        cb.compressedData = o->coded + o->table[i].offset;
        cb.compressedDataLength = len;
        cb.numBitPlanes = 1;                            // (a tree that served a decode comes back with the host's values)
        cb.numPasses = 1;
        cb.passes[0].rate = len ? len - 1 : 0;          // host uses rate + 1 (plugin_bridge.cpp:230)
        cb.passes[0].length = len;
        cb.passes[0].distortionDecrease = 0.0;          // (grk_amd_plugin_tile_fill_distortion: only when the host makes layers)
    }
}

TileOwner* acquire_owner(const grk_amd_tile_params& p)
{
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_tile_cache.size(); ++i)
            if (std::memcmp(&g_tile_cache[i]->params, &p, sizeof p) == 0) {
                TileOwner* o = g_tile_cache[i];
                g_tile_cache.erase(g_tile_cache.begin() + (long)i);
                return o;
            }
    }
    return make_owner(p);
}

void release_owner(TileOwner* o)
{
    if (!o) return;
    if (o->no_cache) { delete o; return; }
    // a decode's buffer (16 KB per block + the file: ~0.8 GB pinned for an 8K frame) does not stay with the kept tree; an encode's
    // (the coded bytes of a frame) does, within a budget over the whole cache
    if (o->served_decode || o->coded_cap > kTileCacheBytes) { o->free_coded(); o->served_decode = false; }
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        size_t held = o->coded_cap;
        for (TileOwner* t : g_tile_cache) held += t->coded_cap;
        for (size_t i = 0; held > kTileCacheBytes && i < g_tile_cache.size(); ++i) {          // oldest first
            held -= g_tile_cache[i]->coded_cap;
            g_tile_cache[i]->free_coded();
        }
        if (g_tile_cache.size() < kTileCacheMax) { g_tile_cache.push_back(o); return; }
        // full: the oldest goes (another geometry has taken over)
        TileOwner* old = g_tile_cache.front();
        g_tile_cache.erase(g_tile_cache.begin());
        g_tile_cache.push_back(o);
        o = old;
    }
    delete o;
}

void drop_tile_cache()
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (TileOwner* o : g_tile_cache) delete o;
    g_tile_cache.clear();
}

// pixels of an image the plugin loads itself: pinned when a context exists (the upload is then one DMA at the link's rate)
// (the batch reader takes one per file: the pinned ones are recycled through a two-slot pool -- hipHostMalloc / hipHostFree per
//  file cost milliseconds each, and the free waits for the device while the device lock is held)
struct PinnedSlot { uint8_t* p = nullptr; size_t cap = 0; };
std::mutex g_pin_mu;
PinnedSlot g_pin_pool[2];
void drop_pinned_pool()
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto& sl : g_pin_pool) { if (sl.p) grk_amd_host_free(nullptr, sl.p); sl = PinnedSlot{}; }
}
struct HostPixels {
    uint8_t* p = nullptr; size_t n = 0, cap = 0; bool pinned = false;
    HostPixels() = default;
    HostPixels(const HostPixels&) = delete;
    HostPixels& operator=(const HostPixels&) = delete;
    ~HostPixels() { reset(); }
    void reset()
    {
        if (p && pinned) {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            PinnedSlot* sl = !g_pin_pool[0].p ? &g_pin_pool[0] : !g_pin_pool[1].p ? &g_pin_pool[1]
                             : (g_pin_pool[0].cap <= g_pin_pool[1].cap ? &g_pin_pool[0] : &g_pin_pool[1]);
            if (!sl->p) { sl->p = p; sl->cap = cap; p = nullptr; }
            else if (sl->cap < cap) { std::swap(sl->p, p); std::swap(sl->cap, cap); }      // keep the larger, free the smaller below
        }
        if (p) { if (pinned) grk_amd_host_free(nullptr, p); else std::free(p); }
        p = nullptr; n = 0; cap = 0;
    }
    bool alloc(size_t bytes)
    {
        reset();
        if (g_ctx) {
            std::lock_guard<std::mutex> lk(g_pin_mu);
            int best = -1;                                                                 // the smallest kept buffer that fits
            for (int i = 0; i < 2; ++i)
                if (g_pin_pool[i].p && g_pin_pool[i].cap >= bytes && (best < 0 || g_pin_pool[i].cap < g_pin_pool[best].cap)) best = i;
            if (best >= 0) { p = g_pin_pool[best].p; cap = g_pin_pool[best].cap; g_pin_pool[best] = PinnedSlot{}; }
        }
        if (p) { pinned = true; n = bytes; return true; }
        p = g_ctx ? static_cast<uint8_t*>(grk_amd_host_alloc(g_ctx, bytes)) : nullptr;
        pinned = p != nullptr;
        if (!p) p = static_cast<uint8_t*>(std::malloc(bytes ? bytes : 1));
        n = p ? bytes : 0; cap = n;
        return p != nullptr;
    }
    uint8_t* data() const { return p; }
    size_t size() const { return n; }
};

// ---- minimal PNM (P5/P6, binary) reader: enough for plugin_encode's "read params->infile" -----------
bool read_pnm(const char* path, HostPixels& planar, uint32_t& w, uint32_t& h, uint32_t& comps, uint32_t& prec)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    auto token = [&](char* buf, size_t n) -> bool {
        int ch;
        for (;;) {
            ch = std::fgetc(f);
            if (ch == '#') { while ((ch = std::fgetc(f)) != EOF && ch != '\n') {} continue; }
            if (ch == EOF) return false;
            if (ch > ' ') break;
        }
        size_t i = 0;
        while (ch != EOF && ch > ' ' && i + 1 < n) { buf[i++] = (char)ch; ch = std::fgetc(f); }
        buf[i] = 0;
        return i > 0;
    };
    char t[32];
    bool ok = token(t, sizeof t) && t[0] == 'P' && (t[1] == '5' || t[1] == '6') && t[2] == 0;
    comps = ok && t[1] == '6' ? 3 : 1;
    unsigned long maxv = 0;
    ok = ok && token(t, sizeof t) && (w = (uint32_t)std::strtoul(t, nullptr, 10)) > 0;
    ok = ok && token(t, sizeof t) && (h = (uint32_t)std::strtoul(t, nullptr, 10)) > 0;
    ok = ok && token(t, sizeof t) && (maxv = std::strtoul(t, nullptr, 10)) > 0 && maxv < 65536;
    if (!ok) { std::fclose(f); return false; }
    prec = 1; while ((1ul << prec) <= maxv) ++prec;
    const size_t bps = prec > 8 ? 2 : 1, n = (size_t)w * h;
    std::vector<uint8_t> raw(n * comps * bps);
    ok = std::fread(raw.data(), 1, raw.size(), f) == raw.size();
    std::fclose(f);
    if (!ok) return false;
    if (!planar.alloc(raw.size())) return false;
    uint8_t* const dst = planar.data();
    for (uint32_t c = 0; c < comps; ++c)
        for (size_t i = 0; i < n; ++i) {
            if (bps == 1) dst[c * n + i] = raw[i * comps + c];
            else {   // PNM 16-bit is big endian; the tile buffer is host endian
                const uint8_t* s = &raw[(i * comps + c) * 2];
                reinterpret_cast<uint16_t*>(dst)[c * n + i] = (uint16_t)((s[0] << 8) | s[1]);
            }
        }
    return true;
}

// does the tile grid cell anchored at (tx0, ty0) cover the image area, which starts at (image_offset_x0, image_offset_y0)
// (grk_compress -d, stored as grk_image x0 / y0 by the host's image readers)?  With sub-sampled components (grk_compress -s dx,dy:
// every component of a PNM alike) the area on the reference grid is (w - 1) dx + 1 wide (image_format/PNMFormat.cpp:388-392).
bool single_tile(const gra_cparameters* cp, uint32_t w, uint32_t h)
{
    const uint64_t ox = cp->image_offset_x0, oy = cp->image_offset_y0;
    const uint64_t gw = (uint64_t)(w - 1) * cp->subsampling_dx + 1, gh = (uint64_t)(h - 1) * cp->subsampling_dy + 1;
    return !cp->tile_size_on || !(cp->tx0 > ox || cp->ty0 > oy || (uint64_t)cp->tx0 + cp->t_width < ox + gw ||
                                  (uint64_t)cp->ty0 + cp->t_height < oy + gh);
}

// does the host make quality layers from the passes' rates and distortions (TileProcessor::needs_rate_control,
// tile/TileProcessor.cpp:75-90; grk_compress -r / -q set cp_disto_alloc / cp_fixed_quality and one entry per layer)?
bool wants_rate_control(const gra_cparameters* cp)
{
    for (uint32_t l = 0; l < std::min<uint32_t>(std::max<uint32_t>(cp->tcp_numlayers, 1u), 100u); ++l)
        if ((cp->cp_disto_alloc && cp->tcp_rates[l] > 0.0) || (cp->cp_fixed_quality && cp->tcp_distoratio[l] > 0.0)) return true;
    return cp->tcp_numlayers > 1;
}

// multi = false: the parameters of THE tile of a single-tile image (what the plugin protocol can carry, D3);
// multi = true: the base parameters of an image of several tiles (tile size / origin filled per tile by grk_amd_layout_tile)
bool params_from_cparameters(const gra_cparameters* cp, uint32_t w, uint32_t h, uint32_t comps, uint32_t prec,
                             grk_amd_tile_params& p, bool multi = false)
{
    if (!cp->isHT || !(cp->cblk_sty & GRA_CBLKSTY_HT)) return false;             // hot path = HTJ2K only
    if (!multi && !single_tile(cp, w, h)) return false;
    if (cp->numpocs || cp->roi_compno >= 0) return false;
    // Quality layers / rate targets: the HOST forms the layers (its Tier-2, from the rates and the distortion decreases the tile tree
    // carries: gpu_step fills them); this library's own writer -- the route an image of several tiles takes -- writes one layer
    if (multi && wants_rate_control(cp)) return false;
    // Sub-sampled components, every component alike (all a PNM can carry): component c of the tile is [ceil(x0 / dx), ceil(x1 / dx))
    // (tile/TileProcessor.cpp:605-612) -- w x h samples whose origin is ceil(offset / d); the sub-sampling itself is the host's SIZ.
    // The several-tiles route writes SIZ itself, with XRsiz = YRsiz = 1: declined there
    if (cp->subsampling_dx < 1 || cp->subsampling_dy < 1 || cp->subsampling_dx > 255 || cp->subsampling_dy > 255) return false;
    if (multi && (cp->subsampling_dx != 1 || cp->subsampling_dy != 1)) return false;
    // (an offset that is not a multiple of the factor: the component the host derives, ceil(x1 / dx) - ceil(x0 / dx), is a column
    //  short of the file's -- the host's own business)
    if (cp->image_offset_x0 % cp->subsampling_dx || cp->image_offset_y0 % cp->subsampling_dy) return false;
    if (cp->numresolution < 1 || cp->numresolution > GRK_AMD_MAX_LEVELS + 1) return false;
    auto lg = [](uint32_t v) { int e = 0; while ((1u << e) < v) ++e; return e; };
    std::memset(&p, 0, sizeof p);
    p.tile_w = w; p.tile_h = h; p.num_comps = (uint16_t)comps; p.prec = (uint8_t)prec; p.sgnd = 0;
    // the tile = the image area, wherever it lies; in the component's own coordinates
    p.tile_x0 = (cp->image_offset_x0 + cp->subsampling_dx - 1) / cp->subsampling_dx;
    p.tile_y0 = (cp->image_offset_y0 + cp->subsampling_dy - 1) / cp->subsampling_dy;
    p.irreversible = cp->irreversible ? 1 : 0;
    // tcp_mct as grk_compress leaves it: 255 = "not set" (the library then applies RCT/ICT to >= 3 components,
    // CodeStreamCompress.cpp:345-352), 0 / 1 as given, 2 = custom array MCT (mct_data) -- outside the hot path
    if (cp->tcp_mct == 2 || cp->mct_data) return false;
    p.mct = cp->tcp_mct == 255 ? (comps >= 3 ? 1 : 0) : (cp->tcp_mct ? 1 : 0);
    p.num_levels = (uint8_t)(cp->numresolution - 1);
    p.cblk_w_exp = (uint8_t)lg(cp->cblockw_init ? cp->cblockw_init : 64);
    p.cblk_h_exp = (uint8_t)lg(cp->cblockh_init ? cp->cblockh_init : 64);
    // precincts (grk_compress -c): sizes from the highest resolution down, the last one halved for the resolutions beyond
    // the list, exponent = floor(log2), at least 1 -- CodeStreamCompress.cpp:475-514
    if ((cp->csty & 1u) && cp->res_spec) {
        auto fl = [](uint32_t v) { uint32_t e = 0; while (v >>= 1) ++e; return e; };
        const uint32_t rs = std::min<uint32_t>(cp->res_spec, GRA_J2K_MAXRLVLS);
        for (uint32_t q = 0; q < cp->numresolution; ++q) {
            const uint32_t r = cp->numresolution - 1 - q;
            const uint32_t pw = q < rs ? cp->prcw_init[q] : cp->prcw_init[rs - 1] >> (q - (rs - 1));
            const uint32_t ph = q < rs ? cp->prch_init[q] : cp->prch_init[rs - 1] >> (q - (rs - 1));
            const uint32_t ex = pw < 1 ? 1 : fl(pw), ey = ph < 1 ? 1 : fl(ph);
            if (ex > 15 || ey > 15) return false;
            // (a precinct of ONE sample: exponent byte 0, which grk_amd_tile_params reads as "the default 15 / 15" -- the host
            //  would cut 1 x 1 precincts where we built one; its CPU path takes such a list, as on the decode side)
            if ((ex | (ey << 4)) == 0) return false;
            p.precinct_exp[r] = (uint8_t)(ex | (ey << 4));
        }
    }
    if (multi) { p.tile_w = std::min(w, cp->t_width); p.tile_h = std::min(h, cp->t_height); p.tile_x0 = p.tile_y0 = 0; }
    return grk_amd_tile_num_blocks(&p) > 0;
}

// One file through the protocol in three steps, so that the batch mode can overlap them across files:
//   load      read + de-interleave the PNM                                                  (disk, host)
//   gpu_step  H2D, encode, D2H, the grk_plugin_tile tree (+ the self-check image)            (GPU; holds g_mu)
//   host_step the host's callback: its own Tier-2 over our blocks, the file                  (host library)
struct EncodeJob {
    std::string in, out;
    HostPixels px;
    uint32_t w = 0, h = 0, comps = 0, prec = 0;
    grk_amd_tile_params p{};
    gra_plugin_tile* tile = nullptr;
    void* dbg_image = nullptr;
    void (*unref)(void*) = nullptr;
};

bool load_step(EncodeJob& j) { return read_pnm(j.in.c_str(), j.px, j.w, j.h, j.comps, j.prec); }

bool gpu_step(gra_cparameters* cp, EncodeJob& j, Dev* dev = nullptr)
{
    if (!params_from_cparameters(cp, j.w, j.h, j.comps, j.prec, j.p)) return false;
    grk_amd_ctx* const ctx = dev ? dev->ctx : g_ctx;
    std::lock_guard<std::mutex> lk(dev ? *dev->mu : g_mu);
    j.tile = grk_amd_plugin_tile_create(ctx, &j.p, j.px.data(), 0);
    if (!j.tile) return false;
    if (wants_rate_control(cp) && grk_amd_plugin_tile_fill_distortion(ctx, j.tile) != GRK_AMD_OK) {
        grk_amd_plugin_tile_destroy(j.tile); j.tile = nullptr;
        return false;
    }
    j.px.reset();                               // the pixels are on the device / coded: the host callback loads its own copy
    // self-check mode: the "image" the host gets holds our sub-band coefficients (it skips its own DC shift / MCT / DWT and
    // codes them with its own Tier-1).  The image object is made by the host library itself (grk_image_new, resolved from
    // the process we were loaded into) so that the host can treat it as any other.
    if (g_debug_state & GRA_PLUGIN_STATE_DEBUG) {
        typedef gra_image* (*image_new_fn)(uint16_t, gra_image_cmptparm*, int32_t, bool);
        auto image_new = reinterpret_cast<image_new_fn>(dlsym(RTLD_DEFAULT, "grk_image_new"));
        j.unref = reinterpret_cast<void (*)(void*)>(dlsym(RTLD_DEFAULT, "grk_object_unref"));
        const grk_amd_tile_params& p = j.p;
        auto fail = [&]() { grk_amd_plugin_tile_destroy(j.tile); j.tile = nullptr; return false; };
        if (!image_new || !j.unref) return fail();
        std::vector<gra_image_cmptparm> cps(j.comps);
        for (auto& c : cps) { c.dx = cp->subsampling_dx; c.dy = cp->subsampling_dy; c.w = j.w; c.stride = 0; c.h = j.h; c.x0 = p.tile_x0; c.y0 = p.tile_y0; c.prec = (uint8_t)j.prec; c.sgnd = false; }
        gra_image* img = image_new((uint16_t)j.comps, cps.data(), j.comps >= 3 ? 1 /* GRK_CLRSPC_SRGB */ : 2 /* GRK_CLRSPC_GRAY */, true);
        if (!img) return fail();
        img->x0 = cp->image_offset_x0; img->y0 = cp->image_offset_y0;
        img->x1 = img->x0 + (j.w - 1) * cp->subsampling_dx + 1; img->y1 = img->y0 + (j.h - 1) * cp->subsampling_dy + 1;
        bool ok = true;
        for (uint32_t c = 0; c < j.comps && ok; ++c)
            ok = img->comps[c].data && grk_amd_fetch_coefficients(ctx, c, img->comps[c].data, img->comps[c].stride) == GRK_AMD_OK;
        if (!ok) { j.unref(&img->obj); return fail(); }
        j.dbg_image = img;
    }
    return true;
}

int32_t host_step(gra_cparameters* cp, EncodeJob& j, gra_encode_callback cb)
{
    gra_encode_callback_info info{};
    info.input_file_name = j.in.c_str();
    info.outputFileNameIsRelative = false;
    info.output_file_name = j.out.c_str();
    info.compressor_parameters = cp;
    info.image = static_cast<gra_image*>(j.dbg_image);     // nullptr: the host callback loads the image itself (grk_compress.cpp:1636)
    info.tile = j.tile;
    info.error_code = 0;
    cb(&info);
    if (j.dbg_image) j.unref(&static_cast<gra_image*>(j.dbg_image)->obj);
    grk_amd_plugin_tile_destroy(j.tile);
    j.tile = nullptr; j.dbg_image = nullptr;
    return info.error_code;
}

// An image of SEVERAL tiles.  The plugin protocol attaches one grk_plugin_tile to every tile of an image (D3), so the host
// cannot be handed the blocks tile by tile; but the whole file is within reach: every tile through the hot path
// (grk_amd_encode_image: tiles grouped by geometry, one batch per group) and the codestream through our own Tier-2 writer,
// which writes what the reference writes byte for byte (SIZ / COD / QCD / TLM / PLT / SOP / EPH, the progression orders,
// precincts).  Only raw codestreams (.j2k / .j2c / .jpc): the JP2 boxes stay with the host.  Returns 0 (handled: the file
// is written, the host's callback is not needed) or -1 (the host takes its CPU path).
int32_t encode_multi_tile(gra_cparameters* cp, EncodeJob& j, Dev* dev = nullptr)
{
    if (g_debug_state & GRA_PLUGIN_STATE_DEBUG) return -1;
    const size_t dot = j.out.rfind('.');
    if (dot == std::string::npos) return -1;
    std::string ext = j.out.substr(dot + 1);
    for (auto& ch : ext) ch = (char)std::tolower((unsigned char)ch);
    if (ext != "j2k" && ext != "j2c" && ext != "jpc") return -1;
    grk_amd_tile_params base;
    if (!params_from_cparameters(cp, j.w, j.h, j.comps, j.prec, base, true)) return -1;
    if (cp->prog_order < 0 || cp->prog_order > 4 || cp->cp_num_comments) return -1;
    // what this writer does not write the host's way stays with the host (its CPU path): tile-part division (grk_compress -u),
    // profiles / extensions (-Z ...: rsiz beyond the JPH flag the library sets for HT itself), size caps, rate or quality targets
    if (cp->tp_on || (cp->rsiz & ~0x4000u) || cp->max_cs_size || cp->max_comp_size) return -1;
    for (uint32_t l = 0; l < std::max<uint32_t>(cp->tcp_numlayers, 1u); ++l)
        if (cp->tcp_rates[l] != 0.0 || cp->tcp_distoratio[l] != 0.0) return -1;
    grk_amd_image_layout im{cp->image_offset_x0, cp->image_offset_y0, cp->image_offset_x0 + j.w, cp->image_offset_y0 + j.h,
                            cp->tx0, cp->ty0, cp->t_width, cp->t_height};
    const uint32_t flags = (cp->writeTLM ? GRK_AMD_CS_TLM : 0u) | (cp->writePLT ? GRK_AMD_CS_PLT : 0u) |
                           ((cp->csty & 2u) ? GRK_AMD_CS_SOP : 0u) | ((cp->csty & 4u) ? GRK_AMD_CS_EPH : 0u) |
                           GRK_AMD_CS_PROG((uint32_t)cp->prog_order);
    std::vector<uint8_t> out(j.px.size() * 2 + (1u << 20));
    int64_t n;
    {
        std::lock_guard<std::mutex> lk(dev ? *dev->mu : g_mu);
        n = grk_amd_encode_image(dev ? dev->ctx : g_ctx, &im, &base, j.px.data(), flags, out.data(), out.size());
    }
    if (n <= 0) return -1;
    FILE* f = std::fopen(j.out.c_str(), "wb");
    if (!f) return -1;
    const bool ok = std::fwrite(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok ? 0 : -1;
}

int32_t encode_file(gra_cparameters* cp, const char* in, const char* out, gra_encode_callback cb)
{
    if (!g_ctx || !cp || !cb || !in || !out) return -1;
    EncodeJob j;
    j.in = in; j.out = out;
    if (!load_step(j)) return -1;
    if (!single_tile(cp, j.w, j.h)) return encode_multi_tile(cp, j);
    if (!gpu_step(cp, j)) return -1;
    return host_step(cp, j, cb);
}

// ---- batch mode: a worker thread walks the input directory ----------------------------------------
std::thread g_batch;
std::atomic<bool> g_batch_done{true}, g_batch_stop{false};

// ---- the stream's own main header: QCD (guard bits, exponents) and the file size ---------------------------------
// The host hands a plugin every block's numbps but not the band's (plugin_bridge.cpp:63-76), and the HT decoder needs
// their difference (missing_msbs).  Grok's own HT streams carry the exponents of HTParams.cpp:248-312 with one guard bit
// (D4 included), which is what grk_amd_tile_layout models; another encoder's stream need not.  So the band numbps come
// from the codestream itself: the file `grk_decompress -i` names in parameters->infile.
struct StreamHeader {
    uint64_t file_size = 0;
    uint32_t guard_bits = 0, qstyle = 0;
    std::vector<uint16_t> words;          // SPqcd values in band order (8-bit expn << 3 for style 0)
    bool overrides = false;               // QCC / COC / RGN / POC in the main header: per-component deviations
};
bool read_stream_header(const char* path, StreamHeader& h)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::vector<uint8_t> b(1u << 20);
    b.resize(std::fread(b.data(), 1, b.size(), f));
    bool ok = std::fseek(f, 0, SEEK_END) == 0;
    const long sz = std::ftell(f);
    std::fclose(f);
    if (!ok || sz <= 0) return false;
    h.file_size = (uint64_t)sz;
    size_t at = 0;
    auto be16 = [&](size_t i) { return (uint32_t)(b[i] << 8 | b[i + 1]); };
    auto be32 = [&](size_t i) { return (uint32_t)b[i] << 24 | (uint32_t)b[i + 1] << 16 | (uint32_t)b[i + 2] << 8 | b[i + 3]; };
    if (b.size() >= 12 && be32(0) == 12 && be32(4) == 0x6A502020u) {          // JP2: walk the boxes to the codestream
        for (;;) {
            if (at + 8 > b.size()) return false;
            uint64_t len = be32(at);
            const uint32_t type = be32(at + 4);
            size_t hdr = 8;
            if (len == 1) { if (at + 16 > b.size()) return false; len = (uint64_t)be32(at + 8) << 32 | be32(at + 12); hdr = 16; }
            if (type == 0x6A703263u) { at += hdr; break; }                    // 'jp2c'
            if (len < hdr) return false;                                      // (0 = to the end of the file: no codestream box follows)
            if (len > b.size() - at) return false;                            // (a length from the file: never past what was read,
            at += (size_t)len;                                                //  never wrapping back -- the lock is held here)
        }
    }
    if (at + 4 > b.size() || be16(at) != 0xFF4F) return false;
    at += 2;
    bool have_qcd = false;
    while (at + 4 <= b.size()) {
        const uint32_t m = be16(at), len = be16(at + 2);
        if (m == 0xFF90 || m == 0xFF93) break;                                // SOT / SOD: end of the main header
        if (m < 0xFF00 || len < 2 || at + 2 + len > b.size()) return false;
        const size_t d = at + 4, n = len - 2;
        if (m == 0xFF5C && n >= 1) {                                          // QCD
            h.guard_bits = b[d] >> 5; h.qstyle = b[d] & 0x1Fu;
            h.words.clear();
            if (h.qstyle == 0) for (size_t i = 1; i < n; ++i) h.words.push_back(b[d + i]);
            else for (size_t i = 1; i + 1 < n; i += 2) h.words.push_back((uint16_t)be16(d + i));
            have_qcd = true;
        } else if (m == 0xFF5D || m == 0xFF53 || m == 0xFF5E || m == 0xFF5F) {
            h.overrides = true;
        }
        at += 2 + len;
    }
    // the first tile-part header as well: a COD / COC / QCD / QCC / RGN / POC there overrides the main header for that tile,
    // and band_numbps (with it every block's missing_msbs) would come out of the wrong exponents
    if (at + 4 <= b.size() && be16(at) == 0xFF90) {
        at += 2 + be16(at + 2);
        while (at + 4 <= b.size()) {
            const uint32_t m = be16(at), len = be16(at + 2);
            if (m == 0xFF93 || m < 0xFF00 || len < 2) break;
            if (m == 0xFF52 || m == 0xFF53 || m == 0xFF5C || m == 0xFF5D || m == 0xFF5E || m == 0xFF5F) h.overrides = true;
            at += 2 + len;
        }
    }
    return have_qcd;
}

// ---- decode: the callback record of plugin_decompress (plugin/plugin_interface.h:86-130).  C++ on purpose -- it
//      carries two std::string members, so it is no C ABI; plugin and host must share one libstdc++.  Its layout is
//      checked against the reference's own struct in oracle/ref_harness/abi_check.cpp.
struct DecodeCallbackInfo {
    size_t deviceId = 0;
    gra_init_decompressors_func init_decompressors_func = nullptr;
    std::string inputFile, outputFile;
    int32_t decod_format = 0, cod_format = 0;             // GRK_UNK_FMT: the host takes them from its own parameters
    void* stream = nullptr; void* codec = nullptr;
    void* decompressor_parameters = nullptr;
    gra_header_info header_info;
    gra_image* image = nullptr;
    bool plugin_owns_image = false;
    gra_plugin_tile* tile = nullptr;
    int32_t error_code = 0;
    uint32_t decompress_flags = 0;
    void* user_data = nullptr;
};
typedef int32_t (*DecodeUserCallback)(DecodeCallbackInfo*);

gra_header_info g_dec_header;           // what the host's header parser told init_decompressors_func
gra_image* g_dec_image = nullptr;
int dec_init_decompressors(gra_header_info* h, gra_image* img)
{
    if (!h || !img) return 1;
    g_dec_header = *h;
    g_dec_image = img;
    return 0;
}

// The plugin side of Grok's decode protocol (grk_decompress.cpp:792-1008 is the host side):
//   1. GRK_DECODE_HEADER: the host opens the stream, reads the main header and calls init_decompressors_func
//   2. GRK_DECODE_T2 with our tile tree attached: the host runs Tier-2 and decompress_synch_plugin_with_host copies
//      every code-block's bytes, numbps and pass count into the tree (plugin_bridge.cpp:24-80); T1 and everything
//      after it are skipped on the host (TileProcessor.cpp:786-789, CodeStreamDecompress.cpp:935-936)
//   3. block decode, inverse DWT, inverse MCT on the GPU; the pixels go into the host's grk_image
//   4. GRK_DECODE_POST_T1: the host stores the image;  5. GRK_PLUGIN_DECODE_CLEAN
// Anything outside the hot path's scope is declined (non-zero) and the host decodes on its CPU.
// in_path / out_path: batch mode -- the host's callback takes them as input_file_name / output_file_name (grok.cpp:698-725),
// otherwise it reads parameters->infile / outfile
int32_t decompress_file(void* params, DecodeUserCallback cb, const char* in_path = nullptr, const char* out_path = nullptr)
{
    if (!g_ctx || !cb) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    DecodeCallbackInfo info;
    std::memset(&info.header_info, 0, sizeof(info.header_info));
    info.decompressor_parameters = params;
    if (in_path) info.inputFile = in_path;
    if (out_path) info.outputFile = out_path;
    info.init_decompressors_func = dec_init_decompressors;
    info.decompress_flags = GRA_DECODE_HEADER;
    g_dec_image = nullptr;
    auto clean = [&](int32_t rc) {
        info.decompress_flags = GRA_PLUGIN_DECODE_CLEAN;
        (void)cb(&info);
        return rc;
    };
    if (cb(&info) != 0 || !g_dec_image) return clean(-1);
    const gra_header_info& h = g_dec_header;
    gra_image* img = g_dec_image;
    // the stream's main header, from the file the host was pointed at (grk_decompress -i: parameters->infile,
    // grk_decompress.cpp:552).  A host that decodes from memory gives us nothing to read it from: declined.
    StreamHeader sh;
    {
        const gra_decompress_parameters_head* dp = static_cast<const gra_decompress_parameters_head*>(params);
        const char* path = in_path ? in_path : !dp ? nullptr : dp->infile[0] ? dp->infile : dp->core.infile[0] ? dp->core.infile : nullptr;
        if (!path || !read_stream_header(path, sh) || sh.overrides) return clean(-1);
    }
    // the scope of the hot path: one tile (anywhere on the canonical grid), equal full-resolution components, one
    // layer, one codeword segment per block (the host's bridge throws on more); irreversible only for classic
    // blocks (the reference's own HT + 9/7 encoder is broken, D1: there is no stream to be compatible with)
    if (h.t_grid_width * h.t_grid_height != 1 || img->numcomps == 0 ||
        (h.irreversible && (h.cblk_sty & 0x40u)) || (h.cblk_sty & 0x05u) || h.numresolutions == 0)
        return clean(-1);
    const gra_image_comp& c0 = img->comps[0];
    // Components: one precision and signedness; sub-sampled components (SIZ XRsiz / YRsiz) as they come -- all alike (every
    // component then is the same w x h rectangle at ceil(offset / d): one geometry) or each in its own way (4:2:0 ...: every
    // component its own tile-component, the tree built per component, runs of equal factors decoded together)
    bool alike = true;
    for (uint16_t k = 0; k < img->numcomps; ++k) {
        const gra_image_comp& ck = img->comps[k];
        if (ck.dx < 1 || ck.dy < 1 || ck.dx > 255 || ck.dy > 255 || ck.w == 0 || ck.h == 0 || ck.prec != c0.prec || ck.sgnd != c0.sgnd || ck.prec > 16)
            return clean(-1);
        alike = alike && ck.dx == c0.dx && ck.dy == c0.dy && ck.w == c0.w && ck.h == c0.h && ck.x0 == c0.x0 && ck.y0 == c0.y0;
    }
    if (!alike && img->numcomps > 4) return clean(-1);
    grk_amd_tile_params tp{};
    // alike: the tile IS the component rectangle; else: the tile on the reference grid, the components derived from it
    tp.tile_w = alike ? c0.w : img->x1 - img->x0; tp.tile_h = alike ? c0.h : img->y1 - img->y0; tp.num_comps = img->numcomps;
    tp.tile_x0 = alike ? c0.x0 : img->x0; tp.tile_y0 = alike ? c0.y0 : img->y0;
    tp.prec = c0.prec; tp.sgnd = c0.sgnd; tp.irreversible = h.irreversible ? 1 : 0; tp.mct = h.mct ? 1 : 0;
    tp.num_levels = (uint8_t)(h.numresolutions - 1);
    uint32_t ew = 0, eh = 0;
    while ((1u << ew) < h.cblockw_init) ++ew;
    while ((1u << eh) < h.cblockh_init) ++eh;
    tp.cblk_w_exp = (uint8_t)ew; tp.cblk_h_exp = (uint8_t)eh;
    if (h.csty & 1u) {                                      // precinct partition: sizes 2^PPx x 2^PPy per resolution (0 = coarsest)
        for (uint32_t r = 0; r < h.numresolutions; ++r) {
            uint32_t ex = 0, ey = 0;
            while ((1u << ex) < h.prcw_init[r]) ++ex;
            while ((1u << ey) < h.prch_init[r]) ++ey;
            if (ex > 15 || ey > 15 || (ex | (ey << 4)) == 0) return clean(-1);
            tp.precinct_exp[r] = (uint8_t)(ex | (ey << 4));
        }
    }
    tp.reserved[0] = (h.cblk_sty & 0x40u) ? 0 : 1;         // HT bit clear: classic Part-1 blocks
    tp.reserved[1] = h.cblk_sty & 0x3Fu;
    std::vector<grk_amd_tile_params> cps;                 // !alike: every component's rectangle (num_comps = 1)
    uint8_t cdx[4] = {1, 1, 1, 1}, cdy[4] = {1, 1, 1, 1};
    std::vector<grk_amd_block> layout;
    int64_t nb = 0;
    if (alike) {
        nb = grk_amd_tile_num_blocks(&tp);
        if (nb <= 0) return clean(-1);
        layout.resize((size_t)nb);
        if (grk_amd_tile_layout(&tp, layout.data(), (uint64_t)nb, nullptr) != nb) return clean(-1);
    } else {
        if (tp.mct) return clean(-1);                     // (a colour transform across component sizes: no encoder writes that)
        const grk_amd_image_layout iml{tp.tile_x0, tp.tile_y0, tp.tile_x0 + tp.tile_w, tp.tile_y0 + tp.tile_h, tp.tile_x0, tp.tile_y0, tp.tile_w, tp.tile_h};
        cps.resize(tp.num_comps);
        for (uint32_t c = 0; c < tp.num_comps; ++c) {
            const gra_image_comp& ck = img->comps[c];
            cdx[c] = (uint8_t)ck.dx; cdy[c] = (uint8_t)ck.dy;
            if (grk_amd_layout_tile_comp(&iml, &tp, ck.dx, ck.dy, 0, &cps[c]) != GRK_AMD_OK) return clean(-1);
            cps[c].num_comps = 1; cps[c].mct = 0;
            if (cps[c].tile_w != ck.w || cps[c].tile_h != ck.h) return clean(-1);      // (the host's component is not the rectangle SIZ implies)
            const int64_t nbc = grk_amd_tile_num_blocks(&cps[c]);
            if (nbc <= 0) return clean(-1);
            const size_t at = layout.size();
            layout.resize(at + (size_t)nbc);
            if (grk_amd_tile_layout(&cps[c], layout.data() + at, (uint64_t)nbc, nullptr) != nbc) return clean(-1);
            nb += nbc;
        }
    }
    // a tree whose blocks own buffers the host can copy into: nominal block area x 4 bytes, as the host allocates
    // for its own code-blocks (t1/T1Structs.cpp:292-307)
    // The host copies getSegBuffersLen() bytes into a block's buffer without asking how large it is
    // (plugin_bridge.cpp:69-71 copy_to_contiguous_buffer), so a crafted stream that signals a longer block writes past
    // its slot.  No block is longer than the file it comes from: a tail of that size behind the last slot keeps every
    // such write inside the allocation, and the lengths are checked against the slots after the Tier-2 callback.
    std::vector<grk_amd_coded_block> slots((size_t)nb);
    std::vector<uint64_t> slot_cap((size_t)nb);
    uint64_t cap = 0;
    for (size_t i = 0; i < (size_t)nb; ++i) {
        slots[i].offset = cap; slots[i].length = 0; slots[i].missing_msbs = 0;
        slot_cap[i] = (uint64_t)(layout[i].x1 - layout[i].x0) * (layout[i].y1 - layout[i].y0) * 4u + 16u;
        cap += slot_cap[i];
    }
    // band numbps of an HT stream from ITS quantisation marker (reversible: 8-bit exponents; Quantizer.cpp:49-51)
    std::vector<uint8_t> band_numbps;
    if (!tp.reserved[0]) {
        const size_t nbands = 3u * tp.num_levels + 1u;
        if (sh.qstyle != 0 || sh.words.size() < nbands) return clean(-1);
        for (size_t b = 0; b < nbands; ++b) {
            const int v = (int)(sh.words[b] >> 3) + (int)sh.guard_bits - 1;
            if (v < 1 || v > 31) return clean(-1);
            band_numbps.push_back((uint8_t)v);
        }
    }
    TileOwner* const owner = alike ? acquire_owner(tp) : make_owner(tp, &cps);
    if (owner) { owner->served_decode = true; owner->no_cache = !alike; }
    if (!owner || !owner->ensure_coded(g_ctx, cap + sh.file_size + 64)) { release_owner(owner); return clean(-1); }
    std::memset(owner->coded, 0, cap + sh.file_size + 64);
    owner->table = slots;
    patch_owner(owner);
    gra_plugin_tile* tree = &owner->tile;
    for (auto* b : owner->block_ptr) { b->numBitPlanes = 0; b->numPasses = 0; }
    auto done = [&](int32_t rc) { grk_amd_plugin_tile_destroy(tree); return clean(rc); };
    info.tile = tree;
    // T2 alone cannot be asked for: without GRK_DECODE_POST_T1 the host never advances to the next tile-part
    // (CodeStreamDecompress.cpp:968-973 skips findNextTile) and its tile loop then fails with "no SOT marker found"
    // (:452-461, :2076) AFTER Tier-2 and the synch have run -- reference defect D11.  With POST_T1 set the host also
    // runs its inverse MCT + DC shift over the (empty) tile buffers and hands them to the image (cheap next to T1 and
    // the DWT, which stay skipped: TileProcessor.cpp:786-818); the pixels are overwritten below.
    info.decompress_flags = GRA_DECODE_T2 | GRA_DECODE_POST_T1;
    tree->decompress_flags = GRA_DECODE_T2 | GRA_DECODE_POST_T1;
    if (cb(&info) != 0) return done(-1);
    {
        const auto& bl = reinterpret_cast<TileOwner*>(tree)->blocks;
        for (size_t i = 0; i < (size_t)nb; ++i)
            if (bl[i].compressedDataLength > slot_cap[i]) return done(-1);       // overran its slot: the CPU decoder takes it
    }
    const size_t bps = (tp.prec + 7u) / 8u;
    std::vector<size_t> plane_at(tp.num_comps + 1u, 0);
    for (uint32_t c = 0; c < tp.num_comps; ++c)
        plane_at[c + 1] = plane_at[c] + (alike ? (size_t)tp.tile_w * tp.tile_h : (size_t)cps[c].tile_w * cps[c].tile_h) * bps;
    std::vector<uint8_t> px(plane_at[tp.num_comps]);
    const int drc = alike ? grk_amd_plugin_tile_decode_qcd(g_ctx, &tp, tree, band_numbps.empty() ? nullptr : band_numbps.data(),
                                                           (uint32_t)band_numbps.size(), px.data(), 0)
                          : grk_amd_plugin_tile_decode_subsampled(g_ctx, &tp, cdx, cdy, tree, band_numbps.empty() ? nullptr : band_numbps.data(),
                                                                  (uint32_t)band_numbps.size(), px.data());
    if (drc != GRK_AMD_OK) return done(-1);
    img = info.image ? info.image : img;
    for (uint16_t k = 0; k < img->numcomps; ++k) {
        gra_image_comp& ck = img->comps[k];
        const uint32_t pw = alike ? tp.tile_w : cps[k].tile_w;
        if (!ck.data) {                                   // the host skipped post-T1, so nothing was allocated
            ck.stride = (ck.w + 31u) & ~31u;
            void* mem = nullptr;
            if (posix_memalign(&mem, 64, (size_t)ck.stride * ck.h * sizeof(int32_t)) != 0) return done(-1);
            ck.data = static_cast<int32_t*>(mem);         // freed by the host with the image (grk_aligned_free = free)
        }
        for (uint32_t y = 0; y < ck.h; ++y) {
            int32_t* dst = ck.data + (size_t)y * ck.stride;
            const uint8_t* src = px.data() + plane_at[k] + (size_t)y * pw * bps;
            for (uint32_t x = 0; x < ck.w; ++x) {
                if (bps == 1) dst[x] = tp.sgnd ? (int32_t)(int8_t)src[x] : (int32_t)src[x];
                else { uint16_t v; std::memcpy(&v, src + 2 * x, 2); dst[x] = tp.sgnd ? (int32_t)(int16_t)v : (int32_t)v; }
            }
        }
    }
    info.decompress_flags = GRA_DECODE_POST_T1;
    tree->decompress_flags = GRA_DECODE_POST_T1;
    const int32_t rc = cb(&info);
    info.tile = nullptr;
    return done(rc == 0 ? 0 : -1);
}

// ---- batch decode (plugin/plugin_interface.h:131-143; the host side: grk_decompress.cpp:874-900): a worker thread walks the
//      input directory; a stream outside the hot path's scope is handed back to the host's own decoder in the same callback
//      protocol (all stages in one call, grok.h:1254 GRK_DECODE_ALL), so that every file of the directory comes out
std::thread g_dbatch;
std::atomic<bool> g_dbatch_done{true}, g_dbatch_stop{false};
std::atomic<int> g_dbatch_gpu{0}, g_dbatch_cpu{0}, g_dbatch_failed{0};
struct { std::string in, out; void* params = nullptr; DecodeUserCallback cb = nullptr; } g_dbatch_job;

const char* out_extension(int32_t cod_format)
{
    switch (cod_format) {           // GRK_SUPPORTED_FILE_FMT, grok.h:59-72
    case 3: return ".ppm"; case 4: return ".pgx"; case 5: return ".pam"; case 6: return ".bmp"; case 7: return ".tif";
    case 8: return ".raw"; case 9: return ".png"; case 10: return ".rawl"; case 11: return ".jpg";
    default: return ".ppm";
    }
}

int32_t plugin_exit() { drop_tile_cache(); drop_pinned_pool(); return 0; }
void* plugin_create(gra_minpf_object_params*) { return nullptr; }
int32_t plugin_destroy(void*) { return 0; }

} // namespace

extern "C" {

#define GRA_EXPORT __attribute__((visibility("default")))

GRA_EXPORT gra_plugin_tile* grk_amd_plugin_tile_create(grk_amd_ctx* ctx, const grk_amd_tile_params* p,
                                                       const void* pixels, int on_device)
{
    if (!ctx || !p || !pixels) return nullptr;
    TileOwner* o = acquire_owner(*p);            // a kept tree of this geometry, or a new one
    if (!o) return nullptr;
    uint64_t total = 0;
    bool ok = grk_amd_encode_tiles(ctx, p, 1, pixels, on_device, o->table.data(), &total) == GRK_AMD_OK;
    for (size_t i = 0; ok && i < o->table.size(); ++i)
        if (o->table[i].length > 65535) ok = false;    // host keeps rates in uint16_t (plugin_bridge.cpp:174, D7)
    ok = ok && o->ensure_coded(ctx, total ? total : 1);
    ok = ok && (!total || grk_amd_fetch_coded(ctx, o->coded, total) == GRK_AMD_OK);      // (pinned: one DMA)
    if (!ok) { release_owner(o); return nullptr; }
    patch_owner(o);
    return &o->tile;
}

// The library-level drop-in for an image whose components are sub-sampled each in its own way (4:2:0 ...): `p` = the tile on the
// REFERENCE grid (tile_x0 / tile_y0 / tile_w / tile_h), component c = [ceil(x0 / dx_c), ceil(x1 / dx_c)) x ... of its own samples
// (tile/TileProcessor.cpp:605-612), `planes` = the components back to back, each tight at its own size.  Runs of consecutive
// components with equal factors are coded together (MCT only for a run that holds components 0..2, else off as the reference has it:
// CodeStreamCompress.cpp:434-447); the tree carries every component's own resolutions / precincts / blocks.
GRA_EXPORT gra_plugin_tile* grk_amd_plugin_tile_create_subsampled(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const uint8_t* comp_dx,
                                                                 const uint8_t* comp_dy, const void* planes)
{
    if (!ctx || !p || !comp_dx || !comp_dy || !planes || p->num_comps == 0) return nullptr;
    const uint32_t nc = p->num_comps, bps = (p->prec + 7u) / 8u;
    const grk_amd_image_layout im{p->tile_x0, p->tile_y0, p->tile_x0 + p->tile_w, p->tile_y0 + p->tile_h, p->tile_x0, p->tile_y0, p->tile_w, p->tile_h};
    std::vector<grk_amd_tile_params> cps(nc);
    std::vector<size_t> plane_at(nc + 1, 0);
    for (uint32_t c = 0; c < nc; ++c) {
        if (grk_amd_layout_tile_comp(&im, p, comp_dx[c], comp_dy[c], 0, &cps[c]) != GRK_AMD_OK) return nullptr;
        cps[c].num_comps = 1; cps[c].mct = 0;
        plane_at[c + 1] = plane_at[c] + (size_t)cps[c].tile_w * cps[c].tile_h * bps;
    }
    TileOwner* o = make_owner(*p, &cps);           // (not cached: the cache is keyed by the tile's parameters alone)
    if (!o) return nullptr;
    bool ok = true;
    size_t row = 0;
    uint64_t used = 0;
    const bool mct = p->mct && nc >= 3 && comp_dx[0] == comp_dx[1] && comp_dx[1] == comp_dx[2] && comp_dy[0] == comp_dy[1] && comp_dy[1] == comp_dy[2];
    for (uint32_t c0 = 0; ok && c0 < nc;) {
        uint32_t n = 1;
        while (c0 + n < nc && comp_dx[c0 + n] == comp_dx[c0] && comp_dy[c0 + n] == comp_dy[c0]) ++n;
        grk_amd_tile_params pr = cps[c0];
        pr.num_comps = (uint16_t)n; pr.mct = (mct && c0 == 0 && n >= 3) ? 1 : 0;
        const int64_t nbl = grk_amd_tile_num_blocks(&pr);
        uint64_t total = 0;
        ok = nbl > 0 && row + (size_t)nbl <= o->table.size() &&
             grk_amd_encode_tiles(ctx, &pr, 1, (const uint8_t*)planes + plane_at[c0], 0, o->table.data() + row, &total) == GRK_AMD_OK;
        for (size_t i = row; ok && i < row + (size_t)nbl; ++i) {
            if (o->table[i].length > 65535) ok = false;
            o->table[i].offset += used;
        }
        if (ok && total) {
            // (the bytes of the runs one behind the other: a run's encode reuses the context's arena)
            const size_t need = used + total;
            if (need > o->coded_cap) {
                uint8_t* old = o->coded; const size_t old_cap = o->coded_cap; const bool old_pinned = o->coded_pinned;
                o->coded = nullptr; o->coded_cap = 0;
                ok = o->ensure_coded(ctx, need * 2);
                if (ok && used) std::memcpy(o->coded, old, used);
                if (old) { if (old_pinned) grk_amd_host_free(nullptr, old); else std::free(old); }
                (void)old_cap;
            }
            ok = ok && grk_amd_fetch_coded(ctx, o->coded + used, total) == GRK_AMD_OK;
        }
        used += total; row += (size_t)nbl; c0 += n;
    }
    ok = ok && row == o->table.size() && (o->coded || o->ensure_coded(ctx, 1));
    if (!ok) { delete o; return nullptr; }
    patch_owner(o);
    o->no_cache = true;                            // (release_owner: the cache is keyed by the tile's parameters alone)
    return &o->tile;
}

GRA_EXPORT int grk_amd_plugin_tile_fill_distortion(grk_amd_ctx* ctx, gra_plugin_tile* tile)
{
    if (!ctx || !tile) return GRK_AMD_ERR_INVALID;
    TileOwner* o = reinterpret_cast<TileOwner*>(tile);
    std::vector<double> dd(o->blocks.size());
    const int rc = grk_amd_block_distortion(ctx, dd.data(), dd.size());
    if (rc) return rc;
    for (size_t i = 0; i < dd.size(); ++i) o->blocks[i].passes[0].distortionDecrease = dd[i];
    return GRK_AMD_OK;
}

GRA_EXPORT void grk_amd_plugin_tile_destroy(gra_plugin_tile* tile)
{
    release_owner(reinterpret_cast<TileOwner*>(tile));      // kept for the next frame of this geometry
}

GRA_EXPORT int grk_amd_plugin_tile_decode(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const gra_plugin_tile* tile,
                                          void* pixels, int pixels_on_device)
{
    return grk_amd_plugin_tile_decode_qcd(ctx, p, tile, nullptr, 0, pixels, pixels_on_device);
}

// components [comp0, comp0 + p->num_comps) of the tree, which all have p's geometry (the whole tree: comp0 = 0)
static int decode_tree_comps(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const gra_plugin_tile* tile, uint32_t comp0,
                             const uint8_t* band_numbps, uint32_t nbands, void* pixels, int pixels_on_device);

GRA_EXPORT int grk_amd_plugin_tile_decode_qcd(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const gra_plugin_tile* tile,
                                              const uint8_t* band_numbps, uint32_t nbands, void* pixels, int pixels_on_device)
{
    if (!ctx || !p || !tile || !pixels) return GRK_AMD_ERR_INVALID;
    if (tile->numComponents != p->num_comps) return GRK_AMD_ERR_INVALID;
    return decode_tree_comps(ctx, p, tile, 0, band_numbps, nbands, pixels, pixels_on_device);
}

// The decode counterpart of grk_amd_plugin_tile_create_subsampled: `p` = the tile on the reference grid, component c of the tree has
// the geometry of [ceil(x0 / dx_c), ceil(x1 / dx_c)) x ...; `planes` receives the components back to back, each tight at its own size.
// Runs of components with equal factors are decoded together (the inverse colour transform only for a run that holds components
// 0..2 of a stream that signals it -- an encoder cannot have applied it across sizes).
GRA_EXPORT int grk_amd_plugin_tile_decode_subsampled(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const uint8_t* comp_dx,
                                                     const uint8_t* comp_dy, const gra_plugin_tile* tile, const uint8_t* band_numbps,
                                                     uint32_t nbands, void* planes)
{
    if (!ctx || !p || !comp_dx || !comp_dy || !tile || !planes || tile->numComponents != p->num_comps) return GRK_AMD_ERR_INVALID;
    const uint32_t nc = p->num_comps, bps = (p->prec + 7u) / 8u;
    const grk_amd_image_layout im{p->tile_x0, p->tile_y0, p->tile_x0 + p->tile_w, p->tile_y0 + p->tile_h, p->tile_x0, p->tile_y0, p->tile_w, p->tile_h};
    size_t at = 0;
    for (uint32_t c0 = 0; c0 < nc;) {
        uint32_t n = 1;
        while (c0 + n < nc && comp_dx[c0 + n] == comp_dx[c0] && comp_dy[c0 + n] == comp_dy[c0]) ++n;
        grk_amd_tile_params pr;
        int rc = grk_amd_layout_tile_comp(&im, p, comp_dx[c0], comp_dy[c0], 0, &pr);
        if (rc) return rc;
        pr.num_comps = (uint16_t)n;
        pr.mct = (p->mct && c0 == 0 && n >= 3) ? 1 : 0;
        if (p->mct && c0 == 0 && n < 3 && nc >= 3) return GRK_AMD_ERR_UNSUPPORTED;     // (a colour transform across sizes: no encoder writes that)
        rc = decode_tree_comps(ctx, &pr, tile, c0, band_numbps, nbands, (uint8_t*)planes + at, 0);
        if (rc) return rc;
        at += (size_t)pr.tile_w * pr.tile_h * n * bps;
        c0 += n;
    }
    return GRK_AMD_OK;
}

static int decode_tree_comps(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const gra_plugin_tile* tile, uint32_t comp0,
                             const uint8_t* band_numbps, uint32_t nbands, void* pixels, int pixels_on_device)
{
    if (band_numbps && nbands != 3u * p->num_levels + 1u) return GRK_AMD_ERR_INVALID;
    const int64_t nb = grk_amd_tile_num_blocks(p);
    if (nb <= 0) return (int)(nb ? nb : GRK_AMD_ERR_UNSUPPORTED);
    std::vector<grk_amd_block> layout((size_t)nb);
    if (grk_amd_tile_layout(p, layout.data(), (uint64_t)nb, nullptr) != nb) return GRK_AMD_ERR_INVALID;
    if (comp0 + p->num_comps > tile->numComponents) return GRK_AMD_ERR_INVALID;
    // walk the tree in the enumeration order both sides share: comp -> resolution -> band -> precinct -> block
    std::vector<grk_amd_coded_block> table((size_t)nb);
    std::vector<uint8_t> coded;
    std::vector<float> steps;            // irreversible: the bands' step sizes; the host's synch stores half (plugin_bridge.cpp:40)
    size_t i = 0;
    for (uint32_t c = comp0; c < comp0 + p->num_comps; ++c) {
        const gra_plugin_tile_component* tc = tile->tileComponents[c];
        for (uint32_t r = 0; r < tc->numResolutions; ++r) {
            const gra_plugin_resolution* res = tc->resolutions[r];
            for (uint32_t b = 0; b < res->numBands; ++b) {
                const gra_plugin_band* band = res->band[b];
                steps.push_back(band->stepsize * 2.0f);
                for (uint64_t pr = 0; pr < band->numPrecincts; ++pr) {
                    const gra_plugin_precinct* prec = band->precincts[pr];
                    for (uint64_t k = 0; k < prec->numBlocks; ++k) {
                        if (i >= (size_t)nb) return GRK_AMD_ERR_INVALID;
                        const gra_plugin_code_block* cb = prec->blocks[k];
                        grk_amd_coded_block& row = table[i];
                        row.offset = coded.size();
                        row.length = cb->compressedData ? cb->compressedDataLength : 0;
                        const uint32_t nbp = (uint32_t)cb->numBitPlanes;
                        if (p->reserved[0]) row.missing_msbs = row.length ? (nbp | ((uint32_t)cb->numPasses << 8)) : 0;
                        else {                                                // band numbps - block numbps
                            const uint32_t bn = band_numbps ? band_numbps[layout[i].res ? 3u * layout[i].res - 2u + (layout[i].band - 1u) : 0u]
                                                            : layout[i].kmax;
                            if (row.length && nbp > bn) return GRK_AMD_ERR_INVALID;
                            row.missing_msbs = row.length ? bn - nbp : 0;
                        }
                        if (row.length) coded.insert(coded.end(), cb->compressedData, cb->compressedData + row.length);
                        coded.resize((coded.size() + 15u) & ~(size_t)15u);
                        ++i;
                    }
                }
            }
        }
    }
    if (i != (size_t)nb) return GRK_AMD_ERR_INVALID;
    coded.resize(coded.size() + 16);
    if (p->irreversible && grk_amd_set_decode_steps(ctx, steps.data(), (uint32_t)steps.size()) != GRK_AMD_OK) return GRK_AMD_ERR_INVALID;
    const int rc = grk_amd_decode_tiles(ctx, p, 1, table.data(), coded.data(), coded.size(), 0, pixels, pixels_on_device);
    if (p->irreversible) (void)grk_amd_set_decode_steps(ctx, nullptr, 0);
    return rc;
}

GRA_EXPORT gra_minpf_exit_func minpf_post_load_plugin(const char*, const gra_minpf_platform_services* services)
{
    if (!services || !services->registerObject) return nullptr;
    gra_minpf_register_params rp;
    rp.version.major = 1;                 // the loader insists on major == 1 (minpf_plugin_manager.cpp:47-72)
    rp.version.minor = 0;
    rp.createFunc = plugin_create;
    rp.destroyFunc = plugin_destroy;
    if (services->registerObject("grok_amd MI355X tile processor", &rp) < 0) return nullptr;
    return plugin_exit;
}

GRA_EXPORT bool plugin_init(gra_plugin_init_info info)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_verbose = info.verbose;
    if (const char* e = std::getenv("GRK_AMD_PLUGIN_DEBUG")) g_debug_state = std::atoi(e) ? GRA_PLUGIN_STATE_DEBUG : GRA_PLUGIN_STATE_NO_DEBUG;
    if (g_ctx) return true;
    const int rc = grk_amd_create(info.deviceId, info.verbose ? 1 : 0, &g_ctx);
    if (rc != GRK_AMD_OK) {
        if (g_verbose) std::fprintf(stderr, "[grok_amd plugin] no usable MI355X (rc=%d): host falls back to CPU\n", rc);
        g_ctx = nullptr;
        return false;
    }
    // the devices of the batch mode: the one Grok named first (it shares g_mu with the single-file entry points), then the
    // node's other GPUs, or what GRK_AMD_PLUGIN_DEVICES lists after it (e.g. "0,0": a second context on GPU 0)
    g_devs.clear();
    g_devs.emplace_back(new Dev());
    g_devs[0]->ctx = g_ctx; g_devs[0]->mu = &g_mu;
    std::vector<int> more;
    if (const char* e = std::getenv("GRK_AMD_PLUGIN_DEVICES")) {
        bool first = true;
        for (const char* q = e; *q;) {
            char* end = nullptr;
            const long v = std::strtol(q, &end, 10);
            if (end == q) break;
            if (!first) more.push_back((int)v);          // (the first entry is deviceId's place)
            first = false;
            q = *end == ',' ? end + 1 : end;
        }
    } else {
        const int n = grk_amd_device_count();
        for (int d = 0; d < n; ++d) if (d != info.deviceId) more.push_back(d);
    }
    for (int d : more) {
        grk_amd_ctx* c = nullptr;
        if (grk_amd_create(d, info.verbose ? 1 : 0, &c) != GRK_AMD_OK) continue;       // (a GPU that is not there is not used)
        g_devs.emplace_back(new Dev());
        g_devs.back()->ctx = c; g_devs.back()->mu = &g_devs.back()->own;
    }
    if (g_verbose) std::fprintf(stderr, "[grok_amd plugin] %zu device context(s)\n", g_devs.size());
    return true;
}

// (for tests and embedders: how many device contexts the batch mode spreads files over)
GRA_EXPORT uint32_t grk_amd_plugin_num_devices(void) { return (uint32_t)g_devs.size(); }

GRA_EXPORT int32_t plugin_encode(gra_cparameters* params, gra_encode_callback callback)
{
    if (!params) return -1;
    return encode_file(params, params->infile, params->outfile, callback);
}

GRA_EXPORT int32_t plugin_batch_encode(const char* input_dir, const char* output_dir, gra_cparameters* params,
                                       gra_encode_callback callback)
{
    if (!g_ctx || !input_dir || !output_dir || !params || !callback) return -1;
    if (!g_batch_done.load()) return -1;
    if (g_batch.joinable()) g_batch.join();
    g_batch_done = false; g_batch_stop = false;
    std::string in(input_dir), out(output_dir);
    gra_cparameters* cp = params;
    g_batch = std::thread([in, out, cp, callback]() {
        // the files of the directory, then three overlapped stages over them (a file is in one stage at a time, every stage
        // works on one file at a time): while the GPU codes file n, file n + 1 is being read and de-interleaved and the host
        // library runs its Tier-2 and writes file n - 1.  Hand-over slots hold one job each, so at most three images are
        // in flight.
        std::vector<std::pair<std::string, std::string>> files;
        if (DIR* d = opendir(in.c_str())) {
            while (dirent* e = readdir(d)) {
                std::string name(e->d_name);
                const size_t dot = name.rfind('.');
                if (dot == std::string::npos) continue;
                const std::string ext = name.substr(dot);
                if (ext != ".pgm" && ext != ".ppm" && ext != ".pnm") continue;
                files.emplace_back(in + "/" + name, out + "/" + name.substr(0, dot) + ".j2k");
            }
            closedir(d);
        }
        struct Slot {
            std::mutex m; std::condition_variable cv; std::unique_ptr<EncodeJob> job; bool closed = false;
            void put(std::unique_ptr<EncodeJob> j) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !job; }); job = std::move(j); cv.notify_all(); }
            std::unique_ptr<EncodeJob> take() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return job || closed; }); auto j = std::move(job); cv.notify_all(); return j; }
            void close() { std::lock_guard<std::mutex> lk(m); closed = true; cv.notify_all(); }
        } loaded, coded;
        std::thread reader([&]() {
            for (const auto& f : files) {
                if (g_batch_stop.load()) break;
                auto j = std::make_unique<EncodeJob>();
                j->in = f.first; j->out = f.second;
                if (load_step(*j)) loaded.put(std::move(j));
            }
            loaded.close();
        });
        std::thread writer([&]() {
            while (auto j = coded.take()) host_step(cp, *j, callback);
        });
        // one GPU stage per device context: whichever is free takes the next loaded file (files are independent: replicas,
        // no exchange -- SURVEY.md 8(e) "single-tile configs: replicas only")
        std::vector<std::thread> gpus;
        for (size_t d = 0; d < g_devs.size(); ++d)
            gpus.emplace_back([&, d]() {
                Dev* dev = g_devs[d].get();
                while (auto j = loaded.take()) {
                    if (g_batch_stop.load()) continue;          // (drain the reader)
                    if (!single_tile(cp, j->w, j->h)) { (void)encode_multi_tile(cp, *j, dev); continue; }     // several tiles: the whole file here
                    if (gpu_step(cp, *j, dev)) coded.put(std::move(j));
                }
            });
        for (auto& g : gpus) g.join();
        coded.close();
        reader.join();
        writer.join();
        g_batch_done = true;
    });
    return 0;
}

GRA_EXPORT bool plugin_is_batch_complete(void) { return g_batch_done.load() && g_dbatch_done.load(); }

GRA_EXPORT void plugin_stop_batch_encode(void)
{
    g_batch_stop = true;
    if (g_batch.joinable()) g_batch.join();
    g_batch_done = true;
}

// Decode: Grok's plugin protocol end to end (decompress_file above); what is outside the hot path's scope is declined
// and the host keeps its CPU decoder (grk_decompress.cpp falls back when the plugin returns non-zero).  The batch
// variants stay declined.
GRA_EXPORT int32_t plugin_decompress(void* decompress_parameters, gra_decode_callback callback)
{
    return decompress_file(decompress_parameters, reinterpret_cast<DecodeUserCallback>(callback));
}
// layout facts of the C++ callback record for the ABI check (oracle/ref_harness/abi_check.cpp, tests)
GRA_EXPORT size_t grk_amd_plugin_decode_info_layout(int which)
{
    switch (which) {
    case 0: return sizeof(DecodeCallbackInfo);
    case 1: return offsetof(DecodeCallbackInfo, init_decompressors_func);
    case 2: return offsetof(DecodeCallbackInfo, inputFile);
    case 3: return offsetof(DecodeCallbackInfo, outputFile);
    case 4: return offsetof(DecodeCallbackInfo, decod_format);
    case 5: return offsetof(DecodeCallbackInfo, stream);
    case 6: return offsetof(DecodeCallbackInfo, codec);
    case 7: return offsetof(DecodeCallbackInfo, decompressor_parameters);
    case 8: return offsetof(DecodeCallbackInfo, header_info);
    case 9: return offsetof(DecodeCallbackInfo, image);
    case 10: return offsetof(DecodeCallbackInfo, plugin_owns_image);
    case 11: return offsetof(DecodeCallbackInfo, tile);
    case 12: return offsetof(DecodeCallbackInfo, error_code);
    case 13: return offsetof(DecodeCallbackInfo, decompress_flags);
    case 14: return offsetof(DecodeCallbackInfo, user_data);
    default: return 0;
    }
}
GRA_EXPORT int32_t plugin_init_batch_decompress(const char* input_dir, const char* output_dir, void* decompress_parameters,
                                                gra_decode_callback callback)
{
    if (!g_ctx || !input_dir || !output_dir || !decompress_parameters || !callback) return -1;
    if (!g_dbatch_done.load()) return -1;
    if (g_dbatch.joinable()) g_dbatch.join();
    g_dbatch_job.in = input_dir; g_dbatch_job.out = output_dir; g_dbatch_job.params = decompress_parameters;
    g_dbatch_job.cb = reinterpret_cast<DecodeUserCallback>(callback);
    return 0;
}
GRA_EXPORT int32_t plugin_batch_decompress(void)
{
    if (!g_ctx || !g_dbatch_job.cb || !g_dbatch_done.load()) return -1;
    if (g_dbatch.joinable()) g_dbatch.join();
    g_dbatch_done = false; g_dbatch_stop = false;
    g_dbatch_gpu = 0; g_dbatch_cpu = 0; g_dbatch_failed = 0;
    g_dbatch = std::thread([]() {
        const auto job = g_dbatch_job;
        const auto* dp = static_cast<const gra_decompress_parameters_head*>(job.params);
        std::vector<std::string> names;
        if (DIR* d = opendir(job.in.c_str())) {
            while (dirent* e = readdir(d)) {
                const std::string name(e->d_name);
                const size_t dot = name.rfind('.');
                if (dot == std::string::npos) continue;
                const std::string ext = name.substr(dot);
                if (ext == ".j2k" || ext == ".j2c" || ext == ".jp2" || ext == ".jph" || ext == ".jhc") names.push_back(name);
            }
            closedir(d);
        }
        std::sort(names.begin(), names.end());
        for (const auto& name : names) {
            if (g_dbatch_stop.load()) break;
            const std::string src = job.in + "/" + name;
            const std::string dst = job.out + "/" + name.substr(0, name.rfind('.')) + out_extension(dp->cod_format);
            if (decompress_file(job.params, job.cb, src.c_str(), dst.c_str()) == 0) { ++g_dbatch_gpu; continue; }
            // outside the hot path: the host decodes this one itself, all stages in one call
            DecodeCallbackInfo info;
            std::memset(&info.header_info, 0, sizeof(info.header_info));
            info.decompressor_parameters = job.params;
            info.inputFile = src; info.outputFile = dst;
            info.decompress_flags = GRA_DECODE_HEADER | GRA_DECODE_T2 | GRA_DECODE_T1 | GRA_DECODE_POST_T1;
            const int32_t rc = job.cb(&info);
            info.decompress_flags = GRA_PLUGIN_DECODE_CLEAN;
            (void)job.cb(&info);
            if (rc == 0) ++g_dbatch_cpu; else ++g_dbatch_failed;
        }
        g_dbatch_done = true;
    });
    return 0;
}
GRA_EXPORT void plugin_stop_batch_decompress(void)
{
    g_dbatch_stop = true;
    if (g_dbatch.joinable()) g_dbatch.join();
    g_dbatch_done = true;
}
// how the last decode batch went: files decoded on the GPU, handed back to the host's decoder, failed
GRA_EXPORT void grk_amd_plugin_batch_decode_counts(int32_t* gpu, int32_t* cpu, int32_t* failed)
{
    if (gpu) *gpu = g_dbatch_gpu.load();
    if (cpu) *cpu = g_dbatch_cpu.load();
    if (failed) *failed = g_dbatch_failed.load();
}

GRA_EXPORT uint32_t plugin_get_debug_state(void) { return g_debug_state; }
GRA_EXPORT void plugin_debug_mqc_next_cxd(void*, uint32_t) {}
GRA_EXPORT void plugin_debug_next_cxd(void*, uint32_t) {}
GRA_EXPORT void plugin_debug_mqc_next_plane(void*) {}

} // extern "C"
