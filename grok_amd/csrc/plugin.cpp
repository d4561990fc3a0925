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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>

namespace {

grk_amd_ctx* g_ctx = nullptr;
bool g_verbose = false;
std::mutex g_mu;

// ---- the tile tree ------------------------------------------------------------------------------
struct TileOwner {
    gra_plugin_tile tile{};
    std::vector<gra_plugin_tile_component> comps;   std::vector<gra_plugin_tile_component*> comp_ptr;
    std::vector<gra_plugin_resolution> ress;        std::vector<gra_plugin_resolution*> res_ptr;
    std::vector<gra_plugin_band> bands;             std::vector<gra_plugin_band*> band_ptr;
    std::vector<gra_plugin_precinct> precs;         std::vector<gra_plugin_precinct*> prec_ptr;
    std::vector<gra_plugin_code_block> blocks;      std::vector<gra_plugin_code_block*> block_ptr;
    std::vector<uint8_t> coded;
};
static_assert(offsetof(TileOwner, tile) == 0, "tile must be the first member: destroy() casts back");

gra_plugin_tile* build_tree(const grk_amd_tile_params& p, const std::vector<grk_amd_block>& layout,
                            const std::vector<grk_amd_coded_block>& table, std::vector<uint8_t>&& coded)
{
    auto* o = new TileOwner();
    o->coded = std::move(coded);
    const size_t nb = layout.size();
    const uint32_t nres = p.num_levels + 1u;
    const size_t nbands_c = 3 * p.num_levels + 1;
    o->comps.resize(p.num_comps); o->comp_ptr.resize(p.num_comps);
    o->ress.resize((size_t)p.num_comps * nres); o->res_ptr.resize(o->ress.size());
    o->bands.resize((size_t)p.num_comps * nbands_c); o->band_ptr.resize(o->bands.size());
    o->precs.resize(o->bands.size()); o->prec_ptr.resize(o->bands.size());
    o->blocks.resize(nb); o->block_ptr.resize(nb);
    std::memset(o->blocks.data(), 0, nb * sizeof(gra_plugin_code_block));
    for (size_t i = 0; i < nb; ++i) {
        const grk_amd_block& b = layout[i];
        gra_plugin_code_block& cb = o->blocks[i];
        cb.x0 = b.x0; cb.y0 = b.y0; cb.x1 = b.x1; cb.y1 = b.y1;
        cb.numPix = (b.x1 - b.x0) * (b.y1 - b.y0);
        cb.compressedData = o->coded.data() + table[i].offset;
        cb.compressedDataLength = table[i].length;
        cb.numBitPlanes = 1;                     // T1HT::compress sets cblk->numbps = 1 (T1HT.cpp:123)
        cb.numPasses = 1;
        cb.passes[0].rate = table[i].length ? table[i].length - 1 : 0;   // host uses rate+1 (plugin_bridge.cpp:230)
        cb.passes[0].length = table[i].length;
        cb.passes[0].distortionDecrease = 0.0;
        o->block_ptr[i] = &cb;
    }
    size_t bi = 0, blk = 0;
    for (uint32_t c = 0; c < p.num_comps; ++c) {
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
                B.numPrecincts = 1;
                B.precincts = &o->prec_ptr[bi];
                o->prec_ptr[bi] = &o->precs[bi];
                // blocks of this band are contiguous in enumeration order
                size_t first = blk;
                while (blk < nb && layout[blk].comp == c && layout[blk].res == r && layout[blk].band == orient) ++blk;
                o->precs[bi].numBlocks = blk - first;
                o->precs[bi].blocks = first < nb ? &o->block_ptr[first] : nullptr;
                B.stepsize = blk > first ? layout[first].stepsize : 1.0f;
            }
        }
    }
    o->tile.decompress_flags = 0;
    o->tile.numComponents = p.num_comps;
    o->tile.tileComponents = o->comp_ptr.data();
    return &o->tile;
}

// ---- minimal PNM (P5/P6, binary) reader: enough for plugin_encode's "read params->infile" -----------
bool read_pnm(const char* path, std::vector<uint8_t>& planar, uint32_t& w, uint32_t& h, uint32_t& comps, uint32_t& prec)
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
    planar.resize(raw.size());
    for (uint32_t c = 0; c < comps; ++c)
        for (size_t i = 0; i < n; ++i) {
            if (bps == 1) planar[c * n + i] = raw[i * comps + c];
            else {   // PNM 16-bit is big endian; the tile buffer is host endian
                const uint8_t* s = &raw[(i * comps + c) * 2];
                reinterpret_cast<uint16_t*>(planar.data())[c * n + i] = (uint16_t)((s[0] << 8) | s[1]);
            }
        }
    return true;
}

bool params_from_cparameters(const gra_cparameters* cp, uint32_t w, uint32_t h, uint32_t comps, uint32_t prec,
                             grk_amd_tile_params& p)
{
    if (!cp->isHT || !(cp->cblk_sty & GRA_CBLKSTY_HT)) return false;             // hot path = HTJ2K only
    if (cp->tile_size_on && (cp->t_width < w || cp->t_height < h)) return false;    // single tile (D3)
    if (cp->tcp_numlayers > 1 || cp->numpocs || cp->res_spec || cp->roi_compno >= 0) return false;
    if (cp->subsampling_dx != 1 || cp->subsampling_dy != 1 || cp->image_offset_x0 || cp->image_offset_y0) return false;
    if (cp->numresolution < 1 || cp->numresolution > GRK_AMD_MAX_LEVELS + 1) return false;
    auto lg = [](uint32_t v) { int e = 0; while ((1u << e) < v) ++e; return e; };
    std::memset(&p, 0, sizeof p);
    p.tile_w = w; p.tile_h = h; p.num_comps = (uint16_t)comps; p.prec = (uint8_t)prec; p.sgnd = 0;
    p.irreversible = cp->irreversible ? 1 : 0;
    p.mct = cp->tcp_mct ? 1 : 0;
    p.num_levels = (uint8_t)(cp->numresolution - 1);
    p.cblk_w_exp = (uint8_t)lg(cp->cblockw_init ? cp->cblockw_init : 64);
    p.cblk_h_exp = (uint8_t)lg(cp->cblockh_init ? cp->cblockh_init : 64);
    return grk_amd_tile_num_blocks(&p) > 0;
}

int32_t encode_file(gra_cparameters* cp, const char* in, const char* out, gra_encode_callback cb)
{
    if (!g_ctx || !cp || !cb) return -1;
    std::vector<uint8_t> px;
    uint32_t w, h, comps, prec;
    if (!read_pnm(in, px, w, h, comps, prec)) return -1;
    grk_amd_tile_params p;
    if (!params_from_cparameters(cp, w, h, comps, prec, p)) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    gra_plugin_tile* tile = grk_amd_plugin_tile_create(g_ctx, &p, px.data(), 0);
    if (!tile) return -1;
    gra_encode_callback_info info{};
    info.input_file_name = in;
    info.outputFileNameIsRelative = false;
    info.output_file_name = out;
    info.compressor_parameters = cp;
    info.image = nullptr;                 // the host callback loads the image itself (grk_compress.cpp:1636)
    info.tile = tile;
    info.error_code = 0;
    cb(&info);
    grk_amd_plugin_tile_destroy(tile);
    return info.error_code;
}

// ---- batch mode: a worker thread walks the input directory ----------------------------------------
std::thread g_batch;
std::atomic<bool> g_batch_done{true}, g_batch_stop{false};

int32_t plugin_exit() { return 0; }
void* plugin_create(gra_minpf_object_params*) { return nullptr; }
int32_t plugin_destroy(void*) { return 0; }

} // namespace

extern "C" {

#define GRA_EXPORT __attribute__((visibility("default")))

GRA_EXPORT gra_plugin_tile* grk_amd_plugin_tile_create(grk_amd_ctx* ctx, const grk_amd_tile_params* p,
                                                       const void* pixels, int on_device)
{
    if (!ctx || !p || !pixels) return nullptr;
    const int64_t nb = grk_amd_tile_num_blocks(p);
    if (nb <= 0) return nullptr;
    std::vector<grk_amd_block> layout((size_t)nb);
    if (grk_amd_tile_layout(p, layout.data(), (uint64_t)nb, nullptr) != nb) return nullptr;
    std::vector<grk_amd_coded_block> table((size_t)nb);
    uint64_t total = 0;
    if (grk_amd_encode_tiles(ctx, p, 1, pixels, on_device, table.data(), &total) != GRK_AMD_OK) return nullptr;
    for (const auto& t : table)
        if (t.length > 65535) return nullptr;          // host keeps rates in uint16_t (plugin_bridge.cpp:174, D7)
    std::vector<uint8_t> coded(total ? total : 1);
    if (total && grk_amd_fetch_coded(ctx, coded.data(), total) != GRK_AMD_OK) return nullptr;
    return build_tree(*p, layout, table, std::move(coded));
}

GRA_EXPORT void grk_amd_plugin_tile_destroy(gra_plugin_tile* tile)
{
    delete reinterpret_cast<TileOwner*>(tile);
}

GRA_EXPORT int grk_amd_plugin_tile_decode(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const gra_plugin_tile* tile,
                                          void* pixels, int pixels_on_device)
{
    if (!ctx || !p || !tile || !pixels) return GRK_AMD_ERR_INVALID;
    const int64_t nb = grk_amd_tile_num_blocks(p);
    if (nb <= 0) return (int)(nb ? nb : GRK_AMD_ERR_UNSUPPORTED);
    std::vector<grk_amd_block> layout((size_t)nb);
    if (grk_amd_tile_layout(p, layout.data(), (uint64_t)nb, nullptr) != nb) return GRK_AMD_ERR_INVALID;
    if (tile->numComponents != p->num_comps) return GRK_AMD_ERR_INVALID;
    // walk the tree in the enumeration order both sides share: comp -> resolution -> band -> precinct -> block
    std::vector<grk_amd_coded_block> table((size_t)nb);
    std::vector<uint8_t> coded;
    size_t i = 0;
    for (uint32_t c = 0; c < tile->numComponents; ++c) {
        const gra_plugin_tile_component* tc = tile->tileComponents[c];
        for (uint32_t r = 0; r < tc->numResolutions; ++r) {
            const gra_plugin_resolution* res = tc->resolutions[r];
            for (uint32_t b = 0; b < res->numBands; ++b) {
                const gra_plugin_band* band = res->band[b];
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
                        else row.missing_msbs = layout[i].kmax >= nbp ? layout[i].kmax - nbp : 0;   // band numbps - block numbps
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
    return grk_amd_decode_tiles(ctx, p, 1, table.data(), coded.data(), coded.size(), 0, pixels, pixels_on_device);
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
    if (g_ctx) return true;
    const int rc = grk_amd_create(info.deviceId, info.verbose ? 1 : 0, &g_ctx);
    if (rc != GRK_AMD_OK) {
        if (g_verbose) std::fprintf(stderr, "[grok_amd plugin] no usable MI355X (rc=%d): host falls back to CPU\n", rc);
        g_ctx = nullptr;
        return false;
    }
    return true;
}

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
        if (DIR* d = opendir(in.c_str())) {
            while (dirent* e = readdir(d)) {
                if (g_batch_stop.load()) break;
                std::string name(e->d_name);
                const size_t dot = name.rfind('.');
                if (dot == std::string::npos) continue;
                const std::string ext = name.substr(dot);
                if (ext != ".pgm" && ext != ".ppm" && ext != ".pnm") continue;
                const std::string src = in + "/" + name, dst = out + "/" + name.substr(0, dot) + ".j2k";
                encode_file(cp, src.c_str(), dst.c_str(), callback);
            }
            closedir(d);
        }
        g_batch_done = true;
    });
    return 0;
}

GRA_EXPORT bool plugin_is_batch_complete(void) { return g_batch_done.load(); }

GRA_EXPORT void plugin_stop_batch_encode(void)
{
    g_batch_stop = true;
    if (g_batch.joinable()) g_batch.join();
    g_batch_done = true;
}

// Decode side is not on the device yet (SURVEY.md §8a rows a13-a17 are "next"): decline, the host
// keeps its CPU decoder (grk_decompress.cpp falls back when the plugin returns non-zero).
GRA_EXPORT int32_t plugin_decompress(void*, gra_decode_callback) { return -1; }
GRA_EXPORT int32_t plugin_init_batch_decompress(const char*, const char*, void*, gra_decode_callback) { return -1; }
GRA_EXPORT int32_t plugin_batch_decompress(void) { return -1; }
GRA_EXPORT void plugin_stop_batch_decompress(void) {}

GRA_EXPORT uint32_t plugin_get_debug_state(void) { return GRA_PLUGIN_STATE_NO_DEBUG; }
GRA_EXPORT void plugin_debug_mqc_next_cxd(void*, uint32_t) {}
GRA_EXPORT void plugin_debug_next_cxd(void*, uint32_t) {}
GRA_EXPORT void plugin_debug_mqc_next_plane(void*) {}

} // extern "C"
