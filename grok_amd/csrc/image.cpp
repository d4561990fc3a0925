// grok_amd/csrc/image.cpp -- a whole image of any tile layout through the tile processor (host side, on top of the
// C ABI's own entry points).
//
// grk_amd_encode_tiles codes a batch of tiles that share ONE geometry.  An image whose tile grid does not sit on
// multiples of 2^levels x code-block size -- image offsets (grk_image x0 / y0), tile sizes such as 1000 x 1000, ragged
// last rows and columns -- has tiles of several geometries: sub-band sizes differ by a sample, bands start with partial
// code-blocks, and where a resolution begins on an odd coordinate the lifting starts with a high-pass sample
// (tile/TileProcessor.cpp:100-170 tile rectangle; tile/TileComponent.cpp:131-138 bands; WaveletFwd.cpp:884-905).
// Here the tiles are grouped by geometry, each group goes through grk_amd_encode_tiles as one batch, and the
// tile-parts are written in tile order (codestream/CodeStreamCompress.cpp:535-603).
#include "../../include/grok_amd.h"
#include "geometry.h"
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace grk_amd;

extern "C" int grk_amd_same_tile_geometry(const grk_amd_tile_params* a, const grk_amd_tile_params* b)
{
    if (!a || !b) return GRK_AMD_ERR_INVALID;
    TileGeom ga, gb;
    int rc = build_tile_geom(*a, ga); if (rc) return rc;
    rc = build_tile_geom(*b, gb); if (rc) return rc;
    return same_geometry(ga, gb) ? 1 : 0;
}

extern "C" int64_t grk_amd_encode_image(grk_amd_ctx* ctx, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                        const void* pixels, uint32_t flags, uint8_t* out, uint64_t cap)
{
    if (!ctx || !im || !base || !pixels || !out) return GRK_AMD_ERR_INVALID;
    const int64_t nt = grk_amd_layout_num_tiles(im);
    if (nt < 0) return nt;
    const uint32_t ntiles = (uint32_t)nt;
    const uint32_t W = im->x1 - im->x0, H = im->y1 - im->y0;
    const uint32_t bps = (base->prec + 7u) / 8u, nc = base->num_comps;
    std::vector<grk_amd_tile_params> tp(ntiles);
    std::vector<TileGeom> geoms;                       // one per group
    std::vector<std::vector<uint32_t>> groups;
    for (uint32_t t = 0; t < ntiles; ++t) {
        int rc = grk_amd_layout_tile(im, base, t, &tp[t]);
        if (rc) return rc;
        TileGeom g;
        rc = build_tile_geom(tp[t], g);
        if (rc) return rc;
        size_t k = 0;
        for (; k < geoms.size(); ++k) if (same_geometry(geoms[k], g)) break;
        if (k == geoms.size()) { geoms.push_back(std::move(g)); groups.emplace_back(); }
        groups[k].push_back(t);
    }
    // Tier-2 on the device (grk_amd_assemble_device; GRK_AMD_IMAGE_T2=host: the host writer below, which is also where a layout beyond
    // the device writer's tables goes): every group's finished tile-parts appended in the context's output buffer, then -- their sizes
    // known -- the main header and each tile-part fetched to its place
    const char* const et2 = std::getenv("GRK_AMD_IMAGE_T2");
    if (!(et2 && std::strcmp(et2, "host") == 0)) {
        std::vector<uint32_t> part_len(ntiles, 0);
        std::vector<uint64_t> dev_at(ntiles, 0);
        std::vector<uint8_t> staging;
        uint64_t used = 0;
        int64_t rc = GRK_AMD_OK;
        for (size_t k = 0; k < groups.size() && rc >= 0; ++k) {
            const auto& G = groups[k];
            const grk_amd_tile_params& p = tp[G[0]];
            const size_t tile_bytes = (size_t)p.tile_w * p.tile_h * nc * bps;
            staging.resize(tile_bytes * G.size());
            for (size_t i = 0; i < G.size(); ++i) {
                const grk_amd_tile_params& q = tp[G[i]];
                const size_t ox = q.tile_x0 - im->x0, oy = q.tile_y0 - im->y0;
                for (uint32_t c = 0; c < nc; ++c)
                    for (uint32_t y = 0; y < q.tile_h; ++y)
                        std::memcpy(&staging[i * tile_bytes + ((size_t)c * q.tile_h + y) * q.tile_w * bps],
                                    (const uint8_t*)pixels + (((size_t)c * H + oy + y) * W + ox) * bps, (size_t)q.tile_w * bps);
            }
            rc = grk_amd_encode_tiles(ctx, &p, (uint32_t)G.size(), staging.data(), 0, nullptr, nullptr);
            if (rc < 0) return rc;
            std::vector<uint32_t> lens(G.size());
            rc = grk_amd_assemble_device(ctx, &p, (uint32_t)G.size(), G.data(), flags, used, lens.data());
            if (rc < 0) break;
            for (size_t i = 0; i < G.size(); ++i) { part_len[G[i]] = lens[i]; dev_at[G[i]] = used; used += lens[i]; }
        }
        if (rc >= 0) {
            if ((flags & GRK_AMD_CS_TLM) && ntiles > 255) return GRK_AMD_ERR_UNSUPPORTED;
            const int64_t hdr = grk_amd_write_main_header_layout(im, base, flags, part_len.data(), out, cap);
            if (hdr < 0) return hdr;
            uint64_t at = (uint64_t)hdr;
            for (uint32_t t = 0; t < ntiles; ++t) at += part_len[t];
            if (at + 2 > cap) return GRK_AMD_ERR_OVERFLOW;
            at = (uint64_t)hdr;
            for (uint32_t t = 0; t < ntiles; ++t) {
                // (tile-parts that lie one behind the other on the device as in the file go in one piece)
                uint32_t t1 = t;
                uint64_t n = part_len[t];
                while (t1 + 1 < ntiles && dev_at[t1 + 1] == dev_at[t] + n) { n += part_len[t1 + 1]; ++t1; }
                const int fr = grk_amd_fetch_assembled(ctx, dev_at[t], n, out + at);
                if (fr) return fr;
                at += n; t = t1;
            }
            out[at++] = 0xFF; out[at++] = 0xD9;
            return (int64_t)at;
        }
        if (rc != GRK_AMD_ERR_UNSUPPORTED) return rc;
    }
    // per tile: its rows and where its group's coded bytes start in `coded`
    std::vector<std::vector<grk_amd_coded_block>> rows(ntiles);
    std::vector<uint8_t> coded, staging;
    for (size_t k = 0; k < groups.size(); ++k) {
        const auto& G = groups[k];
        const grk_amd_tile_params& p = tp[G[0]];
        const size_t tile_bytes = (size_t)p.tile_w * p.tile_h * nc * bps;
        staging.resize(tile_bytes * G.size());
        for (size_t i = 0; i < G.size(); ++i) {
            const grk_amd_tile_params& q = tp[G[i]];
            const size_t ox = q.tile_x0 - im->x0, oy = q.tile_y0 - im->y0;
            for (uint32_t c = 0; c < nc; ++c)
                for (uint32_t y = 0; y < q.tile_h; ++y)
                    std::memcpy(&staging[i * tile_bytes + ((size_t)c * q.tile_h + y) * q.tile_w * bps],
                                (const uint8_t*)pixels + (((size_t)c * H + oy + y) * W + ox) * bps, (size_t)q.tile_w * bps);
        }
        const uint64_t bpt = (uint64_t)geoms[k].blocks_per_comp * nc;
        std::vector<grk_amd_coded_block> table(bpt * G.size());
        uint64_t total = 0;
        int rc = grk_amd_encode_tiles(ctx, &p, (uint32_t)G.size(), staging.data(), 0, table.data(), &total);
        if (rc) return rc;
        const size_t at = coded.size();
        coded.resize(at + total);
        rc = grk_amd_fetch_coded(ctx, coded.data() + at, total);
        if (rc) return rc;
        for (size_t i = 0; i < G.size(); ++i) {
            rows[G[i]].assign(table.begin() + i * bpt, table.begin() + (i + 1) * bpt);
            for (auto& r : rows[G[i]]) r.offset += at;
        }
    }
    std::vector<grk_amd_coded_block> all;
    for (uint32_t t = 0; t < ntiles; ++t) all.insert(all.end(), rows[t].begin(), rows[t].end());
    return grk_amd_write_codestream_layout(im, base, all.data(), coded.data(), flags, out, cap);
}

// ---- sub-sampled components (4:2:2, 4:2:0, ...) ---------------------------------------------------------------------------
// Components of different size are tile-components of different geometry: a RUN of consecutive components with the same factors
// is coded as one unit -- the tile's rectangle in their coordinates, `n` components, the multiple component transform only for a
// run that holds components 0..2 --, the units of all tiles are grouped by geometry, every group is one grk_amd_encode_tiles batch,
// and the codestream writer takes each component with its own geometry (grk_amd_write_codestream_subsampled).
extern "C" int64_t grk_amd_encode_image_subsampled(grk_amd_ctx* ctx, const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                                   const uint8_t* comp_dx, const uint8_t* comp_dy, const void* pixels, uint32_t flags,
                                                   uint8_t* out, uint64_t cap)
{
    if (!ctx || !im || !base || !pixels || !out || !comp_dx || !comp_dy) return GRK_AMD_ERR_INVALID;
    const int64_t nt = grk_amd_layout_num_tiles(im);
    if (nt < 0) return nt;
    const uint32_t ntiles = (uint32_t)nt, nc = base->num_comps;
    const uint32_t bps = (base->prec + 7u) / 8u;
    for (uint32_t c = 0; c < nc; ++c) if (!comp_dx[c] || !comp_dy[c]) return GRK_AMD_ERR_INVALID;
    struct Run { uint32_t c0, n; };
    std::vector<Run> runs;
    for (uint32_t c = 0; c < nc; ++c) {
        if (!runs.empty() && comp_dx[c] == comp_dx[runs.back().c0] && comp_dy[c] == comp_dy[runs.back().c0]) runs.back().n++;
        else runs.push_back(Run{c, 1});
    }
    // (MCT over components of different size: switched off, as the reference does with a warning, CodeStreamCompress.cpp:434-447)
    const bool mct = base->mct && nc >= 3 && runs[0].n >= 3;
    // the image's components in `pixels`: component c is ceil(x1 / dx) - ceil(x0 / dx) columns wide, planes back to back
    auto cdiv = [](uint64_t a, uint64_t b) { return (a + b - 1) / b; };
    std::vector<uint64_t> cw(nc), ch(nc), cx0(nc), cy0(nc), plane_at(nc + 1, 0);
    for (uint32_t c = 0; c < nc; ++c) {
        cx0[c] = cdiv(im->x0, comp_dx[c]); cy0[c] = cdiv(im->y0, comp_dy[c]);
        cw[c] = cdiv(im->x1, comp_dx[c]) - cx0[c]; ch[c] = cdiv(im->y1, comp_dy[c]) - cy0[c];
        plane_at[c + 1] = plane_at[c] + cw[c] * ch[c] * bps;
    }
    struct Unit { uint32_t tile, run; grk_amd_tile_params p; size_t group; };
    std::vector<Unit> units;
    std::vector<TileGeom> geoms;
    std::vector<grk_amd_tile_params> gparams;
    std::vector<std::vector<size_t>> groups;
    for (uint32_t t = 0; t < ntiles; ++t)
        for (uint32_t k = 0; k < runs.size(); ++k) {
            Unit u{t, k, {}, 0};
            int rc = grk_amd_layout_tile_comp(im, base, comp_dx[runs[k].c0], comp_dy[runs[k].c0], t, &u.p);
            if (rc) return rc;
            u.p.num_comps = (uint16_t)runs[k].n;
            u.p.mct = (mct && k == 0) ? 1 : 0;
            TileGeom g;
            rc = build_tile_geom(u.p, g);
            if (rc) return rc;
            size_t gi = 0;
            for (; gi < geoms.size(); ++gi)
                if (gparams[gi].num_comps == u.p.num_comps && gparams[gi].mct == u.p.mct && same_geometry(geoms[gi], g)) break;
            if (gi == geoms.size()) { geoms.push_back(std::move(g)); gparams.push_back(u.p); groups.emplace_back(); }
            u.group = gi;
            groups[gi].push_back(units.size());
            units.push_back(u);
        }
    std::vector<std::vector<grk_amd_coded_block>> rows(units.size());
    std::vector<uint8_t> coded, staging;
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const auto& G = groups[gi];
        const grk_amd_tile_params& p = units[G[0]].p;
        const size_t unit_bytes = (size_t)p.tile_w * p.tile_h * p.num_comps * bps;
        staging.resize(unit_bytes * G.size());
        for (size_t i = 0; i < G.size(); ++i) {
            const Unit& u = units[G[i]];
            for (uint32_t k = 0; k < u.p.num_comps; ++k) {
                const uint32_t c = runs[u.run].c0 + k;
                const size_t ox = u.p.tile_x0 - cx0[c], oy = u.p.tile_y0 - cy0[c];
                for (uint32_t y = 0; y < u.p.tile_h; ++y)
                    std::memcpy(&staging[i * unit_bytes + ((size_t)k * u.p.tile_h + y) * u.p.tile_w * bps],
                                (const uint8_t*)pixels + plane_at[c] + ((oy + y) * cw[c] + ox) * bps, (size_t)u.p.tile_w * bps);
            }
        }
        const uint64_t bpu = (uint64_t)geoms[gi].blocks_per_comp * p.num_comps;
        std::vector<grk_amd_coded_block> table(bpu * G.size());
        uint64_t total = 0;
        int rc = grk_amd_encode_tiles(ctx, &p, (uint32_t)G.size(), staging.data(), 0, table.data(), &total);
        if (rc) return rc;
        const size_t at = coded.size();
        coded.resize(at + total);
        rc = grk_amd_fetch_coded(ctx, coded.data() + at, total);
        if (rc) return rc;
        for (size_t i = 0; i < G.size(); ++i) {
            rows[G[i]].assign(table.begin() + i * bpu, table.begin() + (i + 1) * bpu);
            for (auto& r : rows[G[i]]) r.offset += at;
        }
    }
    std::vector<grk_amd_coded_block> all;                       // tile-major, within a tile component-major (the runs in order)
    for (size_t i = 0; i < units.size(); ++i) all.insert(all.end(), rows[i].begin(), rows[i].end());
    return grk_amd_write_codestream_subsampled(im, base, comp_dx, comp_dy, all.data(), coded.data(), flags, out, cap);
}
