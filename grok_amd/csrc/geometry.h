// grok_amd/csrc/geometry.h -- host-side tile geometry and quantiser parameters.
//
// Restates, for the hot path's case (dx=dy=1, a tile-component anywhere on the canonical grid, one
// precinct per resolution), what the reference derives in
//   tile/TileComponent.cpp:69-170 (resolutions / bands),  util/util.cpp:34-59 (band windows),
//   t1/T1Structs.cpp:109-136, :449-493 (precinct + code-block grid),
//   tile/TileComponentWindowBuffer.h:287-313 (block origin in the Mallat plane),
//   codestream/HTParams.cpp:248-312 + codestream/Quantizer.cpp:26-66 (exponents, Kmax, step).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/grok_amd.h"

namespace grk_amd {

struct BandGeom {
    uint8_t  orient;        // 0 LL 1 HL 2 LH 3 HH
    uint32_t x0, y0;        // origin in the band's own coordinates (0 for a tile at the origin)
    uint32_t w, h;          // band size
    uint32_t ox, oy;        // origin in the Mallat plane
    struct Prec { uint32_t gw, gh, first_block; };      // code-block grid of the band's part of a precinct (0 x 0: none),
    std::vector<Prec> prec;                             // first block (index within the component); [npw * nph] of the resolution
    uint8_t  kmax;          // numbps
    uint16_t qcd;           // SPqcd word (expn<<3 | or expn<<11|mant)
    float    stepsize;      // band->stepsize on the encoder side
    uint32_t first_block;   // index (within the component) of the band's first block
    uint32_t num_blocks;
};
struct ResGeom {
    uint32_t x0, y0;        // origin on the resolution's grid: its parity picks the lifting variant (odd start: the
                            // first sample is a high-pass one, WaveletFwd.cpp:884-905)
    uint32_t w, h;
    uint32_t ppx, ppy;      // precinct exponents of the resolution (15, 15: one precinct)
    uint32_t npw, nph;      // its precinct grid (0 x 0 for a resolution without samples)
    uint32_t num_bands;
    BandGeom band[3];
};
struct TileGeom {
    grk_amd_tile_params p;
    uint32_t stride;                 // plane row stride in elements
    uint64_t plane_elems;            // elements per plane
    std::vector<ResGeom> res;        // levels+1, coarsest first
    std::vector<grk_amd_block> blocks_comp0;   // blocks of one component
    uint32_t blocks_per_comp;
    uint16_t qcd_words[3 * GRK_AMD_MAX_LEVELS + 1];
    uint32_t num_bands_total;
};

// returns GRK_AMD_OK or an error code
int build_tile_geom(const grk_amd_tile_params& p, TileGeom& g);

// the same sub-band partition, block partition and lifting variants: what one batch of grk_amd_encode_tiles needs
bool same_geometry(const TileGeom& a, const TileGeom& b);

// decomposition level l (0 = the tile itself) is resolution L - l
inline const ResGeom& level_geom(const TileGeom& g, uint32_t l) { return g.res[g.p.num_levels - l]; }
inline bool on_even_grid(const TileGeom& g)
{
    for (uint32_t l = 0; l < g.p.num_levels; ++l) if ((level_geom(g, l).x0 | level_geom(g, l).y0) & 1u) return false;
    return true;
}

// t2_writer.cpp: a tile-part as literal bytes + a segment list (grk_amd_plan_tile_part, include/grok_amd.h); returns its length
int64_t plan_tile_part(const grk_amd_tile_params& p, uint32_t tile_index, uint32_t flags, const grk_amd_coded_block* tile_table,
                       std::vector<uint8_t>& lit, std::vector<grk_amd_tp_segment>& segs);

// ---- Tier-2 on the device (kernels_t2.hip; the host's share in t2_writer.cpp) -----------------------------------------------------
// One packet of a tile in progression order, as the header kernel meets it: up to three bands' code-block grids (rows of the
// tile's table, raster order within a band), each with the height of its tag trees and its Kmax, and where the packet's raw header
// bits and stuffed header bytes go in the tile's scratch areas (bounds that hold for any block lengths below 2^29).
struct T2Packet {
    uint32_t row0;                    // first row of the packet's component in the tile's table
    uint32_t nbands, nblocks;
    uint32_t first_block[3], gw[3], gh[3], kmax[3], height[3];
    uint32_t u_at, u_words;           // raw header bits: 32-bit words of the tile's scratch
    uint32_t h_at;                    // stuffed header: bytes of the tile's header scratch
};
struct T2Plan {
    std::vector<T2Packet> packets;
    std::vector<uint32_t> packet_of_block;        // [rows of a tile]
    uint32_t u_words = 0, h_bytes = 0;            // scratch per tile
};
constexpr uint32_t kT2MaxLenBits = 29;            // block lengths the device writer takes (a coded block is a few KB)
int t2_device_plan(const TileGeom& g, uint32_t flags, T2Plan& out);
inline uint32_t ceil_div_pow2(uint32_t v, uint32_t n) { return (uint32_t)(((uint64_t)v + (1ull << n) - 1) >> n); }

} // namespace grk_amd
