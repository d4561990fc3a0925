// grok_amd/csrc/t2_writer.cpp -- host-side Tier-2 + codestream assembly.
//
// The Grok plugin protocol hands ONE grk_plugin_tile to every tile of an image (defect D3), so
// multi-tile images -- the tile-sharded multi-GPU configuration -- cannot go through
// grk_compress_with_plugin.  This writer produces the same bytes the reference would:
//   main header : CodeStreamCompress.cpp:722-753 (SOC, SIZ markers/SIZMarker.cpp, CAP :936-981,
//                 COD :1060-1105, QCD Quantizer.cpp:87-121, COM :296-309)
//   tile part   : SOT markers/SOTMarker.cpp:41-72, SOD, then one packet per (resolution, component)
//                 in LRCP order (t2/PacketIter.cpp:805-829) -- 1 layer, 1 precinct per resolution,
//                 no SOP/EPH, every code-block included with its single HT cleanup pass
//   packet      : T2Compress::compressPacket t2/T2Compress.cpp:123-333; tag trees t1/TagTree.cpp:170-218;
//                 bit stuffing t1/BitIO.cpp:46-175
// It is O(#code-blocks) host work plus one memcpy of the coded bytes.
#include "../../include/grok_amd.h"
#include "geometry.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

using namespace grk_amd;

// p == nullptr: only counts (ovf stays false).  PLAN mode (lit != nullptr): nothing is copied -- what the writer writes itself (marker
// segments, packet headers) collects in `lit`, and the output is described as a list of segments, each either a run of those
// literal bytes or a code-block's bytes in the coded buffer (grk_amd_plan_tile_part: the caller places them, on whatever
// threads or device it likes, once it knows where the tile-part goes).
struct Out {
    uint8_t* p; uint64_t cap; uint64_t n = 0; bool ovf = false;
    std::vector<uint8_t>* lit = nullptr; std::vector<grk_amd_tp_segment>* segs = nullptr;
    void u8(uint32_t v)
    {
        if (lit) {
            if (segs->empty() || segs->back().kind != 0 || segs->back().dst + segs->back().len != n)
                segs->push_back(grk_amd_tp_segment{n, (uint64_t)lit->size(), 0u, 0u});
            lit->push_back((uint8_t)v); segs->back().len++;
        } else if (p) { if (n < cap) p[n] = (uint8_t)v; else ovf = true; }
        ++n;
    }
    void u16(uint32_t v) { u8(v >> 8); u8(v & 0xFF); }
    void u32(uint32_t v) { u16(v >> 16); u16(v & 0xFFFF); }
    void bytes(const uint8_t* s, uint64_t len)               // bytes of the writer's own
    {
        if (lit) { for (uint64_t i = 0; i < len; ++i) u8(s[i]); return; }
        if (p) { if (n + len <= cap) std::memcpy(p + n, s, len); else ovf = true; }
        n += len;
    }
    void body(const uint8_t* coded, uint64_t off, uint64_t len)   // a code-block's bytes
    {
        if (lit) { if (len) segs->push_back(grk_amd_tp_segment{n, off, (uint32_t)len, 1u}); n += len; return; }
        if (p) { if (n + len <= cap) std::memcpy(p + n, coded + off, len); else ovf = true; }
        n += len;
    }
    // (a patch lies in the SOT marker segment or the TLM: literal bytes written in one run -- `lit_at` = where that run began in lit,
    //  `run_dst` = its first byte's place in the output)
    void patch32(uint64_t at, uint32_t v, uint64_t lit_at = 0, uint64_t run_dst = 0)
    {
        uint8_t* q = nullptr;
        if (lit) { const uint64_t i = lit_at + (at - run_dst); if (i + 4 <= lit->size()) q = lit->data() + i; }
        else if (p && at + 4 <= cap) q = p + at;
        if (q) { q[0] = (uint8_t)(v >> 24); q[1] = (uint8_t)(v >> 16); q[2] = (uint8_t)(v >> 8); q[3] = (uint8_t)v; }
    }
};

// MSB-first packet-header bit writer: a byte following 0xFF carries 7 bits (BitIO.cpp:46-175).  Bits collect in a 64-bit
// register and leave a byte at a time; BitIO's own form (one call per bit, the byte written when the NEXT bit arrives) gives the same
// bytes: a full byte is the same byte whenever it is written, and its flush -- the byte in work as it is, then an empty byte if that
// one was 0xFF -- is flush() below.
struct HeaderBits {
    Out& o; uint64_t acc = 0; int n = 0; uint32_t last = 0;
    explicit HeaderBits(Out& out) : o(out) {}
    void drain()
    {
        for (;;) {
            const int take = last == 0xFF ? 7 : 8;
            if (n < take) break;
            last = (uint32_t)(acc >> (n - take)) & ((1u << take) - 1u);
            o.u8(last);
            n -= take;
        }
    }
    void put(uint32_t v, int k) { acc = (acc << k) | v; n += k; drain(); }      // k <= 32 bits of v, MSB first
    void bit(uint32_t b) { put(b & 1u, 1); }
    void ones(int k) { put((uint32_t)((1ull << k) - 1ull), k); }                  // k <= 32
    void zeros(int k) { while (k > 0) { const int t = k > 32 ? 32 : k; put(0, t); k -= t; } }
    void comma(int k) { while (k > 31) { ones(31); k -= 31; } put(((1u << k) - 1u) << 1, k + 1); }   // k ones, then a zero
    void flush()
    {
        if (n) { const int take = last == 0xFF ? 7 : 8; last = (uint32_t)(acc & ((1ull << n) - 1ull)) << (take - n); o.u8(last); n = 0; }
        else if (last == 0xFF) { o.u8(0); last = 0; }
    }
};

// Tag trees (ISO 15444-1 B.10.2; t1/TagTree.cpp:170-218) as this writer meets them.  A packet of the single layer includes every
// code-block, and every block of a band has the same number of missing bit-planes (numbps = 1 is signalled for an HT block,
// T1HT.cpp:123: Kmax - 1 of them): BOTH trees of a band's precinct -- inclusion, all leaves 0 against threshold 1; zero bit-planes,
// all leaves v -- are uniform.  Coding leaf (x, y) then emits, from the root down, one '1' for every node on its path that no
// earlier leaf (raster order) has passed -- the nodes (x >> l, y >> l) with the low l bits of x and y zero: 1 + min(ctz x, ctz y)
// of them, at most the tree's height -- and, for the zero-bit-plane tree's root alone, v '0's in front of its '1'.  (The general
// procedure -- per node a lower bound raised bit by bit to the threshold -- reduces to exactly this when no node's value is
// below its parent's.)
inline int tag_tree_height(uint32_t gw, uint32_t gh)
{
    int h = 1;
    while ((uint64_t)gw * gh > 1) { gw = (gw + 1) >> 1; gh = (gh + 1) >> 1; ++h; }
    return h;
}
inline int new_nodes(uint32_t x, uint32_t y, int height)
{
    const uint32_t m = x | y;
    const int z = m ? __builtin_ctz(m) : 31;
    return z + 1 < height ? z + 1 : height;
}

int floor_log2(uint32_t v) { int r = 0; while (v >>= 1) ++r; return r; }

// tlm_at: where the Ptlm fields of the TLM marker segment start (0: none written)
void write_main_header(Out& o, const TileGeom& g, const grk_amd_image_layout& im, uint32_t flags, uint32_t ntiles, uint64_t* tlm_at,
                       const uint8_t* comp_dx = nullptr, const uint8_t* comp_dy = nullptr)
{
    if (tlm_at) *tlm_at = 0;
    const grk_amd_tile_params& p = g.p;
    o.u16(0xFF4F);                                                     // SOC
    o.u16(0xFF51); o.u16(38 + 3 * p.num_comps); o.u16(0x4000);         // SIZ, Rsiz: HTJ2K (Part 15)
    o.u32(im.x1); o.u32(im.y1); o.u32(im.x0); o.u32(im.y0);            // Xsiz Ysiz XOsiz YOsiz (markers/SIZMarker.cpp)
    o.u32(im.t_width); o.u32(im.t_height); o.u32(im.tx0); o.u32(im.ty0);
    o.u16(p.num_comps);
    for (uint32_t c = 0; c < p.num_comps; ++c) {                       // Ssiz, XRsiz, YRsiz
        o.u8((p.prec - 1) | (p.sgnd ? 0x80 : 0)); o.u8(comp_dx && comp_dx[c] ? comp_dx[c] : 1); o.u8(comp_dy && comp_dy[c] ? comp_dy[c] : 1);
    }
    // CAP (CodeStreamCompress.cpp:936-981; MAGBp HTParams.cpp:313-329)
    uint32_t B = 0;
    const uint32_t nb = 3 * p.num_levels + 1;
    for (uint32_t i = 0; i < nb; ++i) {
        if (!p.irreversible) B = std::max<uint32_t>(B, (uint32_t)(g.qcd_words[i] >> 3) + 1 - 1);
        else {
            uint32_t lev = p.num_levels - (i ? (i - 1) / 3 : 0);
            B = std::max<uint32_t>(B, (uint32_t)(g.qcd_words[i] >> 11) + 1 - lev);
        }
    }
    uint32_t Bp = B <= 8 ? 0 : (B < 28 ? B - 8 : (B < 48 ? 13 + (B >> 2) : 31));
    o.u16(0xFF50); o.u16(8); o.u32(0x00020000); o.u16((p.irreversible ? 0x0020 : 0) | Bp);
    // COD: Scod = SOP (2) | EPH (4), SGcod = progression order, one layer, MCT
    bool prt = false;                       // user-defined precincts: Scod bit 0 and one size byte per resolution
    for (uint32_t r = 0; r <= p.num_levels; ++r) prt = prt || p.precinct_exp[r] != 0;
    o.u16(0xFF52); o.u16(12 + (prt ? p.num_levels + 1u : 0u));
    o.u8((prt ? 1u : 0u) | ((flags & GRK_AMD_CS_SOP) ? 2u : 0u) | ((flags & GRK_AMD_CS_EPH) ? 4u : 0u));
    o.u8((flags >> GRK_AMD_CS_PROG_SHIFT) & 7u); o.u16(1); o.u8(p.mct ? 1 : 0);
    o.u8(p.num_levels); o.u8(p.cblk_w_exp - 2); o.u8(p.cblk_h_exp - 2); o.u8(0x40); o.u8(p.irreversible ? 0 : 1);
    for (uint32_t r = 0; prt && r <= p.num_levels; ++r) o.u8(p.precinct_exp[r] ? p.precinct_exp[r] : 0xFFu);
    // QCD: one guard bit
    if (!p.irreversible) {
        o.u16(0xFF5C); o.u16(3 + nb); o.u8(0x20);
        for (uint32_t i = 0; i < nb; ++i) o.u8(g.qcd_words[i] & 0xFF);
    } else {
        o.u16(0xFF5C); o.u16(3 + 2 * nb); o.u8(0x22);
        for (uint32_t i = 0; i < nb; ++i) o.u16(g.qcd_words[i]);
    }
    // TLM (markers/LengthMarkers.cpp:166-189, written between QCD and COM: CodeStreamCompress.cpp:734): Ztlm 0, Stlm 0x50 =
    // one-byte tile index + four-byte tile-part length per tile-part; the lengths are patched in once they are known
    if (flags & GRK_AMD_CS_TLM) {
        o.u16(0xFF55); o.u16(4 + 5 * ntiles); o.u8(0); o.u8(0x50);
        if (tlm_at) *tlm_at = o.n;
        for (uint32_t t = 0; t < ntiles; ++t) { o.u8(t); o.u32(0); }
    }
    // COM (the reference's default comment, so that whole files compare byte for byte)
    static const char kCom[] = "Created by Grok     version 8.0.2";
    const uint32_t cl = (uint32_t)std::strlen(kCom);
    o.u16(0xFF64); o.u16(4 + cl); o.u16(1);
    o.bytes(reinterpret_cast<const uint8_t*>(kCom), cl);
}

// sop: the packet's number in the tile (SOP marker segment in front, T2Compress.cpp:149-164) or < 0; eph: EPH after the header
// the packet of precinct `pi` of resolution r
void write_packet(Out& o, const TileGeom& g, uint32_t r, uint32_t pi, const grk_amd_coded_block* comp_table, const uint8_t* coded,
                  int32_t sop = -1, bool eph = false)
{
    const ResGeom& R = g.res[r];
    if (sop >= 0) { o.u16(0xFF91); o.u16(4); o.u16((uint32_t)sop & 0xFFFFu); }
    HeaderBits hb(o);
    hb.bit(1);
    for (uint32_t bi = 0; bi < R.num_bands; ++bi) {
        const BandGeom& B = R.band[bi];
        const BandGeom::Prec& P = B.prec[pi];
        if (!P.gw || !P.gh) continue;
        const int height = tag_tree_height(P.gw, P.gh);
        const grk_amd_coded_block* cb = comp_table + P.first_block;
        for (uint32_t y = 0; y < P.gh; ++y)
            for (uint32_t x = 0; x < P.gw; ++x, ++cb) {
                const int nn = new_nodes(x, y, height);
                hb.ones(nn);                                          // inclusion: the path's new nodes
                if (!(x | y)) hb.zeros((int)B.kmax - 1);              // zero bit-planes: the root's value ...
                hb.ones(nn);                                          // ... and the path's new nodes
                const uint32_t len = cb->length;
                int inc = floor_log2(len) + 1 - 3;
                if (inc < 0) inc = 0;
                hb.put(0, 1);                                         // one coding pass
                hb.comma(inc);                                        // Lblock raised from 3 to what the length needs
                hb.put(len, 3 + inc);
            }
    }
    hb.flush();
    if (eph) o.u16(0xFF92);
    for (uint32_t bi = 0; bi < R.num_bands; ++bi) {
        const BandGeom::Prec& P = R.band[bi].prec[pi];
        for (uint32_t k = 0; k < P.gw * P.gh; ++k) {
            const grk_amd_coded_block& cb = comp_table[P.first_block + k];
            o.body(coded, cb.offset, cb.length);
        }
    }
}

// The tiles of an image (ISO 15444-1 B.3; the reference: TileProcessor::init, tile/TileProcessor.cpp:100-170): tile
// (tx, ty) of the grid anchored at (tx0, ty0) is its cell clipped to the image area.
struct Layout { grk_amd_image_layout im; uint32_t tcols, trows; };
int check_layout(const grk_amd_image_layout* im, Layout& l)
{
    if (!im || im->x1 <= im->x0 || im->y1 <= im->y0 || !im->t_width || !im->t_height) return GRK_AMD_ERR_INVALID;
    if (im->tx0 > im->x0 || im->ty0 > im->y0) return GRK_AMD_ERR_INVALID;                              // B.3: XTOsiz <= XOsiz
    if ((uint64_t)im->tx0 + im->t_width <= im->x0 || (uint64_t)im->ty0 + im->t_height <= im->y0) return GRK_AMD_ERR_INVALID;
    l.im = *im;
    l.tcols = (uint32_t)(((uint64_t)im->x1 - im->tx0 + im->t_width - 1) / im->t_width);
    l.trows = (uint32_t)(((uint64_t)im->y1 - im->ty0 + im->t_height - 1) / im->t_height);
    if ((uint64_t)l.tcols * l.trows > 65535) return GRK_AMD_ERR_UNSUPPORTED;
    return GRK_AMD_OK;
}
void tile_of(const Layout& l, const grk_amd_tile_params& base, uint32_t t, grk_amd_tile_params& p)
{
    const uint32_t tx = t % l.tcols, ty = t / l.tcols;
    const uint64_t cx0 = (uint64_t)l.im.tx0 + (uint64_t)tx * l.im.t_width, cy0 = (uint64_t)l.im.ty0 + (uint64_t)ty * l.im.t_height;
    const uint64_t x0 = std::max<uint64_t>(cx0, l.im.x0), y0 = std::max<uint64_t>(cy0, l.im.y0);
    const uint64_t x1 = std::min<uint64_t>(cx0 + l.im.t_width, l.im.x1), y1 = std::min<uint64_t>(cy0 + l.im.t_height, l.im.y1);
    p = base;
    p.tile_x0 = (uint32_t)x0; p.tile_y0 = (uint32_t)y0;
    p.tile_w = (uint32_t)(x1 - x0); p.tile_h = (uint32_t)(y1 - y0);
}
grk_amd_image_layout plain_layout(const grk_amd_tile_params& p, uint32_t img_w, uint32_t img_h)
{
    return grk_amd_image_layout{p.tile_x0, p.tile_y0, p.tile_x0 + img_w, p.tile_y0 + img_h, p.tile_x0, p.tile_y0, p.tile_w, p.tile_h};
}

// SOT, (PLT,) SOD and the packets of one tile in the progression order of `flags`; returns the tile-part's length.
// One layer.  `cg[c]` is the geometry of component c's tile-component (all the same object for an image without sub-sampling),
// `row0[c]` the first of its rows in the tile's table.  The five orders (ISO 15444-1 B.12.1; t2/PacketIter.cpp:805-1100) are
//   LRCP, RLCP  resolution -> component -> precinct (raster)
//   RPCL        resolution -> precinct position -> component
//   PCRL        precinct position -> component -> resolution
//   CPRL        component -> precinct position -> resolution
// where a precinct's position is its top-left corner on the REFERENCE grid (component coordinates times the component's
// sub-sampling factors), clipped to the tile -- the (y, x) at which the standard's position loops meet it; positions are walked
// in raster order, several resolutions / components can share one.
struct Pk { uint32_t c, r, pi; uint64_t x, y; };
std::vector<Pk> packet_order(const std::vector<const TileGeom*>& cg, const uint8_t* comp_dx, const uint8_t* comp_dy, uint32_t gx0, uint32_t gy0,
                             uint32_t order)
{
    const uint32_t ncomp = (uint32_t)cg.size();
    const grk_amd_tile_params& p = cg[0]->p;
    std::vector<Pk> prec;                                   // every precinct of the tile: component-major, resolution-major, raster
    for (uint32_t c = 0; c < ncomp; ++c) {
        const TileGeom& g = *cg[c];
        const uint64_t dx = comp_dx && comp_dx[c] ? comp_dx[c] : 1, dy = comp_dy && comp_dy[c] ? comp_dy[c] : 1;
        for (uint32_t r = 0; r <= p.num_levels; ++r) {
            const ResGeom& R = g.res[r];
            const uint32_t sh = p.num_levels - r;
            for (uint32_t pj = 0; pj < R.nph; ++pj)
                for (uint32_t pi = 0; pi < R.npw; ++pi) {
                    const uint64_t cx = (((uint64_t)((R.x0 >> R.ppx) + pi) << R.ppx) << sh) * dx, cy = (((uint64_t)((R.y0 >> R.ppy) + pj) << R.ppy) << sh) * dy;
                    prec.push_back(Pk{c, r, pj * R.npw + pi, std::max<uint64_t>(cx, gx0), std::max<uint64_t>(cy, gy0)});
                }
        }
    }
    // (stable sorts of the component-major, resolution-major, raster list: the keys named, everything else in that order)
    if (order <= 1)                // resolution, component, precinct
        std::stable_sort(prec.begin(), prec.end(), [](const Pk& a, const Pk& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; });
    else if (order == 2)           // resolution, position, component
        std::stable_sort(prec.begin(), prec.end(), [](const Pk& a, const Pk& b) {
            return a.r != b.r ? a.r < b.r : a.y != b.y ? a.y < b.y : a.x != b.x ? a.x < b.x : a.c < b.c; });
    else if (order == 3)           // position, component, resolution
        std::stable_sort(prec.begin(), prec.end(), [](const Pk& a, const Pk& b) {
            return a.y != b.y ? a.y < b.y : a.x != b.x ? a.x < b.x : a.c != b.c ? a.c < b.c : a.r < b.r; });
    else                           // component, position, resolution
        std::stable_sort(prec.begin(), prec.end(), [](const Pk& a, const Pk& b) {
            return a.c != b.c ? a.c < b.c : a.y != b.y ? a.y < b.y : a.x != b.x ? a.x < b.x : a.r < b.r; });
    return prec;
}

// a packet's length as the PLT marker segment carries it: a big-endian base-128 number (continuation bit 0x80)
void plt_length(std::vector<uint8_t>& body, uint64_t v)
{
    uint8_t tmp[10]; int k = 0;
    tmp[k++] = (uint8_t)(v & 0x7F);
    while (v >>= 7) tmp[k++] = (uint8_t)((v & 0x7F) | 0x80);
    while (k) body.push_back(tmp[--k]);
}

// PLT (markers/LengthMarkers.cpp:313-374, written in front of SOD: TileProcessor.cpp:719-726): Zplt 0, then every packet's length.
// Marker segments of at most 65535 bytes (Lplt is 16 bits: Lplt, Zplt and 65532 bytes of lengths), Zplt = 0, 1, ...;
// a packet's length (1-5 bytes, the last one without the continuation bit) is never split over two segments.
// 18 packets of a 6-resolution RGB tile need < 100 bytes; small precincts multiply that: the reference starts another
// segment near 64 KiB as well (LengthMarkers.cpp:313-331 -- its continuation segments lack the Zplt byte, a defect
// this writer does not copy: T.800 A.7.3).
void write_plt(Out& o, const std::vector<uint8_t>& body)
{
    size_t at = 0;
    uint32_t z = 0;
    do {
        size_t end = at, next = at;
        while (next < body.size()) {
            size_t e = next;
            while (body[e] & 0x80) ++e;                        // one length: bytes with the continuation bit, then one without
            ++e;
            if (e - at > 65532) break;
            end = next = e;
        }
        if (z > 255) { o.ovf = true; break; }                  // (Zplt is one byte: > 16 MB of packet lengths in one tile-part does not fit the syntax)
        o.u16(0xFF58); o.u16((uint32_t)(3 + (end - at))); o.u8((uint8_t)z++);
        o.bytes(body.data() + at, end - at);
        at = end;
    } while (at < body.size());
}

uint64_t write_tile_part(Out& o, const std::vector<const TileGeom*>& cg, const std::vector<uint64_t>& row0, const uint8_t* comp_dx,
                         const uint8_t* comp_dy, uint32_t gx0, uint32_t gy0, uint32_t t, uint32_t flags, const grk_amd_coded_block* tt,
                         const uint8_t* coded)
{
    const uint64_t sot = o.n, sot_lit = o.lit ? o.lit->size() : 0;
    const uint32_t order = (flags >> GRK_AMD_CS_PROG_SHIFT) & 7u;
    const bool sop = (flags & GRK_AMD_CS_SOP) != 0, eph = (flags & GRK_AMD_CS_EPH) != 0;
    const std::vector<Pk> seq = packet_order(cg, comp_dx, comp_dy, gx0, gy0, order);
    auto packets = [&](Out& dst, std::vector<uint8_t>* plt) {
        int32_t n = 0;
        for (const Pk& q : seq) {
            const uint64_t at = dst.n;
            write_packet(dst, *cg[q.c], q.r, q.pi, tt + row0[q.c], coded, sop ? n : -1, eph);
            ++n;
            if (plt) plt_length(*plt, dst.n - at);
        }
    };
    o.u16(0xFF90); o.u16(10); o.u16(t); o.u32(0); o.u8(0); o.u8(1);
    if (flags & GRK_AMD_CS_PLT) {
        // the packets are sized with a counting pass
        std::vector<uint8_t> body;
        Out cnt{nullptr, 0};
        packets(cnt, &body);
        write_plt(o, body);
    }
    o.u16(0xFF93);
    packets(o, nullptr);
    o.patch32(sot + 6, (uint32_t)(o.n - sot), sot_lit, sot);
    return o.n - sot;
}

// an image without sub-sampling: every component has the tile's own geometry, comp c's rows start at c * blocks_per_comp
uint64_t write_tile_part(Out& o, const TileGeom& g, uint32_t t, uint32_t flags, const grk_amd_coded_block* tt, const uint8_t* coded)
{
    std::vector<const TileGeom*> cg(g.p.num_comps, &g);
    std::vector<uint64_t> row0(g.p.num_comps);
    for (uint32_t c = 0; c < g.p.num_comps; ++c) row0[c] = (uint64_t)c * g.blocks_per_comp;
    return write_tile_part(o, cg, row0, nullptr, nullptr, g.p.tile_x0, g.p.tile_y0, t, flags, tt, coded);
}

} // namespace

extern "C" int64_t grk_amd_layout_num_tiles(const grk_amd_image_layout* im)
{
    Layout l;
    const int rc = check_layout(im, l);
    return rc != GRK_AMD_OK ? rc : (int64_t)l.tcols * l.trows;
}

extern "C" int grk_amd_layout_tile(const grk_amd_image_layout* im, const grk_amd_tile_params* base, uint32_t tile_index,
                                   grk_amd_tile_params* out)
{
    Layout l;
    const int rc = check_layout(im, l);
    if (rc != GRK_AMD_OK) return rc;
    if (!base || !out || tile_index >= l.tcols * l.trows) return GRK_AMD_ERR_INVALID;
    tile_of(l, *base, tile_index, *out);
    return GRK_AMD_OK;
}

extern "C" int64_t grk_amd_write_codestream_layout(const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                                   const grk_amd_coded_block* table, const uint8_t* coded, uint32_t flags,
                                                   uint8_t* out, uint64_t cap)
{
    if (!base || !table || !coded || !out) return GRK_AMD_ERR_INVALID;
    Layout l;
    int rc = check_layout(im, l);
    if (rc != GRK_AMD_OK) return rc;
    const uint32_t ntiles = l.tcols * l.trows;
    if ((flags & GRK_AMD_CS_TLM) && ntiles > 255) return GRK_AMD_ERR_UNSUPPORTED;     // (one-byte Ttlm, as the reference writes it)
    TileGeom g;
    grk_amd_tile_params p;
    Out o{out, cap};
    uint64_t tlm_at = 0, row = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        tile_of(l, *base, t, p);
        rc = build_tile_geom(p, g);
        if (rc != GRK_AMD_OK) return rc;
        if (t == 0) write_main_header(o, g, l.im, flags, ntiles, &tlm_at);
        const uint64_t len = write_tile_part(o, g, t, flags, table + row, coded);
        row += (uint64_t)g.blocks_per_comp * p.num_comps;
        if (tlm_at) o.patch32(tlm_at + 5ull * t + 1, (uint32_t)len);
    }
    o.u16(0xFFD9);
    if (o.ovf) return GRK_AMD_ERR_OVERFLOW;
    return (int64_t)o.n;
}

// ---- sub-sampled components ------------------------------------------------------------------------------------------
// Component c of a tile [x0, x1) x [y0, y1) of the reference grid is [ceil(x0 / dx_c), ceil(x1 / dx_c)) x [ceil(y0 / dy_c),
// ceil(y1 / dy_c)) in its own coordinates (tile/TileProcessor.cpp:605-612); everything below the tile-component -- resolutions,
// bands, precincts, code-blocks -- is derived from that rectangle as for any tile (tile/TileComponent.cpp:69-170).
namespace {
int tile_comp_of(const Layout& l, const grk_amd_tile_params& base, uint32_t dx, uint32_t dy, uint32_t t, grk_amd_tile_params& p)
{
    tile_of(l, base, t, p);
    dx = dx ? dx : 1; dy = dy ? dy : 1;
    const uint64_t x0 = p.tile_x0, y0 = p.tile_y0, x1 = x0 + p.tile_w, y1 = y0 + p.tile_h;
    const uint64_t cx0 = (x0 + dx - 1) / dx, cy0 = (y0 + dy - 1) / dy, cx1 = (x1 + dx - 1) / dx, cy1 = (y1 + dy - 1) / dy;
    if (cx1 <= cx0 || cy1 <= cy0) return GRK_AMD_ERR_UNSUPPORTED;      // (a tile without a sample of this component)
    p.tile_x0 = (uint32_t)cx0; p.tile_y0 = (uint32_t)cy0; p.tile_w = (uint32_t)(cx1 - cx0); p.tile_h = (uint32_t)(cy1 - cy0);
    return GRK_AMD_OK;
}
} // namespace

extern "C" int grk_amd_layout_tile_comp(const grk_amd_image_layout* im, const grk_amd_tile_params* base, uint32_t dx, uint32_t dy,
                                        uint32_t tile_index, grk_amd_tile_params* out)
{
    Layout l;
    const int rc = check_layout(im, l);
    if (rc != GRK_AMD_OK) return rc;
    if (!base || !out || tile_index >= l.tcols * l.trows || dx > 255 || dy > 255) return GRK_AMD_ERR_INVALID;
    return tile_comp_of(l, *base, dx, dy, tile_index, *out);
}

extern "C" int64_t grk_amd_write_codestream_subsampled(const grk_amd_image_layout* im, const grk_amd_tile_params* base,
                                                       const uint8_t* comp_dx, const uint8_t* comp_dy,
                                                       const grk_amd_coded_block* table, const uint8_t* coded, uint32_t flags,
                                                       uint8_t* out, uint64_t cap)
{
    if (!base || !table || !coded || !out || !comp_dx || !comp_dy) return GRK_AMD_ERR_INVALID;
    Layout l;
    int rc = check_layout(im, l);
    if (rc != GRK_AMD_OK) return rc;
    const uint32_t ntiles = l.tcols * l.trows, nc = base->num_comps;
    if ((flags & GRK_AMD_CS_TLM) && ntiles > 255) return GRK_AMD_ERR_UNSUPPORTED;
    // the multiple component transform needs its three components on one grid: otherwise the reference switches it off with a
    // warning (CodeStreamCompress.cpp:434-447), and COD says so
    grk_amd_tile_params based = *base;
    if (based.mct && nc >= 3 && !(comp_dx[0] == comp_dx[1] && comp_dx[1] == comp_dx[2] && comp_dy[0] == comp_dy[1] && comp_dy[1] == comp_dy[2]))
        based.mct = 0;
    base = &based;
    std::vector<TileGeom> geoms(nc);
    std::vector<const TileGeom*> cg(nc);
    std::vector<uint64_t> row0(nc);
    grk_amd_tile_params p, pt;
    Out o{out, cap};
    uint64_t tlm_at = 0, row = 0;
    for (uint32_t t = 0; t < ntiles; ++t) {
        tile_of(l, *base, t, pt);
        uint64_t rows = 0;
        for (uint32_t c = 0; c < nc; ++c) {
            if ((rc = tile_comp_of(l, *base, comp_dx[c], comp_dy[c], t, p)) != GRK_AMD_OK) return rc;
            if ((rc = build_tile_geom(p, geoms[c])) != GRK_AMD_OK) return rc;
            cg[c] = &geoms[c]; row0[c] = rows; rows += geoms[c].blocks_per_comp;
        }
        if (t == 0) write_main_header(o, geoms[0], l.im, flags, ntiles, &tlm_at, comp_dx, comp_dy);
        const uint64_t len = write_tile_part(o, cg, row0, comp_dx, comp_dy, pt.tile_x0, pt.tile_y0, t, flags, table + row, coded);
        row += rows;
        if (tlm_at) o.patch32(tlm_at + 5ull * t + 1, (uint32_t)len);
    }
    o.u16(0xFFD9);
    if (o.ovf) return GRK_AMD_ERR_OVERFLOW;
    return (int64_t)o.n;
}

extern "C" int64_t grk_amd_write_codestream_ex(const grk_amd_tile_params* p, uint32_t img_w, uint32_t img_h,
                                               const grk_amd_coded_block* table, const uint8_t* coded, uint32_t flags,
                                               uint8_t* out, uint64_t cap)
{
    if (!p) return GRK_AMD_ERR_INVALID;
    const grk_amd_image_layout im = plain_layout(*p, img_w, img_h);
    // this entry point serves tables of ONE grk_amd_encode_tiles batch: every tile has to have *p's geometry
    Layout l;
    int rc = check_layout(&im, l);
    if (rc != GRK_AMD_OK) return rc;
    TileGeom g0, g;
    grk_amd_tile_params q;
    rc = build_tile_geom(*p, g0);
    if (rc != GRK_AMD_OK) return rc;
    for (uint32_t t = 1; t < l.tcols * l.trows; ++t) {
        tile_of(l, *p, t, q);
        if ((rc = build_tile_geom(q, g)) != GRK_AMD_OK) return rc;
        if (!same_geometry(g0, g)) return GRK_AMD_ERR_UNSUPPORTED;
    }
    return grk_amd_write_codestream_layout(&im, p, table, coded, flags, out, cap);
}

extern "C" int64_t grk_amd_write_codestream(const grk_amd_tile_params* p, uint32_t img_w, uint32_t img_h,
                                            const grk_amd_coded_block* table, const uint8_t* coded,
                                            uint8_t* out, uint64_t cap)
{
    return grk_amd_write_codestream_ex(p, img_w, img_h, table, coded, 0, out, cap);
}

extern "C" int64_t grk_amd_write_main_header_layout(const grk_amd_image_layout* im, const grk_amd_tile_params* base, uint32_t flags,
                                                    const uint32_t* tile_part_bytes, uint8_t* out, uint64_t cap)
{
    if (!base) return GRK_AMD_ERR_INVALID;
    Layout l;
    int rc = check_layout(im, l);
    if (rc != GRK_AMD_OK) return rc;
    const uint32_t ntiles = l.tcols * l.trows;
    if (flags & GRK_AMD_CS_TLM) { if (ntiles > 255) return GRK_AMD_ERR_UNSUPPORTED; if (!tile_part_bytes) return GRK_AMD_ERR_INVALID; }
    TileGeom g;
    grk_amd_tile_params p;
    tile_of(l, *base, 0, p);
    rc = build_tile_geom(p, g);
    if (rc != GRK_AMD_OK) return rc;
    Out o{out, cap};
    uint64_t tlm_at = 0;
    write_main_header(o, g, l.im, flags, ntiles, &tlm_at);
    for (uint32_t t = 0; tlm_at && t < ntiles; ++t) o.patch32(tlm_at + 5ull * t + 1, tile_part_bytes[t]);
    if (o.ovf) return GRK_AMD_ERR_OVERFLOW;
    return (int64_t)o.n;
}

extern "C" int64_t grk_amd_write_main_header(const grk_amd_tile_params* p, uint32_t img_w, uint32_t img_h, uint32_t flags,
                                             const uint32_t* tile_part_bytes, uint8_t* out, uint64_t cap)
{
    if (!p) return GRK_AMD_ERR_INVALID;
    const grk_amd_image_layout im = plain_layout(*p, img_w, img_h);
    return grk_amd_write_main_header_layout(&im, p, flags, tile_part_bytes, out, cap);
}

extern "C" int64_t grk_amd_write_tile_part(const grk_amd_tile_params* p, uint32_t tile_index, uint32_t flags,
                                           const grk_amd_coded_block* tile_table, const uint8_t* coded, uint8_t* out, uint64_t cap)
{
    if (!p || !tile_table || (out && !coded)) return GRK_AMD_ERR_INVALID;
    TileGeom g;
    int rc = build_tile_geom(*p, g);
    if (rc != GRK_AMD_OK) return rc;
    Out o{out, cap};
    write_tile_part(o, g, tile_index, flags, tile_table, coded);
    if (o.ovf) return GRK_AMD_ERR_OVERFLOW;
    return (int64_t)o.n;
}

namespace grk_amd {
int64_t plan_tile_part(const grk_amd_tile_params& p, uint32_t tile_index, uint32_t flags, const grk_amd_coded_block* tile_table,
                       std::vector<uint8_t>& lit, std::vector<grk_amd_tp_segment>& segs)
{
    TileGeom g;
    const int rc = build_tile_geom(p, g);
    if (rc != GRK_AMD_OK) return rc;
    lit.clear(); segs.clear();
    lit.reserve((size_t)g.blocks_per_comp * p.num_comps * 5 + 256);
    segs.reserve((size_t)g.blocks_per_comp * p.num_comps + 64);
    Out o{nullptr, 0};
    o.lit = &lit; o.segs = &segs;
    write_tile_part(o, g, tile_index, flags, tile_table, nullptr);
    return (int64_t)o.n;
}
} // namespace grk_amd

namespace grk_amd {
// What the header kernel needs to know about a tile's packets (no sub-sampling: every component has the tile's geometry).
int t2_device_plan(const TileGeom& g, uint32_t flags, T2Plan& out)
{
    const grk_amd_tile_params& p = g.p;
    std::vector<const TileGeom*> cg(p.num_comps, &g);
    const std::vector<Pk> seq = packet_order(cg, nullptr, nullptr, p.tile_x0, p.tile_y0, (flags >> GRK_AMD_CS_PROG_SHIFT) & 7u);
    out.packets.clear();
    out.packet_of_block.assign((size_t)g.blocks_per_comp * p.num_comps, 0xFFFFFFFFu);
    uint64_t u = 0, h = 0;
    for (const Pk& q : seq) {
        const ResGeom& R = g.res[q.r];
        T2Packet k{};
        k.row0 = q.c * g.blocks_per_comp;
        uint64_t bits = 1;
        for (uint32_t bi = 0; bi < R.num_bands; ++bi) {
            const BandGeom& B = R.band[bi];
            const BandGeom::Prec& P = B.prec[q.pi];
            if (!P.gw || !P.gh) continue;
            const uint32_t b = k.nbands++;
            k.first_block[b] = P.first_block; k.gw[b] = P.gw; k.gh[b] = P.gh; k.kmax[b] = B.kmax;
            k.height[b] = (uint32_t)tag_tree_height(P.gw, P.gh);
            const uint64_t n = (uint64_t)P.gw * P.gh;
            for (uint64_t i = 0; i < n; ++i) out.packet_of_block[k.row0 + P.first_block + i] = (uint32_t)out.packets.size();
            k.nblocks += (uint32_t)n;
            // per block: two paths of at most `height` nodes, the pass bit, Lblock's comma code and the length
            bits += n * (2ull * k.height[b] + 1 + (kT2MaxLenBits - 3 + 1) + kT2MaxLenBits) + B.kmax;
        }
        if (bits > 0x7FFFFFFFull) return GRK_AMD_ERR_UNSUPPORTED;
        k.u_at = (uint32_t)u; k.u_words = (uint32_t)((bits + 31) / 32 + 2);
        k.h_at = (uint32_t)h;
        u += (k.u_words + 31u) & ~31ull;                         // (packets start on 128-byte lines)
        h += ((bits + 6) / 7 + 2 + 15) & ~15ull;                 // a stuffed byte carries at least 7 bits
        if (u > 0x3FFFFFFFull || h > 0xFFFFFFFFull) return GRK_AMD_ERR_UNSUPPORTED;
        out.packets.push_back(k);
    }
    for (uint32_t v : out.packet_of_block) if (v == 0xFFFFFFFFu) return GRK_AMD_ERR_INVALID;     // (every block lies in one precinct)
    out.u_words = (uint32_t)u; out.h_bytes = (uint32_t)h;
    return GRK_AMD_OK;
}

} // namespace grk_amd

extern "C" int64_t grk_amd_plan_tile_part(const grk_amd_tile_params* p, uint32_t tile_index, uint32_t flags,
                                          const grk_amd_coded_block* tile_table, uint8_t* literal, uint64_t literal_cap, uint64_t* literal_len,
                                          grk_amd_tp_segment* segments, uint64_t segment_cap, uint64_t* num_segments)
{
    if (!p || !tile_table || !literal_len || !num_segments) return GRK_AMD_ERR_INVALID;
    std::vector<uint8_t> lit;
    std::vector<grk_amd_tp_segment> segs;
    const int64_t total = grk_amd::plan_tile_part(*p, tile_index, flags, tile_table, lit, segs);
    if (total < 0) return total;
    struct { uint64_t n; } o{(uint64_t)total};
    *literal_len = lit.size(); *num_segments = segs.size();
    if (literal && segments) {
        if (lit.size() > literal_cap || segs.size() > segment_cap) return GRK_AMD_ERR_OVERFLOW;
        std::memcpy(literal, lit.data(), lit.size());
        std::memcpy(segments, segs.data(), segs.size() * sizeof(grk_amd_tp_segment));
    }
    return (int64_t)o.n;
}

// Random access into a codestream (the reader's side of markers/LengthMarkers.cpp:91-164): where every tile-part starts
// and how long it is -- from the TLM marker segments when the main header carries them (no byte of a tile-part is
// touched), else by hopping from SOT to SOT over Psot.
extern "C" int64_t grk_amd_locate_tile_parts(const uint8_t* cs, uint64_t len, uint64_t* offsets, uint32_t* lengths,
                                             uint16_t* tile_index, uint64_t cap, int* used_tlm)
{
    if (!cs || len < 4 || cs[0] != 0xFF || cs[1] != 0x4F) return GRK_AMD_ERR_INVALID;
    auto be16 = [&](uint64_t i) { return (uint32_t)(cs[i] << 8 | cs[i + 1]); };
    auto be32 = [&](uint64_t i) { return (uint32_t)cs[i] << 24 | (uint32_t)cs[i + 1] << 16 | (uint32_t)cs[i + 2] << 8 | cs[i + 3]; };
    std::vector<std::pair<uint32_t, uint32_t>> tlm;       // (tile index or 0xFFFFFFFF = "next", length), in marker order
    uint64_t at = 2;
    bool have_tlm = false;
    while (at + 4 <= len) {
        const uint32_t m = be16(at);
        if (m == 0xFF90) break;
        const uint32_t l = be16(at + 2);
        if (m < 0xFF00 || l < 2 || at + 2 + l > len) return GRK_AMD_ERR_INVALID;
        if (m == 0xFF55 && l >= 4) {
            have_tlm = true;
            const uint32_t stlm = cs[at + 5], st = (stlm >> 4) & 3u, sp = (stlm >> 6) & 1u;
            const uint32_t rec = st + (sp ? 4u : 2u);
            if (st == 3) return GRK_AMD_ERR_INVALID;
            for (uint64_t q = at + 6; q + rec <= at + 2 + l; q += rec) {
                const uint32_t ti = st == 0 ? 0xFFFFFFFFu : st == 1 ? cs[q] : be16(q);
                tlm.emplace_back(ti, sp ? be32(q + st) : be16(q + st));
            }
        }
        at += 2 + l;
    }
    if (at + 12 > len || be16(at) != 0xFF90) return GRK_AMD_ERR_INVALID;
    if (used_tlm) *used_tlm = have_tlm ? 1 : 0;
    uint64_t n = 0;
    if (have_tlm) {
        uint32_t next = 0;
        for (const auto& e : tlm) {
            const uint32_t ti = e.first == 0xFFFFFFFFu ? next : e.first;
            if (at + e.second > len) return GRK_AMD_ERR_INVALID;
            if (n < cap) { if (offsets) offsets[n] = at; if (lengths) lengths[n] = e.second; if (tile_index) tile_index[n] = (uint16_t)ti; }
            ++n; at += e.second; next = ti + 1;
        }
        return (int64_t)n;
    }
    while (at + 12 <= len && be16(at) == 0xFF90) {
        const uint32_t psot = be32(at + 6);
        const uint64_t l = psot ? psot : len - 2 - at;            // Psot 0: the last tile-part runs to EOC
        if (l < 14 || at + l > len) return GRK_AMD_ERR_INVALID;
        if (n < cap) { if (offsets) offsets[n] = at; if (lengths) lengths[n] = (uint32_t)l; if (tile_index) tile_index[n] = (uint16_t)be16(at + 4); }
        ++n; at += l;
    }
    return (int64_t)n;
}
