"""Test-side Tier-2 reader (NOT part of the product: codestream parsing stays with the host library,
SURVEY.md §2).  It recovers, from a codestream of the restricted class our hot path serves -- one
quality layer, LRCP, one precinct per resolution, no SOP/EPH/PPM/PPT, one tile-part per tile -- what
Grok's T2 hands a decode plugin per code-block (plugin/plugin_bridge.cpp:63-76): the bytes, the number
of coding passes and the zero bit-planes.  Written from ITU-T T.800 Annex A (markers) and Annex B
(packet headers, tag trees); used to decode streams of the *reference's own encoders* on the GPU."""
import struct

import numpy as np


class Bits:
    """Packet-header bit reader: MSB first, a byte following 0xFF carries 7 bits (B.10.1)."""

    def __init__(self, data, pos):
        self.d, self.pos, self.cur, self.n = data, pos, 0, 0

    def bit(self):
        if self.n == 0:
            prev_ff = self.cur == 0xFF
            self.cur = self.d[self.pos]
            self.pos += 1
            self.n = 7 if prev_ff else 8
        self.n -= 1
        return (self.cur >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def align(self):
        if self.cur == 0xFF:                      # a header ending on 0xFF is followed by a stuffed byte
            self.pos += 1
        self.n = 0
        self.cur = 0
        return self.pos


class TagTree:
    """B.10.2: quad-tree of minima; decode(x, y, threshold) -> (value < threshold is known, value so far)."""

    def __init__(self, w, h):
        self.dims, self.nodes = [], []
        while True:
            self.dims.append((w, h))
            self.nodes.append([[0, False] for _ in range(w * h)])      # [lower bound, value known]
            if w == 1 and h == 1:
                break
            w, h = (w + 1) // 2, (h + 1) // 2

    def decode(self, br, x, y, threshold):
        low = 0
        for lv in range(len(self.dims) - 1, -1, -1):
            w, _ = self.dims[lv]
            node = self.nodes[lv][(y >> lv) * w + (x >> lv)]
            if node[0] < low:
                node[0] = low
            while not node[1] and node[0] < threshold:
                if br.bit():
                    node[1] = True
                else:
                    node[0] += 1
            low = node[0]
        leaf = self.nodes[0][y * self.dims[0][0] + x]
        return leaf[1] and leaf[0] < threshold, leaf[0]


def _cdp2(v, n):
    return (v + (1 << n) - 1) >> n


def parse(cs):
    """-> dict(W, H, C, prec, levels, irreversible, mct, ht, guard, qcd [(expn, mant)...],
               blocks {(comp, res, band, idx): (bytes, numpasses, zero_bitplanes)})  for a single-tile stream."""
    cs = bytes(cs)
    assert cs[:2] == b"\xff\x4f"
    pos, info = 2, {}
    while True:
        marker, = struct.unpack(">H", cs[pos:pos + 2])
        if marker == 0xFF90:        # SOT
            break
        ln, = struct.unpack(">H", cs[pos + 2:pos + 4])
        body = cs[pos + 4:pos + 2 + ln]
        if marker == 0xFF51:        # SIZ
            _, xs, ys, xo, yo, xt, yt, xto, yto, nc = struct.unpack(">HIIIIIIIIH", body[:36])
            info.update(W=xs - xo, H=ys - yo, C=nc, prec=(body[36] & 0x7F) + 1, tw=xt, th=yt, x0=xo, y0=yo)
            assert xto + xt >= xs and yto + yt >= ys, "single-tile streams only"
        elif marker == 0xFF52:      # COD
            scod, prog, layers, mct, levels, cbw, cbh, sty, xf = struct.unpack(">BBHBBBBBB", body[:10])
            assert prog in (0, 1) and layers == 1, "LRCP / RLCP, 1 layer only"
            prc = [(b & 15, b >> 4) for b in body[10:10 + levels + 1]] if scod & 1 else [(15, 15)] * (levels + 1)
            info.update(levels=levels, cbw=cbw + 2, cbh=cbh + 2, mct=mct, cblk_sty=sty, irreversible=int(xf == 0),
                        ht=int((sty & 0x40) != 0), prc=prc, scod=scod)
        elif marker == 0xFF5C:      # QCD
            sq = body[0]
            info["guard"] = sq >> 5
            if (sq & 0x1F) == 0:    # no quantisation: 8-bit exponents
                info["qcd"] = [(b >> 3, 0) for b in body[1:]]
            else:                   # scalar expounded
                assert (sq & 0x1F) == 2
                info["qcd"] = [(w >> 11, w & 0x7FF) for w in struct.unpack(">%dH" % ((len(body) - 1) // 2), body[1:])]
        pos += 2 + ln
    # ---- the single tile-part
    _, lsot, isot, psot, tp, tn = struct.unpack(">HHHIBB", cs[pos:pos + 12])
    end = pos + psot if psot else len(cs) - 2
    pos += 12
    while cs[pos:pos + 2] != b"\xff\x93":       # tile-part header markers up to SOD
        ln, = struct.unpack(">H", cs[pos + 2:pos + 4])
        pos += 2 + ln
    pos += 2
    W, H, L = info["W"], info["H"], info["levels"]
    state = {}
    blocks = {}
    segments = {}                 # key -> [(bytes, passes), ...] codeword segments (B.10.7.2: TERMALL / LAZY split them)
    sty = info["cblk_sty"]

    def split_passes(npass):
        if sty & 0x04:            # TERMALL: every pass terminated (T2Decompress.cpp:169-186)
            return [1] * npass
        if sty & 0x01:            # LAZY: 10 passes, then raw pair / cleanup alternately
            out, cap = [], 10
            while npass > 0:
                out.append(min(cap, npass)); npass -= out[-1]
                cap = 2 if cap in (10, 1) else 1
            return out
        return [npass]

    X0, Y0 = info["x0"], info["y0"]

    def band_rect(r, b):
        """(x0, y0, x1, y1) of the band in its own coordinates (B.5: the tile = the image area, anywhere on the grid)."""
        n = L - r + (1 if r > 0 else 0)
        if r == 0:
            return _cdp2(X0, L), _cdp2(Y0, L), _cdp2(X0 + W, L), _cdp2(Y0 + H, L)
        bx, by = b & 1, b >> 1
        return (_cdp2(max(X0 - (1 << (n - 1)) * bx, 0), n), _cdp2(max(Y0 - (1 << (n - 1)) * by, 0), n),
                _cdp2(max(X0 + W - (1 << (n - 1)) * bx, 0), n), _cdp2(max(Y0 + H - (1 << (n - 1)) * by, 0), n))

    sop, eph = bool(info.get("scod", 0) & 2), bool(info.get("scod", 0) & 4)
    for r in range(L + 1):
        rx0, rx1, ry0, ry1 = _cdp2(X0, L - r), _cdp2(X0 + W, L - r), _cdp2(Y0, L - r), _cdp2(Y0 + H, L - r)
        if rx1 == rx0 or ry1 == ry0:
            continue              # a resolution without samples has no precinct and no packet
        # precincts: the cells of the 2^PPx x 2^PPy grid anchored at the origin of the resolution's coordinates (B.6); in the
        # bands of a resolution r > 0 a precinct is half as large, and a code-block is never larger than that
        ppx, ppy = info["prc"][r]
        npw, nph = _cdp2(rx1, ppx) - (rx0 >> ppx), _cdp2(ry1, ppy) - (ry0 >> ppy)
        sub = 1 if r else 0
        bpx, bpy = ppx - sub, ppy - sub
        psx, psy = ((rx0 >> ppx) << ppx) >> sub, ((ry0 >> ppy) << ppy) >> sub
        cexp = min(info["cbw"], bpx), min(info["cbh"], bpy)
        bands = [0] if r == 0 else [1, 2, 3]
        grids = {}                # band -> [(gw, gh, first index)] per precinct
        for b in bands:
            bx0, by0, bx1, by1 = band_rect(r, b)
            lst, first = [], 0
            for pj in range(nph):
                for pi in range(npw):
                    qx0, qy0 = psx + (pi << bpx), psy + (pj << bpy)
                    cx0, cx1 = max(qx0, bx0), min(qx0 + (1 << bpx), bx1)
                    cy0, cy1 = max(qy0, by0), min(qy0 + (1 << bpy), by1)
                    if cx0 >= cx1 or cy0 >= cy1:
                        lst.append((0, 0, first))
                        continue
                    gw, gh = _cdp2(cx1, cexp[0]) - (cx0 >> cexp[0]), _cdp2(cy1, cexp[1]) - (cy0 >> cexp[1])
                    lst.append((gw, gh, first))
                    first += gw * gh
            grids[b] = lst
        for c in range(info["C"]):
          for pk in range(npw * nph):
            if sop:
                assert cs[pos:pos + 2] == b"\xff\x91"
                pos += 6
            br = Bits(cs, pos)
            nonempty = br.bit()
            todo = []
            for b in bands:
                gw, gh, first = grids[b][pk]
                if gw == 0 or gh == 0:
                    continue
                key = (c, r, b, pk)
                if key not in state:
                    state[key] = (TagTree(gw, gh), TagTree(gw, gh), {})
                incl, zbp, lblock = state[key]
                for loc in range(gw * gh):
                    idx = first + loc
                    x, y = loc % gw, loc // gw
                    if not nonempty:
                        blocks[(c, r, b, idx)] = (b"", 0, 0)
                        continue
                    ok, _ = incl.decode(br, x, y, 1)
                    if not ok:
                        blocks[(c, r, b, idx)] = (b"", 0, 0)
                        continue
                    z = 0
                    while True:
                        okz, val = zbp.decode(br, x, y, z + 1)
                        if okz:
                            break
                        z += 1
                    zero_bp = val
                    # number of coding passes (Table B.4)
                    if not br.bit():
                        npass = 1
                    elif not br.bit():
                        npass = 2
                    else:
                        v = br.bits(2)
                        if v < 3:
                            npass = 3 + v
                        else:
                            v = br.bits(5)
                            npass = 6 + v if v < 31 else 37 + br.bits(7)
                    lb = lblock.get(loc, 3)
                    while br.bit():
                        lb += 1
                    lblock[loc] = lb
                    # one length per codeword segment, lblock + floor(log2(passes in the segment)) bits each; HT: the
                    # reference's encoder emits the single cleanup pass as one segment
                    segs = [(br.bits(lb + int(np.floor(np.log2(k)))), k) for k in ([npass] if info["ht"] else split_passes(npass))]
                    segments[(c, r, b, idx)] = segs
                    todo.append(((c, r, b, idx), npass, zero_bp, sum(a for a, _ in segs)))
            pos = br.align()
            if eph:
                assert cs[pos:pos + 2] == b"\xff\x92"
                pos += 2
            for key, npass, zero_bp, ln in todo:
                blocks[key] = (cs[pos:pos + ln], npass, zero_bp)
                pos += ln
    assert pos <= end + 2, "packet parsing ran past the tile-part"
    info["blocks"] = blocks
    info["segments"] = segments
    return info


def decode_table(info, layout_blocks, part1):
    """Rows for grk_amd_decode_tiles in the enumeration order of grk_amd_tile_layout (comp -> res -> band ->
    raster): (offset, length, missing_msbs | numbps | numpasses << 8) + the concatenated bytes."""
    rows, chunks, off = [], [], 0
    counters = {}
    for b in layout_blocks:
        key3 = (b.comp, b.res, b.band)
        idx = counters.get(key3, 0)
        counters[key3] = idx + 1
        data, npass, zbp = info["blocks"][(b.comp, b.res, b.band, idx)]
        bi = 0 if b.res == 0 else 3 * b.res - 2 + (b.band - 1)
        expn = info["qcd"][bi][0]
        band_numbps = expn + info["guard"] - 1
        if part1:
            extra = ((band_numbps - zbp) | (npass << 8)) if len(data) else 0
        else:
            extra = zbp
        rows.append((off, len(data), extra))
        pad = data + b"\0" * (-len(data) % 16 + 16)
        chunks.append(pad)
        off += len(pad)
    return rows, b"".join(chunks)


def segment_list(info, layout_blocks):
    """Codeword segments per block, [[(bytes, passes), ...], ...], in the order of grk_amd_tile_layout."""
    out, counters = [], {}
    for b in layout_blocks:
        key3 = (b.comp, b.res, b.band)
        idx = counters.get(key3, 0)
        counters[key3] = idx + 1
        out.append([sg for sg in info["segments"].get((b.comp, b.res, b.band, idx), []) if sg[1]])
    return out


def parse_cod_layers(cs):
    """Number of quality layers the main header's COD marker segment declares (A.6.1: SGcod = progression order, layers (2 bytes), MCT)."""
    i = 2
    while i + 4 <= len(cs):
        m, n = struct.unpack(">HH", cs[i:i + 4])
        if m == 0xFF52:
            return struct.unpack(">H", cs[i + 6:i + 8])[0]
        if m == 0xFF90 or m == 0xFF93:
            break
        i += 2 + n
    raise ValueError("no COD marker segment in the main header")
