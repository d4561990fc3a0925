"""CPU: precincts (grk_compress -c; COD precinct sizes) -- the code-block partition they cut, the packet per precinct.

  * the product's host geometry with precinct exponents == the oracle's enumeration (random sizes, origins, exponents),
  * oracle tiles + the product's Tier-2 writer == the files grk_compress writes with -c (several size lists, LRCP and RLCP,
    multi-tile, image offsets, TLM + PLT + SOP + EPH),
  * RPCL / PCRL / CPRL walk the precincts' positions on the canonical grid: == grk_compress -p files as well."""
import numpy as np
import pytest

import grok_amd as G
import oracle as O
import refharness as R
import synth
from grok_amd.capi import CODED_DTYPE

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


def exps_from_sizes(sizes, levels):
    """grk_compress -c list (highest resolution first, last entry halved beyond the list; CodeStreamCompress.cpp:475-514)
    -> [(PPx, PPy)] for r = 0 .. levels"""
    out = []
    for q in range(levels + 1):
        if q < len(sizes):
            pw, ph = sizes[q]
        else:
            pw, ph = sizes[-1][0] >> (q - (len(sizes) - 1)), sizes[-1][1] >> (q - (len(sizes) - 1))
        out.append((1 if pw < 1 else int(np.floor(np.log2(pw))), 1 if ph < 1 else int(np.floor(np.log2(ph)))))
    return out[::-1]


def test_layout_with_precincts_matches_oracle_enumeration():
    rng = np.random.default_rng(11)
    for _ in range(60):
        w, h = int(rng.integers(1, 500)), int(rng.integers(1, 500))
        lv = int(rng.integers(0, 6))
        org = (int(rng.integers(0, 300)), int(rng.integers(0, 300))) if rng.integers(0, 2) else (0, 0)
        prc = [(int(rng.integers(1, 9)), int(rng.integers(1, 9))) for _ in range(lv + 1)]
        p = G.TileParams.make(w, h, 1, 8, lv, origin=org, precincts=prc)
        blocks, _ = G.tile_layout(p)
        want = O.enumerate_blocks(w, h, lv, origin=org, precincts=prc)
        assert len(blocks) == len(want), (w, h, lv, org, prc)
        for b, o in zip(blocks, want):
            assert (b.px, b.py, b.x1 - b.x0, b.y1 - b.y0, b.res, b.band) == (o.x, o.y, o.w, o.h, o.res, o.band), (w, h, lv, org, prc)
        # precinct indices are non-decreasing within a band and below the resolution's precinct count
        counts = (np.zeros(lv + 1, np.uint32))
        G.lib().grk_amd_tile_precincts(p, counts.ctypes.data)
        last = {}
        for b in blocks:
            assert b.precinct < counts[b.res]
            assert b.precinct >= last.get((b.res, b.band), 0)
            last[(b.res, b.band)] = b.precinct
    # the default (15, 15) is what no precinct list gives
    a, _ = G.tile_layout(G.TileParams.make(300, 200, 1, 8, 4))
    b, _ = G.tile_layout(G.TileParams.make(300, 200, 1, 8, 4, precincts=[(15, 15)] * 5))
    assert [(x.px, x.py, x.x0, x.x1, x.precinct) for x in a] == [(x.px, x.py, x.x0, x.x1, x.precinct) for x in b]


def oracle_codestream_prc(px, prec, L, layout, prc, flags=0):
    Cn, H, W = px.shape
    base = G.TileParams.make(1, 1, Cn, prec, L, precincts=prc)
    tabs, chunks, off = [], [], 0
    for p in G.layout_tiles(layout, base):
        ox, oy = p.tile_x0 - layout.x0, p.tile_y0 - layout.y0
        tile = np.ascontiguousarray(px[:, oy:oy + p.tile_h, ox:ox + p.tile_w])
        _, lens, coded = O.encode_tile_rev(tile, prec, L, origin=(p.tile_x0, p.tile_y0), precincts=prc)
        assert len(lens) == G.lib().grk_amd_tile_num_blocks(p)
        t = np.zeros(len(lens), CODED_DTYPE)
        t["length"] = lens
        t["offset"] = off + np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(lens) else 0
        off += int(lens.sum())
        tabs.append(t)
        chunks.append(coded)
    return G.write_codestream_layout(layout, base, np.concatenate(tabs), np.concatenate(chunks), flags)


CASES = [   # W, H, TW, TH, levels, offset, -c sizes (highest resolution first), progression, extra flags
    (256, 192, 256, 192, 4, (0, 0), [(128, 128)], 0, 0),                       # halved below: 64, 32, 16, 8
    (256, 192, 256, 192, 4, (0, 0), [(256, 256), (128, 128), (64, 64)], 1, 0),  # RLCP
    (300, 210, 128, 128, 3, (0, 0), [(64, 32), (32, 64)], 0, G.CS_TLM | G.CS_PLT),
    (257, 129, 300, 200, 5, (33, 95), [(128, 64), (64, 64), (16, 16)], 0, G.CS_SOP | G.CS_EPH),
    (199, 159, 100, 100, 2, (1, 1), [(32, 32)], 1, G.CS_PLT),
    (96, 80, 96, 80, 3, (0, 0), [(16, 16)], 0, 0),                               # code-blocks of 8 x 8 down to 2 x 2
]


@needs_ref
@pytest.mark.parametrize("W,H,TW,TH,L,off,sizes,order,extra", CASES)
def test_oracle_codestream_with_precincts_is_the_reference_file(monkeypatch, W, H, TW, TH, L, off, sizes, order, extra):
    from test_offgrid_cpu import ref_defects
    px = synth.g2(3, H, W, 8, seed=W + L)
    layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
    assert ref_defects(layout, L) == (False, False)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    monkeypatch.setenv("REF_PRECINCTS", ",".join("%d,%d" % s for s in sizes))
    monkeypatch.setenv("REF_PROG_ORDER", str(order))
    monkeypatch.setenv("REF_WRITE_TLM", "1" if extra & G.CS_TLM else "0")
    monkeypatch.setenv("REF_WRITE_PLT", "1" if extra & G.CS_PLT else "0")
    monkeypatch.setenv("REF_CSTY", str((2 if extra & G.CS_SOP else 0) | (4 if extra & G.CS_EPH else 0)))
    want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
    got = oracle_codestream_prc(px, 8, L, layout, exps_from_sizes(sizes, L), G.CS_PROG(order) | extra)
    assert got == want
    assert np.array_equal(R.decode(got, 3, H, W), px.astype(np.int32))


@needs_ref
@pytest.mark.parametrize("order", [2, 3, 4])
@pytest.mark.parametrize("W,H,TW,TH,L,off,sizes", [(256, 192, 256, 192, 4, (0, 0), [(128, 128)]), (300, 210, 128, 128, 3, (0, 0), [(64, 32), (32, 64)]),
                                                   (257, 129, 300, 200, 5, (33, 95), [(128, 64), (64, 64), (16, 16)]),
                                                   (199, 159, 100, 100, 2, (1, 1), [(32, 32)]), (256, 256, 256, 256, 5, (0, 0), [(256, 256), (64, 64), (128, 128)])])
def test_position_first_orders_with_precincts_are_the_reference_files(monkeypatch, order, W, H, TW, TH, L, off, sizes):
    """RPCL, PCRL, CPRL with several precincts per resolution: packets in the order of the precincts' positions on the
    canonical grid (resolutions with different precinct sizes interleave) == grk_compress -p; with PLT."""
    from test_offgrid_cpu import ref_defects
    px = synth.g2(3, H, W, 8, seed=W + L + order)
    layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
    assert ref_defects(layout, L) == (False, False)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    monkeypatch.setenv("REF_PRECINCTS", ",".join("%d,%d" % s for s in sizes))
    monkeypatch.setenv("REF_PROG_ORDER", str(order))
    # (PLT only where every band of every precinct has samples: the reference's PLT entries of packets with an empty band
    #  are too large, D15 -- e.g. +5 each for the 24-row tile of the third layout -- while the packets agree)
    plt = off == (0, 0) and W % TW == 0
    monkeypatch.setenv("REF_WRITE_PLT", "1" if plt else "0")
    want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
    got = oracle_codestream_prc(px, 8, L, layout, exps_from_sizes(sizes, L), G.CS_PROG(order) | (G.CS_PLT if plt else 0))
    assert got == want
    assert np.array_equal(R.decode(got, 3, H, W), px.astype(np.int32))


@needs_ref
def test_plt_of_one_sample_wide_tiles_with_precincts_reference_defect_d15(monkeypatch):
    """D15: for the one-sample-wide last tile column of a 200 x 160 image at offset (1, 1) in 100 x 100 tiles with 32 x 32
    precincts, grk_compress -L writes PLT entries for the highest resolution's packets that are 10 too large each -- they sum
    to 721 for 481 bytes of packets.  Everything else of the file, those tile-parts' packets included, is byte-identical to
    ours; our PLT sums to the bytes that are there."""
    W, H, L = 200, 160, 2
    px = synth.g2(3, H, W, 8, seed=W + L)
    layout = G.ImageLayout.make(W, H, 100, 100, offset=(1, 1))
    for k, v in (("REF_IMG_X0", "1"), ("REF_IMG_Y0", "1"), ("REF_PRECINCTS", "32,32"), ("REF_WRITE_PLT", "1")):
        monkeypatch.setenv(k, v)
    want, _ = R.encode(px, 8, TW=100, TH=100, numres=L + 1, mode=1)
    got = oracle_codestream_prc(px, 8, L, layout, exps_from_sizes([(32, 32)], L), G.CS_PLT)
    assert len(got) == len(want)
    wa, _ = G.locate_tile_parts(want)
    for (at, ln, t) in wa:
        w, g = want[at:at + ln], got[at:at + ln]
        assert w[:14] == g[:14] and w[12:14] == b"\xff\x58"
        lplt = (w[14] << 8) | w[15]
        sod = 12 + 2 + lplt
        assert w[sod:] == g[sod:]                                   # SOD + every packet: identical
        def total(body):
            n = v = 0
            for b in body:
                v = (v << 7) | (b & 0x7F)
                if not b & 0x80:
                    n += v; v = 0
            return n
        ours, theirs = total(g[17:sod]), total(w[17:sod])
        assert ours == ln - sod - 2
        assert (theirs == ours) == (t not in (2, 5))               # the one-sample-wide tiles: the reference's sums are off


@needs_ref
def test_random_layouts_precincts_orders_against_the_reference(monkeypatch):
    """A sweep: random image sizes, offsets, tile sizes, level counts, precinct size lists, progression orders, SOP / EPH --
    oracle tiles + the product's writer == grk_compress, and grk_decompress returns the image (where the reference is free
    of its own defects D13 / D14, tests/test_offgrid_cpu.py)."""
    from test_offgrid_cpu import ref_defects
    rng = np.random.default_rng(4242)
    checked = 0
    for _ in range(60):
        W, H = int(rng.integers(8, 200)), int(rng.integers(8, 200))
        off = (int(rng.integers(0, 40)), int(rng.integers(0, 40))) if rng.integers(0, 2) else (0, 0)
        TW, TH = int(rng.integers(max(off[0] + 1, 32), 220)), int(rng.integers(max(off[1] + 1, 32), 220))
        L = int(rng.integers(1, 6))
        Cn = int(rng.choice([1, 3]))
        nlist = int(rng.integers(1, 4))
        sizes = [(1 << int(rng.integers(3, 9)), 1 << int(rng.integers(3, 9))) for _ in range(nlist)]
        order = int(rng.integers(0, 5))
        csty = int(rng.choice([0, 2, 4, 6]))
        layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
        d13, d14 = ref_defects(layout, L)
        if d13:
            continue
        prc = exps_from_sizes(sizes, L)
        if any(e[0] < 1 or e[1] < 1 for e in prc[1:]):
            continue
        px = synth.g2(Cn, H, W, 8, seed=int(rng.integers(1, 1000)))
        for k, v in (("REF_IMG_X0", off[0]), ("REF_IMG_Y0", off[1]), ("REF_PROG_ORDER", order), ("REF_CSTY", csty)):
            monkeypatch.setenv(k, str(v))
        monkeypatch.setenv("REF_PRECINCTS", ",".join("%d,%d" % s for s in sizes))
        want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
        flags = G.CS_PROG(order) | (G.CS_SOP if csty & 2 else 0) | (G.CS_EPH if csty & 4 else 0)
        got = oracle_codestream_prc(px, 8, L, layout, prc, flags)
        assert got == want, (W, H, TW, TH, L, off, Cn, sizes, order, csty)
        if not d14:
            assert np.array_equal(R.decode(got, Cn, H, W), px.astype(np.int32)), (W, H, TW, TH, L, off, Cn, sizes, order, csty)
        checked += 1
    assert checked >= 35
