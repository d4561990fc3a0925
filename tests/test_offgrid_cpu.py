"""CPU: tiles and images OFF the origin -- odd-start lifting, band coordinates, partial first code-blocks, empty
resolutions (SURVEY.md Appendix A.3 "odd-start variant").

  * the oracle's lines / levels with an origin == the reference's own kernels (WaveletFwd.cpp:884-905, :782-842; harness
    ref_dwt53_row / ref_dwt97_row with even = 0, ref_dwt*_fwd_at),
  * the product's host geometry (grk_amd_tile_layout with tile_x0 / tile_y0) == the oracle's enumeration,
  * oracle tiles + the product's Tier-2 writer with an image layout == the bytes grk_compress writes for an image with
    image offsets / tile sizes that are no multiple of 2^levels (1-sample-wide edge tiles, empty resolutions included),
  * the forward / inverse pair is lossless for every origin."""
import numpy as np
import pytest

import grok_amd as G
import oracle as O
import refharness as R
import synth
from grok_amd.capi import CODED_DTYPE

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("irrev", [False, True])
def test_odd_start_line_matches_reference_kernel(irrev):
    rng = np.random.default_rng(5)
    L = R.lib()
    for n in list(range(1, 40)) + [63, 64, 65, 127, 500, 1001]:
        for even in (1, 0):
            row = rng.integers(-2000, 2000, n).astype(np.float32 if irrev else np.int32)
            want = row.copy()
            (L.ref_dwt97_row if irrev else L.ref_dwt53_row)(want.ctypes.data, n, even)
            got = O.dwt_row(row, 1 - even, irrev)
            assert np.array_equal(got.view(np.int32), want.view(np.int32)), (n, even)


@needs_ref
@pytest.mark.parametrize("irrev", [False, True])
def test_levels_with_origin_match_reference_kernels(irrev):
    rng = np.random.default_rng(6)
    L = R.lib()
    for (w, h, lv, x0, y0) in [(37, 29, 3, 1, 1), (64, 64, 5, 3, 7), (100, 50, 4, 1000, 1), (1, 17, 1, 5, 0), (19, 1, 2, 3, 3),
                               (33, 47, 5, 17, 33), (2, 2, 2, 1, 1), (125, 125, 5, 875, 125)]:
        a = rng.integers(-500, 500, (h, w)).astype(np.float32 if irrev else np.int32)
        want = a.copy()
        (L.ref_dwt97_fwd_at if irrev else L.ref_dwt53_fwd_at)(want.ctypes.data, w, h, w, lv, x0, y0)
        got = (O.dwt97_fwd if irrev else O.dwt53_fwd)(a, lv, origin=(x0, y0))
        assert np.array_equal(got.view(np.int32), want.view(np.int32)), (w, h, lv, x0, y0)
        if not irrev:
            assert np.array_equal(O.dwt53_inv(got, lv, origin=(x0, y0)), a)
        # (the inverse 9/7 with an origin is pinned against grk_decompress in test_oracle_ebcot.py: Grok's forward and inverse
        #  9/7 differ by the band gains the quantiser absorbs, so the pair alone is no round trip)


def test_layout_with_origin_matches_oracle_enumeration():
    rng = np.random.default_rng(7)
    cases = [(1000, 1000, 5, 1000, 3000), (999, 999, 5, 1, 1), (1, 1000, 5, 2000, 1), (64, 64, 3, 33, 95), (257, 129, 4, 63, 65)]
    for _ in range(40):
        cases.append((int(rng.integers(1, 400)), int(rng.integers(1, 400)), int(rng.integers(0, 6)), int(rng.integers(0, 300)),
                      int(rng.integers(0, 300))))
    for (w, h, lv, x0, y0) in cases:
        p = G.TileParams.make(w, h, 1, 8, lv, origin=(x0, y0))
        blocks, _ = G.tile_layout(p)
        want = O.enumerate_blocks(w, h, lv, origin=(x0, y0))
        assert len(blocks) == len(want), (w, h, lv, x0, y0)
        for b, o in zip(blocks, want):
            assert (b.px, b.py, b.x1 - b.x0, b.y1 - b.y0, b.res, b.band) == (o.x, o.y, o.w, o.h, o.res, o.band), (w, h, lv, x0, y0)
    # at the origin nothing changes
    a, _ = G.tile_layout(G.TileParams.make(300, 200, 1, 8, 4))
    b, _ = G.tile_layout(G.TileParams.make(300, 200, 1, 8, 4, origin=(0, 0)))
    assert [(x.px, x.py, x.x0, x.x1) for x in a] == [(x.px, x.py, x.x0, x.x1) for x in b]


def oracle_image_codestream(px, prec, L, layout, flags=0):
    """Oracle tiles (each with its own origin) -> the product's writer with the image layout."""
    Cn, H, W = px.shape
    base = G.TileParams.make(1, 1, Cn, prec, L)
    tabs, chunks, off = [], [], 0
    for p in G.layout_tiles(layout, base):
        ox, oy = p.tile_x0 - layout.x0, p.tile_y0 - layout.y0
        tile = np.ascontiguousarray(px[:, oy:oy + p.tile_h, ox:ox + p.tile_w])
        _, lens, coded = O.encode_tile_rev(tile, prec, L, origin=(p.tile_x0, p.tile_y0))
        assert len(lens) == G.lib().grk_amd_tile_num_blocks(p)
        t = np.zeros(len(lens), CODED_DTYPE)
        t["length"] = lens
        t["offset"] = off + np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(lens) else 0
        off += int(lens.sum())
        tabs.append(t)
        chunks.append(coded)
    return G.write_codestream_layout(layout, base, np.concatenate(tabs), np.concatenate(chunks), flags)


def _cd(v, n):
    return (v + (1 << n) - 1) >> n


def ref_defects(layout, L):
    """Which of the reference's own defects a layout runs into (so that a test knows what it may compare):
    D13 -- encoder: a level of zero WIDTH on an odd origin rewrites a coefficient of the level before (see the D13 test);
    D14 -- decoder: vertical 5/3 synthesis of a level of height 2 on an odd origin stores its second row into the next
           COLUMN (decompress_v_53, WaveletReverse.cpp:649-656: dest[1] instead of dest[strideDest])."""
    d13 = d14 = False
    for p in G.layout_tiles(layout, G.TileParams.make(1, 1, 1, 8, L)):
        for l in range(L):
            lx, ly = _cd(p.tile_x0, l), _cd(p.tile_y0, l)
            cw, ch = _cd(p.tile_x0 + p.tile_w, l) - lx, _cd(p.tile_y0 + p.tile_h, l) - ly
            d13 |= cw == 0 and (lx & 1) == 1 and ch > 0
            d14 |= ch == 2 and (ly & 1) == 1 and cw > 0
    return d13, d14


OFFGRID = [   # W, H, TW, TH, levels, image offset
    (250, 230, 100, 100, 3, (1, 1)),        # tiles at 1 / 100 / 200: odd starts, partial first blocks, ragged last tiles
    (300, 200, 125, 125, 5, (0, 0)),        # tile pitch no multiple of 2^levels, image at the origin (the 1000 x 1000 case, 1/8)
    (130, 70, 137, 73, 4, (7, 3)),          # one tile, image off the origin
    (96, 96, 200, 200, 5, (33, 95)),
    (200, 200, 100, 100, 2, (1, 1)),        # the last column / row of tiles is ONE sample wide / high
    (67, 200, 32, 100, 5, (0, 1)),          # 1-sample-HIGH last tiles, lone high-pass rows, resolutions without a sample
]


@needs_ref
@pytest.mark.parametrize("W,H,TW,TH,L,off", OFFGRID)
def test_oracle_offgrid_codestream_is_the_reference_file(monkeypatch, W, H, TW, TH, L, off):
    px = synth.g2(3, H, W, 8, seed=77)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
    assert ref_defects(layout, L) == (False, False)
    want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
    got = oracle_image_codestream(px, 8, L, layout)
    assert got == want
    assert np.array_equal(R.decode(got, 3, H, W), px.astype(np.int32))


@needs_ref
def test_random_layouts_against_the_reference(monkeypatch):
    """A sweep over small images, offsets, tile sizes and level counts: where the reference is free of D13 the bytes are
    its bytes, where it is free of D13 and D14 its decoder returns the image from our file."""
    rng = np.random.default_rng(99)
    checked = clean = 0
    for _ in range(60):
        W, H = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        off = (int(rng.integers(0, 40)), int(rng.integers(0, 40)))
        TW, TH = int(rng.integers(max(off[0] + 1, 8), 100)), int(rng.integers(max(off[1] + 1, 8), 100))
        L = int(rng.integers(0, 6))
        Cn = int(rng.choice([1, 3]))
        layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
        d13, d14 = ref_defects(layout, L)
        px = synth.g2(Cn, H, W, 8, seed=int(rng.integers(1, 1000)))
        monkeypatch.setenv("REF_IMG_X0", str(off[0]))
        monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
        got = oracle_image_codestream(px, 8, L, layout)
        if not d13:
            want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
            assert got == want, (W, H, TW, TH, L, off, Cn)
            checked += 1
        if not d14:
            assert np.array_equal(R.decode(got, Cn, H, W), px.astype(np.int32)), (W, H, TW, TH, L, off, Cn)
        clean += not (d13 or d14)
    assert checked >= 40 and clean >= 30


@needs_ref
def test_height_two_level_on_an_odd_origin_reference_decoder_defect_d14(monkeypatch):
    """D14 (see ref_defects): the file is the reference encoder's own, byte for byte, the oracle's inverse chain returns the
    image from it -- grk_decompress does not."""
    W, H, L, off = 130, 70, 2, (7, 3)
    px = synth.g2(3, H, W, 8, seed=77)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    layout = G.ImageLayout.make(W, H, 130, 70, offset=off)
    assert ref_defects(layout, L) == (False, True)
    want, _ = R.encode(px, 8, TW=130, TH=70, numres=L + 1, mode=1)
    assert oracle_image_codestream(px, 8, L, layout) == want
    bad = R.decode(want, 3, H, W) != px
    assert bad.any() and not bad[:, :67, :].any()               # the tile row [70, 73): rows 67..69 of the image
    for p in G.layout_tiles(layout, G.TileParams.make(1, 1, 3, 8, L)):
        ox, oy = p.tile_x0 - off[0], p.tile_y0 - off[1]
        tile = px[:, oy:oy + p.tile_h, ox:ox + p.tile_w].astype(np.int32) - 128
        y, cb, cr = O.rct_fwd(tile[0], tile[1], tile[2])
        for pl in (y, cb, cr):
            org = (p.tile_x0, p.tile_y0)
            assert np.array_equal(O.dwt53_inv(O.dwt53_fwd(pl, L, origin=org), L, origin=org), pl)


@needs_ref
def test_zero_width_resolution_on_an_odd_origin_reference_defect_d13(monkeypatch):
    """D13: a resolution of ZERO width whose origin is odd -- the tiles at x0 = 200 of a 1-sample-wide last tile column with
    5 levels: [ceil(200 / 16), ceil(201 / 16)) = [13, 13) -- still runs the reference's odd-start row function, whose
    width == 0 case falls into the general branch and rewrites row[0] from row[1] (WaveletFwd.cpp:884-905: only width == 1
    is special-cased).  row[0] is a real coefficient of the level before, so grk_compress writes a file its own decoder
    does not turn back into the image.  Nothing to be compatible with: the oracle (and the GPU) transform nothing where
    there is nothing, every other tile-part is byte-identical, and the reference DECODER reads our file back exactly."""
    W = H = 200
    px = synth.g2(3, H, W, 8, seed=77)
    monkeypatch.setenv("REF_IMG_X0", "1")
    monkeypatch.setenv("REF_IMG_Y0", "1")
    want, _ = R.encode(px, 8, TW=100, TH=100, numres=6, mode=1)
    got = oracle_image_codestream(px, 8, 5, G.ImageLayout.make(W, H, 100, 100, offset=(1, 1)))
    assert np.array_equal(R.decode(got, 3, H, W), px.astype(np.int32))
    try:                                                         # the reference's own file: last column wrong, or (the
        bad = R.decode(want, 3, H, W) != px                      # rewritten coefficient depends on what lies beside the
        assert bad.any() and not bad[:, :, :199].any()           # plane) a block its decoder rejects
    except RuntimeError:
        pass
    wa, _ = G.locate_tile_parts(want)
    ga, _ = G.locate_tile_parts(got)
    same = [want[a[0] + 12:a[0] + a[1]] == got[b[0] + 12:b[0] + b[1]] for a, b in zip(wa, ga)]
    assert same == [True, True, False, True, True, False, True, True, True]


@needs_ref
def test_offgrid_16bit_gray_with_markers(monkeypatch):
    W, H, off = 150, 90, (5, 9)
    px = synth.g2(1, H, W, 12, seed=3)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    monkeypatch.setenv("REF_WRITE_TLM", "1")
    monkeypatch.setenv("REF_WRITE_PLT", "1")
    want, _ = R.encode(px, 12, TW=64, TH=64, numres=4, mode=1)
    got = oracle_image_codestream(px, 12, 3, G.ImageLayout.make(W, H, 64, 64, offset=off), flags=G.CS_TLM | G.CS_PLT)
    assert got == want


def test_same_tile_geometry_groups():
    base = G.TileParams.make(1, 1, 3, 8, 5)
    # pitch 1024 at the origin: every tile like the first
    tiles = G.layout_tiles(G.ImageLayout.make(4096, 2048, 1024, 1024), base)
    assert len(tiles) == 8 and all(G.same_tile_geometry(tiles[0], t) for t in tiles)
    # pitch 1000, one level: the sub-bands start at 500 t on a code-block grid of 64 -> the partition repeats every 16 tiles
    base1 = G.TileParams.make(1, 1, 3, 8, 1)
    tiles = G.layout_tiles(G.ImageLayout.make(17000, 1000, 1000, 1000), base1)
    assert [t for t in range(17) if G.same_tile_geometry(tiles[0], tiles[t])] == [0, 16]
    # the old whole-image entry point serves one batch: it refuses what needs per-tile geometry
    p = G.TileParams.make(1000, 1000, 3, 8, 5)
    n = G.lib().grk_amd_tile_num_blocks(p)
    t = np.zeros(5 * n, CODED_DTYPE)
    with pytest.raises(RuntimeError):
        G.write_codestream(p, 5000, 1000, t, np.zeros(16, np.uint8))
