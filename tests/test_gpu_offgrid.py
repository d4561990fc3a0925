"""-m gpu: tiles and images OFF the origin on the GPU: odd-start 5/3 and 9/7 in K2 / K6, band
coordinates and partial first code-blocks in the geometry, image layouts in the writer.

Done = GPU == reference on 1000 x 1000 tiles and on image_offset_x0 / y0 = 1 (WaveletFwd.cpp:884-905, :812-815,
TileComponent.cpp:131-138); the CPU side of the same (oracle == reference) is tests/test_offgrid_cpu.py."""
import numpy as np
import pytest

import grok_amd as G
import oracle as O
import refharness as R
import synth
import gpuutil as U
import j2kparse as J
from test_offgrid_cpu import oracle_image_codestream, ref_defects, OFFGRID

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")

# W, H, levels, origin: odd starts at some or all levels, lone low- / high-pass rows and columns, strips with an odd start
ORIGIN_CASES = [(8, 8, 1, (1, 1)), (64, 64, 3, (1, 0)), (65, 33, 3, (0, 1)), (100, 77, 5, (33, 95)), (17, 1, 2, (3, 3)), (1, 9, 2, (5, 0)),
                (1, 17, 3, (5, 0)), (3, 3, 1, (1, 1)), (2, 2, 2, (1, 1)), (255, 257, 5, (1, 1)), (512, 512, 5, (7, 21)), (1000, 1000, 5, (1000, 3000)),
                (999, 999, 5, (1, 1)), (1500, 700, 4, (449, 3)), (4, 600, 3, (2, 1)), (1, 1, 3, (7, 7)), (1, 1, 1, (1, 0)), (40, 1, 2, (0, 1)),
                (125, 125, 5, (875, 125)), (2, 300, 4, (31, 31))]


@pytest.mark.parametrize("W,H,L,org", ORIGIN_CASES)
def test_dwt53_with_origin(W, H, L, org):
    rng = np.random.default_rng(W * 7 + H + org[0])
    a = rng.integers(-300, 300, size=(2, H, W)).astype(np.int32)
    p = G.TileParams.make(W, H, 1, 8, L, mct=False, origin=org)
    d_in = U.upload_planes(a, p)
    d_out = U.dev_planes(p, 2)
    U.ctx().stage_dwt_fwd(p, 2, d_in.data_ptr(), d_out.data_ptr())
    U.ctx().synchronize()
    got = U.planes_to_numpy(d_out, p, 2)
    for k in range(2):
        want = O.dwt53_fwd(a[k], L, origin=org)
        assert np.array_equal(got[k], want), "plane %d" % k
    # and back on the GPU: the pair is lossless for every origin
    d_back = U.dev_planes(p, 2)
    U.ctx().stage_dwt_inv(p, 2, d_out.data_ptr(), d_back.data_ptr())
    U.ctx().synchronize()
    assert np.array_equal(U.planes_to_numpy(d_back, p, 2), a)


@pytest.mark.parametrize("W,H,L,org", ORIGIN_CASES)
def test_dwt97_with_origin_bit_exact(W, H, L, org):
    rng = np.random.default_rng(W * 11 + H + org[1])
    f = (rng.standard_normal((2, H, W)) * 200).astype(np.float32)
    p = G.TileParams.make(W, H, 1, 8, L, irreversible=True, mct=False, origin=org)
    d_in = U.upload_planes(f.view(np.int32), p)
    d_out = U.dev_planes(p, 2)
    U.ctx().stage_dwt_fwd(p, 2, d_in.data_ptr(), d_out.data_ptr())
    U.ctx().synchronize()
    got = U.planes_to_numpy(d_out, p, 2)
    for k in range(2):
        want = O.dwt97_fwd(f[k], L, origin=org).view(np.int32)
        assert np.array_equal(got[k], want), "plane %d: max ulp %d" % (k, np.abs(got[k].astype(np.int64) - want).max())


@pytest.mark.parametrize("W,H,L,org", ORIGIN_CASES)
@pytest.mark.parametrize("irrev", [False, True])
def test_idwt_with_origin(W, H, L, org, irrev):
    rng = np.random.default_rng(W * 13 + H + org[0] + irrev)
    if irrev:
        a = (rng.standard_normal((2, H, W)) * 200).astype(np.float32)
    else:
        a = rng.integers(-3000, 3000, size=(2, H, W)).astype(np.int32)         # arbitrary Mallat content (odd lone samples too)
    p = G.TileParams.make(W, H, 1, 8, L, irreversible=irrev, mct=False, origin=org)
    d_in = U.upload_planes(a.view(np.int32), p)
    d_out = U.dev_planes(p, 2)
    U.ctx().stage_dwt_inv(p, 2, d_in.data_ptr(), d_out.data_ptr())
    U.ctx().synchronize()
    got = U.planes_to_numpy(d_out, p, 2)
    for k in range(2):
        want = (O.dwt97_inv(a[k], L, origin=org) if irrev else O.dwt53_inv(a[k], L, origin=org)).view(np.int32)
        assert np.array_equal(got[k], want), "plane %d" % k


def _tile_blocks_equal_oracle(p, px):
    """One tile through the whole GPU path == the oracle's blocks, byte for byte."""
    table, coded = U.ctx().encode_host(p, px)
    _, lens, ocoded = O.encode_tile_rev(px, p.prec, p.num_levels, origin=(p.tile_x0, p.tile_y0))
    assert np.array_equal(table["length"], lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for i in range(len(lens)):
        o = int(table["offset"][i])
        assert bytes(coded[o:o + int(lens[i])]) == bytes(ocoded[off[i]:off[i + 1]]), "block %d" % i
    return table, coded


@pytest.mark.parametrize("C,W,H,prec,L,org", [(3, 100, 77, 8, 5, (33, 95)), (3, 999, 999, 8, 5, (1, 1)), (1, 1000, 1000, 12, 5, (1000, 3000)),
                                             (3, 1, 40, 8, 3, (200, 1)), (3, 130, 1, 8, 4, (7, 73)), (3, 257, 129, 16, 4, (63, 65))])
def test_tile_at_origin_equals_oracle_and_round_trips(C, W, H, prec, L, org):
    px = synth.g2(C, H, W, prec, seed=W + org[0])
    p = G.TileParams.make(W, H, C, prec, L, origin=org)
    table, coded = _tile_blocks_equal_oracle(p, px)
    back = U.ctx().decode_host(p, table, coded)
    assert np.array_equal(back[0], px)


@needs_ref
@pytest.mark.parametrize("W,H,TW,TH,L,off", OFFGRID + [
    (2000, 2000, 1000, 1000, 5, (0, 0)),            # the 1000 x 1000 tiles of the verdict: four tiles, four geometries
    (1999, 1999, 1000, 1000, 5, (1, 1)),            # image_offset_x0 = y0 = 1: tiles [1, 1000) and [1000, 2000)
    (1500, 900, 1024, 1024, 5, (3, 5)),             # ragged last tiles, everything odd
])
def test_encode_image_is_the_reference_file(monkeypatch, W, H, TW, TH, L, off):
    px = synth.g2(3, H, W, 8, seed=77)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
    assert ref_defects(layout, L) == (False, False)
    want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
    got = U.ctx().encode_image(layout, G.TileParams.make(1, 1, 3, 8, L), px)
    assert got == want
    assert np.array_equal(R.decode(got, 3, H, W), px.astype(np.int32))


@needs_ref
@pytest.mark.parametrize("W,H,T,off", [(2000, 2000, 1000, (1, 1)), (200, 200, 100, (1, 1))])
def test_encode_image_with_one_sample_wide_last_tiles(monkeypatch, W, H, T, off):
    """Image offset (1, 1) and a tile size that divides the image: the last tile column / row is ONE sample wide (lone
    high-pass samples, empty resolutions).  GPU == oracle, always; == grk_compress where its encoder is not damaged (D13:
    the 200 x 200 case, tests/test_offgrid_cpu.py); grk_decompress reads ours back exactly."""
    px = synth.g2(3, H, W, 8, seed=5)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    layout = G.ImageLayout.make(W, H, T, T, offset=off)
    d13, d14 = ref_defects(layout, 5)
    assert d13 == (W == 200) and not d14
    got = U.ctx().encode_image(layout, G.TileParams.make(1, 1, 3, 8, 5), px, flags=G.CS_TLM)
    assert got == oracle_image_codestream(px, 8, 5, layout, flags=G.CS_TLM)
    assert np.array_equal(R.decode(got, 3, H, W), px.astype(np.int32))
    if not d13:
        monkeypatch.setenv("REF_WRITE_TLM", "1")
        want, _ = R.encode(px, 8, TW=T, TH=T, numres=6, mode=1)
        assert got == want


def _gpu_decode_stream_with_offset(cs, part1):
    info = J.parse(cs)
    p = G.TileParams.make(info["W"], info["H"], info["C"], info["prec"], info["levels"], irreversible=bool(info["irreversible"]),
                          mct=bool(info["mct"]), part1=part1, cblksty=info["cblk_sty"] & 0x3F if part1 else 0,
                          origin=(info["x0"], info["y0"]))
    blocks, _ = G.tile_layout(p)
    rows, data = J.decode_table(info, blocks, part1)
    table = np.array(rows, dtype=G.capi.CODED_DTYPE)
    c = U.ctx()
    c.set_decode_qcd([(e << 11) | m for e, m in info["qcd"]] if info["irreversible"] else [])
    try:
        return c.decode_host(p, table, data)[0]
    finally:
        c.set_decode_qcd([])


@needs_ref
@pytest.mark.parametrize("off", [(1, 1), (7, 0), (32, 33), (95, 1)])
@pytest.mark.parametrize("ht,irrev", [(1, 0), (0, 0), (0, 1)])
def test_decode_reference_stream_off_the_origin(monkeypatch, off, ht, irrev):
    """grk_compress -d x0,y0 streams (HT 5/3, Part-1 5/3, Part-1 9/7) decoded on the GPU == grk_decompress."""
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    done = 0
    for (C, H, W, numres) in [(3, 75, 131, 5), (3, 300, 517, 6), (1, 1, 40, 3), (1, 37, 1, 4), (1, 2, 2, 2), (1, 3, 40, 3), (1, 40, 3, 3),
                              (1, 1, 1, 2), (1, 5, 1, 3), (1, 1, 5, 3)]:
        d13, d14 = ref_defects(G.ImageLayout.make(W, H, W + off[0], H + off[1], offset=off), numres - 1)
        if d13 or (d14 and not irrev):
            continue
        px = synth.g2(C, H, W, 8, seed=off[0] + numres)
        cs, _ = R.encode(px, 8, TW=W + off[0], TH=H + off[1], numres=numres, mode=1, ht=ht, irrev=irrev)
        got = _gpu_decode_stream_with_offset(cs, part1=not ht).astype(np.int32)
        assert np.array_equal(got, R.decode(cs, C, H, W)), (C, H, W, numres)
        if not irrev:
            assert np.array_equal(got, px.astype(np.int32))
        done += 1
    assert done >= 7


@pytest.mark.parametrize("C,W,H,L,org,irrev", [(1, 64, 64, 3, (1, 1), False), (3, 469, 311, 5, (33, 95), False), (3, 300, 200, 4, (7, 0), True),
                                                  (1, 999, 999, 5, (1, 1), False), (3, 130, 70, 2, (7, 3), False), (1, 40, 1, 2, (0, 1), False)])
def test_region_decode_off_the_origin_equals_crop_of_full_decode(C, W, H, L, org, irrev):
    """grk_amd_decode_region of a tile anywhere on the canonical grid: the window's pixels == the same crop of the full decode
    (the planes are poisoned with another image's coefficients first, so a block or strip wrongly skipped would show)."""
    rng = np.random.default_rng(W + org[0])
    px = synth.g2(C, H, W, 8, seed=5 + W)
    p = G.TileParams.make(W, H, C, 8, L, irreversible=irrev, origin=org)
    c = U.ctx()
    table, coded = c.encode_host(p, px)
    t2, c2 = c.encode_host(p, synth.g2(C, H, W, 8, seed=77 + W))
    full = c.decode_host(p, table, coded)[0]
    if not irrev:
        assert np.array_equal(full, px)
    wins = [(0, 0, W, H), (0, 0, min(W, 9), min(H, 7)), (W - min(W, 5), H - min(H, 3), W, H)]
    for _ in range(6):
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
        wins.append((x0, y0, int(rng.integers(x0 + 1, W + 1)), int(rng.integers(y0 + 1, H + 1))))
    for (x0, y0, x1, y1) in wins:
        c.decode_host(p, t2, c2)                                  # poison
        got = c.decode_region_host(p, table, coded, x0, y0, x1, y1)
        assert np.array_equal(got, full[:, y0:y1, x0:x1]), (x0, y0, x1, y1)


def test_random_layouts_gpu_equals_oracle_and_decodes():
    """A sweep like tests/test_offgrid_cpu.py's, on the GPU: random image sizes, offsets, tile sizes, level counts -- the
    codestream of grk_amd_encode_image == oracle tiles through the same writer, and every tile decodes back to its pixels
    (its own table rows, parameters with its origin; region decode of a random window of it == the crop)."""
    rng = np.random.default_rng(2024)
    c = U.ctx()
    for it in range(30):
        W, H = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        off = (int(rng.integers(0, 70)), int(rng.integers(0, 70)))
        TW, TH = int(rng.integers(max(off[0] + 1, 16), 260)), int(rng.integers(max(off[1] + 1, 16), 260))
        L = int(rng.integers(0, 6))
        Cn = int(rng.choice([1, 3]))
        prec = int(rng.choice([8, 8, 12]))
        layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
        px = synth.g2(Cn, H, W, prec, seed=int(rng.integers(1, 1000)))
        base = G.TileParams.make(1, 1, Cn, prec, L)
        got = c.encode_image(layout, base, px)
        assert got == oracle_image_codestream(px, prec, L, layout), (W, H, TW, TH, L, off, Cn, prec)
        for p in G.layout_tiles(layout, base):
            ox, oy = p.tile_x0 - off[0], p.tile_y0 - off[1]
            tile = np.ascontiguousarray(px[:, oy:oy + p.tile_h, ox:ox + p.tile_w])
            table, coded = c.encode_host(p, tile)
            back = c.decode_host(p, table, coded)[0]
            assert np.array_equal(back, tile), (W, H, TW, TH, L, off, Cn, prec, p.tile_x0, p.tile_y0)
            if L >= 1:
                x0, y0 = int(rng.integers(0, p.tile_w)), int(rng.integers(0, p.tile_h))
                x1, y1 = int(rng.integers(x0 + 1, p.tile_w + 1)), int(rng.integers(y0 + 1, p.tile_h + 1))
                win = c.decode_region_host(p, table, coded, x0, y0, x1, y1)
                assert np.array_equal(win, tile[:, y0:y1, x0:x1]), (W, H, TW, TH, L, off, p.tile_x0, p.tile_y0, x0, y0, x1, y1)
