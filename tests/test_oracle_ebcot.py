"""CPU: the Part-1 (EBCOT/MQ) block-decoder oracle (oracle/ebcot_oracle.c) pinned
against the REAL reference T1 (oracle/_ref): blocks coded by Grok's own T1::compress_cblk are decoded by
both, sample for sample, and truncated pass sequences as well."""
import numpy as np
import pytest

import oracle as O
import refharness as R

pytestmark = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


def _block(rng, w, h, bits, mode):
    mag = rng.integers(0, 1 << bits, size=(h, w))
    if mode == 1:
        mag = mag >> rng.integers(0, bits + 1, size=(h, w))
    elif mode == 2:
        mag = np.where(rng.random((h, w)) < 0.95, 0, mag & 7)
    elif mode == 3:
        mag = np.zeros((h, w), np.int64); mag[h // 2, w // 3] = 5
    elif mode == 4:
        mag = np.full((h, w), (1 << bits) - 1)
    sign = np.where(rng.random((h, w)) < 0.5, -1, 1)
    return (mag * sign).astype(np.int32)


@pytest.mark.parametrize("w,h,bits", [(64, 64, 8), (64, 64, 12), (32, 32, 10), (37, 3, 8), (1, 1, 5), (5, 64, 9), (64, 1, 7),
                                      (2, 2, 3), (63, 31, 14), (4, 4, 1), (13, 7, 16)])
@pytest.mark.parametrize("orient", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_t1_decode_equals_reference(w, h, bits, orient, mode):
    rng = np.random.default_rng(w * 31 + h * 5 + bits * 3 + orient + 7 * mode)
    coef = _block(rng, w, h, bits, mode)
    cb, npass, nbps = R.t1_encode_block(coef, orient)
    ref = R.t1_decode_block(cb, npass, nbps, orient, w, h)
    got = O.t1_decode_block(cb, npass, nbps, orient, w, h)
    assert np.array_equal(got, ref)
    assert np.array_equal(O.t1_dequant_rev(got), coef)            # all passes decoded: lossless


@pytest.mark.parametrize("keep", [1, 2, 3, 4, 7, 11])
def test_t1_decode_truncated_pass_sequence(keep):
    """Decoding fewer passes than were coded (what a quality-layer truncation leaves) matches too."""
    rng = np.random.default_rng(keep)
    coef = _block(rng, 64, 64, 10, 1)
    cb, npass, nbps = R.t1_encode_block(coef, 3)
    k = min(keep, npass)
    assert np.array_equal(O.t1_decode_block(cb, k, nbps, 3, 64, 64), R.t1_decode_block(cb, k, nbps, 3, 64, 64))


# ---- code-block styles and codeword segments (row a13: LAZY / RESET / TERMALL / VSC / PTERM / SEGSYM) ----------
STYLES = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x01 | 0x04, 0x02 | 0x08 | 0x20, 0x01 | 0x02 | 0x04 | 0x08 | 0x10 | 0x20, 0x04 | 0x10]


@pytest.mark.parametrize("sty", STYLES)
@pytest.mark.parametrize("w,h,bits,mode", [(64, 64, 11, 0), (64, 64, 9, 1), (32, 32, 8, 2), (37, 13, 10, 1), (64, 6, 7, 0),
                                           (5, 64, 12, 1), (1, 1, 6, 0), (64, 64, 3, 0)])
def test_t1_decode_styles_equal_reference(w, h, bits, mode, sty):
    rng = np.random.default_rng(w * 13 + h * 7 + bits + sty * 101 + mode)
    for orient in (0, 1, 3):
        coef = _block(rng, w, h, bits, mode)
        cb, segs, nbps = R.t1_encode_block_sty(coef, orient, sty)
        if sty & 0x04:
            assert all(n == 1 for _, n in segs)                     # TERMALL: one segment per pass
        ref = R.t1_decode_block_sty(cb, segs, nbps, orient, sty, w, h)
        got, bad = O.t1_decode_block_sty(cb, segs, nbps, orient, sty, w, h)
        assert bad == 0
        assert np.array_equal(got, ref)
        assert np.array_equal(O.t1_dequant_rev(got), coef)


def test_t1_decode_styles_truncated_segments():
    """Dropping trailing segments (a layer truncation at a terminated pass) decodes like the reference."""
    rng = np.random.default_rng(77)
    coef = _block(rng, 64, 64, 10, 1)
    for sty in (0x04, 0x01 | 0x04, 0x01):
        cb, segs, nbps = R.t1_encode_block_sty(coef, 2, sty)
        for keep in (1, 2, len(segs) // 2, len(segs) - 1):
            part = segs[:max(keep, 1)]
            n = sum(a for a, _ in part)
            assert np.array_equal(O.t1_decode_block_sty(cb[:n], part, nbps, 2, sty, 64, 64)[0],
                                  R.t1_decode_block_sty(cb[:n], part, nbps, 2, sty, 64, 64))


# ---- whole Part-1 streams of the reference encoder through the test-side Tier-2 reader ------------------
import grok_amd as G
import chain
import j2kparse as J
import synth


def _oracle_decode_stream(cs, part1):
    info = J.parse(cs)
    W, H, Cn, prec, L, irrev = info["W"], info["H"], info["C"], info["prec"], info["levels"], bool(info["irreversible"])
    org = (info["x0"], info["y0"])
    p = G.TileParams.make(W, H, Cn, prec, L, irreversible=irrev, mct=bool(info["mct"]), part1=part1, origin=org,
                          precincts=info["prc"] if info["scod"] & 1 else None)
    blocks, _ = G.tile_layout(p)
    rows, data = J.decode_table(info, blocks, part1)
    seglist = J.segment_list(info, blocks)
    mall = [np.zeros((H, W), np.float32 if irrev else np.int32) for _ in range(Cn)]
    for (off, ln, extra), b, segs in zip(rows, blocks, seglist):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        if not ln:
            continue
        expn, mant = info["qcd"][chain.band_index(b)]
        if part1:
            if info["cblk_sty"]:
                v, bad = O.t1_decode_block_sty(data[off:off + ln], segs, extra & 0xFF, b.band, info["cblk_sty"], bw, bh)
                assert bad == 0
            else:
                v = O.t1_decode_block(data[off:off + ln], extra >> 8, extra & 0xFF, b.band, bw, bh)
            if irrev:
                v = O.t1_dequant_irrev(v, np.float32((1.0 + mant / 2048.0) * 2.0 ** (prec - expn)))
            else:
                v = O.t1_dequant_rev(v)
        else:
            sm = O.ht_decode_block(data[off:off + ln], extra, bw, bh)
            v = O.ht_dequant_rev(sm, extra)
        mall[b.comp][b.py:b.py + bh, b.px:b.px + bw] = v
    planes = [O.dwt97_inv(m, L, origin=org) if irrev else O.dwt53_inv(m, L, origin=org) for m in mall]
    planes = [pl.view(np.int32) if irrev else pl for pl in planes]
    return np.stack(O.color_inv_store(planes, prec, irrev, bool(info["mct"])))


@pytest.mark.parametrize("C,H,W,prec,numres", [(3, 96, 160, 8, 5), (1, 128, 128, 8, 4), (3, 128, 128, 12, 6)])
@pytest.mark.parametrize("irrev", [0, 1])
def test_reference_part1_stream_oracle_chain_equals_grk_decompress(C, H, W, prec, numres, irrev):
    """BASELINE configs[4] shape on the CPU: a Part-1 stream written by grk_compress, its blocks recovered by the
    test-side T2 reader, decoded by the oracle chain (MQ decode -> ShiftFilter/ScaleFilter -> inverse 5/3 | 9/7
    -> inverse RCT/ICT) == grk_decompress, pixel for pixel."""
    px = synth.g2(C, H, W, prec)
    cs, _ = R.encode(px, prec, numres=numres, mode=1, ht=0, irrev=irrev)
    assert np.array_equal(_oracle_decode_stream(cs, True), R.decode(cs, C, H, W))


@pytest.mark.parametrize("off", [(1, 1), (7, 0), (32, 33), (95, 1)])
@pytest.mark.parametrize("irrev", [0, 1])
def test_reference_stream_off_the_origin_oracle_chain_equals_grk_decompress(monkeypatch, off, irrev):
    """An image whose area starts off the origin (grk_compress -d): odd-start synthesis (WaveletReverse.cpp:595-606,
    :1011-1062), band coordinates and partial first code-blocks on the decode side."""
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    from test_offgrid_cpu import ref_defects
    done = 0
    for (C, H, W, numres) in [(3, 75, 131, 5), (1, 1, 40, 3), (1, 37, 1, 4), (1, 2, 2, 2), (1, 3, 40, 3), (1, 40, 3, 3), (1, 37, 1, 2),
                              (1, 1, 1, 2), (1, 5, 1, 3), (1, 1, 5, 3)]:
        d13, d14 = ref_defects(G.ImageLayout.make(W, H, W + off[0], H + off[1], offset=off), numres - 1)
        if d13 or (d14 and not irrev):      # the reference's own defects (tests/test_offgrid_cpu.py); D14 is in the 5/3 only
            continue
        px = synth.g2(C, H, W, 8, seed=off[0] + numres)
        cs, _ = R.encode(px, 8, TW=W + off[0], TH=H + off[1], numres=numres, mode=1, ht=0, irrev=irrev)     # (one tile)
        assert np.array_equal(_oracle_decode_stream(cs, True), R.decode(cs, C, H, W)), (C, H, W, numres)
        done += 1
    assert done >= 7


@pytest.mark.parametrize("sizes", ["128,128", "64,32,32,64", "256,256,64,64,16,16"])
@pytest.mark.parametrize("ht,irrev", [(1, 0), (0, 0), (0, 1)])
def test_reference_stream_with_precincts_oracle_chain_equals_grk_decompress(monkeypatch, sizes, ht, irrev):
    """grk_compress -c streams (HT 5/3, Part-1 5/3, Part-1 9/7; with SOP + EPH, RLCP, an image offset): the code-block
    partition cut by the precincts and the packet per precinct on the decode side."""
    monkeypatch.setenv("REF_PRECINCTS", sizes)
    for (C, H, W, numres, off, order, csty) in [(3, 192, 256, 5, (0, 0), 0, 0), (1, 130, 77, 4, (0, 0), 1, 6), (3, 100, 150, 4, (33, 95), 0, 2)]:
        monkeypatch.setenv("REF_IMG_X0", str(off[0]))
        monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
        monkeypatch.setenv("REF_PROG_ORDER", str(order))
        monkeypatch.setenv("REF_CSTY", str(csty))
        px = synth.g2(C, H, W, 8, seed=numres)
        cs, _ = R.encode(px, 8, TW=W + off[0], TH=H + off[1], numres=numres, mode=1, ht=ht, irrev=irrev)
        got = _oracle_decode_stream(cs, not ht)
        assert np.array_equal(got, R.decode(cs, C, H, W)), (C, H, W, numres, off)
        if not irrev:
            assert np.array_equal(got, px.astype(np.int32))


@pytest.mark.parametrize("sty", [0x01, 0x02, 0x04, 0x08, 0x20, 0x01 | 0x04, 0x3F])
@pytest.mark.parametrize("irrev", [0, 1])
def test_reference_part1_styled_stream_oracle_chain_equals_grk_decompress(sty, irrev):
    """The same with `grk_compress -M`: code-block styles and their codeword segments through Tier-2."""
    px = synth.g2(3, 128, 192, 10)
    cs, _ = R.encode(px, 10, numres=4, mode=1, ht=0, irrev=irrev, cblksty=sty)
    info = J.parse(cs)
    assert info["cblk_sty"] == sty
    assert np.array_equal(_oracle_decode_stream(cs, True), R.decode(cs, 3, 128, 192))


def test_reference_ht_stream_through_t2_reader():
    px = synth.g2(3, 96, 160, 8)
    cs, _ = R.encode(px, 8, numres=5, mode=1)
    assert np.array_equal(_oracle_decode_stream(cs, False), px.astype(np.int32))
