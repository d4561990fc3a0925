"""CPU: the pointer marker segments (TLM in the main header, PLT in the tile-part headers -- SURVEY.md §8f N4's
"PLT/TLM random access", codestream/markers/LengthMarkers.cpp) as the reference's encoder writes them with
grk_compress -X / -L, the tile-part pieces a parallel writer puts together, and the tile-part locator."""
import os

import numpy as np
import pytest

import grok_amd as G
import cshelp
import oracle as O
import refharness as R
import synth
from grok_amd.capi import CODED_DTYPE

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref (the real reference) not built here")


def _oracle_tiles(px, prec, L, TW, TH):
    C, H, W = px.shape
    p = G.TileParams.make(TW, TH, C, prec, L)
    tabs, chunks, off = [], [], 0
    for ty in range(H // TH):
        for tx in range(W // TW):
            tile = np.ascontiguousarray(px[:, ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW])
            _, lens, coded = O.encode_tile_rev(tile, prec, L)
            t = np.zeros(len(lens), CODED_DTYPE)
            t["length"] = lens
            t["offset"] = off + np.concatenate([[0], np.cumsum(lens)[:-1]])
            off += int(lens.sum())
            tabs.append(t)
            chunks.append(coded)
    return p, np.concatenate(tabs), np.concatenate(chunks)


@needs_ref
@pytest.mark.parametrize("tlm,plt", [(1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("C,H,W,TW,TH,L", [(3, 256, 256, 128, 128, 3), (1, 192, 320, 320, 192, 4), (3, 128, 384, 128, 128, 2)])
def test_tlm_plt_as_grk_compress_writes_them(monkeypatch, tlm, plt, C, H, W, TW, TH, L):
    px = synth.g2(C, H, W, 8)
    monkeypatch.setenv("REF_WRITE_TLM", str(tlm))
    monkeypatch.setenv("REF_WRITE_PLT", str(plt))
    want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
    p, table, coded = _oracle_tiles(px, 8, L, TW, TH)
    got = G.write_codestream(p, W, H, table, coded, flags=(G.CS_TLM if tlm else 0) | (G.CS_PLT if plt else 0))
    assert got == want
    assert np.array_equal(R.decode(got, C, H, W), px.astype(np.int32))


def test_parallel_writer_pieces_make_the_same_file_and_the_locator_finds_them():
    """main header + every tile-part written on its own (sizes first: TLM needs them) + EOC == the one-call writer;
    grk_amd_locate_tile_parts returns the same places from the TLM marker as from hopping over the SOTs."""
    px = synth.g2(3, 256, 384, 8)
    p, table, coded = _oracle_tiles(px, 8, 3, 128, 128)
    ntiles, bpt = 6, len(table) // 6
    for flags in (0, G.CS_TLM, G.CS_PLT, G.CS_TLM | G.CS_PLT):
        whole = G.write_codestream(p, 384, 256, table, coded, flags=flags)
        sizes = [G.write_tile_part(p, t, table[t * bpt:(t + 1) * bpt], None, flags=flags, size_only=True) for t in range(ntiles)]
        parts = [G.write_tile_part(p, t, table[t * bpt:(t + 1) * bpt], coded, flags=flags) for t in range(ntiles)]
        assert [len(x) for x in parts] == sizes
        hdr = G.write_main_header(p, 384, 256, flags=flags, tile_part_bytes=sizes)
        assert hdr + b"".join(parts) + b"\xff\xd9" == whole
        where, used_tlm = G.locate_tile_parts(whole)
        assert used_tlm == bool(flags & G.CS_TLM)
        assert [w[2] for w in where] == list(range(ntiles)) and [w[1] for w in where] == sizes
        at = len(hdr)
        for (o, n, _), part in zip(where, parts):
            assert o == at and whole[o:o + n] == part
            at += n


@needs_ref
def test_locator_on_reference_streams(monkeypatch):
    px = synth.g2(3, 256, 256, 8)
    plain, _ = R.encode(px, 8, TW=128, TH=128, numres=4, mode=1)
    monkeypatch.setenv("REF_WRITE_TLM", "1")
    with_tlm, _ = R.encode(px, 8, TW=128, TH=128, numres=4, mode=1)
    a, ua = G.locate_tile_parts(plain)
    b, ub = G.locate_tile_parts(with_tlm)
    assert (ua, ub) == (False, True) and len(a) == len(b) == 4
    assert [x[1:] for x in a] == [x[1:] for x in b]
    shift = len(with_tlm) - len(plain)
    assert [x[0] + shift for x in a] == [x[0] for x in b]


@pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("order", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("csty", [0, 2, 4, 6])
def test_progression_orders_sop_eph_are_the_reference_files(monkeypatch, order, csty):
    """grk_compress -p LRCP|RLCP|RPCL|PCRL|CPRL, -S (SOP), -E (EPH): with one layer and one precinct per resolution the five
    orders are two loop nests; SOP numbers the packets of a tile, EPH closes every packet header.  Oracle tiles + the
    product's writer == the reference's file, with PLT + TLM as well (the packet lengths include the markers); multi-tile."""
    import cshelp
    px = synth.g2(3, 192, 256, 8, seed=order * 7 + csty)
    monkeypatch.setenv("REF_PROG_ORDER", str(order))
    monkeypatch.setenv("REF_CSTY", str(csty))
    flags = G.CS_PROG(order) | (G.CS_SOP if csty & 2 else 0) | (G.CS_EPH if csty & 4 else 0)
    for tlmplt in (0, 1):
        monkeypatch.setenv("REF_WRITE_TLM", str(tlmplt))
        monkeypatch.setenv("REF_WRITE_PLT", str(tlmplt))
        f = flags | ((G.CS_TLM | G.CS_PLT) if tlmplt else 0)
        want, _ = R.encode(px, 8, TW=128, TH=64, numres=4, mode=1)
        got = cshelp.oracle_codestream(px, 8, 3, 128, 64, flags=f)
        assert got == want, (order, csty, tlmplt)
    assert np.array_equal(R.decode(got, 3, 192, 256), px.astype(np.int32))


@needs_ref
def test_plt_longer_than_one_marker_segment_is_split(monkeypatch):
    """Small precincts multiply the packets: 2048 x 2048 x 3 with 16 x 16 precincts has 65 280 of them, ~125 KB of packet
    lengths -- more than Lplt (16 bits) can announce (the writer used to wrap the length).  The lengths go into
    marker segments Zplt = 0, 1, ... of at most 65535 bytes, no length split between two; they add up to the tile-part's
    packets; the reference's decoder reads the file back.  (The reference's own continuation segments lack the Zplt byte,
    LengthMarkers.cpp:313-331 -- a defect -- so its -L file is not the yardstick here; without PLT the files are equal.)"""
    from test_precincts_cpu import oracle_codestream_prc, exps_from_sizes
    W = H = 2048
    L = 3
    px = synth.g2(3, H, W, 8, seed=5)
    layout = G.ImageLayout.make(W, H, W, H)
    sizes = [(16, 16)] * (L + 1)
    monkeypatch.setenv("REF_PRECINCTS", ",".join("%d,%d" % s for s in sizes))
    want, _ = R.encode(px, 8, TW=W, TH=H, numres=L + 1, mode=1)
    prc = exps_from_sizes(sizes, L)
    plain = oracle_codestream_prc(px, 8, L, layout, prc, 0)
    assert plain == want
    got = oracle_codestream_prc(px, 8, L, layout, prc, G.CS_PLT)
    sot = got.index(b"\xff\x90")
    psot = int.from_bytes(got[sot + 6:sot + 10], "big")
    at, z, lens, segs = sot + 12, 0, [], 0
    while got[at:at + 2] == b"\xff\x58":
        lplt = int.from_bytes(got[at + 2:at + 4], "big")
        assert 3 <= lplt <= 65535 and got[at + 4] == z
        body = got[at + 5:at + 2 + lplt]
        assert body[-1] < 0x80                                   # a segment ends with the end of a length
        v = 0
        for b in body:
            v = (v << 7) | (b & 0x7F)
            if b < 0x80:
                lens.append(v)
                v = 0
        at += 2 + lplt
        z += 1
        segs += 1
    assert segs >= 2 and got[at:at + 2] == b"\xff\x93"
    assert len(lens) == 3 * sum((W >> (L - r)) // 16 * ((H >> (L - r)) // 16) for r in range(L + 1))
    assert sum(lens) == sot + psot - (at + 2)                    # the packets are what is left of the tile-part
    # the same packets as without PLT
    assert got[at + 2:sot + psot] == plain[plain.index(b"\xff\x93") + 2:-2]
    assert np.array_equal(R.decode(got, 3, H, W), px.astype(np.int32))


def test_tile_part_plan_materialises_to_the_written_tile_part():
    """grk_amd_plan_tile_part: the literal bytes + segment list of a tile-part, placed, are the bytes grk_amd_write_tile_part writes
    (every flag combination that changes the tile-part: PLT, SOP, EPH, the progression orders, precincts)."""
    import ctypes as C
    import oracle as O
    rng = np.random.default_rng(11)
    L = G.lib()
    L.grk_amd_plan_tile_part.restype = C.c_int64
    L.grk_amd_plan_tile_part.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    SEG = np.dtype([("dst", "<u8"), ("src", "<u8"), ("len", "<u4"), ("kind", "<u4")])
    for W, H, Cn, Lv, flags, precincts in ((256, 192, 3, 3, 0, None), (300, 200, 1, 4, G.CS_PLT | G.CS_SOP | G.CS_EPH | G.CS_PROG(2), None),
                                           (256, 256, 3, 3, G.CS_PROG(4) | G.CS_PLT, [(5, 5)] * 4), (64, 64, 3, 0, G.CS_PROG(3), None)):
        p = G.TileParams.make(W, H, Cn, 8, Lv, precincts=precincts)
        nb = L.grk_amd_tile_num_blocks(C.byref(p))
        t = np.zeros(nb, G.capi.CODED_DTYPE)
        t["length"] = rng.integers(0, 3000, nb)
        order = rng.permutation(nb)                       # (blocks anywhere in the coded buffer)
        offs = np.zeros(nb, np.int64)
        offs[order] = np.concatenate([[0], np.cumsum(t["length"][order].astype(np.int64))[:-1]])
        t["offset"] = offs
        coded = rng.integers(0, 256, int(t["length"].astype(np.int64).sum()) + 1, dtype=np.uint8)
        want = G.write_tile_part(p, 3, t, coded, flags=flags)
        nlit, nseg = C.c_uint64(0), C.c_uint64(0)
        total = L.grk_amd_plan_tile_part(C.byref(p), 3, flags, t.ctypes.data, None, 0, C.byref(nlit), None, 0, C.byref(nseg))
        assert total == len(want)
        lit = np.zeros(nlit.value, np.uint8)
        segs = np.zeros(nseg.value, SEG)
        assert L.grk_amd_plan_tile_part(C.byref(p), 3, flags, t.ctypes.data, lit.ctypes.data, lit.size, C.byref(nlit),
                                        segs.ctypes.data, segs.size, C.byref(nseg)) == total
        out = np.zeros(total, np.uint8)
        covered = 0
        for s in segs:
            src = coded if s["kind"] else lit
            out[int(s["dst"]):int(s["dst"]) + int(s["len"])] = src[int(s["src"]):int(s["src"]) + int(s["len"])]
            covered += int(s["len"])
        assert covered == total and out.tobytes() == want
