"""-m gpu: precincts on the GPU path -- the code-block partition they cut (blocks down to 2 x 2), the packets per precinct,
through the C ABI and through Grok's plugin protocol (grk_compress -c)."""
import numpy as np
import pytest

import grok_amd as G
import oracle as O
import refharness as R
import synth
import gpuutil as U
from test_precincts_cpu import exps_from_sizes, oracle_codestream_prc, CASES

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


@pytest.mark.parametrize("C,W,H,prec,L,org,sizes", [(3, 256, 192, 8, 4, (0, 0), [(128, 128)]), (1, 300, 210, 12, 3, (0, 0), [(64, 32), (32, 64)]),
                                                   (3, 257, 129, 8, 5, (33, 95), [(128, 64), (64, 64), (16, 16)]), (3, 96, 80, 8, 3, (0, 0), [(16, 16)]),
                                                   (3, 1024, 768, 8, 5, (0, 0), [(256, 256)])])
def test_tile_with_precincts_equals_oracle_and_round_trips(C, W, H, prec, L, org, sizes):
    px = synth.g2(C, H, W, prec, seed=W + L)
    prc = exps_from_sizes(sizes, L)
    p = G.TileParams.make(W, H, C, prec, L, origin=org, precincts=prc)
    table, coded = U.ctx().encode_host(p, px)
    _, lens, ocoded = O.encode_tile_rev(px, prec, L, origin=org, precincts=prc)
    assert np.array_equal(table["length"], lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for i in range(len(lens)):
        o = int(table["offset"][i])
        assert bytes(coded[o:o + int(lens[i])]) == bytes(ocoded[off[i]:off[i + 1]]), "block %d" % i
    assert np.array_equal(U.ctx().decode_host(p, table, coded)[0], px)
    x0, y0, x1, y1 = W // 3, H // 4, W // 3 + max(1, W // 5), H // 4 + max(1, H // 3)
    if L >= 1:
        assert np.array_equal(U.ctx().decode_region_host(p, table, coded, x0, y0, x1, y1), px[:, y0:y1, x0:x1])


@needs_ref
@pytest.mark.parametrize("W,H,TW,TH,L,off,sizes,order,extra", CASES)
def test_encode_image_with_precincts_is_the_reference_file(monkeypatch, W, H, TW, TH, L, off, sizes, order, extra):
    px = synth.g2(3, H, W, 8, seed=W + L)
    layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    monkeypatch.setenv("REF_PRECINCTS", ",".join("%d,%d" % s for s in sizes))
    monkeypatch.setenv("REF_PROG_ORDER", str(order))
    monkeypatch.setenv("REF_WRITE_TLM", "1" if extra & G.CS_TLM else "0")
    monkeypatch.setenv("REF_WRITE_PLT", "1" if extra & G.CS_PLT else "0")
    monkeypatch.setenv("REF_CSTY", str((2 if extra & G.CS_SOP else 0) | (4 if extra & G.CS_EPH else 0)))
    want, _ = R.encode(px, 8, TW=TW, TH=TH, numres=L + 1, mode=1)
    base = G.TileParams.make(1, 1, 3, 8, L, precincts=exps_from_sizes(sizes, L))
    got = U.ctx().encode_image(layout, base, px, flags=G.CS_PROG(order) | extra)
    assert got == want


@needs_ref
@pytest.mark.parametrize("sizes", [[(128, 128)], [(256, 256), (128, 128), (64, 64)], [(64, 32), (32, 64)]])
def test_plugin_file_protocol_with_precincts(tmp_path, monkeypatch, sizes):
    """grk_compress -c through the plugin: the tile tree carries the blocks precinct by precinct as the host's own structure
    does (compress_synch_with_plugin walks both in step), the host writes the packets: file == the pure-CPU encode."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    monkeypatch.setenv("REF_PRECINCTS", ",".join("%d,%d" % s for s in sizes))
    for Cn, H, W, prec in ((3, 192, 256, 8), (1, 300, 200, 12)):
        px = synth.g2(Cn, H, W, prec)
        path = str(tmp_path / ("in_%d_%d.%s" % (Cn, prec, "pgm" if Cn == 1 else "ppm")))
        R.write_pnm(path, px, prec)
        got = R.plugin_compress_file(px, prec, path, numres=5)
        assert not isinstance(got, int), "plugin refused: %s" % got
        cpu, _ = R.encode(px, prec, numres=5, mode=1)
        assert got == cpu


@needs_ref
@pytest.mark.parametrize("sizes", ["128,128", "64,32,32,64", "256,256,64,64,16,16"])
@pytest.mark.parametrize("ht,irrev", [(1, 0), (0, 0), (0, 1)])
def test_decode_reference_streams_with_precincts(monkeypatch, sizes, ht, irrev):
    """grk_compress -c streams decoded on the GPU through the C ABI and through the plugin's decode protocol == grk_decompress."""
    from test_gpu_decode import _gpu_decode_reference_stream
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    monkeypatch.setenv("REF_PRECINCTS", sizes)
    for (C, H, W, numres, order, csty) in [(3, 192, 256, 5, 0, 0), (1, 130, 77, 4, 1, 6)]:
        monkeypatch.setenv("REF_PROG_ORDER", str(order))
        monkeypatch.setenv("REF_CSTY", str(csty))
        px = synth.g2(C, H, W, 8, seed=numres)
        cs, _ = R.encode(px, 8, numres=numres, mode=1, ht=ht, irrev=irrev)
        want = R.decode(cs, C, H, W)
        got = _gpu_decode_reference_stream(cs, part1=not ht).astype(np.int32)
        assert np.array_equal(got, want), (C, H, W, numres)
        back, stages = R.plugin_decompress(cs, C, H, W)
        assert not isinstance(back, int), "plugin refused: %s (stages %s)" % (back, stages)
        assert np.array_equal(back, want)
