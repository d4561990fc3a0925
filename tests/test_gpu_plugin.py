"""-m gpu: the drop-in boundary, end to end through the REAL Grok library (oracle/_ref):
our plugin's tile tree -> grk_compress_with_plugin() -> file identical to Grok's pure-CPU encode."""
import ctypes as C
import os

import numpy as np
import pytest

import grok_amd as G
import gpuutil as U
import refharness as R
import synth

pytestmark = [pytest.mark.gpu]
needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not shipped")

PLUGIN = os.path.join(os.path.dirname(G.lib_path()), "libgrokj2k_plugin.so")


def _plugin():
    L = C.CDLL(PLUGIN)
    L.grk_amd_plugin_tile_create.restype = C.c_void_p
    L.grk_amd_plugin_tile_create.argtypes = [C.c_void_p, C.POINTER(G.TileParams), C.c_void_p, C.c_int]
    L.grk_amd_plugin_tile_destroy.argtypes = [C.c_void_p]
    return L


@needs_ref
@pytest.mark.parametrize("Cn,H,W,prec,numres,gen", [(1, 512, 512, 8, 4, "g2"), (3, 256, 320, 8, 6, "g2"),
                                                    (3, 512, 512, 8, 6, "g0"), (1, 128, 128, 12, 5, "g2")])
def test_compress_with_plugin_tile_equals_cpu(Cn, H, W, prec, numres, gen):
    px = getattr(synth, gen)(Cn, H, W, prec)
    cpu, _ = R.encode(px, prec, numres=numres, mode=1)
    p = G.TileParams.make(W, H, Cn, prec, numres - 1)
    L = _plugin()
    tile = L.grk_amd_plugin_tile_create(U.ctx()._h, C.byref(p), px.ctypes.data, 0)
    assert tile
    try:
        # rateControlAlgorithm = 1: defect D2 of the reference's plugin path
        via_plugin, _ = R.encode(px, prec, numres=numres, mode=1, rate_algo=1, plugin_tile=tile)
    finally:
        L.grk_amd_plugin_tile_destroy(tile)
    assert via_plugin == cpu


@needs_ref
def test_file_protocol_through_grok_loader(tmp_path):
    """grk_initialize(plugin dir) -> grk_plugin_init -> grk_plugin_compress(params{infile}, cb):
    Grok dlsym()s plugin_encode in our .so, we read the PNM, encode on the GPU and call back."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    for Cn, prec in ((1, 8), (3, 8), (1, 12)):
        px = synth.g2(Cn, 192, 256, prec)
        path = str(tmp_path / ("in_%d_%d.%s" % (Cn, prec, "pgm" if Cn == 1 else "ppm")))
        R.write_pnm(path, px, prec)
        got = R.plugin_compress_file(px, prec, path, numres=5)
        assert not isinstance(got, int), "plugin refused: %s" % got
        cpu, _ = R.encode(px, prec, numres=5, mode=1)
        assert got == cpu
    # a request outside the hot path is declined (non-zero) so that the host falls back to its CPU path
    assert isinstance(R.plugin_compress_file(synth.g2(1, 64, 64, 8), 8, "/nonexistent.pgm", numres=3), int)


@pytest.mark.parametrize("Cn,H,W,prec,L,irrev", [(3, 192, 256, 8, 4, False), (1, 130, 97, 8, 3, False), (3, 128, 160, 12, 3, True),
                                                  (3, 256, 256, 16, 5, True)])
def test_block_distortion_equals_the_oracles(Cn, H, W, prec, L, irrev):
    """The rate-control hook: grk_amd_block_distortion of an encode == the oracle's restatement of T1::getwmsedec's
    weights times the energy of the quantised magnitudes of the oracle's own coefficients -- the same double, bit for bit (the energy
    is an integer sum; the weights are the reference's tables)."""
    import oracle as O
    import chain
    px = synth.g2(Cn, H, W, prec)
    p = G.TileParams.make(W, H, Cn, prec, L, irreversible=irrev)
    c = U.ctx()
    table, coded = c.encode_host(p, px)
    got = c.block_distortion(len(table))
    blocks, _ = G.tile_layout(p)
    planes = [px[k].astype(np.int32) - (1 << (prec - 1)) for k in range(Cn)]
    mct = Cn >= 3
    if irrev:
        if mct:
            planes[:3] = [v.view(np.float32) for v in O.ict_fwd(*planes[:3])]
        else:
            planes = [v.astype(np.float32) for v in planes]
        mall = [O.dwt97_fwd(v, L) for v in planes]
    else:
        if mct:
            planes[:3] = O.rct_fwd(*planes[:3])
        mall = [O.dwt53_fwd(v, L) for v in planes]
    OL = O.lib()
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        sub = np.ascontiguousarray(mall[b.comp][b.py:b.py + bh, b.px:b.px + bw])
        if irrev:
            sm = np.zeros((bh, bw), np.uint32)
            OL.orc_ht_signmag_irrev(sub.ctypes.data, bw, bw, bh, b.kmax, C.c_float(np.float32(1.0) / np.float32(b.stepsize)), sm.ctypes.data)
        else:
            sm = O.signmag(sub, b.kmax)
        want = O.ht_block_distortion(sm, b.kmax, b.band, L - b.res, not irrev, mct, b.comp, b.stepsize)
        assert got[i] == want, (i, b.comp, b.res, b.band, got[i], want)
    assert got.max() > 0


@needs_ref
def test_layered_job_through_grok_loader(tmp_path, monkeypatch):
    """grk_compress -r 20,10,1 on an HTJ2K job through the plugin: the tile tree carries a distortion decrease per block
    (grk_amd_plugin_tile_fill_distortion), the HOST's rate control makes three quality layers of the blocks' single passes
    (TileProcessor::pcrd_bisect_feasible over the plugin's rates and slopes) and writes the file.  What can be said about it without
    a reference to compare with -- the reference's own CPU path leaves an HT block's distortion unset (T1HT.cpp:102-127), so its
    layers are arbitrary -- : all layers together decode to the source exactly; the first layer alone holds the blocks with the
    steepest slopes, i.e. it decodes to a far better picture than the CPU path's first layer, within the size the ratio allows."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    Cn, H, W, prec = 3, 384, 512, 8
    px = synth.g2(Cn, H, W, prec)
    path = str(tmp_path / "in.ppm")
    R.write_pnm(path, px, prec)
    monkeypatch.setenv("REF_LAYERS", "20,10,1")
    got = R.plugin_compress_file(px, prec, path, numres=6)
    assert not isinstance(got, int), "plugin refused: %s" % got
    cpu, _ = R.encode(px, prec, numres=6, mode=1, rate_algo=1)
    import j2kparse as J
    assert J.parse_cod_layers(got) == 3 and J.parse_cod_layers(cpu) == 3
    assert np.array_equal(R.decode(got, Cn, H, W), px.astype(np.int32))            # every layer: lossless
    monkeypatch.setenv("REF_MAX_LAYERS", "1")
    first, first_cpu = R.decode(got, Cn, H, W), R.decode(cpu, Cn, H, W)
    monkeypatch.delenv("REF_MAX_LAYERS")
    p_gpu, p_cpu = synth.psnr_db(first, px, prec), synth.psnr_db(first_cpu, px, prec)
    assert p_gpu > p_cpu + 3.0 and p_gpu > 20.0, (p_gpu, p_cpu)


@needs_ref
@pytest.mark.parametrize("sub,off", [((2, 2), (0, 0)), ((2, 1), (0, 0)), ((1, 2), (4, 6)), ((3, 2), (6, 4))])
def test_subsampled_components_through_grok_loader(tmp_path, monkeypatch, sub, off):
    """grk_compress -s dx,dy (every component of a PNM sub-sampled alike on the reference grid): the plugin codes the w x h
    components as it does any tile -- their origin is ceil(offset / d) -- and the host writes SIZ with XRsiz / YRsiz; the file ==
    the pure-CPU encode of the same job."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    monkeypatch.setenv("REF_SUBSAMPLING", "%d,%d" % sub)
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    for Cn, H, W, prec in ((3, 191, 255, 8), (1, 80, 300, 12)):
        px = synth.g2(Cn, H, W, prec)
        path = str(tmp_path / ("in_%d.%s" % (Cn, "pgm" if Cn == 1 else "ppm")))
        R.write_pnm(path, px, prec)
        TW, TH = off[0] + (W - 1) * sub[0] + 1, off[1] + (H - 1) * sub[1] + 1           # one tile: the whole reference grid
        got = R.plugin_compress_file(px, prec, path, numres=5, TW=TW, TH=TH)
        assert not isinstance(got, int), "plugin refused: %s" % got
        cpu, _ = R.encode(px, prec, TW=TW, TH=TH, numres=5, mode=1)
        assert got == cpu
        assert np.array_equal(R.decode(cpu, Cn, H, W), px.astype(np.int32))


@needs_ref
@pytest.mark.parametrize("Cn,H,W,prec,numres,ht,sty,irrev", [(3, 192, 256, 8, 5, 1, 0, 0), (1, 128, 128, 8, 4, 1, 0, 0), (3, 100, 77, 12, 3, 1, 0, 0),
                                                              (3, 128, 192, 8, 4, 0, 0, 0), (1, 96, 160, 10, 3, 0, 0x02 | 0x08 | 0x20, 0),
                                                              (3, 128, 192, 12, 5, 0, 0, 1), (3, 96, 160, 8, 4, 0, 0x20, 1)])
def test_decode_protocol_through_grok_loader(Cn, H, W, prec, numres, ht, sty, irrev):
    """grk_initialize(plugin dir) -> grk_plugin_init -> grk_plugin_decompress(params, cb): Grok dlsym()s
    plugin_decompress in our .so; the host parses the header and runs Tier-2 INTO OUR TILE TREE
    (decompress_synch_plugin_with_host), skips its own T1 / inverse DWT / inverse MCT, and gets the pixels of the
    GPU decode back in its grk_image.  HT and classic (Part-1, also with single-segment code-block styles) streams
    written by grk_compress, lossless and (classic blocks: BASELINE configs[4]'s shape) ICT + 9/7 with the band step
    sizes taken from the tree; result == grk_decompress on the CPU (== the source when lossless)."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    px = synth.g2(Cn, H, W, prec)
    cs, _ = R.encode(px, prec, numres=numres, mode=1, ht=ht, cblksty=sty, irrev=irrev)
    got, stages = R.plugin_decompress(cs, Cn, H, W)
    assert not isinstance(got, int), "plugin refused: %s (stages %s)" % (got, stages)
    assert stages == [1, 1, 1, 1]
    assert np.array_equal(got, R.decode(cs, Cn, H, W))
    if not irrev:
        assert np.array_equal(got, px.astype(np.int32))


@needs_ref
@pytest.mark.parametrize("off", [(1, 1), (33, 95)])
def test_protocols_with_an_image_off_the_origin(tmp_path, monkeypatch, off):
    """grk_compress -d x0,y0 (image_offset_x0 / y0 in the parameters, grk_image x0 / y0 in the host's callback): the tile
    is the image area wherever it lies -- odd-start transforms, partial first code-blocks (the decline
    is gone).  The file == the pure-CPU encode; and the decode protocol returns grk_decompress's pixels for such a stream."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    for Cn, H, W, prec in ((3, 191, 255, 8), (1, 77, 300, 12)):
        px = synth.g2(Cn, H, W, prec)
        path = str(tmp_path / ("in_%d_%d.%s" % (Cn, prec, "pgm" if Cn == 1 else "ppm")))
        R.write_pnm(path, px, prec)
        got = R.plugin_compress_file(px, prec, path, numres=6, TW=W + off[0], TH=H + off[1])
        assert not isinstance(got, int), "plugin refused: %s" % got
        cpu, _ = R.encode(px, prec, TW=W + off[0], TH=H + off[1], numres=6, mode=1)
        assert got == cpu
        back, stages = R.plugin_decompress(cpu, Cn, H, W)
        assert not isinstance(back, int), "plugin refused: %s (stages %s)" % (back, stages)
        assert np.array_equal(back, px.astype(np.int32))


@needs_ref
@pytest.mark.parametrize("case", [
    # (C, H, W, prec, TW, TH, offset, env)
    (3, 200, 300, 8, 128, 128, (0, 0), {}),
    (1, 250, 190, 12, 100, 64, (0, 0), {"REF_WRITE_TLM": "1", "REF_WRITE_PLT": "1"}),     # (no empty bands: D15)
    (3, 150, 220, 8, 96, 80, (7, 5), {"REF_PROG_ORDER": "2", "REF_CSTY": "6"}),
    (3, 130, 130, 8, 64, 64, (0, 0), {"REF_PRECINCTS": "64,64,32,32", "REF_PROG_ORDER": "4"}),
])
def test_image_of_several_tiles_through_the_plugin(tmp_path, monkeypatch, case):
    """The plugin protocol cannot carry an image of several tiles (D3: one grk_plugin_tile per image), so the plugin writes
    the whole .j2k itself -- every tile through the hot path, the codestream through its own writer -- and answers 0
    without the host's callback; the file equals the one grk_compress writes on its CPU path, tile grid, image offset,
    TLM / PLT, SOP / EPH, progression order and precincts included.  A .jp2 target stays with the host (non-zero)."""
    Cn, H, W, prec, TW, TH, off, env = case
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    monkeypatch.setenv("REF_IMG_X0", str(off[0]))
    monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    px = synth.g2(Cn, H, W, prec)
    path = str(tmp_path / ("t.%s" % ("pgm" if Cn == 1 else "ppm")))
    R.write_pnm(path, px, prec)
    got = R.plugin_compress_file(px, prec, path, numres=4, TW=TW, TH=TH)
    assert not isinstance(got, int), "plugin refused: %s" % got
    cpu, _ = R.encode(px, prec, TW=TW, TH=TH, numres=4, mode=1)
    assert got == cpu
    assert np.array_equal(R.decode(got, Cn, H, W), px.astype(np.int32))


@needs_ref
def test_file_protocol_with_mct_not_set_on_the_command_line(tmp_path, monkeypatch):
    """grk_compress hands the plugin tcp_mct = 255 ("not set") unless -Y was given (grk_compress.cpp:1836; resolved
    only inside its callback, :1817-1820): the plugin resolves it the same way -- grayscale is coded without MCT, RGB
    with -- instead of declining every grayscale image.  tcp_mct = 2 (custom array MCT) is declined."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    monkeypatch.setenv("REF_TCP_MCT", "255")
    for Cn in (1, 3):
        px = synth.g2(Cn, 128, 192, 8)
        path = str(tmp_path / ("m%d.%s" % (Cn, "pgm" if Cn == 1 else "ppm")))
        R.write_pnm(path, px, 8)
        got = R.plugin_compress_file(px, 8, path, numres=4)
        assert not isinstance(got, int), "plugin refused: %s" % got
        cpu, _ = R.encode(px, 8, numres=4, mode=1)
        assert got == cpu
    monkeypatch.setenv("REF_TCP_MCT", "2")
    assert isinstance(R.plugin_compress_file(px, 8, path, numres=4), int)


@needs_ref
def test_plugin_self_check_mode_the_reference_compares_every_block(tmp_path, monkeypatch):
    """GRK_PLUGIN_STATE_DEBUG (grok.h:1719-1739; ours under GRK_AMD_PLUGIN_DEBUG=1): the host skips its own DC shift / MCT / DWT,
    runs its own Tier-1 over the sub-band coefficients the plugin hands it, and compares every code-block -- bytes, rates,
    passes, bounding boxes, step sizes -- with the plugin's (plugin_bridge.cpp:138-252): not one warning; with a single
    coefficient changed behind the plugin's back (test hook of the harness) it reports the block."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    cases = ((3, 8, 5, (0, 0)), (1, 8, 4, (0, 0)), (1, 12, 3, (0, 0)), (3, 8, 5, (1, 1)), (1, 8, 4, (33, 95)))   # (last two: image off the origin)
    H, W = 192, 256

    def offsets(off):
        monkeypatch.setenv("REF_IMG_X0", str(off[0]))
        monkeypatch.setenv("REF_IMG_Y0", str(off[1]))
    # the pure-CPU files first: the debug state is a property of the loaded plugin, and the host's CPU path obeys it too
    # (TileProcessor.cpp:676-690 skips DC shift / MCT / DWT whenever it is set)
    cpu = {}
    for k in cases:
        offsets(k[3])
        cpu[k] = R.encode(synth.g2(k[0], H, W, k[1]), k[1], TW=W + k[3][0], TH=H + k[3][1], numres=k[2], mode=1)[0]
    monkeypatch.setenv("GRK_AMD_PLUGIN_DEBUG", "1")
    assert R.plugin_init(0) == 1
    assert R.plugin_debug_state() == 1
    try:
        for Cn, prec, numres, off in cases:
            offsets(off)
            px = synth.g2(Cn, H, W, prec)
            path = str(tmp_path / ("dbg_%d_%d.%s" % (Cn, prec, "pgm" if Cn == 1 else "ppm")))
            R.write_pnm(path, px, prec)
            R.warning_count()
            got = R.plugin_compress_file(px, prec, path, numres=numres, TW=W + off[0], TH=H + off[1])
            n, last = R.warning_count()
            assert not isinstance(got, int), "plugin refused: %s" % got
            assert n == 0, "the reference disagrees with the plugin: %d warnings, last: %s" % (n, last)
            assert got == cpu[(Cn, prec, numres, off)]
        offsets((0, 0))
        monkeypatch.setenv("REF_DEBUG_PERTURB", "1")
        R.warning_count()
        px = synth.g2(1, H, W, 12)
        path = str(tmp_path / "dbg_1_12.pgm")
        got = R.plugin_compress_file(px, 12, path, numres=3)
        n, last = R.warning_count()
        assert n >= 1 and "differ" in last, (n, last)
    finally:
        monkeypatch.setenv("GRK_AMD_PLUGIN_DEBUG", "0")
        R.plugin_init(0)                       # back to production state for the tests that follow
    assert R.plugin_debug_state() == 0


def _patch_guard_bits(cs, guard):
    """The same codestream with another number of guard bits in QCD: band numbps = expn + G - 1 moves, the blocks'
    zero-bit-plane counts in the packet headers stay, so every block's numbps moves with it and missing_msbs (their
    difference, all the HT decoder uses) does not -- a legal stream with the same pixels, but not Grok's QCD."""
    b = bytearray(cs)
    i = b.index(b"\xff\x5c")
    assert b[i + 4] >> 5 == 1
    b[i + 4] = (b[i + 4] & 0x1F) | (guard << 5)
    return bytes(b)


@needs_ref
@pytest.mark.parametrize("Cn,H,W,prec,numres", [(3, 192, 256, 8, 5), (1, 128, 128, 12, 4)])
def test_decode_protocol_takes_band_numbps_from_the_stream(Cn, H, W, prec, numres):
    """an HT stream whose QCD is not the one this library's encoder models (here: 2 and 3 guard bits) must
    decode through the plugin to what grk_decompress makes of it -- the host hands over block numbps only, the band's
    comes from the stream's own QCD marker, read from the file the host was pointed at."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    px = synth.g2(Cn, H, W, prec)
    cs, _ = R.encode(px, prec, numres=numres, mode=1, ht=1)
    for guard in (2, 3):
        alt = _patch_guard_bits(cs, guard)
        ref = R.decode(alt, Cn, H, W)
        assert np.array_equal(ref, px.astype(np.int32))
        got, stages = R.plugin_decompress(alt, Cn, H, W)
        assert not isinstance(got, int), "plugin refused: %s (stages %s)" % (got, stages)
        assert np.array_equal(got, ref)


@needs_ref
def test_decode_protocol_declines_without_a_file_to_read_the_qcd_from():
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    px = synth.g2(1, 128, 128, 8)
    cs, _ = R.encode(px, 8, numres=3, mode=1, ht=1)
    got, stages = R.plugin_decompress(cs, 1, 128, 128, as_file=False)
    assert isinstance(got, int) and got != 0 and stages[3] == 1
    got, _ = R.plugin_decompress(cs, 1, 128, 128)
    assert np.array_equal(got, px.astype(np.int32))


@needs_ref
def test_batch_compress_through_grok_loader(tmp_path):
    """grk_plugin_batch_compress (grok.cpp:683-707; `grk_compress -y dir -a dir`): our worker walks the directory with three
    overlapped stages -- read + de-interleave the next file, GPU encode of the current one, the host's callback (its Tier-2 +
    file write) for the one before -- and every output file == the pure-CPU encode of its image.  Images of different sizes,
    component counts and depths, one file the hot path declines (it is skipped, the batch goes on)."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); outd.mkdir()
    imgs = {}
    for i, (Cn, H, W, prec) in enumerate([(3, 192, 256, 8), (1, 300, 200, 8), (3, 257, 129, 8), (1, 128, 128, 12), (3, 512, 384, 8),
                                          (3, 64, 64, 8), (1, 1000, 40, 8)]):
        px = synth.g2(Cn, H, W, prec, seed=100 + i)
        name = "img%02d" % i
        R.write_pnm(str(ind / (name + (".pgm" if Cn == 1 else ".ppm"))), px, prec)
        imgs[name] = (px, prec)
    (ind / "broken.pgm").write_bytes(b"P5\n10 10\n255\nshort")
    (ind / "notes.txt").write_text("not an image")
    n = R.plugin_batch_compress(str(ind), str(outd), numres=6)
    assert n == len(imgs), n
    for name, (px, prec) in imgs.items():
        got = (outd / (name + ".j2k")).read_bytes()
        cpu, _ = R.encode(px, prec, numres=6, mode=1)
        assert got == cpu, name
    assert not (outd / "broken.j2k").exists()


@needs_ref
def test_batch_decompress_through_grok_loader(tmp_path):
    """grk_plugin_init_batch_decompress + grk_plugin_batch_decompress (plugin/plugin_interface.h:131-143; grk_decompress
    -y <dir> -a <dir>, grk_decompress.cpp:874-900): the plugin's worker walks the directory, decodes what is inside the hot
    path on the GPU and hands the rest (here an HT + 9/7 stream and a multi-tile one) back to the host's own decoder in the
    same callback protocol -- every file of the directory comes out, pixel-identical to grk_decompress."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); outd.mkdir()
    want = {}
    jobs = [("a_rgb", dict(ht=1), (3, 192, 256, 8)), ("b_mono12", dict(ht=1), (1, 128, 128, 12)), ("c_classic", dict(ht=0), (3, 96, 160, 8)),
            ("d_classic97", dict(ht=0, irrev=1), (3, 128, 192, 10)), ("e_ht97", dict(ht=1, irrev=1), (3, 128, 128, 8)),
            ("f_tiles", dict(ht=1, TW=64, TH=64), (3, 128, 128, 8))]
    for name, kw, (Cn, H, W, prec) in jobs:
        cs, _ = R.encode(synth.g2(Cn, H, W, prec), prec, numres=4, mode=1, **kw)
        (ind / (name + ".j2k")).write_bytes(cs)
        want[name] = R.decode(cs, Cn, H, W)
    assert R.plugin_batch_decompress(str(ind), str(outd)) == 0
    L = C.CDLL(PLUGIN)
    g, c, f = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    L.grk_amd_plugin_batch_decode_counts(C.byref(g), C.byref(c), C.byref(f))
    assert (g.value, c.value, f.value) == (4, 2, 0)
    for name in want:
        got = R.read_batch_output(str(outd / (name + ".raw")))
        assert np.array_equal(got, want[name]), name


@needs_ref
def test_decode_protocol_declines_outside_the_hot_path():
    """HT + 9/7 streams (D1), multi-segment code-block styles and multi-tile images are answered with non-zero:
    the host keeps its CPU decoder (grk_decompress.cpp:953-955)."""
    assert R.plugin_load() == 1
    assert R.plugin_init(0) == 1
    px = synth.g2(3, 128, 128, 8)
    for kw in (dict(irrev=1, ht=1), dict(ht=0, cblksty=0x04), dict(TW=64, TH=64)):
        cs, _ = R.encode(px, 8, numres=3, mode=1, **kw)
        got, stages = R.plugin_decompress(cs, 3, 128, 128)
        assert isinstance(got, int) and got != 0
        assert stages[3] == 1                       # the host was told to clean up


@pytest.mark.parametrize("Cn,H,W,prec,L", [(1, 256, 256, 8, 3), (3, 128, 192, 8, 4), (3, 64, 96, 12, 2)])
def test_plugin_tile_decode_round_trip(Cn, H, W, prec, L):
    """The decode counterpart at the plugin-tile level: a grk_plugin_tile tree carrying what the host's Tier-2
    puts there (compressedData, compressedDataLength, numBitPlanes, numPasses; plugin_bridge.cpp:63-76) ->
    grk_amd_plugin_tile_decode -> the source pixels.  (No oracle/_ref needed: the tree comes from our encoder.)"""
    px = synth.g2(Cn, H, W, prec)
    p = G.TileParams.make(W, H, Cn, prec, L)
    Lp = _plugin()
    Lp.grk_amd_plugin_tile_decode.restype = C.c_int
    Lp.grk_amd_plugin_tile_decode.argtypes = [C.c_void_p, C.POINTER(G.TileParams), C.c_void_p, C.c_void_p, C.c_int]
    tile = Lp.grk_amd_plugin_tile_create(U.ctx()._h, C.byref(p), px.ctypes.data, 0)
    assert tile
    try:
        out = np.zeros_like(px)
        rc = Lp.grk_amd_plugin_tile_decode(U.ctx()._h, C.byref(p), tile, out.ctypes.data, 0)
        assert rc == 0
        assert np.array_equal(out, px)
    finally:
        Lp.grk_amd_plugin_tile_destroy(tile)


@needs_ref
def test_batch_compress_spreads_files_over_device_contexts(tmp_path):
    """The batch mode runs one GPU stage per device context (plugin.cpp: the device Grok named + the node's other GPUs; here
    GRK_AMD_PLUGIN_DEVICES=0,0,0 -- three contexts on the one GPU of the box -- so that the multi-device code path runs): every
    output file == the pure-CPU encode of its image.  In a process of its own: the plugin's device list is fixed at plugin_init."""
    import subprocess
    import sys
    script = r'''
import sys, os, ctypes as C
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import numpy as np, refharness as R, synth, grok_amd as G
assert R.plugin_load() == 1 and R.plugin_init(0) == 1
P = C.CDLL(os.path.join(os.path.dirname(G.lib_path()), "libgrokj2k_plugin.so"))
P.grk_amd_plugin_num_devices.restype = C.c_uint32
assert P.grk_amd_plugin_num_devices() == 3, P.grk_amd_plugin_num_devices()
ind, outd = sys.argv[1], sys.argv[2]
imgs = {}
for i in range(9):
    Cn, H, W = (3, 256 + 64 * (i %% 3), 320) if i %% 2 == 0 else (1, 200, 500 + 10 * i)
    px = synth.g2(Cn, H, W, 8, seed=300 + i)
    name = "f%%02d" %% i
    R.write_pnm(os.path.join(ind, name + (".pgm" if Cn == 1 else ".ppm")), px, 8)
    imgs[name] = px
assert R.plugin_batch_compress(ind, outd, numres=6) == len(imgs)
for name, px in imgs.items():
    cpu, _ = R.encode(px, 8, numres=6, mode=1)
    assert open(os.path.join(outd, name + ".j2k"), "rb").read() == cpu, name
print("ok")
''' % {"tests": os.path.dirname(os.path.abspath(__file__)), "root": os.path.dirname(os.path.dirname(os.path.abspath(__file__)))}
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir(); outd.mkdir()
    env = dict(os.environ, GRK_AMD_PLUGIN_DEVICES="0,0,0")
    r = subprocess.run([sys.executable, "-c", script, str(ind), str(outd)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
