"""-m gpu: every HIP kernel against the CPU oracle, through the C-ABI, bit-exact."""
import numpy as np
import pytest
import torch

import grok_amd as G
import oracle as O
import synth
import gpuutil as U

pytestmark = pytest.mark.gpu


def oracle_ingest(px, prec, mct, irrev):
    C, H, W = px.shape
    planes = [px[c].astype(np.int32) - (1 << (prec - 1)) for c in range(C)]
    if mct:
        if irrev:
            f = O.ict_fwd(*planes[:3])
            planes[:3] = [v.view(np.int32) for v in f]
        else:
            planes[:3] = O.rct_fwd(*planes[:3])
    elif irrev:
        planes = [p.astype(np.float32).view(np.int32) for p in planes]
    if mct and irrev and C > 3:
        planes[3] = planes[3].astype(np.float32).view(np.int32)
    return np.stack(planes)


@pytest.mark.parametrize("C,H,W,prec,irrev", [
    (3, 64, 64, 8, 0), (3, 33, 70, 8, 0), (1, 17, 5, 8, 0), (3, 128, 256, 16, 0),
    (3, 64, 64, 8, 1), (3, 50, 101, 12, 1), (1, 64, 64, 8, 1), (4, 32, 36, 8, 0), (4, 32, 36, 10, 1)])
def test_ingest_mct(C, H, W, prec, irrev):
    rng = np.random.default_rng(C * 1000 + W)
    px = rng.integers(0, 1 << prec, size=(2, C, H, W)).astype(np.uint8 if prec <= 8 else np.uint16)
    p = G.TileParams.make(W, H, C, prec, 0, irreversible=bool(irrev))
    d_px = U.to_dev(px.reshape(-1).view(np.uint8))
    d_pl = U.dev_planes(p, 2 * C)
    U.ctx().stage_ingest_mct(p, 2, d_px.data_ptr(), d_pl.data_ptr())
    U.ctx().synchronize()
    got = U.planes_to_numpy(d_pl, p, 2 * C).reshape(2, C, H, W)
    for t in range(2):
        want = oracle_ingest(px[t], prec, C >= 3, bool(irrev))
        assert np.array_equal(got[t], want)


DWT_CASES = [(8, 8, 1), (64, 64, 3), (65, 33, 3), (100, 77, 5), (17, 1, 2), (1, 9, 2), (3, 3, 1),
             (2, 2, 1), (255, 257, 5), (512, 512, 5), (1024, 1024, 5), (1500, 700, 4), (4, 600, 3)]


@pytest.mark.parametrize("W,H,L", DWT_CASES)
def test_dwt53(W, H, L):
    rng = np.random.default_rng(W * 7 + H)
    a = rng.integers(-300, 300, size=(2, H, W)).astype(np.int32)
    p = G.TileParams.make(W, H, 1, 8, L, mct=False)
    d_in = U.upload_planes(a, p)
    d_out = U.dev_planes(p, 2)
    U.ctx().stage_dwt_fwd(p, 2, d_in.data_ptr(), d_out.data_ptr())
    U.ctx().synchronize()
    got = U.planes_to_numpy(d_out, p, 2)
    for k in range(2):
        assert np.array_equal(got[k], O.dwt53_fwd(a[k], L)), "plane %d" % k


@pytest.mark.parametrize("W,H,L", DWT_CASES)
def test_dwt97_bit_exact(W, H, L):
    """north_star asks for <= 1 ULP per sub-band coefficient; we hold 0 ULP (same op order, no FMA)."""
    rng = np.random.default_rng(W * 11 + H)
    f = (rng.standard_normal((2, H, W)) * 200).astype(np.float32)
    p = G.TileParams.make(W, H, 1, 8, L, irreversible=True, mct=False)
    d_in = U.upload_planes(f.view(np.int32), p)
    d_out = U.dev_planes(p, 2)
    U.ctx().stage_dwt_fwd(p, 2, d_in.data_ptr(), d_out.data_ptr())
    U.ctx().synchronize()
    got = U.planes_to_numpy(d_out, p, 2)
    for k in range(2):
        want = O.dwt97_fwd(f[k], L).view(np.int32)
        assert np.array_equal(got[k], want), "plane %d: max ulp %d" % (k, np.abs(got[k].astype(np.int64) - want).max())


def _ht_case(W, H, L, C, prec, mode, seed):
    """Random Mallat planes -> HIP block bytes vs oracle block bytes."""
    rng = np.random.default_rng(seed)
    p = G.TileParams.make(W, H, C, prec, L)
    blocks, _ = G.tile_layout(p)
    planes = np.zeros((C, H, W), np.int32)
    for b in blocks:
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        mag = rng.integers(0, 1 << b.kmax, size=(bh, bw))
        if mode == 1:
            mag = mag >> rng.integers(0, b.kmax + 1, size=(bh, bw))
        elif mode == 2:
            mag = np.where(rng.random((bh, bw)) < 0.93, 0, mag & 7)
        elif mode == 3:
            mag = np.zeros((bh, bw), np.int64)
        elif mode == 4:
            mag = np.full((bh, bw), (1 << b.kmax) - 1)      # 0xFF-heavy streams: stuffing paths
        sign = np.where(rng.random((bh, bw)) < 0.5, -1, 1)
        planes[b.comp, b.py:b.py + bh, b.px:b.px + bw] = (mag * sign).astype(np.int32)
    d_m = U.upload_planes(planes, p)
    c = U.ctx()
    c.stage_ht_encode(p, 1, d_m.data_ptr())
    table, tot = c.fetch_table(len(blocks))
    coded = c.fetch_coded(tot)
    got = U.split_blocks(table, coded)
    bad = []
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        sm = O.signmag(planes[b.comp, b.py:b.py + bh, b.px:b.px + bw], b.kmax)
        want = O.ht_encode_sm(sm, b.kmax)
        if got[i] != want:
            bad.append((i, b.res, b.band, bw, bh, len(got[i]), len(want)))
    assert not bad, "mismatching blocks (idx,res,band,w,h,len_gpu,len_oracle): %s" % bad[:8]


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_ht_blocks_512(mode):
    _ht_case(512, 512, 3, 1, 8, mode, 100 + mode)


@pytest.mark.parametrize("W,H,L,C,prec", [(200, 120, 2, 3, 8), (130, 67, 5, 1, 12), (1024, 1024, 5, 3, 8),
                                           (64, 64, 0, 1, 16), (37, 3, 1, 1, 8), (1, 1, 0, 1, 8)])
def test_ht_blocks_ragged(W, H, L, C, prec):
    _ht_case(W, H, L, C, prec, 1, W + H)


@pytest.mark.parametrize("C,H,W,prec,L,gen", [(1, 512, 512, 8, 3, "g2"), (1, 512, 512, 8, 3, "g0"),
                                               (3, 256, 384, 8, 5, "g2"), (3, 128, 128, 16, 4, "g2")])
def test_encode_tile_blocks_vs_oracle(C, H, W, prec, L, gen):
    px = getattr(synth, gen)(C, H, W, prec)
    p = G.TileParams.make(W, H, C, prec, L)
    table, coded = U.ctx().encode_host(p, px)
    got = U.split_blocks(table, coded)
    blocks, lens, ocoded = O.encode_tile_rev(px, prec, L)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    assert len(got) == len(blocks)
    bad = [i for i in range(len(blocks)) if got[i] != bytes(ocoded[off[i]:off[i + 1]])]
    assert not bad, "blocks differing from oracle: %s" % bad[:10]


@pytest.mark.parametrize("value", [0, 1, 16, 127, 128, 129, 255])
@pytest.mark.parametrize("C,H,W,L", [(3, 256, 384, 2), (1, 512, 512, 3)])
def test_flat_frames_blocks_vs_oracle(value, C, H, W, L):
    """A frame of ONE value: every band but LL is empty (quads without bits: the lanes that stay out of the LDS atomics), and LL is
    a constant -- for value 0 that is -128, MagSgn bytes 0xFF throughout: the run form of the stuffing step, block after block."""
    px = np.full((C, H, W), value, np.uint8)
    p = G.TileParams.make(W, H, C, 8, L)
    table, coded = U.ctx().encode_host(p, px)
    got = U.split_blocks(table, coded)
    blocks, lens, ocoded = O.encode_tile_rev(px, 8, L)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bad = [i for i in range(len(blocks)) if got[i] != bytes(ocoded[off[i]:off[i + 1]])]
    assert not bad, "blocks differing from oracle: %s" % bad[:10]
    back = U.ctx().decode_host(p, table, coded)
    assert np.array_equal(np.asarray(back).reshape(px.shape), px)


# ---- whole files: HIP hot path + product Tier-2 == Grok's CPU encoder output (golden md5 / fixtures)
import hashlib
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpu_codestream(px, prec, L, TW=None, TH=None, cblk=(6, 6)):
    C, H, W = px.shape
    TW, TH = TW or W, TH or H
    p = G.TileParams.make(TW, TH, C, prec, L, cblk=cblk)
    tiles = [px[:, ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW] for ty in range(H // TH) for tx in range(W // TW)]
    batch = np.ascontiguousarray(np.stack(tiles))
    table, coded = U.ctx().encode_host(p, batch, ntiles=len(tiles))
    return G.write_codestream(p, W, H, table, coded)


@pytest.mark.parametrize("gen,C,W,H,L,md5", [
    ("g0", 1, 512, 512, 3, "8f2ec0f22e10fbeb97c3bf515d7ad976"),
    ("g2", 1, 512, 512, 3, "0ea91840e2b0d964ce8e570148a07642"),        # BASELINE configs[0]
    ("g0", 3, 512, 512, 5, "8d81ed0576b981cba0c0111e5f9724a5"),
    ("g0", 1, 64, 64, 0, "4afbe3defe07e7ca0e8d4c1d5433b84d"),
    ("g0", 1, 128, 128, 1, "8ae843154d7f6f1283e796a9bd029758"),
    ("g2", 3, 1024, 1024, 5, "2ec6724e8acd2796841140b37cf5ca72"),
    ("g2", 3, 4096, 4096, 5, "9acbfe328b6611cfec34020db95f3c5b"),      # BASELINE configs[1]
    ("g2", 3, 8192, 8192, 5, "7e5275ef3d61edd7b95332bef74a986c")])     # BASELINE metric shape (md5 of oracle/_ref run)
def test_gpu_codestream_md5_vs_grok(gen, C, W, H, L, md5):
    cs = gpu_codestream(getattr(synth, gen)(C, H, W), 8, L)
    assert hashlib.md5(cs).hexdigest() == md5


def test_gpu_codestream_md5_multitile():
    full = np.tile(synth.g2(3, 1024, 1024), (1, 2, 2))
    cs = gpu_codestream(full, 8, 5, 1024, 1024)
    assert len(cs) == 6145625 and hashlib.md5(cs).hexdigest() == "6677d0490f0ae2b26da59a69925f0e3b"


@pytest.mark.parametrize("name,gen,shape,prec,L,tile", [
    ("g2_1x256x256_r4", "g2", (1, 256, 256), 8, 3, None), ("g2_3x192x160_r4", "g2", (3, 160, 192), 8, 3, None),
    ("g2_3x256x256_t128_r4", "g2", (3, 256, 256), 8, 3, 128), ("g2u16_1x128x128_r5", "g2", (1, 128, 128), 12, 4, None)])
def test_gpu_codestream_fixture(name, gen, shape, prec, L, tile):
    cs = gpu_codestream(getattr(synth, gen)(*shape, prec), prec, L, tile, tile)
    assert cs == open(os.path.join(GOLD, name + ".j2k"), "rb").read()


def test_gpu_8k_properties():
    """BASELINE full size (8192x8192x3): size-independent checks -- every block's Scup is sane, the
    tile-replication invariant holds (a 2x2 replicated 4K image codes each quadrant's interior
    blocks identically is NOT true for DWT, so instead:) encoding is deterministic and the whole
    file matches an independent second encode through a fresh context."""
    px = synth.g2(3, 8192, 8192, 8)
    p = G.TileParams.make(8192, 8192, 3, 8, 5)
    t1, c1 = U.ctx().encode_host(p, px)
    ctx2 = G.Context(0)
    t2, c2 = ctx2.encode_host(p, px)
    ctx2.close()
    assert np.array_equal(t1["length"], t2["length"])
    b1, b2 = U.split_blocks(t1, c1), U.split_blocks(t2, c2)
    assert b1 == b2
    assert len(b1) == 49152
    for b in b1[::97]:
        scup = (b[-1] << 4) | (b[-2] & 0xF)
        assert 2 <= scup <= len(b) and scup <= 4079
    # spot-check 64 blocks against the oracle (reversible path is integer-exact end to end)
    import ctypes as C
    planes = [px[c].astype(np.int32) - 128 for c in range(3)]
    planes = O.rct_fwd(*planes)
    blocks, _ = G.tile_layout(p)
    mall = [O.dwt53_fwd(pl.reshape(8192, 8192), 5) for pl in planes]
    rng = np.random.default_rng(1)
    for i in rng.choice(len(blocks), 64, replace=False):
        b = blocks[i]
        sm = O.signmag(mall[b.comp][b.py:b.py + (b.y1 - b.y0), b.px:b.px + (b.x1 - b.x0)], b.kmax)
        assert b1[i] == O.ht_encode_sm(sm, b.kmax), "block %d" % i


# ---- irreversible path (ICT + 9/7 + dead-zone quantiser + HT): BASELINE configs[2] shape -----------
import chain


@pytest.mark.parametrize("C,H,W,prec,L", [(3, 128, 192, 8, 3), (3, 256, 256, 16, 5), (1, 100, 77, 12, 2), (4, 64, 96, 10, 3)])
def test_encode_irreversible_blocks_vs_oracle(C, H, W, prec, L):
    """Whole irreversible encode on the GPU (fused ingest + 9/7, quantiser inside K3) == the oracle chain,
    block for block; the oracle chain's codestreams are pinned to grk_decompress in test_oracle_decode.py
    (the reference's own HT 9/7 *encoder* is unusable as an oracle: defect D1)."""
    px = synth.g2(C, H, W, prec)
    p, blocks, qcd, otable, ocoded = chain.encode_tile_oracle(px, prec, L, irrev=True)
    table, coded = U.ctx().encode_host(p, px)
    got = U.split_blocks(table, coded)
    want = [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]
    bad = [i for i in range(len(blocks)) if got[i] != want[i]]
    assert not bad, "blocks differing from the oracle: %s" % bad[:10]


@pytest.mark.parametrize("cblk", [(5, 5), (6, 5), (4, 4), (5, 6), (2, 2)])
def test_gpu_codestream_other_block_sizes_vs_grok(cblk):
    """Code-block sizes other than 64x64 (COD SPcod): the GPU file equals Grok's CPU encode byte for byte."""
    import refharness as R
    if not R.have_ref():
        pytest.skip("oracle/_ref not shipped")
    px = synth.g2(3, 128, 192, 8)
    want, _ = R.encode(px, 8, numres=4, mode=1, cblk=(1 << cblk[0], 1 << cblk[1]))
    assert gpu_codestream(px, 8, 3, cblk=cblk) == want


@pytest.mark.parametrize("C,H,W,L,sgnd,mct", [(3, 512, 640, 5, False, True), (1, 300, 517, 3, False, False), (3, 257, 129, 2, False, True),
                                              (3, 256, 256, 5, True, True), (4, 192, 320, 4, False, True), (3, 1024, 1024, 1, False, False)])
def test_int16_planes_equal_int32_planes(C, H, W, L, sgnd, mct, monkeypatch):
    """8-bit reversible encodes keep int16 LL / Mallat planes between K2 and K3 (half the bytes; context.hip
    planes16_ok).  The same tile through a context with GRK_AMD_PLANES16=0 (int32 planes), with and without the
    K3/DWT overlap, and through the oracle chain: identical blocks.  Extreme pixels (0 / 255 checkerboards)
    push the coefficients to the bound the 16-bit planes are sized for."""
    rng = np.random.default_rng(H * 7 + W)
    px = synth.g2(C, H, W, 8)
    yy, xx = np.mgrid[0:H, 0:W]
    px[:, :H // 2] = np.where(((yy[:H // 2] // 3 + xx[:H // 2] // 5 + rng.integers(0, 2)) & 1) == 0, 0, 255).astype(np.uint8)
    if sgnd:
        px = (px.astype(np.int32) - 128).astype(np.int8)
    p = G.TileParams.make(W, H, C, 8, L, sgnd=sgnd, mct=mct)
    got = {}
    for planes16, overlap in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("GRK_AMD_PLANES16", planes16)
        monkeypatch.setenv("GRK_AMD_OVERLAP", overlap)
        c = G.Context(0)
        try:
            t, coded = c.encode_host(p, px)
            got[(planes16, overlap)] = U.split_blocks(t, coded)
        finally:
            c.close() if hasattr(c, "close") else None
    assert got[("1", "1")] == got[("0", "1")] == got[("1", "0")]
    _, _, _, otable, ocoded = chain.encode_tile_oracle(px, 8, L, mct=mct, sgnd=sgnd)
    want = [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]
    assert got[("1", "1")] == want


@pytest.mark.parametrize("cell", [1, 2, 4, 8, 0])
def test_packed_int16_dwt_at_the_extremes(cell, monkeypatch):
    """Levels whose every intermediate stays inside 16 bits run on packed int16 pairs (kernels_dwt.hip strip_pk;
    context.hip pk16_level_ok gives the levels).  Content made to drive the lifting sums as far as 8-bit pixels can --
    0 / 255 checkerboards with cells of 1, 2, 4, 8 pixels (each resonates with one level), and binary noise (cell 0) -- through
    5 levels, with different phases per component so that the RCT's chroma reaches +-255: blocks identical to the 32-bit
    arithmetic (GRK_AMD_DWT_PK=0) and to the oracle chain."""
    H, W, L = 512, 1024, 5
    rng = np.random.default_rng(cell)
    yy, xx = np.mgrid[0:H, 0:W]
    if cell:
        base = (((yy // cell) + (xx // cell)) & 1).astype(np.uint8) * 255
        px = np.stack([base, 255 - base, np.roll(base, cell, axis=1)])
    else:
        px = (rng.integers(0, 2, size=(3, H, W)) * 255).astype(np.uint8)
    px = np.ascontiguousarray(px)
    p = G.TileParams.make(W, H, 3, 8, L)
    got = {}
    for pk in ("1", "0"):
        monkeypatch.setenv("GRK_AMD_DWT_PK", pk)
        c = G.Context(0)
        t, coded = c.encode_host(p, px)
        got[pk] = U.split_blocks(t, coded)
    assert got["1"] == got["0"]
    _, _, _, otable, ocoded = chain.encode_tile_oracle(px, 8, L, mct=True)
    want = [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]
    assert got["1"] == want


@pytest.mark.parametrize("C,H,W,L", [(3, 16, 256, 1), (1, 18, 260, 1), (3, 50, 1000, 2), (1, 34, 1924, 1), (3, 130, 2044, 3), (3, 64, 4100, 2),
                                     (3, 96, 964, 1), (1, 200, 3844, 2)])
def test_packed_dwt_strip_geometries(C, H, W, L):
    """The packed 5/3 kernels (four columns per lane, strips of up to 960 columns shared evenly) at widths that leave
    ragged last strips, one-lane last strips, mirrored halo groups on both sides: encode == oracle chain block for block,
    and the decode (packed inverse kernels) returns the pixels."""
    px = synth.g2(C, H, W, 8)
    px[:, :, -3:] = 255 - px[:, :, -3:]            # (something to mirror at the right edge)
    p = G.TileParams.make(W, H, C, 8, L)
    t, coded = U.ctx().encode_host(p, px)
    got = U.split_blocks(t, coded)
    _, _, _, otable, ocoded = chain.encode_tile_oracle(px, 8, L, mct=(C >= 3))
    want = [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]
    assert got == want
    back = U.ctx().decode_host(p, t, coded)
    assert np.array_equal(np.asarray(back).reshape(px.shape), px)


@pytest.mark.parametrize("seed", range(10))
def test_packed_dwt_random_shapes(seed):
    """Random shapes on the packed kernels' side of the launch conditions (width a multiple of 4 from 256 up, even height from
    16 up) and just off it (odd heights, widths that are not: the 32-bit kernels take those levels), 1 / 3 / 4 components,
    1-5 levels, batches of tiles: blocks == oracle chain, decode == pixels."""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.choice([1, 3, 3, 4]))
    W = int(rng.integers(64, 700)) * 4 + (int(rng.integers(0, 4)) if seed % 3 == 2 else 0)
    H = int(rng.integers(8, 120)) * 2 + (1 if seed % 4 == 3 else 0)
    L = int(rng.integers(1, 6))
    nt = int(rng.choice([1, 1, 2, 3]))
    px = np.stack([synth.g2(C, H, W, 8, seed=seed * 7 + t) for t in range(nt)])
    p = G.TileParams.make(W, H, C, 8, L)
    t, coded = U.ctx().encode_host(p, px, ntiles=nt)
    got = U.split_blocks(t, coded)
    want = []
    for k in range(nt):
        _, _, _, otable, ocoded = chain.encode_tile_oracle(px[k], 8, L, mct=(C >= 3))
        want += [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]
    assert got == want, (C, H, W, L, nt)
    back = U.ctx().decode_host(p, t, coded, ntiles=nt)
    assert np.array_equal(np.asarray(back).reshape(px.shape), px), (C, H, W, L, nt)


def _dev_view(ptr, n, typestr):
    class _H:
        pass
    h = _H()
    h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")


@pytest.mark.parametrize("depth,room,frame_streams", [(1, None, "0"), (2, None, "0"), (1, "0", "0"), (2, "1", "0"), (1, "2", "0"),
                                                      (1, None, None), (2, None, None), (3, "0", "2")])
def test_pipelined_encodes_keep_their_results_until_the_second_next_call(depth, room, frame_streams, monkeypatch):
    """grk_amd_set_pipelining: consecutive encodes overlap (the next one's DWT runs while this one's blocks are still
    being coded), each working in its own buffer set.  Different images back to back without any fetch in between:
    the device-resident results of encode k are read AFTER encode k+1 (two sets) / k+2 (three sets: set_pipelining(2))
    has been issued, and equal a plain encode's.  `room`: GRK_AMD_K3_ROOM -- which K3 launches of the sequence run the
    instance that leaves registers for the next frame's DWT (default: both classes); `frame_streams`: GRK_AMD_FRAME_STREAMS -- "0":
    the DWT chain on the main stream and K3 on the side streams behind events (what large frames take), default / "2": a frame's whole
    chain on one of the side streams in turn (what frames of this size take); the bytes are the same either way."""
    if room is not None:
        monkeypatch.setenv("GRK_AMD_K3_ROOM", room)
    if frame_streams is not None:
        monkeypatch.setenv("GRK_AMD_FRAME_STREAMS", frame_streams)
    p = G.TileParams.make(1024, 768, 3, 8, 5)
    imgs = [synth.g2(3, 768, 1024, 8), (synth.g2(3, 768, 1024, 8)[:, ::-1, :]).copy(), (255 - synth.g2(3, 768, 1024, 8)).astype(np.uint8),
            synth.g2(3, 768, 1024, 8, seed=7), (synth.g2(3, 768, 1024, 8, seed=8)[:, :, ::-1]).copy()]
    want = []
    for im in imgs:
        t, coded = U.ctx().encode_host(p, im)
        want.append(U.split_blocks(t, coded))
    c = G.Context(0)
    c.set_pipelining(depth)
    nb = G.lib().grk_amd_tile_num_blocks(p)
    d = [U.to_dev(im.reshape(-1)) for im in imgs]
    held = []
    for k in range(len(imgs)):
        c.encode_tiles(p, 1, d[k].data_ptr(), True, fetch=False)
        held.append((c.coded_device_ptr(), c.table_device_ptr(0), c.table_device_ptr(1), c.table_device_ptr(2)))
        if k >= depth:                   # look at encode k - depth now that `depth` more encodes are in flight
            torch.cuda.synchronize()
            arena_p, off_p, len_p, used_p = held[k - depth]
            used = int(_dev_view(used_p, 1, "<i8").cpu()[0])
            offs = _dev_view(off_p, nb, "<i8").cpu().numpy()
            lens = _dev_view(len_p, nb, "<i4").cpu().numpy()
            arena = _dev_view(arena_p, used, "|u1").cpu().numpy()
            got = [bytes(arena[int(o):int(o) + int(l)]) for o, l in zip(offs, lens)]
            assert got == want[k - depth], "encode %d" % (k - depth)
    assert len({h[0] for h in held}) == depth + 1        # that many coded arenas in rotation
    t, tot = c.fetch_table(nb)           # the last one through the ordinary path (joins the side streams)
    coded = c.fetch_coded(tot)
    assert U.split_blocks(t, coded) == want[-1]
    c.set_pipelining(False)
    c.close()


@pytest.mark.parametrize("frame_streams,hold", [("0", False), ("2", False), ("2", True), ("0", True)])
def test_pipelined_encodes_out_of_one_pixel_buffer_rewritten_in_stream_order(frame_streams, hold, monkeypatch):
    """The pixel-lifetime contract of grk_amd_encode_tiles with device pixels: the pixels are read in the order of the CONTEXT's stream,
    so a caller that gave the context its own stream (grk_amd_set_stream) may overwrite the buffer on that stream right behind the
    call.  With a frame's chain on one of the side streams (GRK_AMD_FRAME_STREAMS, what frames of this size take) level 0 reads the
    pixels there: the context's stream waits for that read.  Ten frames through ONE buffer, each copied in on the caller's stream
    directly behind the previous call, no host synchronisation in between; every frame's blocks equal a plain encode's.
    `hold`: grk_amd_set_pixel_hold(ctx, 1) -- the context leaves that wait out and the caller's stream takes it explicitly
    (grk_amd_stream_wait_pixels) before it refills the buffer."""
    monkeypatch.setenv("GRK_AMD_FRAME_STREAMS", frame_streams)
    W, H = 1536, 1024
    p = G.TileParams.make(W, H, 3, 8, 5)
    imgs = [synth.g2(3, H, W, 8, seed=40 + i) if i % 2 else (255 - synth.g2(3, H, W, 8, seed=40 + i)).astype(np.uint8) for i in range(10)]
    want = []
    for im in imgs:
        t, coded = U.ctx().encode_host(p, im)
        want.append(U.split_blocks(t, coded))
    nb = G.lib().grk_amd_tile_num_blocks(p)
    src = [U.to_dev(im.reshape(-1)) for im in imgs]
    one = torch.empty_like(src[0])
    st = torch.cuda.Stream()
    c = G.Context(0)
    c.set_stream(st.cuda_stream)
    depth = 3
    c.set_pipelining(depth)
    c.set_pixel_hold(hold)
    held = []
    with torch.cuda.stream(st):
        for k in range(len(imgs)):
            if hold and k:
                c.stream_wait_pixels(st.cuda_stream)
            one.copy_(src[k], non_blocking=True)        # on the caller's stream = the context's stream: behind call k - 1
            c.encode_tiles(p, 1, one.data_ptr(), True, fetch=False)
            held.append((c.coded_device_ptr(), c.table_device_ptr(0), c.table_device_ptr(1), c.table_device_ptr(2)))
            if k >= depth:
                c.synchronize()
                torch.cuda.synchronize()
                arena_p, off_p, len_p, used_p = held[k - depth]
                used = int(_dev_view(used_p, 1, "<i8").cpu()[0])
                offs = _dev_view(off_p, nb, "<i8").cpu().numpy()
                lens = _dev_view(len_p, nb, "<i4").cpu().numpy()
                arena = _dev_view(arena_p, used, "|u1").cpu().numpy()
                assert [bytes(arena[int(o):int(o) + int(l)]) for o, l in zip(offs, lens)] == want[k - depth], "frame %d" % (k - depth)
    t, tot = c.fetch_table(nb)
    coded = c.fetch_coded(tot)
    assert U.split_blocks(t, coded) == want[-1]
    c.set_pipelining(False)
    c.set_stream(0)
    c.close()


def test_pipelined_encodes_of_both_forms_in_turn():
    """Calls of one geometry with one tile (a frame's whole chain on one side stream: up to 16 M samples per call) and with eight
    tiles (DWT chain on the main stream, K3 on the side streams) in turn, no fetch in between: each call's device-resident results,
    read after the next call has been issued, equal a plain encode's -- the buffer sets (the LL ping-pong buffers among them) are
    what keeps a running frame's intermediate data from the next call, whichever form it takes."""
    W = H = 1024
    p = G.TileParams.make(W, H, 3, 8, 5)
    base = [synth.g2(3, H, W, 8, seed=20 + i) for i in range(8)]
    calls = [base[:1], base[:8], base[1:2], [b[:, ::-1, :].copy() for b in base], base[2:3]]
    nb1 = G.lib().grk_amd_tile_num_blocks(p)
    want = []
    for tiles in calls:
        per = []
        for im in tiles:
            t, coded = U.ctx().encode_host(p, im)
            per += U.split_blocks(t, coded)
        want.append(per)
    c = G.Context(0)
    c.set_pipelining(1)
    d = [U.to_dev(np.stack(tiles).reshape(-1)) for tiles in calls]
    held = []

    def check(k):
        torch.cuda.synchronize()
        arena_p, off_p, len_p, used_p = held[k]
        nb = nb1 * len(calls[k])
        used = int(_dev_view(used_p, 1, "<i8").cpu()[0])
        offs = _dev_view(off_p, nb, "<i8").cpu().numpy()
        lens = _dev_view(len_p, nb, "<i4").cpu().numpy()
        arena = _dev_view(arena_p, used, "|u1").cpu().numpy()
        assert [bytes(arena[int(o):int(o) + int(l)]) for o, l in zip(offs, lens)] == want[k], "call %d" % k
    for k, tiles in enumerate(calls):
        c.encode_tiles(p, len(tiles), d[k].data_ptr(), True, fetch=False)
        held.append((c.coded_device_ptr(), c.table_device_ptr(0), c.table_device_ptr(1), c.table_device_ptr(2)))
        if k >= 1:
            check(k - 1)
    check(len(calls) - 1)
    c.set_pipelining(False)
    c.close()


@pytest.mark.parametrize("kind", ["noise", "smooth", "mixed"])
def test_ht_encoder_lds_cap_and_fallback(kind, monkeypatch):
    """K3 sizes its LDS streams for what real content needs (occupancy); a block that outgrows them is coded again by
    the fallback launch with worst-case buffers.  White noise overflows the blocks of the fine sub-bands, a smooth image none, a
    mixture some: the blocks always equal the oracle's and those of a context with GRK_AMD_LDS_CAP=0."""
    rng = np.random.default_rng(3)
    H, W = 512, 768
    noise = rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
    smooth = synth.g2(3, H, W, 8)
    px = {"noise": noise, "smooth": smooth, "mixed": np.where((np.arange(W) // 128 % 2 == 0)[None, None, :], noise, smooth)}[kind]
    px = np.ascontiguousarray(px)
    p = G.TileParams.make(W, H, 3, 8, 5)
    nb = G.lib().grk_amd_tile_num_blocks(p)
    got = {}
    for cap in ("1", "0"):
        monkeypatch.setenv("GRK_AMD_LDS_CAP", cap)
        c = G.Context(0)
        t, coded = c.encode_host(p, px)
        got[cap] = U.split_blocks(t, coded)
        handed = int(_dev_view(c.table_device_ptr(3), 24, "<i8").cpu().sum())
        if cap == "0":
            assert handed == 0
        elif kind == "smooth":       # at most the few high-Kmax blocks of the lowest resolutions (the buffers are sized for the typical block)
            assert handed <= sum(1 for b in G.tile_layout(p)[0] if b.res <= 2)
        elif kind == "noise":
            assert handed > nb // 8
        else:
            assert 0 < handed < nb
        c.close()
    assert got["1"] == got["0"]
    _, _, _, otable, ocoded = chain.encode_tile_oracle(px, 8, 5)
    assert got["1"] == [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        C = int(rng.choice([1, 3, 3, 4]))
        W, H = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        prec = int(rng.choice([8, 8, 10, 12, 16]))
        L = int(rng.integers(0, 7))
        irrev = bool(rng.integers(0, 3) == 0)
        sgnd = bool(rng.integers(0, 4) == 0)
        mct = C >= 3 and bool(rng.integers(0, 4) != 0)
        if irrev and not mct and C >= 3:
            continue                                  # (the reference's irreversible path without MCT scales by 2048: D1)
        out.append((C, H, W, prec, L, irrev, sgnd, mct))
    return out


@pytest.mark.parametrize("C,H,W,prec,L,irrev,sgnd,mct", _random_cases(24, 20260926) + _random_cases(24, 7))
def test_random_geometries_encode_and_decode(C, H, W, prec, L, irrev, sgnd, mct):
    """A seeded sweep over odd sizes, 0..6 levels, 1/3/4 components, 8..16 bits, signed, with and without MCT, 5/3 and
    9/7: the encoder's blocks == the oracle chain's, and the decoder returns the source (lossless) or what the oracle's
    decode chain makes of the same blocks (9/7).  Every fast path and its fallback gets exercised by some case:
    FAST / edge DWT strips, 16-bit planes, fused level 0 / last level, capped LDS, K3 / DWT overlap."""
    rng = np.random.default_rng(C * 1000003 + H * 1009 + W)
    u = synth.g2(C, H, W, prec).astype(np.int64)
    u += rng.integers(-3, 4, size=u.shape)                          # a little texture on top of the ramp
    u = np.clip(u, 0, (1 << prec) - 1)
    if sgnd:
        px = (u - (1 << (prec - 1))).astype(np.int8 if prec <= 8 else np.int16)
    else:
        px = u.astype(np.uint8 if prec <= 8 else np.uint16)
    p, blocks, qcd, otable, ocoded = chain.encode_tile_oracle(px, prec, L, irrev=irrev, mct=mct, sgnd=sgnd)
    table, coded = U.ctx().encode_host(p, px)
    got = U.split_blocks(table, coded)
    want = [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]
    bad = [i for i in range(len(want)) if got[i] != want[i]]
    assert not bad, "blocks differing from the oracle chain: %s" % bad[:10]
    try:
        ref = chain.decode_tile_oracle(p, blocks, qcd, otable, ocoded)
    except AssertionError:                # near-full-scale content: the reference's decoder rejects U_q > missing_msbs (D5)
        with pytest.raises(RuntimeError):
            U.ctx().decode_host(p, table, coded)
        return
    back = U.ctx().decode_host(p, table, coded)[0]
    assert np.array_equal(back.view(px.dtype).astype(np.int32), ref.astype(np.int32))
    if not irrev:
        assert np.array_equal(back.view(px.dtype), px)


@pytest.mark.parametrize("C,H,W,L,irrev", [(3, 1024, 1536, 5, False), (3, 700, 1500, 4, True), (1, 333, 2111, 3, False)])
def test_xcd_aware_workgroup_order_changes_nothing_but_the_order(C, H, W, L, irrev, monkeypatch):
    """K2 / K6 map workgroups to (strip, row segment, plane) so that the strips an XCD works on are neighbours
    (GRK_AMD_DWT_XCD, default 1; grids whose size is no multiple of 8 included): identical blocks and identical pixels
    with the plain order of GRK_AMD_DWT_XCD=0, and both equal to the oracle."""
    px = synth.g2(C, H, W, 8, seed=H)
    p = G.TileParams.make(W, H, C, 8, L, irreversible=irrev)
    got = {}
    for xcd in ("1", "0"):
        monkeypatch.setenv("GRK_AMD_DWT_XCD", xcd)
        c = G.Context(0)
        try:
            t, coded = c.encode_host(p, px)
            got[xcd] = (U.split_blocks(t, coded), c.decode_host(p, t, coded)[0])
        finally:
            c.close()
    assert got["1"][0] == got["0"][0]
    assert np.array_equal(got["1"][1], got["0"][1])
    if not irrev:
        assert np.array_equal(got["1"][1], px)
        _, _, _, otable, ocoded = chain.encode_tile_oracle(px, 8, L)
        assert got["1"][0] == [bytes(ocoded[int(o):int(o) + int(l)]) for o, l in zip(otable["offset"], otable["length"])]
