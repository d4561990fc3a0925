"""-m gpu: Tier-2 on the device (grk_amd_assemble_device, kernels_t2.hip) against the host writer (grk_amd_write_tile_part,
t2_writer.cpp -- itself pinned on grk_compress's files by test_length_markers / test_gpu_precincts): the tile-parts the device
assembles are the host writer's bytes, for every progression order, with SOP / EPH / PLT, precincts down to a handful of blocks,
tiles off the grid, batches of several tiles, appended batches, deep content (long headers with many 0xFF) and empty blocks."""
import numpy as np
import pytest

import grok_amd as G
import gpuutil as U
import synth
from test_precincts_cpu import exps_from_sizes

pytestmark = pytest.mark.gpu


def _host_parts(p, indices, table, coded, flags):
    bpt = len(table) // len(indices)
    return [G.write_tile_part(p, int(t), table[i * bpt:(i + 1) * bpt], coded, flags) for i, t in enumerate(indices)]


def _check(p, px, indices, flags, ntiles=1):
    c = U.ctx()
    table, coded = c.encode_host(p, px, ntiles=ntiles)
    n, lens = c.assemble_device(p, indices, flags)
    want = _host_parts(p, indices, table, coded, flags)
    assert [int(v) for v in lens] == [len(w) for w in want]
    got = bytes(c.fetch_assembled(0, n))
    assert n == sum(len(w) for w in want)
    at = 0
    for i, w in enumerate(want):
        if got[at:at + len(w)] != w:
            g = np.frombuffer(got[at:at + len(w)], np.uint8)
            first = int(np.nonzero(g != np.frombuffer(w, np.uint8))[0][0])
            raise AssertionError("tile-part %d differs from byte %d of %d" % (i, first, len(w)))
        at += len(w)
    return want


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("extra", [0, G.CS_PLT, G.CS_SOP | G.CS_EPH, G.CS_PLT | G.CS_SOP | G.CS_EPH])
def test_orders_and_markers(order, extra):
    px = synth.g2(3, 192, 256, 8, seed=order + extra)
    p = G.TileParams.make(256, 192, 3, 8, 4, precincts=exps_from_sizes([(128, 128), (64, 64)], 4))
    _check(p, px, [5], G.CS_PROG(order) | extra)


@pytest.mark.parametrize("C,W,H,prec,L,org,sizes,irrev", [
    (3, 1024, 768, 8, 5, (0, 0), None, False),               # one precinct per resolution: 768 blocks in the largest packet
    (1, 300, 210, 12, 3, (0, 0), [(64, 32), (32, 64)], False),
    (3, 257, 129, 8, 5, (33, 95), [(128, 64), (64, 64), (16, 16)], False),      # off the grid, blocks of 16 x 16 and smaller
    (3, 96, 80, 8, 3, (0, 0), [(16, 16)], False),
    (3, 640, 512, 16, 5, (0, 0), None, True),                # 16-bit ICT + 9/7: long blocks, long Lblock codes
    (1, 64, 64, 8, 0, (0, 0), None, False),                  # no DWT level: one packet of one block
    (3, 33, 17, 8, 2, (7, 3), None, False),
])
def test_geometries(C, W, H, prec, L, org, sizes, irrev):
    px = synth.g2(C, H, W, prec, seed=W + L)
    prc = exps_from_sizes(sizes, L) if sizes else None
    p = G.TileParams.make(W, H, C, prec, L, origin=org, precincts=prc, irreversible=irrev)
    _check(p, px, [0], G.CS_PLT)


def test_headers_longer_than_one_window_and_than_one_workgroup_pass():
    """2048 x 2048 x 3, 5 levels: the top resolution's packets hold 3 072 blocks each -- three passes of a 1 024-lane workgroup over the
    blocks, a raw header of ~90 000 bits; with 16 x 16 code-blocks 12 288 blocks per packet: several 16 KB windows of raw bits, the
    chain's state carried from window to window."""
    px = synth.g2(3, 2048, 2048, 8, seed=11)
    p = G.TileParams.make(2048, 2048, 3, 8, 5)
    _check(p, px, [0], 0)
    p = G.TileParams.make(2048, 2048, 3, 8, 5, cblk=(4, 4))
    _check(p, px, [0], G.CS_SOP)


def test_flat_and_empty_blocks():
    """A flat frame: every block of the detail bands is empty (length 0: Lblock stays 3, three zero bits of length), and the headers are
    runs of ones -- 0xFF after 0xFF, the densest stuffing there is."""
    px = np.full((3, 512, 512), 128, np.uint8)
    p = G.TileParams.make(512, 512, 3, 8, 5)
    _check(p, px, [0], G.CS_PLT | G.CS_EPH)
    px = np.zeros((1, 256, 256), np.uint8)
    p = G.TileParams.make(256, 256, 1, 8, 3, cblk=(3, 3))
    _check(p, px, [9], 0)


def test_batches_of_tiles_and_appended_batches():
    c = U.ctx()
    px = synth.g2(3, 4 * 128, 160, 8, seed=4).reshape(4, 3, 128, 160)          # four tiles of 160 x 128
    p = G.TileParams.make(160, 128, 3, 8, 3)
    flags = G.CS_PLT | G.CS_PROG(2)
    want = _check(p, px, [3, 0, 65534, 7], flags, ntiles=4)
    # a second batch of another geometry appended behind the first: what lay below dst_offset is kept
    n0 = sum(len(w) for w in want)
    px2 = synth.g2(1, 2 * 96, 96, 12, seed=5).reshape(2, 1, 96, 96)
    p2 = G.TileParams.make(96, 96, 1, 12, 2)
    table2, coded2 = c.encode_host(p2, px2, ntiles=2)
    n1, lens2 = c.assemble_device(p2, [1, 2], flags, dst_offset=n0)
    want2 = _host_parts(p2, [1, 2], table2, coded2, flags)
    assert n1 == sum(len(w) for w in want2)
    assert bytes(c.fetch_assembled(0, n0 + n1)) == b"".join(want) + b"".join(want2)


def test_refuses_what_it_cannot_assemble():
    c = U.ctx()
    px = synth.g2(1, 64, 64, 8, seed=1)
    p = G.TileParams.make(64, 64, 1, 8, 2)
    c.encode_host(p, px)
    q = G.TileParams.make(64, 64, 1, 8, 3)
    with pytest.raises(RuntimeError):
        c.assemble_device(q, [0])                       # not the call before it
    with pytest.raises(RuntimeError):
        c.assemble_device(p, [0, 1])                    # not its number of tiles
    n, _ = c.assemble_device(p, [0])
    with pytest.raises(RuntimeError):
        c.assemble_device(p, [0], dst_offset=n + 1)     # a hole in the output


def test_plt_of_many_packets_splits_into_marker_segments_on_the_device():
    """16 x 16 precincts of a 2048 x 1024 x 3 tile: ~30 000 packets, each length one to two bytes in PLT -- a handful of marker segments
    (the host's write_plt and the device's frame kernel have to agree on where they split)."""
    px = synth.g2(3, 1024, 2048, 8, seed=21)
    p = G.TileParams.make(2048, 1024, 3, 8, 3, precincts=exps_from_sizes([(16, 16)], 3))
    want = _check(p, px, [2], G.CS_PLT | G.CS_SOP | G.CS_PROG(3))
    assert want[0].count(b"\xff\x58") >= 2


def test_asynchronous_form_on_a_stream_of_the_callers_with_pipelined_encodes():
    """grk_amd_assemble_device_async: nothing waited for, results read from the device tables; frames alternate between two inputs while
    the encoder rotates three buffer sets -- every frame's tile-parts are still intact two frames later."""
    import torch
    c = G.Context(0)
    try:
        p = G.TileParams.make(320, 256, 3, 8, 4)
        flags = G.CS_PLT | G.CS_EPH
        idx = [4, 9, 1]
        frames = [synth.g2(3, 3 * 256, 320, 8, seed=s).reshape(3, 3, 256, 320) for s in (1, 2)]
        want = []
        for px in frames:
            table, coded = c.encode_host(p, px, ntiles=3)
            want.append(b"".join(_host_parts(p, idx, table, coded, flags)))
        d_px = [U.to_dev(px.reshape(-1).view(np.uint8)) for px in frames]
        st, t2 = torch.cuda.Stream(), torch.cuda.Stream()
        c.set_stream(st.cuda_stream)
        c.set_pipelining(2)
        held = []
        for f in range(6):
            with torch.cuda.stream(st):
                c.encode_tiles(p, 3, d_px[f & 1].data_ptr(), True, fetch=False)
            c.assemble_device_async(p, idx, flags, t2.cuda_stream)
            held.append((f, c.assembled_device_ptr(), c.assembled_table_ptr(0), c.assembled_table_ptr(1), c.assembled_table_ptr(2)))
            if len(held) == 3:                      # frame f - 2: its output set is reused by the NEXT call, not before
                g, out, dst, ln, tot = held.pop(0)
                t2.synchronize()
                total = U.from_dev_ptr(tot, 16).view(np.uint64)
                assert int(total[0]) == len(want[g & 1])
                lens = U.from_dev_ptr(ln, 12).view(np.uint32)
                offs = U.from_dev_ptr(dst, 24).view(np.uint64)
                assert [int(v) for v in offs] == [0, int(lens[0]), int(lens[0]) + int(lens[1])]
                assert bytes(U.from_dev_ptr(out, int(total[0]))) == want[g & 1], "frame %d" % g
        c.set_pipelining(False)
        c.synchronize()
    finally:
        c.close()


def test_encode_image_takes_the_device_route_and_the_host_writer_agrees(monkeypatch):
    """grk_amd_encode_image assembles on the device by default; GRK_AMD_IMAGE_T2=host is the host writer: the same file, for an offset
    tiling with nine geometry groups, markers and a position-major progression order."""
    c = U.ctx()
    W, H, TW, TH, L = 1000, 900, 384, 320, 3
    px = synth.g2(3, H, W, 8, seed=31)
    layout = G.ImageLayout.make(W, H, TW, TH, offset=(100, 60))
    base = G.TileParams.make(1, 1, 3, 8, L, precincts=exps_from_sizes([(64, 64)], L))
    flags = G.CS_TLM | G.CS_PLT | G.CS_SOP | G.CS_PROG(3)
    dev = c.encode_image(layout, base, px, flags)
    monkeypatch.setenv("GRK_AMD_IMAGE_T2", "host")
    assert c.encode_image(layout, base, px, flags) == dev


@pytest.mark.parametrize("C,W,H,prec,L,org,cblk", [
    (1, 1, 1, 8, 1, (0, 0), (6, 6)),          # one sample: resolutions without a band, packets that are just the "not empty" bit
    (3, 5, 3, 8, 3, (0, 0), (6, 6)),          # more levels than the tile has rows to halve
    (4, 130, 70, 8, 2, (3, 1), (5, 4)),       # four components (no MCT for the fourth), 32 x 16 blocks, an odd origin
    (2, 64, 64, 12, 5, (0, 0), (2, 2)),       # 4 x 4 code-blocks: 256 blocks per band at the top, headers of a few bits each
    (1, 2048, 8, 8, 1, (0, 0), (6, 2)),       # 64 x 4 blocks
])
def test_degenerate_geometries(C, W, H, prec, L, org, cblk):
    px = synth.g2(C, H, W, prec, seed=W * 7 + H)
    p = G.TileParams.make(W, H, C, prec, L, origin=org, cblk=cblk)
    for flags in (0, G.CS_PLT | G.CS_SOP | G.CS_EPH | G.CS_PROG(4)):
        _check(p, px, [1], flags)
