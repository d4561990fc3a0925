"""CPU, world_size 2 over gloo: the N>1 path of bench.py / grok_amd.dist -- tile sharding, header
broadcast, gather of coded tile-parts, codestream assembly on rank 0 -- produces the same file
as a single process (= the reference encoder's file, golden fixture)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import grok_amd as G
import grok_amd.dist as D
import cshelp
import oracle as O
import synth
from grok_amd.capi import CODED_DTYPE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _encode_my_tiles(px, TW, TH, tcols, mine, prec, L):
    """Stand-in for the GPU encode of a rank's tiles: the oracle.  -> (table, coded bytes as a tensor)"""
    tabs, chunks, off = [], [], 0
    for t in mine:
        ty, tx = divmod(t, tcols)
        tile = np.ascontiguousarray(px[:, ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW])
        _, lens, coded = O.encode_tile_rev(tile, prec, L)
        tt = np.zeros(len(lens), CODED_DTYPE)
        tt["length"] = lens
        tt["offset"] = off + np.concatenate([[0], np.cumsum(lens)[:-1]])
        off += int(lens.sum())
        tabs.append(tt)
        chunks.append(coded)
    table = np.concatenate(tabs) if tabs else np.zeros(0, CODED_DTYPE)
    coded = torch.from_numpy(np.concatenate(chunks)) if chunks else torch.zeros(0, dtype=torch.uint8)
    return table, coded


def _worker(rank, world, port, q, W, H, TW, TH, L, frames, tmpfile, depth=1, lag=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    tcols, trows = W // TW, H // TH
    ntiles = tcols * trows
    p0 = G.TileParams.make(TW, TH, 3, 8, L) if rank == 0 else G.TileParams.make(1, 1, 1, 8, 0)
    p = D.broadcast_params(p0, dev)                      # everybody now holds rank 0's parameters
    assert (p.tile_w, p.tile_h, p.num_comps, p.num_levels) == (TW, TH, 3, L)
    bpt = G.lib().grk_amd_tile_num_blocks(p)
    mine = D.shard_tiles(ntiles, world, rank)            # uneven when world does not divide the tile count
    # ---- a sequence of frames through the pipeline: counts exchange of frame f, gather of frame f - 1, rotating writer
    pipe = D.FramePipeline(dev, depth=depth, lag=lag)    # depth > 1: one process group per gather in flight + one for the counts
    files = {}
    for f in range(frames):
        px = synth.g2(3, H, W, 8, seed=12345 + f)
        table, coded = _encode_my_tiles(px, TW, TH, tcols, mine, 8, L)
        offs = torch.from_numpy(table["offset"].astype(np.int64))
        lens = torch.from_numpy(table["length"].astype(np.int32))
        pipe.submit(f, torch.tensor([coded.numel()]), offs, lens, coded)
        for done, parts, root in pipe.pop_completed():   # the gathers this submit issued (frame f - lag)
            assert done == f - lag and root == done % world
            if root == rank:                             # this rank is that frame's writer
                ft, fc = D.merge_tile_parts(D.parts_to_numpy(parts), ntiles, bpt)
                files[done] = G.write_codestream(p, W, H, ft, fc)
    parts, root = pipe.flush()
    assert root == (frames - 1) % world
    for done, parts, root in pipe.pop_completed():
        if root == rank:
            ft, fc = D.merge_tile_parts(D.parts_to_numpy(parts), ntiles, bpt)
            files[done] = G.write_codestream(p, W, H, ft, fc)
    # ---- the parallel-writer route over the C ABI pieces: every rank sizes and writes its own tile-parts; the sizes are
    #      exchanged, so that each rank knows its offsets in the file (and rank 0 can write TLM into the main header)
    px = synth.g2(3, H, W, 8, seed=12345)
    table, coded = _encode_my_tiles(px, TW, TH, tcols, mine, 8, L)
    cbytes = coded.numpy()
    flags = G.CS_TLM | G.CS_PLT
    my_sizes = torch.zeros(ntiles, dtype=torch.int64)
    for k, t in enumerate(mine):
        my_sizes[t] = G.write_tile_part(p, t, table[k * bpt:(k + 1) * bpt], None, flags=flags, size_only=True)
    dist.all_reduce(my_sizes)                             # disjoint supports: the sum is the gather
    sizes = [int(v) for v in my_sizes]
    hdr = G.write_main_header(p, W, H, flags=flags, tile_part_bytes=sizes)
    starts = np.concatenate([[len(hdr)], len(hdr) + np.cumsum(sizes)]).astype(np.int64)
    fd = os.open(tmpfile, os.O_RDWR | os.O_CREAT)
    if rank == 0:
        os.pwrite(fd, hdr, 0)
        os.pwrite(fd, b"\xff\xd9", int(starts[-1]))
    for k, t in enumerate(mine):
        os.pwrite(fd, G.write_tile_part(p, t, table[k * bpt:(k + 1) * bpt], cbytes, flags=flags), int(starts[t]))
    os.close(fd)
    dist.barrier()
    q.put((rank, files))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, W, H, TW, TH, L, frames, tmp_path, depth=1, lag=1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    tmpfile = str(tmp_path / "parallel.j2k")
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, W, H, TW, TH, L, frames, tmpfile, depth, lag)) for r in range(world)]
    for pr in procs:
        pr.start()
    files = {}
    for _ in range(world):
        _, fs = q.get(timeout=300)
        files.update(fs)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    return files, open(tmpfile, "rb").read()


def test_two_rank_tile_sharding_matches_reference_file(tmp_path):
    files, parallel = _run(2, 256, 256, 128, 128, 3, 3, tmp_path)
    want = open(os.path.join(GOLD, "g2_3x256x256_t128_r4.j2k"), "rb").read()
    assert files[0] == want                                # frame 0 = the golden image (Grok's own file)
    assert sorted(files) == [0, 1, 2]
    for f in (1, 2):                                       # the other frames: == a single process over the same pixels
        assert files[f] == cshelp.oracle_codestream(synth.g2(3, 256, 256, 8, seed=12345 + f), 8, 3, 128, 128)
    # the parallel writers' file is the golden stream with TLM + PLT added: same tile-parts, and the reference decodes it
    p = G.TileParams.make(128, 128, 3, 8, 3)
    where, used_tlm = G.locate_tile_parts(parallel)
    assert used_tlm and [w[2] for w in where] == [0, 1, 2, 3]
    import refharness as R
    if R.have_ref():
        assert np.array_equal(R.decode(parallel, 3, 256, 256), synth.g2(3, 256, 256, 8).astype(np.int32))


def test_four_ranks_seven_tiles_uneven_shards(tmp_path):
    """world 4, 7 tiles (ranks own 2, 2, 2, 1): exact-size gathers with different byte counts and table rows per rank,
    a different writer for every frame, and the parallel-writer file."""
    W, H, T, L = 7 * 64, 64, 64, 2
    files, parallel = _run(4, W, H, T, T, L, 5, tmp_path)
    assert sorted(files) == [0, 1, 2, 3, 4]
    for f in range(5):
        assert files[f] == cshelp.oracle_codestream(synth.g2(3, H, W, 8, seed=12345 + f), 8, L, T, T)
    where, used_tlm = G.locate_tile_parts(parallel)
    assert used_tlm and len(where) == 7
    import refharness as R
    if R.have_ref():
        assert np.array_equal(R.decode(parallel, 3, H, W), synth.g2(3, H, W, 8).astype(np.int32))
        assert np.array_equal(R.decode(files[3], 3, H, W), synth.g2(3, H, W, 8, seed=12348).astype(np.int32))


@pytest.mark.parametrize("world,depth,lag,frames", [(2, 3, 1, 7), (4, 4, 2, 9), (2, 1, 3, 6)])
def test_gathers_of_several_frames_in_flight(tmp_path, world, depth, lag, frames):
    """FramePipeline(depth = k): frame f's gather on communicator f mod k, the counts on one of their own; more frames than
    slots, so every slot and every writer comes round more than once.  Every frame's file == a single process's."""
    W, H, T, L = 7 * 64, 64, 64, 2
    files, _ = _run(world, W, H, T, T, L, frames, tmp_path, depth=depth, lag=lag)
    assert sorted(files) == list(range(frames))
    for f in range(frames):
        assert files[f] == cshelp.oracle_codestream(synth.g2(3, H, W, 8, seed=12345 + f), 8, L, T, T), "frame %d" % f


def test_world_8_cfg4_shape_scaled_down(tmp_path):
    """BASELINE configs[3]'s shape at the world size its scaling curve ends with, scaled down to what eight CPU ranks do in seconds:
    255 tiles of 64 x 64 (a 17 x 15 grid: ranks 0-6 own 32 tiles, rank 7 owns 31), four gathers in flight two frames behind the
    encoder, ten frames -- every writer comes round, every communicator slot twice.  Every frame's file == a single process's, and
    the parallel writers' file decodes to the image."""
    W, H, T, L = 17 * 64, 15 * 64, 64, 2
    frames = 10
    files, parallel = _run(8, W, H, T, T, L, frames, tmp_path, depth=4, lag=2)
    assert sorted(files) == list(range(frames))
    for f in range(frames):
        assert files[f] == cshelp.oracle_codestream(synth.g2(3, H, W, 8, seed=12345 + f), 8, L, T, T), "frame %d" % f
    where, used_tlm = G.locate_tile_parts(parallel)
    assert used_tlm and len(where) == 255
    import refharness as R
    if R.have_ref():
        assert np.array_equal(R.decode(parallel, 3, H, W), synth.g2(3, H, W, 8).astype(np.int32))


def test_shard_tiles_partition():
    for n, w in ((256, 8), (7, 3), (1, 4)):
        seen = sorted(t for r in range(w) for t in D.shard_tiles(n, w, r))
        assert seen == list(range(n))
