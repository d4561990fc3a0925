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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    px = synth.g2(3, 256, 256, 8)
    TW = TH = 128
    ntiles = 4
    p0 = G.TileParams.make(TW, TH, 3, 8, 3) if rank == 0 else G.TileParams.make(1, 1, 1, 8, 0)
    p = D.broadcast_params(p0, dev)                      # everybody now holds rank 0's parameters
    assert (p.tile_w, p.tile_h, p.num_comps, p.num_levels) == (128, 128, 3, 3)
    bpt = G.lib().grk_amd_tile_num_blocks(p)
    mine = D.shard_tiles(ntiles, world, rank)
    tabs, chunks, off = [], [], 0
    for t in mine:                                       # stand-in for the GPU encode: the oracle
        ty, tx = divmod(t, 2)
        tile = np.ascontiguousarray(px[:, ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW])
        _, lens, coded = O.encode_tile_rev(tile, 8, 3)
        tt = np.zeros(len(lens), CODED_DTYPE)
        tt["length"] = lens
        tt["offset"] = off + np.concatenate([[0], np.cumsum(lens)[:-1]])
        off += int(lens.sum())
        tabs.append(tt)
        chunks.append(coded)
    table = np.concatenate(tabs)
    coded = torch.from_numpy(np.concatenate(chunks))
    parts, _ = D.gather_tile_parts(table, coded, dev, dst=0)
    # the device-resident variant bench.py uses (tables as tensors, one host synchronisation)
    parts2, _ = D.gather_tile_parts_device(torch.tensor([coded.numel()]), torch.from_numpy(table["offset"].astype(np.int64)),
                                           torch.from_numpy(table["length"].astype(np.int32)), coded, dst=0)
    # the per-step exchange of the parallel-writer design: byte counts only; every rank learns its offset
    counts, my_off = D.exchange_tile_part_offsets(torch.tensor([coded.numel()]))
    all_sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(all_sizes, torch.tensor([coded.numel()], dtype=torch.int64))
    assert [int(v) for v in counts] == [int(v) for v in all_sizes]
    assert int(my_off) == sum(int(v) for v in all_sizes[:rank])
    if rank == 0:
        full_table, full_coded = D.merge_tile_parts(parts, ntiles, bpt)
        cs = G.write_codestream(p, 256, 256, full_table, full_coded)
        t2, c2 = D.merge_tile_parts(D.parts_to_numpy(parts2), ntiles, bpt)
        assert G.write_codestream(p, 256, 256, t2, c2) == cs
        q.put(cs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_sharding_matches_reference_file():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    cs = q.get(timeout=240)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    want = open(os.path.join(GOLD, "g2_3x256x256_t128_r4.j2k"), "rb").read()
    assert cs == want


def test_shard_tiles_partition():
    for n, w in ((256, 8), (7, 3), (1, 4)):
        seen = sorted(t for r in range(w) for t in D.shard_tiles(n, w, r))
        assert seen == list(range(n))
