"""Tile sharding across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Tiles of a JPEG 2000 image are independent, so the data path needs no collective:
rank r encodes tiles {t : t mod R == r}.  The only exchanges are
  * broadcast of the coding-parameter blob from rank 0 (a few bytes, once), and
  * the gather of the coded tile-parts to the rank that writes the codestream:
    all_gather of byte counts, then gather of the (padded) coded arenas.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from .capi import CODED_DTYPE, TileParams


def shard_tiles(ntiles, world, rank):
    """Tile indices owned by `rank` (round-robin keeps edge tiles spread over ranks)."""
    return list(range(rank, ntiles, world))


def broadcast_params(params, device, src=0):
    """Rank `src`'s TileParams to everybody (the 'header blob' of the north star)."""
    n = C.sizeof(TileParams)
    if dist.get_rank() == src:
        t = torch.tensor(list(bytes(params)), dtype=torch.uint8, device=device)
    else:
        t = torch.zeros(n, dtype=torch.uint8, device=device)
    dist.broadcast(t, src)
    return TileParams.from_buffer_copy(bytes(t.cpu().numpy().tobytes()))


def gather_tile_parts(table, coded, device, dst=0, scratch=None):
    """Gather this rank's coded blocks on rank `dst`.

    table : numpy CODED_DTYPE rows of this rank's blocks (offsets relative to `coded`)
    coded : 1-D uint8 torch tensor on `device` holding the coded bytes (may be longer than needed)
    returns on dst: list over ranks of (table, coded uint8 tensor); elsewhere None.
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    used = int((table["offset"] + table["length"]).max()) if len(table) else 0
    meta = torch.tensor([used, len(table)], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    sizes = [int(m[0].item()) for m in metas]
    nrows = [int(m[1].item()) for m in metas]
    pad = (max(sizes) + 4095) & ~4095
    maxrows = max(nrows)
    # block tables travel as int64 triples (offset, length, 0)
    tab = torch.zeros(maxrows * 2, dtype=torch.int64, device=device)
    if len(table):
        tab[0:2 * len(table):2] = torch.from_numpy(table["offset"].astype(np.int64)).to(device)
        tab[1:2 * len(table):2] = torch.from_numpy(table["length"].astype(np.int64)).to(device)
    if coded.numel() < pad:
        buf = torch.zeros(pad, dtype=torch.uint8, device=device)
        buf[:coded.numel()] = coded
    else:
        buf = coded[:pad]
    if rank == dst:
        if scratch is None or scratch[0].numel() < pad:
            scratch = [torch.empty(pad, dtype=torch.uint8, device=device) for _ in range(world)]
        bufs = [s[:pad] for s in scratch]
        tabs = [torch.empty_like(tab) for _ in range(world)]
        dist.gather(buf, bufs, dst=dst)
        dist.gather(tab, tabs, dst=dst)
        out = []
        for r in range(world):
            t = np.zeros(nrows[r], CODED_DTYPE)
            tt = tabs[r].cpu().numpy()
            t["offset"] = tt[0:2 * nrows[r]:2]
            t["length"] = tt[1:2 * nrows[r]:2]
            out.append((t, bufs[r][:sizes[r]]))
        return out, scratch
    dist.gather(buf, None, dst=dst)
    dist.gather(tab, None, dst=dst)
    return None, scratch


def gather_tile_parts_device(used, offsets, lengths, arena, dst=0, scratch=None):
    """The same exchange with everything already on the device and ONE host synchronisation (the byte counts,
    which size the transfer): `used` int64[1] bytes used in `arena` (uint8[>= used]), `offsets` int64[n],
    `lengths` int32[n] -- views of the encoder's own device table (grk_amd_table_device_ptr).
    Returns on dst ([(offsets, lengths, coded uint8 tensor) per rank], scratch), elsewhere (None, scratch);
    parts_to_numpy() turns them into what merge_tile_parts() takes."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = arena.device
    meta = torch.cat([used.reshape(1).to(torch.int64), torch.tensor([offsets.numel()], dtype=torch.int64, device=device)])
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    m = torch.stack(metas).cpu()                       # the one synchronisation
    sizes, nrows = [int(v) for v in m[:, 0]], [int(v) for v in m[:, 1]]
    pad = (max(sizes) + 4095) & ~4095
    maxrows = max(nrows)
    if arena.numel() < pad:
        buf = torch.zeros(pad, dtype=torch.uint8, device=device)
        buf[:arena.numel()] = arena
    else:
        buf = arena[:pad]
    if offsets.numel() < maxrows:
        offsets = torch.cat([offsets, offsets.new_zeros(maxrows - offsets.numel())])
        lengths = torch.cat([lengths, lengths.new_zeros(maxrows - lengths.numel())])
    if rank == dst:
        if scratch is None or scratch[0][0].numel() < pad or scratch[1][0].numel() != maxrows:
            scratch = ([torch.empty(pad, dtype=torch.uint8, device=device) for _ in range(world)],
                       [torch.empty(maxrows, dtype=torch.int64, device=device) for _ in range(world)],
                       [torch.empty(maxrows, dtype=torch.int32, device=device) for _ in range(world)])
        bufs = [s[:pad] for s in scratch[0]]
        dist.gather(buf, bufs, dst=dst)
        dist.gather(offsets, scratch[1], dst=dst)
        dist.gather(lengths, scratch[2], dst=dst)
        return [(scratch[1][r][:nrows[r]], scratch[2][r][:nrows[r]], bufs[r][:sizes[r]]) for r in range(world)], scratch
    dist.gather(buf, None, dst=dst)
    dist.gather(offsets, None, dst=dst)
    dist.gather(lengths, None, dst=dst)
    return None, scratch


def exchange_tile_part_offsets(used, counts=None):
    """What a parallel codestream writer needs per step, and all it needs: every rank's coded byte count, from which
    each rank knows where its tile-parts start in the file (exclusive prefix sum; headers are a host-side constant).
    One all_gather of 8 bytes per rank over RCCL, enqueued on the current stream behind the encode -- no host
    synchronisation, no coded byte leaves its GPU (each rank writes its own tile-parts at its offset; funnelling
    N x ~100 MB per step into one GPU would bound the job by that GPU's xGMI ingress instead).
    used: int64[1] device tensor (grk_amd_table_device_ptr(ctx, 2)).  Returns (counts int64[world], my offset int64[1])."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if counts is None:
        counts = torch.zeros(world, dtype=torch.int64, device=used.device)
    dist.all_gather_into_tensor(counts, used.reshape(1).to(torch.int64))
    starts = torch.cumsum(counts, 0) - counts
    return counts, starts[rank:rank + 1]


def parts_to_numpy(parts):
    out = []
    for offs, lens, coded in parts:
        t = np.zeros(offs.numel(), CODED_DTYPE)
        t["offset"] = offs.cpu().numpy().astype(np.uint64)
        t["length"] = lens.cpu().numpy().astype(np.uint32)
        out.append((t, coded))
    return out


def merge_tile_parts(parts, ntiles, blocks_per_tile):
    """Rank-major parts (round-robin tile ownership) -> one tile-ordered table + one byte buffer."""
    world = len(parts)
    table = np.zeros(ntiles * blocks_per_tile, CODED_DTYPE)
    chunks, base = [], 0
    for r, (t, coded) in enumerate(parts):
        c = coded.cpu().numpy() if isinstance(coded, torch.Tensor) else np.asarray(coded)
        for k, tile in enumerate(range(r, ntiles, world)):
            rows = t[k * blocks_per_tile:(k + 1) * blocks_per_tile]
            dstrows = table[tile * blocks_per_tile:(tile + 1) * blocks_per_tile]
            dstrows["length"] = rows["length"]
            dstrows["offset"] = rows["offset"] + base
        chunks.append(c)
        base += len(c)
    return table, (np.concatenate(chunks) if chunks else np.zeros(0, np.uint8))
