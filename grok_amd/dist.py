"""Tile sharding across the GPUs of a node (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Tiles of a JPEG 2000 image are independent (the reference runs them as independent tasks and
writes their tile-parts in index order, codestream/CodeStreamCompress.cpp:535-603), so the data path needs no
collective: rank r encodes tiles {t : t mod R == r}.  The exchanges are
  * broadcast of the coding-parameter blob from rank 0 (a few bytes, once), and
  * per frame, what it takes to make ONE codestream of the ranks' tile-parts:
      exchange_counts   -- all_gather of (bytes used in the coded arena, rows of the block table) per rank.  These are
                           ARENA EXTENTS (they include the allocator's 16-byte alignment of every block and exclude packet
                           headers): they size the transfer below; a tile-part's place in the FILE is known once the
                           writer has run Tier-2 over the gathered block lengths (grk_amd_tile_part_bytes / merge_tile_parts).
      gather_frame      -- every rank's coded bytes and block table to the frame's writer rank, EXACT sizes (grouped
                           send / recv, no padding to the largest rank).  The writer rotates with the frame number
                           (frame f -> rank f mod R): funnelling every frame into one GPU would bound the job by that
                           GPU's xGMI ingress (7 links), rotating spreads the same traffic over all of them.
    The receive sizes have to be known on the host, so a frame's gather is issued one frame late (FramePipeline): the
    counts travel device -> pinned host memory asynchronously behind the encode, and are read when the next frame has
    already been queued -- the GPU never waits for the host.
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


def exchange_counts(used, nrows, out=None):
    """all_gather of this rank's (arena bytes used, block-table rows): int64[world, 2] on the device, no host
    synchronisation.  `used`: int64[1] device tensor (grk_amd_table_device_ptr(ctx, 2)) or an int; `nrows`: int."""
    world = dist.get_world_size()
    dev = used.device if isinstance(used, torch.Tensor) else torch.device("cpu")
    mine = torch.empty(2, dtype=torch.int64, device=dev)
    mine[0:1] = used.reshape(1).to(torch.int64) if isinstance(used, torch.Tensor) else int(used)
    mine[1] = int(nrows)
    if out is None:
        out = torch.empty(world * 2, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, mine)
    return out.view(world, 2)


def gather_frame(counts_host, offsets, lengths, arena, root, bufs=None):
    """Every rank's coded bytes + block table to `root`, exact sizes.

    counts_host : [[bytes used, table rows]] per rank, ON THE HOST (what exchange_counts() gathered)
    offsets     : int64[rows] / lengths: int32[rows] / arena: uint8[>= used] -- this rank's (device) tensors, e.g. views of
                  the encoder's own table and arena (grk_amd_table_device_ptr, grk_amd_coded_device_ptr)
    bufs        : root's receive storage from an earlier call (grown as needed) or None
    Returns on root ([(offsets, lengths, coded) per rank], bufs) -- the root's own part is referenced, not copied --,
    elsewhere (None, None)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [int(c[0]) for c in counts_host]
    rows = [int(c[1]) for c in counts_host]
    dev = arena.device
    if rank != root:
        ops = [dist.P2POp(dist.isend, arena[:sizes[rank]], root),
               dist.P2POp(dist.isend, offsets[:rows[rank]], root),
               dist.P2POp(dist.isend, lengths[:rows[rank]], root)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return None, None
    need_b = sum(s for r, s in enumerate(sizes) if r != root)
    need_r = sum(n for r, n in enumerate(rows) if r != root)
    if bufs is None or bufs[0].numel() < need_b or bufs[1].numel() < need_r:
        bufs = (torch.empty(max(need_b, 1), dtype=torch.uint8, device=dev),
                torch.empty(max(need_r, 1), dtype=torch.int64, device=dev),
                torch.empty(max(need_r, 1), dtype=torch.int32, device=dev))
    parts, ops, ob, orow = [], [], 0, 0
    for r in range(world):
        if r == root:
            parts.append((offsets[:rows[r]], lengths[:rows[r]], arena[:sizes[r]]))
            continue
        cb, co, cl = bufs[0][ob:ob + sizes[r]], bufs[1][orow:orow + rows[r]], bufs[2][orow:orow + rows[r]]
        ob += sizes[r]
        orow += rows[r]
        ops += [dist.P2POp(dist.irecv, cb, r), dist.P2POp(dist.irecv, co, r), dist.P2POp(dist.irecv, cl, r)]
        parts.append((co, cl, cb))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return parts, bufs


class FramePipeline:
    """The per-frame exchange of a sequence of frames, one frame behind the encoder (see the module docstring).

        pipe = FramePipeline(device)
        per frame f:   encode ...;  pipe.submit(f, used, offsets, lengths, arena)     # queues the counts exchange of frame f
                                                                                       # and the gather of frame f - 1
        at the end:    parts = pipe.flush()                                            # gather of the last frame

    `streams`: (encode stream, comm stream) as torch.cuda streams, or None on the CPU (gloo tests: everything in order).
    The caller keeps a frame's tensors valid until the gather of that frame has been issued AND the stream that overwrites
    them next has waited for `pipe.done_event(frame)`.  The gather of frame f is issued while frame f + 1 is being submitted,
    after a host wait for f's byte counts: an encoder that rotates THREE buffer sets (grk_amd_set_pipelining(ctx, 2)) reuses
    f's set for frame f + 3 and so never waits for that host round trip; with two sets it would sit between frames.  The number of block-table rows per rank is a property of the tile geometry: it is exchanged with the
    first frame only; per frame the ranks exchange 8 bytes each (the bytes used in their coded arena), straight out of
    the encoder's own device word -- no kernel, no allocation, no host synchronisation on the submitting side."""

    def __init__(self, device, streams=None):
        self.dev = device
        self.streams = streams
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.rows = None               # block-table rows per rank
        self.pending = None            # (frame, slot, offsets, lengths, arena)
        self.bufs = None
        self.last_parts = None         # on the last frame's writer: [(offsets, lengths, coded)] per rank
        self.last_root = None
        self.gather_done = None        # event: the most recent gather has finished reading its source tensors
        cuda = device.type == "cuda"
        self._host = [torch.empty(self.world, dtype=torch.int64).pin_memory() if cuda else torch.empty(self.world, dtype=torch.int64)
                      for _ in range(2)]
        self._dev = [torch.empty(self.world, dtype=torch.int64, device=device) for _ in range(2)]
        self._ready = [torch.cuda.Event() for _ in range(2)] if cuda else [None, None]
        self._done = {}                # frame -> event: the gather of that frame has read its source tensors


    def _issue_gather(self, pending):
        f, slot, offs, lens, arena = pending
        if self._ready[slot] is not None:          # the counts of frame f are on the host (frame f + 1 is already queued):
            ev = self._ready[slot]                 # polled briefly -- a blocking wait wakes up late, and the next frame's
            spins = 0                              # launches have to be queued while this one runs -- then a blocking wait
            while spins < 20000 and not ev.query():    # (a peer that is late or has failed must not leave this rank
                spins += 1                             # burning a core forever: the collective's own timeout applies)
            if not ev.query():
                ev.synchronize()
        counts = [[int(v), self.rows[r]] for r, v in enumerate(self._host[slot].tolist())]
        root = f % self.world
        mine = self.bufs if self.rank == root else None
        if self.streams is not None:
            with torch.cuda.stream(self.streams[1]):
                parts, bufs = gather_frame(counts, offs, lens, arena, root, mine)
                ev = self._done.pop(f - 4, None) or torch.cuda.Event()
                ev.record(self.streams[1])
                self._done[f] = ev
                self.gather_done = ev
        else:
            parts, bufs = gather_frame(counts, offs, lens, arena, root, mine)
        if bufs is not None:
            self.bufs = bufs
        self.last_parts, self.last_root = parts, root

    def submit(self, frame, used, offsets, lengths, arena, wait_results=None):
        """Queues the counts exchange of `frame` and issues the gather of the frame before it.
        used: int64[1] tensor on the device (the encoder's own word);  wait_results: callable(comm stream handle) that makes
        the comm stream wait for this frame's encode (grk_amd_stream_wait_results) -- CUDA only."""
        if self.rows is None:
            r = torch.tensor([offsets.numel()], dtype=torch.int64, device=self.dev)
            allr = torch.empty(self.world, dtype=torch.int64, device=self.dev)
            dist.all_gather_into_tensor(allr, r)
            self.rows = [int(v) for v in allr.cpu()]
        # first the gather of the frame before (its done-event must not sit behind anything that waits for THIS frame's
        # encode: the encoder's next frame waits for that event before it reuses the buffer set), then this frame's counts
        prev, self.pending = self.pending, None
        if prev is not None:
            self._issue_gather(prev)
        slot = frame & 1
        u = used.reshape(1)
        if u.dtype != torch.int64:
            u = u.to(torch.int64)
        if self.streams is not None:
            comm = self.streams[1]
            if wait_results is not None:
                wait_results(comm.cuda_stream)
            with torch.cuda.stream(comm):
                dist.all_gather_into_tensor(self._dev[slot], u)
                self._host[slot].copy_(self._dev[slot], non_blocking=True)
                self._ready[slot].record(comm)
        else:
            dist.all_gather_into_tensor(self._dev[slot], u)
            self._host[slot].copy_(self._dev[slot])
        self.pending = (frame, slot, offsets, lengths, arena)

    def done_event(self, frame):
        """The event after which the tensors submitted for `frame` may be overwritten (None: nothing to wait for)."""
        return self._done.get(frame)

    def flush(self):
        """Issues the gather of the last submitted frame; returns (parts on that frame's writer else None, writer rank)."""
        if self.pending is not None:
            self._issue_gather(self.pending)
            self.pending = None
        if self.streams is not None:
            self.streams[1].synchronize()
        return self.last_parts, self.last_root


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
