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
import os
import time
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


def gather_frame(counts_host, offsets, lengths, arena, root, bufs=None, group=None):
    """Every rank's coded bytes + block table to `root`, exact sizes.

    counts_host : [[bytes used, table rows]] per rank, ON THE HOST (what exchange_counts() gathered)
    offsets     : int64[rows] / lengths: int32[rows] / arena: uint8[>= used] -- this rank's (device) tensors, e.g. views of
                  the encoder's own table and arena (grk_amd_table_device_ptr, grk_amd_coded_device_ptr)
    bufs        : root's receive storage from an earlier call (grown as needed) or None
    group       : the process group (= communicator) the transfers run on; gathers on different groups run side by side
    Returns on root ([(offsets, lengths, coded) per rank], bufs) -- the root's own part is referenced, not copied --,
    elsewhere (None, None)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [int(c[0]) for c in counts_host]
    rows = [int(c[1]) for c in counts_host]
    dev = arena.device
    if rank != root:
        ops = [dist.P2POp(dist.isend, arena[:sizes[rank]], root, group),
               dist.P2POp(dist.isend, offsets[:rows[rank]], root, group),
               dist.P2POp(dist.isend, lengths[:rows[rank]], root, group)]
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
        ops += [dist.P2POp(dist.irecv, cb, r, group), dist.P2POp(dist.irecv, co, r, group), dist.P2POp(dist.irecv, cl, r, group)]
        parts.append((co, cl, cb))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return parts, bufs


_gather_groups = []          # [0]: the counts' group, [1 + g]: gather slot g

# How long _issue_gather polls for a frame's counts before it blocks (GROK_AMD_GATHER_POLL_US).  The counts of frame f - lag are
# normally there when frame f is submitted; when the host runs ahead of the GPU they arrive within about a frame time.
_POLL_S = float(os.environ.get("GROK_AMD_GATHER_POLL_US", "50")) * 1e-6


def comm_priority():
    """Priority of the streams the exchange's own work is queued on (packing, the waits for the encoder's results): the
    encoder's level.  GROK_AMD_COMM_PRIORITY=-1 moves them to the high level (measured, world size 1 over RCCL, ms per 8K frame
    counts / gather: 0.440 / 0.439 instead of 0.423 / 0.433 on the default hardware queues -- profiles/r04_hw_queues.txt)."""
    return int(os.environ.get("GROK_AMD_COMM_PRIORITY", "0"))


def independent_stream(ctx, device, priority=None, tries=8):
    """A stream for work that WAITS for the encoder (the counts exchange, a gather): one whose kernels are dispatched side by side with
    the context's three streams (Context.streams_side_by_side) -- a stream with a wait at its head holds its dispatch pipe, and when
    that is the main stream's pipe the encode pipeline loses a third (profiles/r06_hw_queues.txt).  The hardware queue a stream gets
    depends on the streams made before it: candidates are made until one passes (the others are dropped afterwards); with
    GRK_AMD_STREAM_PROBE=0, or when none passes, the first one."""
    pr = comm_priority() if priority is None else priority
    first = torch.cuda.Stream(device=device, priority=pr)
    if os.environ.get("GRK_AMD_STREAM_PROBE", "1") == "0":
        return first
    try:
        ctx.probe_streams()
        mine = [ctx.internal_stream(i) for i in range(3)]
        cands = [first]
        for _ in range(tries):
            c = cands[-1]
            if all(m is None or ctx.streams_side_by_side(m, c.cuda_stream) for m in mine):
                return c
            cands.append(torch.cuda.Stream(device=device, priority=pr))
    except Exception:  # noqa: BLE001  (the probe is a convenience: without it the first stream, as before r06)
        pass
    return first


def comm_options():
    """Process-group options: RCCL's OWN streams -- the ones the transfers run on -- at high priority (GROK_AMD_RCCL_PRIORITY=0:
    the backend's default).  The HIP runtime keeps a pool of (by default 4) hardware queues per priority LEVEL, and kernels of
    streams that share a queue run one after the other: at the encoder's level every communicator's stream shares a queue
    with one of the encoder's three, and a transfer of most of a millisecond at the head of a queue holds back the K3 / DWT
    launches behind it.  At the other level the transfers have a pool of their own (and a transfer's few workgroups are placed
    before the coder's many).  At world size 1, where a "transfer" is a local copy, it makes no difference (0.423 / 0.438 against
    0.423 / 0.433 ms per frame): the reason is the N > 1 case, which has not been measured."""
    if int(os.environ.get("GROK_AMD_RCCL_PRIORITY", "-1")) >= 0:
        return None
    try:
        from torch.distributed import ProcessGroupNCCL
        return ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:      # noqa: BLE001  (a build without the NCCL backend: the CPU tests)
        return None


class FramePipeline:
    """The per-frame exchange of a sequence of frames, behind the encoder (see the module docstring), `depth` gathers in flight.

        pipe = FramePipeline(device, streams, depth=k, lag=l)
        per frame f:   encode ...;  pipe.submit(f, used, offsets, lengths, arena)     # queues the counts exchange of frame f
                                                                                       # and the gather of frame f - l
        at the end:    parts = pipe.flush()                                            # the gathers still to be issued

    The gather of frame f goes to the frame's writer, rank f mod R: consecutive frames' gathers have different writers, so a
    rank's sends of frames f, f + 1, ... leave over different xGMI links and the writers' ingress is spread over the node.  One
    gather moves a frame's coded bytes (~100 MB per rank for an 8K tile) over ONE link pair and takes several frame times; with
    `depth` = k of them in flight -- each on its own process group (communicator) and stream, frame f on slot f mod k -- the
    gathers run side by side and the exchange keeps up with an encoder that is k times faster than one link.  The counts (8 bytes
    per rank and frame, needed on the HOST to size the receives) travel on a group and stream of their own, so they never queue
    behind a gather.

    `streams`: (encode stream, comm stream) as torch.cuda streams -- further comm streams are made here --, or None on the CPU
    (gloo tests: everything in order).  The caller keeps a frame's tensors valid until the gather of that frame has been issued
    AND the stream that overwrites them next has waited for `pipe.done_event(frame)`: the gather of frame f is issued while frame
    f + lag is being submitted and may run until about frame f + lag + k, so an encoder that rotates k + lag + 1 buffer sets
    (grk_amd_set_pipelining(ctx, k + lag)) reuses f's set for frame f + k + lag + 1 and never waits.  The number of block-table rows per rank
    is a property of the tile geometry: it is exchanged with the first frame only; per frame the ranks exchange 8 bytes each (the
    bytes used in their coded arena), straight out of the encoder's own device word -- no kernel, no allocation, no host
    synchronisation on the submitting side."""

    def __init__(self, device, streams=None, depth=1, lag=1, ctx=None):
        self.dev = device
        self.streams = streams
        self.depth = max(1, int(depth))
        # the gather of frame f is issued while frame f + lag is being submitted.  lag = 1 makes the submitting host wait for the
        # counts of the frame it queued just before -- i.e. for that frame's encode to FINISH --, which keeps the device queue one
        # frame deep; with lag = 2 the counts are always there already and the host never waits (one more buffer set)
        self.lag = max(1, int(lag))
        self.completed = []            # (frame, parts or None, root) of the gathers issued since the last pop_completed()
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.rows = None               # block-table rows per rank
        self.pending = []              # [(frame, slot, offsets, lengths, arena)] submitted, gather not issued yet (at most lag)
        self.last_parts = None         # on the last frame's writer: [(offsets, lengths, coded)] per rank
        self.last_root = None
        self.gather_done = None        # event: the most recent gather has finished reading its source tensors
        cuda = device.type == "cuda"
        # one communicator per gather in flight (every rank creates the groups in the same order), one for the counts
        # (kept for the process's lifetime: a communicator is expensive to make, and every rank must make them in the same order)
        if self.depth > 1:
            while len(_gather_groups) < self.depth + 1:
                _gather_groups.append(dist.new_group(list(range(self.world)), pg_options=comm_options() if cuda else None))
            self.groups, self.ctrl_group = _gather_groups[1:self.depth + 1], _gather_groups[0]
        else:
            self.groups, self.ctrl_group = [None], None
        if cuda:
            # As FEW streams as the job allows: HIP multiplexes a process's streams onto four hardware queues, the encoder owns three
            # (main + two side streams), and every further stream shares a queue with one of them (measured at world 1, 8K frames:
            # counts-only exchange 0.425 ms per frame; gather depth 4 with the counts on a stream of their own 0.504, with the
            # counts on slot 0's stream 0.453, everything on one stream 0.448).  The counts therefore ride on slot 0's stream.
            self.gstreams = [streams[1]] + [(independent_stream(ctx, device) if ctx is not None else torch.cuda.Stream(device=device, priority=comm_priority()))
                                            for _ in range(self.depth - 1)]
            self.ctrl_stream = streams[1]
        else:
            self.gstreams = [None] * self.depth
            self.ctrl_stream = None
        # a writer's receive storage: one per frame "generation" (f mod (depth + lag)), so that the parts of every gather a
        # submit() or flush() issues stay readable until depth + lag frames later
        self.bufs = [None] * (self.depth + self.lag)
        self._buf_last = [None] * (self.depth + self.lag)      # per receive storage: the event of its last gather on this rank
        nslot = self.lag + 2                               # counts of the frames whose gather has not been issued yet
        self._host = [torch.empty(self.world, dtype=torch.int64).pin_memory() if cuda else torch.empty(self.world, dtype=torch.int64)
                      for _ in range(nslot)]
        self._dev = [torch.empty(self.world, dtype=torch.int64, device=device) for _ in range(nslot)]
        self._ready = [torch.cuda.Event() for _ in range(nslot)] if cuda else [None] * nslot
        self._done = {}                # frame -> event: the gather of that frame has read its source tensors
        self._enc = [torch.cuda.Event() for _ in range(self.lag + 2)] if cuda else []      # frame's results are there

    def _issue_gather(self, pending):
        f, cslot, offs, lens, arena, enc_ev = pending
        if self._ready[cslot] is not None:         # the counts of frame f are on the host (frame f + lag is already queued):
            ev = self._ready[cslot]                # polled for at most _POLL_S -- a blocking wait wakes up late, and the next frame's
            if not ev.query():                     # launches have to be queued while this one runs --, then a blocking wait: a peer
                t_end = time.perf_counter() + _POLL_S  # that is late or has failed must not leave this rank burning a host core
                while time.perf_counter() < t_end and not ev.query():     # (eight ranks share one host)
                    pass
                if not ev.query():
                    ev.synchronize()
        counts = [[int(v), self.rows[r]] for r, v in enumerate(self._host[cslot].tolist())]
        root = f % self.world
        g, b = f % self.depth, f % len(self.bufs)
        mine = self.bufs[b] if self.rank == root else None
        if self.streams is not None:
            before = self._buf_last[b]                         # the gather that used this receive storage last ON THIS RANK (it may
            if before is not None:                             # have run on another gather stream: frames b, b + len(bufs), ... share it,
                self.gstreams[g].wait_event(before)            # and this rank is the writer of only some of them)
            if enc_ev is not None:                             # the frame's encode: long finished (lag frames ago) -- a gather
                self.gstreams[g].wait_event(enc_ev)            # stream never sits waiting for work that is still to run
            with torch.cuda.stream(self.gstreams[g]):
                parts, bufs = gather_frame(counts, offs, lens, arena, root, mine, self.groups[g])
                ev = self._done.pop(f - 2 * (self.depth + self.lag + 2), None) or torch.cuda.Event()
                ev.record(self.gstreams[g])
                self._done[f] = ev
                self.gather_done = ev
                if self.rank == root:
                    self._buf_last[b] = ev
        else:
            parts, bufs = gather_frame(counts, offs, lens, arena, root, mine, self.groups[g])
        if bufs is not None:
            self.bufs[b] = bufs
        self.last_parts, self.last_root = parts, root
        self.completed.append((f, parts, root))
        # (a caller that never pops: entries older than the receive storage's rotation point at bytes that have been overwritten)
        if len(self.completed) > len(self.bufs):
            del self.completed[:len(self.completed) - len(self.bufs)]

    def submit(self, frame, used, offsets, lengths, arena, wait_results=None):
        """Queues the counts exchange of `frame` and issues the gather of the frame before it.
        used: int64[1] tensor on the device (the encoder's own word);  wait_results: callable(stream handle) that makes
        that stream wait for this frame's encode (grk_amd_stream_wait_results) -- CUDA only."""
        if self.rows is None:
            r = torch.tensor([offsets.numel()], dtype=torch.int64, device=self.dev)
            allr = torch.empty(self.world, dtype=torch.int64, device=self.dev)
            dist.all_gather_into_tensor(allr, r)
            self.rows = [int(v) for v in allr.cpu()]
        # first the gather of the frame before (its done-event must not sit behind anything that waits for THIS frame's
        # encode: the encoder's next frame waits for that event before it reuses the buffer set), then this frame's counts
        while len(self.pending) >= self.lag:
            self._issue_gather(self.pending.pop(0))
        cslot = frame % len(self._host)
        u = used.reshape(1)
        if u.dtype != torch.int64:
            u = u.to(torch.int64)
        enc_ev = None
        if self.streams is not None:
            # ONE stream waits for the frame's results as they are queued (the counts' stream, as in the counts-only exchange); the
            # gather streams wait for an event recorded behind that wait, and only when the gather is issued.  (A stream that
            # waits for queued work holds its hardware queue -- HIP multiplexes the streams onto a few -- and an encoder stream
            # behind it on the same queue then loses the overlap of consecutive frames: measured, 0.42 -> 0.55 ms per frame.)
            if wait_results is not None:
                wait_results(self.ctrl_stream.cuda_stream)
            enc_ev = self._enc[frame % len(self._enc)]
            enc_ev.record(self.ctrl_stream)
            with torch.cuda.stream(self.ctrl_stream):
                dist.all_gather_into_tensor(self._dev[cslot], u, group=self.ctrl_group)
                self._host[cslot].copy_(self._dev[cslot], non_blocking=True)
                self._ready[cslot].record(self.ctrl_stream)
        else:
            dist.all_gather_into_tensor(self._dev[cslot], u, group=self.ctrl_group)
            self._host[cslot].copy_(self._dev[cslot])
        self.pending.append((frame, cslot, offsets, lengths, arena, enc_ev))

    def pop_completed(self):
        """[(frame, parts on that frame's writer else None, writer rank)] of the gathers issued since the last call."""
        out, self.completed = self.completed, []
        return out

    def done_event(self, frame):
        """The event after which the tensors submitted for `frame` may be overwritten (None: nothing to wait for)."""
        return self._done.get(frame)

    def flush(self):
        """Issues the gather of the last submitted frame; returns (parts on that frame's writer else None, writer rank)."""
        while self.pending:
            self._issue_gather(self.pending.pop(0))
        if self.streams is not None:
            for st in self.gstreams:
                st.synchronize()
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
