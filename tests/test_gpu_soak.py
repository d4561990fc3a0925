"""-m gpu: longer runs of the two sequence modes (a few seconds each): what a race between frames in flight would need in order to
show -- many frames, different content from frame to frame, results looked at as LATE as the ring of buffer sets allows."""
import hashlib

import numpy as np
import pytest
import torch

import grok_amd as G
import gpuutil as U
import synth

pytestmark = pytest.mark.gpu


def _dev_view(ptr, n, typestr):
    class _H:
        pass
    h = _H()
    h.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(h, device="cuda")


@pytest.mark.parametrize("depth", [1, 3, 7])
def test_many_pipelined_encodes_every_frame_checked_late(depth):
    """240 pipelined encodes of 6 different images over `depth` + 1 buffer sets in rotation, no host fetch in between: frame k's
    device-resident coded bytes + block table are hashed when frame k + depth has been ISSUED (the last moment its buffer set
    is guaranteed untouched) and must equal a plain encode's of the same image."""
    C, H, W, L = 3, 512, 640, 4
    p = G.TileParams.make(W, H, C, 8, L)
    nb = G.lib().grk_amd_tile_num_blocks(p)
    imgs = [synth.g2(C, H, W, 8, seed=s) for s in (1, 2, 3)] + [(255 - synth.g2(C, H, W, 8, seed=4)).astype(np.uint8),
                                                                 np.ascontiguousarray(synth.g2(C, H, W, 8, seed=5)[:, ::-1, :]),
                                                                 np.zeros((C, H, W), np.uint8)]
    want = []
    for im in imgs:
        t, coded = U.ctx().encode_host(p, im)
        want.append(hashlib.md5(b"".join(U.split_blocks(t, coded))).hexdigest())
    c = G.Context(0)
    c.set_pipelining(depth)
    d = [U.to_dev(im.reshape(-1)) for im in imgs]
    held = []
    N = 240

    def check(k):
        arena_p, off_p, len_p, used_p = held[k]
        used = int(_dev_view(used_p, 1, "<i8").cpu()[0])
        offs = _dev_view(off_p, nb, "<i8").cpu().numpy()
        lens = _dev_view(len_p, nb, "<i4").cpu().numpy()
        arena = _dev_view(arena_p, max(used, 1), "|u1").cpu().numpy()
        got = hashlib.md5(b"".join(bytes(arena[int(o):int(o) + int(l)]) for o, l in zip(offs, lens))).hexdigest()
        assert got == want[k % len(imgs)], "frame %d" % k

    for k in range(N):
        c.encode_tiles(p, 1, d[k % len(imgs)].data_ptr(), True, fetch=False)
        held.append((c.coded_device_ptr(), c.table_device_ptr(0), c.table_device_ptr(1), c.table_device_ptr(2)))
        if k >= depth and (k % 7 == 0 or k + 1 == N):      # (a check synchronises: most frames run back to back)
            torch.cuda.synchronize()
            check(k - depth)
    torch.cuda.synchronize()
    for k in range(N - depth, N):
        check(k)
    c.set_pipelining(False)
    c.close()


@pytest.mark.parametrize("flights", [2, 5])
def test_long_decode_sequence_every_frame_equals_its_source(flights):
    """150 HT frames of 5 different images through one context's decode sequence (`flights` in flight), each into a buffer of its
    own generation: every output equals its source image, the status stays clean."""
    C, H, W, L = 3, 256, 320, 3
    p = G.TileParams.make(W, H, C, 8, L)
    enc = G.Context(0)
    frames = []
    for s in range(5):
        px = synth.g2(C, H, W, 8, seed=20 + s)
        table, coded = enc.encode_host(p, px)
        frames.append((px, table, U.to_dev(np.frombuffer(bytes(coded), np.uint8).copy())))
    ring = 2 * flights + 1
    outs = [torch.zeros(C * H * W, dtype=torch.uint8, device="cuda") for _ in range(ring)]
    torch.cuda.synchronize()
    c = G.Context(0)
    c.set_decode_pipelining(flights)
    try:
        last = {}
        for k in range(150):
            px, table, d_c = frames[(k * 3) % 5]
            slot = k % ring
            if slot in last and k % 11 == 0:               # look at what this slot got a full ring ago before it is overwritten
                c.synchronize()
                assert np.array_equal(outs[slot].cpu().numpy().reshape(C, H, W), last[slot]), "frame %d" % (k - ring)
            c.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), outs[slot].data_ptr())
            last[slot] = px
        c.synchronize()
        c.decode_status()
        for slot, px in last.items():
            assert np.array_equal(outs[slot].cpu().numpy().reshape(C, H, W), px), "slot %d" % slot
    finally:
        c.set_decode_pipelining(0)


def test_two_host_threads_with_a_context_each():
    """Contexts are independent: two host threads, each with a context of its own on the one device, encode and decode different
    images at the same time (ctypes calls release the GIL) -- every result equals the single-threaded one."""
    import threading
    C, H, W, L = 3, 384, 512, 4
    p = G.TileParams.make(W, H, C, 8, L)
    imgs = [synth.g2(C, H, W, 8, seed=70 + i) for i in range(4)]
    want = []
    for im in imgs:
        t, coded = U.ctx().encode_host(p, im)
        want.append(U.split_blocks(t, coded))
    errors = []

    def worker(tid):
        try:
            c = G.Context(0)
            for rep in range(12):
                k = (tid + rep) % len(imgs)
                t, coded = c.encode_host(p, imgs[k])
                if U.split_blocks(t, coded) != want[k]:
                    errors.append("thread %d rep %d: encode differs" % (tid, rep))
                back = c.decode_host(p, t, coded)
                if not np.array_equal(np.asarray(back).reshape(imgs[k].shape), imgs[k]):
                    errors.append("thread %d rep %d: decode differs" % (tid, rep))
            c.close()
        except Exception as e:      # noqa: BLE001
            errors.append("thread %d: %r" % (tid, e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
