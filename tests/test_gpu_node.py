"""-m gpu: one image over several GPUs natively (grk_amd_node_*, node.cpp) -- one context + one host thread per device entry,
tiles t -> device t mod R, ONE codestream.  A one-GPU box runs every code path with the device list {0, 0} (two contexts on one
GPU; three for the uneven split): the files equal the single-context file and Grok's own, for both forms of the exchange
(parallel writers / device-to-device gather on a rotating writer)."""
import hashlib
import os

import numpy as np
import pytest

import grok_amd as G
import gpuutil as U
import refharness as R
import synth

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not shipped")


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
@pytest.mark.parametrize("W,H,TW,TH,L,off,prec,irrev,flags", [
    (640, 512, 256, 256, 4, (0, 0), 8, False, 0),                                   # 3 x 2 tiles, ragged right column
    (700, 530, 200, 150, 3, (33, 17), 8, False, G.CS_TLM | G.CS_PLT),               # offsets, 1000-pitch-like tiling, markers
    (512, 384, 128, 128, 3, (0, 0), 12, True, 0),                                   # ICT + 9/7, 12 tiles
])
def test_node_image_equals_single_context_image(devices, W, H, TW, TH, L, off, prec, irrev, flags):
    px = synth.g2(3, H, W, prec, seed=W + len(devices))
    layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
    base = G.TileParams.make(1, 1, 3, prec, L, irreversible=irrev)
    want = U.ctx().encode_image(layout, base, px, flags)
    node = G.Node(devices)
    try:
        assert node.size == len(devices)
        assert bytes(node.encode_image(layout, base, px, flags)) == want
        # the gather form, three frames: the writer device rotates, the file does not change
        for _ in range(3):
            assert bytes(node.encode_image(layout, base, px, flags | G.NODE_GATHER)) == want
        # the image resident on the device: the workers cut their tiles out of it by 2-D device-to-device copies
        d_px = U.to_dev(px.reshape(-1).view(np.uint8))
        for fl in (flags, flags | G.NODE_GATHER):
            assert bytes(node.encode_image_device(layout, base, d_px.data_ptr(), d_px.numel(), 0, fl)) == want
    finally:
        node.close()


def test_node_gather_with_more_geometry_groups_than_buffer_sets():
    """Nine geometry groups (first / middle / last tile column and row all differ: an offset tiling with ragged ends) through the
    gather form: a worker's encodes rotate four buffer sets and the copies to the writer are not waited for group by group, so the
    fifth group must wait for the first one's bytes to have left -- the file equals the single-context one, frame after frame."""
    W, H, TW, TH, L = 1000, 900, 384, 320, 3
    px = synth.g2(3, H, W, 8, seed=9)
    layout = G.ImageLayout.make(W, H, TW, TH, offset=(100, 60))
    base = G.TileParams.make(1, 1, 3, 8, L)
    want = U.ctx().encode_image(layout, base, px, G.CS_TLM)
    for devices in ([0], [0, 0]):
        node = G.Node(devices)
        try:
            for _ in range(3):
                assert bytes(node.encode_image(layout, base, px, G.CS_TLM | G.NODE_GATHER)) == want
        finally:
            node.close()


@needs_ref
def test_node_cfg4_shape_equals_grk_compress():
    """BASELINE configs[3]'s shape at a quarter of its size: 4096 x 4096 as 16 tiles of 1024 x 1024 over two contexts, pixels
    in pinned host memory == the file Grok's CPU encoder writes."""
    W = H = 4096
    T, L = 1024, 5
    px = synth.g2(3, H, W, 8)
    R.lib(threads=os.cpu_count() or 1)
    want, _ = R.encode(px, 8, TW=T, TH=T, numres=L + 1, mode=1)
    node = G.Node([0, 0])
    try:
        hp = U.ctx().host_array(px.size).reshape(px.shape)
        hp[...] = px
        layout = G.ImageLayout.make(W, H, T, T)
        base = G.TileParams.make(1, 1, 3, 8, L)
        for fl in (0, G.NODE_GATHER):
            got = node.encode_image(layout, base, hp, fl)
            assert len(got) == len(want) and hashlib.md5(got).hexdigest() == hashlib.md5(want).hexdigest()
        del hp
    finally:
        node.close()


def test_host_pixels_pinned_and_pageable_give_the_same_blocks():
    """grk_amd_encode_tiles / fetch_coded with host pointers: pinned memory (grk_amd_host_alloc) crosses the link as it lies,
    pageable memory goes through the context's pinned chunks on four copy threads (> 16 MiB) -- same table, same bytes."""
    c = U.ctx()
    W = H = 3072                       # 28 MB of pixels: the staged path
    px = synth.g2(3, H, W, 8, seed=9)
    p = G.TileParams.make(W, H, 3, 8, 5)
    t0, c0 = c.encode_host(p, px)
    hp = c.host_array(px.size).reshape(px.shape)
    hp[...] = px
    t1, c1 = c.encode_host(p, hp)
    assert np.array_equal(t0["length"], t1["length"])
    got0, got1 = U.split_blocks(t0, c0), U.split_blocks(t1, c1)
    assert got0 == got1
    back = c.decode_host(p, t0, c0)
    assert np.array_equal(back[0], px)


def test_c_host_example_over_two_contexts(tmp_path):
    """tests/c/node_example.c (plain C99 + libgrok_amd.so) with the device list {0, 0}: both exchanges and the
    single-context encode give one file."""
    import subprocess
    from test_capi_host import _build_node_example
    exe = _build_node_example(tmp_path)
    r = subprocess.run([exe, "0", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "devices 2" in r.stdout and "identical" in r.stdout, r.stdout + r.stderr


def test_native_rccl_host_one_rank(tmp_path):
    """tests/c/rccl_host_example.cpp (C ABI + RCCL, one process per GPU) as the single rank a one-GPU box can
    run: five pipelined frames, counts all-gathered on the device, the gather to the writer, Tier-2 over the gathered tables --
    frame 0's codestream == the one a single context writes for the image."""
    import subprocess
    from test_capi_host import _build_rccl_host_example
    exe = _build_rccl_host_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ranks 1" in r.stdout and "identical" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
@pytest.mark.parametrize("pinned_out", [False, True])
def test_node_tier2_on_the_device_and_on_the_host_make_one_file(monkeypatch, devices, pinned_out):
    """Parallel writers make their tile-parts on the device by default (grk_amd_assemble_device); GRK_AMD_NODE_T2=host takes the
    host writer's plan instead.  Same file either way and == the single-context image, for one and for nine geometry groups, one and
    three workers, the output in pageable memory (staged through the workers' pinned buffers, or the copy threads for one worker
    and one group) and in pinned memory (every tile-part's DMA straight to its place)."""
    cases = [((640, 512, 256, 256, 4, (0, 0)), G.CS_TLM | G.CS_PLT | G.CS_PROG(2)),          # one to four groups, markers, RPCL
             ((1000, 900, 384, 320, 3, (100, 60)), G.CS_SOP | G.CS_EPH),                     # nine groups
             ((512, 512, 512, 512, 5, (0, 0)), 0)]                                           # one tile: the contiguous route
    c = U.ctx()
    for (W, H, TW, TH, L, off), flags in cases:
        px = synth.g2(3, H, W, 8, seed=W)
        layout = G.ImageLayout.make(W, H, TW, TH, offset=off)
        base = G.TileParams.make(1, 1, 3, 8, L)
        want = c.encode_image(layout, base, px, flags)
        out = c.host_array(px.size * 2 + (1 << 20)) if pinned_out else None
        d_px = U.to_dev(px.reshape(-1).view(np.uint8))
        for route in ("device", "host"):
            monkeypatch.setenv("GRK_AMD_NODE_T2", route)
            node = G.Node(devices)
            try:
                for _ in range(2):
                    assert bytes(node.encode_image(layout, base, px, flags, out=out)) == want, route
                assert bytes(node.encode_image_device(layout, base, d_px.data_ptr(), d_px.numel(), 0, flags, out=out)) == want, route
            finally:
                node.close()
