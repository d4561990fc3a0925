"""-m gpu: the one-block-per-lane Part-1 decoder (K8L: t1_lanes_kernel + t1_recon_kernel, kernels_t1lanes.hip) on the GPU.
The small Part-1 tests of test_gpu_decode.py have too few blocks for it (a call routes blocks to it only in groups of >= 64 with
equal bit-plane / pass counts): these tiles have hundreds of blocks of every edge shape per group, are decoded by both routes --
lanes (pass-synchronous and free-running) and one block per wave -- and compared with the oracle / the reference's T1."""
import os

import numpy as np
import pytest
import torch

import grok_amd as G
import oracle as O
import chain
import gpuutil as U
import refharness as R

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not shipped")]


def _ctx_with(env):
    """A context of its own created under `env` (the Part-1 routing knobs are read when a context is created)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return G.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _tile(rng, W, H, L, bits, prec, keep_fewer=0, irrev=False):
    p = G.TileParams.make(W, H, 1, prec, L, part1=True, irreversible=irrev, mct=False)
    blocks, _ = G.tile_layout(p)
    table = np.zeros(len(blocks), G.capi.CODED_DTYPE)
    chunks, off, want = [], 3, np.zeros((H, W), np.int32)          # (the buffer starts off a dword boundary)
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        mag = rng.integers(1 << (bits - 1), 1 << bits, size=(bh, bw))   # every block: the same number of bit-planes
        mag = np.where(rng.random((bh, bw)) < 0.6, mag >> rng.integers(0, bits + 1, size=(bh, bw)), mag)
        mag[0, 0] = (1 << bits) - 1
        coef = (mag * np.where(rng.random((bh, bw)) < 0.5, -1, 1)).astype(np.int32)
        cb, npass, nbps = R.t1_encode_block(coef, b.band)
        keep = max(1, npass - keep_fewer)
        table["offset"][i] = off; table["length"][i] = len(cb); table["missing_msbs"][i] = nbps | (keep << 8)
        chunks.append(cb + b"\xEE" * (i % 5)); off += len(chunks[-1])
        want[b.py:b.py + bh, b.px:b.px + bw] = O.t1_decode_block(cb, keep, nbps, b.band, bw, bh)
    return p, blocks, table, b"\xEE" * 3 + b"".join(chunks), want


@pytest.mark.parametrize("W,H,L,bits,keep_fewer", [(1000, 520, 1, 5, 0), (777, 650, 2, 3, 0), (1024, 512, 1, 7, 4), (640, 640, 0, 9, 0),
                                                     (900, 333, 1, 2, 1)])
def test_lane_decoder_equals_oracle_on_every_block_shape(W, H, L, bits, keep_fewer):
    rng = np.random.default_rng(W + H + bits)
    p, blocks, table, coded, want = _tile(rng, W, H, L, bits, 12, keep_fewer)
    groups = {}
    for i, b in enumerate(blocks):
        if b.y1 - b.y0 >= 9:
            groups[int(table["missing_msbs"][i])] = groups.get(int(table["missing_msbs"][i]), 0) + 1
    assert max(groups.values()) >= 64, "the case must reach the lane decoder"
    d_c = U.to_dev(np.frombuffer(coded, np.uint8))
    outs = {}
    # (GRK_AMD_T1_LANES=2: the lane decoder wherever it can be used -- by default a call this small goes to the wave decoder alone)
    for name, env in (("lanes", {"GRK_AMD_T1_LANES": "2"}), ("free", {"GRK_AMD_T1_LANES": "2", "GRK_AMD_T1_SYNC": "0"}),
                      ("waves", {"GRK_AMD_T1_LANES": "0"}), ("auto", {})):
        c = _ctx_with(env)
        d_m = U.dev_planes(p, 1)
        c.stage_ht_decode(p, 1, table, d_c.data_ptr(), d_c.numel(), d_m.data_ptr())
        c.synchronize()
        outs[name] = U.planes_to_numpy(d_m, p, 1)[0].copy()
        del c
    half = np.where(want < 0, -((-want) // 2), want // 2)          # ShiftFilter: v / 2 toward zero
    for name, got in outs.items():
        assert np.array_equal(got, half), "%s: %d samples differ" % (name, int((got != half).sum()))


def test_lane_decoder_irreversible_scale():
    rng = np.random.default_rng(3)
    p, blocks, table, coded, want = _tile(rng, 960, 448, 1, 6, 12, 0, irrev=True)
    c = _ctx_with({"GRK_AMD_T1_LANES": "2"})
    c.set_decode_qcd([])
    d_c = U.to_dev(np.frombuffer(coded, np.uint8))
    d_m = U.dev_planes(p, 1)
    c.stage_ht_decode(p, 1, table, d_c.data_ptr(), d_c.numel(), d_m.data_ptr())
    c.synchronize()
    got = U.planes_to_numpy(d_m, p, 1)[0].view(np.float32)
    cw = _ctx_with({"GRK_AMD_T1_LANES": "0"})
    d_m2 = U.dev_planes(p, 1)
    cw.stage_ht_decode(p, 1, table, d_c.data_ptr(), d_c.numel(), d_m2.data_ptr())
    cw.synchronize()
    ref = U.planes_to_numpy(d_m2, p, 1)[0].view(np.float32)
    assert np.array_equal(got.view(np.int32), ref.view(np.int32))            # bit-identical to the wave decoder (checked vs the oracle elsewhere)
    _, qcd = G.tile_layout(p)
    for b in blocks[:40]:                                                    # ... and == the oracle's ScaleFilter on the oracle's integers
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        wq = qcd[chain.band_index(b)]
        step = np.float32((1.0 + (wq & 0x7FF) / 2048.0) * 2.0 ** (12 - (wq >> 11)))
        w = O.t1_dequant_irrev(want[b.py:b.py + bh, b.px:b.px + bw], step)
        assert np.array_equal(got[b.py:b.py + bh, b.px:b.px + bw].view(np.int32), w.view(np.int32))
