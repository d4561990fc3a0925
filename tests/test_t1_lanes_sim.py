"""CPU: the lane logic of the one-block-per-lane Part-1 decoder (grok_amd/csrc/t1_lanes.h) stepped through the kernel's
phases by tests/c/t1_lanes_sim.cpp (64 lanes to a "wave", compiled for the host) against the EBCOT oracle -- the state machine
(passes, stripe hand-over through the block's work area, byte stream refills, plane bitmaps + reconstruction) without a GPU.
The HIP kernel that runs the same header is checked by the -m gpu tests (tests/test_gpu_decode.py, test_gpu_t1_lanes.py)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O
import refharness as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SB = np.dtype([("offset", "<u8"), ("length", "<u4"), ("numbps", "<u4"), ("numpasses", "<u4"), ("w", "<u4"), ("h", "<u4"),
               ("orient", "<u4")], align=True)
_lib = None


def sim():
    global _lib
    if _lib is None:
        out = os.path.join(ROOT, "build", "libt1lsim.so")
        src = os.path.join(ROOT, "tests", "c", "t1_lanes_sim.cpp")
        hdr = os.path.join(ROOT, "grok_amd", "csrc", "t1_lanes.h")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src])
        _lib = C.CDLL(out)
        _lib.t1l_sim_decode.restype = C.c_int
        _lib.t1l_sim_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def decode(blocks, pad_front=0):
    """blocks: [(coded bytes, numpasses, numbps, orient, w, h)] -> [values h x w] through the lane simulator"""
    buf = bytearray(b"\xA5" * pad_front)
    rows = np.zeros(len(blocks), SB)
    for i, (cb, npass, nbps, orient, w, h) in enumerate(blocks):
        rows[i] = (len(buf), len(cb), nbps, npass, w, h, orient)
        buf += cb
        buf += b"\x5A" * (i % 3)                       # blocks at every alignment
    coded = np.frombuffer(bytes(buf), np.uint8)
    out = np.zeros((len(blocks), 64, 64), np.int32)
    st = np.zeros(8, np.uint64)
    rc = sim().t1l_sim_decode(coded.ctypes.data, coded.size, len(blocks), rows.ctypes.data, out.ctypes.data, st.ctypes.data)
    assert rc == 0
    return [out[i, :b[5], :b[4]] for i, b in enumerate(blocks)], st


def _block(rng, w, h, bits, mode):
    mag = rng.integers(0, 1 << bits, size=(h, w))
    if mode == 1:
        mag = mag >> rng.integers(0, bits + 1, size=(h, w))
    elif mode == 2:
        mag = np.where(rng.random((h, w)) < 0.95, 0, mag & 7)
    elif mode == 3:
        mag = np.zeros((h, w), np.int64); mag[h // 2, w // 3] = 5
    elif mode == 4:
        mag = np.full((h, w), (1 << bits) - 1)
    sign = np.where(rng.random((h, w)) < 0.5, -1, 1)
    return (mag * sign).astype(np.int32)


needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")
SHAPES = [(64, 64, 8), (64, 64, 12), (32, 32, 10), (37, 9, 8), (1, 12, 5), (5, 64, 9), (64, 10, 7), (2, 17, 3), (63, 31, 13),
          (4, 12, 1), (13, 11, 14), (64, 61, 6)]


@needs_ref
def test_a_wave_of_mixed_blocks_equals_the_oracle_and_the_reference():
    """64+ blocks of every shape / orientation / content mode side by side in one wave: every lane in another pass and stripe."""
    rng = np.random.default_rng(2026)
    blocks, coefs = [], []
    for k in range(150):
        w, h, bits = SHAPES[k % len(SHAPES)]
        orient, mode = k % 4, (k // 4) % 5
        coef = _block(rng, w, h, bits, mode)
        cb, npass, nbps = R.t1_encode_block(coef, orient)
        if nbps == 0 or npass == 0 or nbps > 14:
            continue
        blocks.append((cb, npass, nbps, orient, w, h)); coefs.append(coef)
    got, st = decode(blocks, pad_front=5)
    for g, b, coef in zip(got, blocks, coefs):
        want = O.t1_decode_block(*b)
        assert np.array_equal(g, want), "block %dx%d bps %d" % (b[4], b[5], b[2])
        assert np.array_equal(g, R.t1_decode_block(b[0], b[1], b[2], b[3], b[4], b[5]))
        assert np.array_equal(O.t1_dequant_rev(g), coef)
    assert st[1] > 0 and st[0] * 64 >= st[1]


@needs_ref
@pytest.mark.parametrize("keep", [1, 2, 3, 4, 5, 6, 7, 11, 12, 13])
def test_truncated_pass_sequences(keep):
    rng = np.random.default_rng(keep)
    blocks = []
    for orient in range(4):
        cb, npass, nbps = R.t1_encode_block(_block(rng, 64, 64, 10, 1), orient)
        blocks.append((cb, min(keep, npass), nbps, orient, 64, 64))
    got, _ = decode(blocks)
    for g, b in zip(got, blocks):
        assert np.array_equal(g, O.t1_decode_block(*b))


def test_garbage_streams_decode_like_the_oracle():
    """Random bytes are a legal MQ stream of some content: the lanes must read them exactly as the reference procedure does
    (0xFF handling, the artificial terminator past the end) and terminate."""
    rng = np.random.default_rng(9)
    blocks = []
    for k in range(70):
        n = int(rng.integers(0, 400))
        cb = bytes(rng.integers(0, 256, n, dtype=np.uint8)) if k % 5 else bytes([0xFF] * n)
        blocks.append((cb, int(rng.integers(1, 20)), int(rng.integers(1, 12)), k % 4, int(rng.integers(1, 65)), int(rng.integers(9, 65))))
    got, _ = decode(blocks, pad_front=3)
    for g, b in zip(got, blocks):
        assert np.array_equal(g, O.t1_decode_block(*b)), "len %d passes %d bps %d" % (len(b[0]), b[1], b[2])


def test_tables_of_the_lane_decoder_match_the_oracle_rules():
    """Folded Table C.2: every entry's Qe and successor entries, both MPS senses."""
    cc = subprocess.run(["g++", "-std=c++17", "-x", "c++", "-", "-o", os.path.join(ROOT, "build", "t1l_tab")], input="""
#include "%s/grok_amd/csrc/t1_lanes.h"
#include <cstdio>
int main() { for (unsigned e = 0; e < 94; ++e) std::printf("%%u\\n", t1l::mq_entry(e)); }
""" % ROOT, text=True, capture_output=True)
    assert cc.returncode == 0, cc.stderr
    vals = [int(v) for v in subprocess.check_output([os.path.join(ROOT, "build", "t1l_tab")]).split()]
    QE = [0x5601, 0x3401, 0x1801, 0x0AC1, 0x0521, 0x0221, 0x5601, 0x5401, 0x4801, 0x3801, 0x3001, 0x2401, 0x1C01, 0x1601, 0x5601, 0x5401,
          0x5101, 0x4801, 0x3801, 0x3401, 0x3001, 0x2801, 0x2401, 0x2201, 0x1C01, 0x1801, 0x1601, 0x1401, 0x1201, 0x1101, 0x0AC1, 0x09C1,
          0x08A1, 0x0521, 0x0441, 0x02A1, 0x0221, 0x0141, 0x0111, 0x0085, 0x0049, 0x0025, 0x0015, 0x0009, 0x0005, 0x0001, 0x5601]
    SW = {0, 6, 14}
    for e, v in enumerate(vals):
        st, mps = e % 47, e // 47
        assert v >> 16 == QE[st] and (v >> 14) & 1 == mps
        assert (v & 0x7F) // 47 == mps                               # after an MPS the sense stays
        assert ((v >> 7) & 0x7F) // 47 == (mps ^ (1 if st in SW else 0))
