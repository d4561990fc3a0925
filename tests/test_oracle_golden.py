"""CPU: pins the oracle (oracle/j2k_oracle.c) to the reference's known answers.

Sources of truth: SURVEY.md Appendix C.1/C.2 (known answers produced by Grok 8.0.2), the fixtures
in tests/golden/ (generated from the real reference by tests/golden/gen_golden.py), and -- when
oracle/_ref is present -- the real reference called live.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle as O
import synth
import refharness as R
from cshelp import oracle_codestream

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


def fnv1a64(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def kat_block(vals, w, h, kmax):
    return O.ht_encode_sm(O.signmag(np.array(vals, np.int64).reshape(h, w), kmax), kmax)


def lcg(s):
    return (s * 1664525 + 1013904223) & 0xFFFFFFFF


# ---- Appendix C.1: HT cleanup encoder known answers -------------------------------------------
def test_kat_4x4():
    v = [((i * 7 + 3) % 23) - 11 for i in range(16)]
    assert kat_block(v, 4, 4, 8).hex() == "6fac34290cd18300e89bb33600"


def test_kat_8x8():
    v = [((i * 7 + 3) % 23) - 11 for i in range(64)]
    want = ("2f01da536c419358714a23650dc8f305ae497803c6f12b54cf620b9a4429"
            "3800a845c95f82ddd312e1a682541f9001")
    assert kat_block(v, 8, 8, 8).hex() == want


def test_kat_zero_blocks():
    assert kat_block([0] * 16, 4, 4, 8).hex() == "f00300"
    assert kat_block([0] * 4096, 64, 64, 10).hex() == "ff7fff7fff7c0800"


def test_kat_64x64_lcg():
    s, v = 1, []
    for _ in range(4096):
        s = lcg(s)
        m = (s >> 20) & 0xFF
        x = -m if (s & 0x80000) else m
        if (s >> 8) & 3:
            x = int(x / 8)           # C truncating division
        v.append(x)
    b = kat_block(v, 64, 64, 10)
    assert len(b) == 4474
    assert "%016x" % fnv1a64(b) == "fbf5b3926d7965a8"
    assert b[:16].hex() == "51cdb390391ddc995a2036c7c3ca5bc0" and b[-4:].hex() == "5261b233"


def test_kat_32x32_lcg():
    s, v = 7, []
    for _ in range(1024):
        s = lcg(s)
        m = (s >> 22) & 0x3F
        v.append(-m if (s & 0x80000) else m)
    b = kat_block(v, 32, 32, 9)
    assert len(b) == 995
    assert "%016x" % fnv1a64(b) == "ad7008c9378e17d4"
    assert b[:16].hex() == "a6b969976daa8a966586de7a2f22eeb0" and b[-4:].hex() == "e4d1b30e"


# ---- Appendix C.2: MCT / DWT known answers -------------------------------------------------------
def test_kat_dwt53_rows():
    a = np.array([10, -3, 7, 22, -15, 4, 9, -1], np.int32)
    O.lib().orc_dwt53_fwd_1d(a.ctypes.data, 8)
    assert a.tolist() == [5, 11, -7, 8, -11, 26, 7, -10]
    a = np.array([10, -3, 7, 22, -15, 4, 9], np.int32)
    O.lib().orc_dwt53_fwd_1d(a.ctypes.data, 7)
    assert a.tolist() == [5, 11, -7, 13, -11, 26, 7]


def test_kat_dwt97_rows():
    a = np.array([10, -3, 7, 22, -15, 4, 9, -1], np.float32)
    O.lib().orc_dwt97_fwd_1d(a.ctypes.data, 8)
    assert ["%08x" % x for x in a.view(np.uint32)] == ["3fe4f3f4", "41217e22", "c03206ac", "40c16895",
                                                      "c17386af", "41f7827a", "4104289a", "c16f4dbc"]
    a = np.array([10, -3, 7, 22, -15, 4, 9], np.float32)
    O.lib().orc_dwt97_fwd_1d(a.ctypes.data, 7)
    assert ["%08x" % x for x in a.view(np.uint32)] == ["3fe4f3f4", "41217e22", "c0608239", "4118a65b",
                                                      "c17386af", "41f7827a", "40b90378"]


def test_kat_dwt53_2d():
    a = np.array([((i * 37 + 11) % 101) - 50 for i in range(64)], np.int32).reshape(8, 8)
    want = np.array([[-57, 26, 8, 3, 13, -50, 51, 37], [14, -5, -9, -11, -44, -62, 51, 37],
                     [9, 4, 1, -17, -50, 39, 44, 24], [6, -2, -25, 30, -24, 51, -6, -77],
                     [-37, 6, 0, 0, 25, 0, 0, 0], [0, -12, -12, 0, 0, -50, 0, 0],
                     [0, 0, -6, 32, 0, 0, -25, -51], [44, 18, -7, -7, 101, 0, 0, 0]], np.int32)
    assert np.array_equal(O.dwt53_fwd(a, 1), want)


def test_kat_mct():
    R_, G_, B_ = [100, 50, 200, 7], [20, 60, 10, 7], [5, 250, 0, 7]
    y, u, v = O.rct_fwd(R_, G_, B_)
    assert y.tolist() == [36, 105, 55, 7] and u.tolist() == [-15, 190, -10, 0] and v.tolist() == [80, -10, 190, 0]
    y, u, v = O.ict_fwd(R_, G_, B_)
    hx = lambda a: ["%08x" % x for x in a.view(np.uint32)]
    assert hx(y) == ["4228d70a", "429d570a", "4283570a", "40e00000"]
    assert hx(u) == ["c1a7fdb0", "42c15fee", "c2143d41", "00000000"]
    assert hx(v) == ["4224e0f6", "c1a39849", "42bfa052", "00000000"]


def test_qcd_exponents_cfg1():
    assert O.rev_exponents(8, 3).tolist() == [10, 11, 11, 12, 11, 11, 11, 10, 10, 11]


# ---- fixtures generated from the real reference ----------------------------------------------------
def test_stage_vectors_fixture():
    z = np.load(os.path.join(GOLD, "stage_vectors.npz"))
    got = O.rct_fwd(*z["rct_in"])
    assert np.array_equal(np.stack(got), z["rct_out"])
    got = O.ict_fwd(*z["ict_in"])
    assert np.array_equal(np.stack([g.view(np.int32) for g in got]), z["ict_out"])
    for i in range(5):
        w, h, lv = z["dwt%d_meta" % i]
        assert np.array_equal(O.dwt53_fwd(z["dwt%d_in53" % i], int(lv)), z["dwt%d_out53" % i])
        assert np.array_equal(O.dwt97_fwd(z["dwt%d_in97" % i], int(lv)).view(np.int32), z["dwt%d_out97" % i])
        assert np.array_equal(O.dwt53_inv(z["dwt%d_out53" % i], int(lv)), z["dwt%d_in53" % i])
    for i in range(int(z["ht_count"][0])):
        w, h, kmax = [int(x) for x in z["ht%d_meta" % i]]
        got = O.ht_encode_sm(O.signmag(z["ht%d_coeff" % i], kmax), kmax)
        assert got == z["ht%d_coded" % i].tobytes(), "HT fixture %d (%dx%d kmax %d)" % (i, w, h, kmax)


GOLD_FILES = [("g0_1x512x512_r4", "g0", (1, 512, 512), 8, 3, None),
              ("g2_1x256x256_r4", "g2", (1, 256, 256), 8, 3, None),
              ("g0_3x512x512_r6", "g0", (3, 512, 512), 8, 5, None),
              ("g2_3x192x160_r4", "g2", (3, 160, 192), 8, 3, None),
              ("g2_3x256x256_t128_r4", "g2", (3, 256, 256), 8, 3, 128),
              ("g2u16_1x128x128_r5", "g2", (1, 128, 128), 12, 4, None)]


@pytest.mark.parametrize("name,gen,shape,prec,L,tile", GOLD_FILES)
def test_codestream_fixture(name, gen, shape, prec, L, tile):
    """oracle hot path + product Tier-2 writer == the reference encoder's file, byte for byte."""
    px = getattr(synth, gen)(*shape, prec)
    cs = oracle_codestream(px, prec, L, tile, tile)
    want = open(os.path.join(GOLD, name + ".j2k"), "rb").read()
    assert cs == want


# whole-file md5s of Grok 8.0.2
MD5S = [("g0", 1, 512, 512, 3, None, "8f2ec0f22e10fbeb97c3bf515d7ad976"),
        ("g2", 1, 512, 512, 3, None, "0ea91840e2b0d964ce8e570148a07642"),       # BASELINE cfg1
        ("g0", 3, 512, 512, 5, None, "8d81ed0576b981cba0c0111e5f9724a5"),
        ("g0", 1, 64, 64, 0, None, "4afbe3defe07e7ca0e8d4c1d5433b84d"),
        ("g0", 1, 128, 128, 1, None, "8ae843154d7f6f1283e796a9bd029758"),
        ("g2", 3, 1024, 1024, 5, None, "2ec6724e8acd2796841140b37cf5ca72")]


@pytest.mark.parametrize("gen,C,W,H,L,tile,md5", MD5S)
def test_codestream_md5(gen, C, W, H, L, tile, md5):
    cs = oracle_codestream(getattr(synth, gen)(C, H, W), 8, L, tile, tile)
    assert hashlib.md5(cs).hexdigest() == md5


def test_codestream_md5_multitile():
    full = np.tile(synth.g0(3, 1024, 1024), (1, 2, 2))
    cs = oracle_codestream(full, 8, 5, 1024, 1024)
    assert len(cs) == 59425 and hashlib.md5(cs).hexdigest() == "c40ea21d1c483639bc15cce9c59e60a6"


# ---- live comparison with the real reference (build container / shipped oracle/_ref) -------------
@needs_ref
@pytest.mark.ref
def test_ht_encoder_vs_reference_random():
    rng = np.random.default_rng(5)
    for trial in range(200):
        w, h, kmax = int(rng.integers(1, 65)), int(rng.integers(1, 65)), int(rng.integers(2, 20))
        mag = rng.integers(0, 1 << kmax, size=(h, w))
        m = trial % 4
        if m == 1: mag = mag >> rng.integers(0, kmax, size=(h, w))
        if m == 2: mag = np.where(rng.random((h, w)) < 0.9, 0, mag)
        if m == 3: mag = mag & 3
        sm = O.signmag(mag * np.where(rng.random((h, w)) < 0.5, -1, 1), kmax)
        assert O.ht_encode_sm(sm, kmax) == R.ht_encode_block(sm, kmax), (w, h, kmax, m)


@needs_ref
@pytest.mark.ref
def test_ht_encoder_decodes_with_reference_decoder():
    rng = np.random.default_rng(6)
    for (w, h, kmax) in [(64, 64, 10), (33, 17, 12), (64, 64, 8)]:
        c = rng.integers(-(1 << (kmax - 2)) + 1, 1 << (kmax - 2), size=(h, w))   # decoder rejects U_q > Kmax-1 (D5)
        coded = O.ht_encode_sm(O.signmag(c, kmax), kmax)
        dec = R.ht_decode_block(coded, kmax - 1, w, h)
        mag = (dec & 0x7FFFFFFF) >> (31 - kmax)
        val = np.where(dec >> 31, -mag.astype(np.int64), mag.astype(np.int64))
        assert np.array_equal(val, c)


@needs_ref
@pytest.mark.ref
def test_dwt_vs_reference_ragged():
    rng = np.random.default_rng(7)
    L = R.lib()
    for (w, h, lv) in [(65, 33, 3), (100, 77, 5), (17, 1, 2), (1, 9, 2), (3, 3, 1), (255, 257, 5)]:
        a = rng.integers(-200, 200, size=(h, w)).astype(np.int32)
        p = a.copy(); L.ref_dwt53_fwd(p.ctypes.data, w, h, w, lv)
        assert np.array_equal(O.dwt53_fwd(a, lv), p)
        f = (rng.standard_normal((h, w)) * 100).astype(np.float32)
        q = f.copy(); L.ref_dwt97_fwd(q.ctypes.data, w, h, w, lv)
        assert np.array_equal(O.dwt97_fwd(f, lv).view(np.int32), q.view(np.int32))


@needs_ref
@pytest.mark.ref
def test_whole_codestream_vs_reference_and_roundtrip():
    px = synth.g2(3, 192, 320, 8)
    ref, _ = R.encode(px, 8, numres=5)
    assert oracle_codestream(px, 8, 4) == ref
    assert np.array_equal(R.decode(ref, 3, 192, 320), px.astype(np.int32))
