"""CPU: the oracle's HT SigProp / MagRef passes (oracle/ht_refine_oracle.c) pinned against the
reference's own decoder called with lengths2 != 0 (ojph_decode_codeblock, ojph_block_decoder.cpp:1627-2100) -- Grok
itself never reaches that code (T1HT.cpp:158-166), and its encoder never writes the passes, so the oracle's small encoder
makes the vectors."""
import numpy as np
import pytest

import oracle as O
import refharness as R

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref (the real reference) not built here")


def make_block(rng, w, h, kmax, mode):
    """-> (mag incl. the refinement bit-plane as LSB, sign)"""
    top = 1 << (kmax - 2)
    if mode == 0:
        mag = rng.integers(0, top, size=(h, w))
    elif mode == 1:                      # sparse: lots of insignificant samples next to significant ones -> SPP members
        mag = np.where(rng.random((h, w)) < 0.7, rng.integers(0, 2, size=(h, w)), rng.integers(0, 16, size=(h, w)))
    elif mode == 2:                      # nothing significant in the cleanup pass: no members at all
        mag = rng.integers(0, 2, size=(h, w))
    elif mode == 3:                      # every sample significant: no SPP bits, MRP for all
        mag = rng.integers(2, top, size=(h, w))
    else:                                # all refinement bits one: 0xFF-heavy segments (stuffing in both directions)
        mag = np.where(rng.random((h, w)) < 0.5, 1, rng.integers(1, 8, size=(h, w)) * 2 + 1)
    sign = rng.integers(0, 2, size=(h, w))
    return mag.astype(np.uint32), sign.astype(np.uint8)


def code_block(mag, sign, kmax, npasses):
    """cleanup pass over mag >> 1 (the oracle's encoder, itself pinned to ojph_encode_codeblock) + the refinement segment"""
    mu = (mag >> 1).astype(np.uint32)
    sm = (sign.astype(np.uint32) << 31) | (mu << np.uint32(30 - kmax))
    sm = np.where(mu == 0, 0, sm).astype(np.uint32)          # (no "negative zero": insignificant samples carry no sign)
    cup = O.ht_encode_sm(sm, kmax)
    seg, spp_len = O.ht_refine_encode(mag, sign, npasses)
    return cup, seg, spp_len


CASES = [(64, 64), (32, 32), (7, 5), (64, 3), (3, 64), (1, 1), (33, 17), (8, 8), (4, 4), (5, 9), (64, 6), (12, 10)]


@needs_ref
@pytest.mark.parametrize("npasses", [2, 3])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_refinement_passes_equal_the_reference_decoder(npasses, mode):
    rng = np.random.default_rng(1000 * npasses + mode)
    n_members = 0
    for (w, h) in CASES:
        for kmax in (6, 9, 12):
            mag, sign = make_block(rng, w, h, kmax, mode)
            cup, seg, spp_len = code_block(mag, sign, kmax, npasses)
            mm = kmax - 1
            ref = R.ht_decode_block_passes(cup + seg, len(cup), len(seg), npasses, mm, w, h)
            assert ref is not None
            base = O.ht_decode_block(cup, mm, w, h)
            assert np.array_equal(base, R.ht_decode_block(cup, mm, w, h))
            got = O.ht_refine_decode(base, mm, seg, npasses)
            assert np.array_equal(got, ref), "w %d h %d kmax %d: %d words differ" % (w, h, kmax, int((got != ref).sum()))
            n_members += spp_len
            # what the passes mean: cleanup-significant samples gain bit-plane p - 1 (MagRef), members that turn
            # significant come out as 1.5 x 2^(p-1) with their sign
            p = 30 - mm
            mu = mag >> 1
            sig = mu != 0
            if npasses == 3:
                want = (sign.astype(np.uint32) << 31) | (mag << np.uint32(p - 1)) | np.uint32(1 << (p - 2))
                assert np.array_equal(got[sig], want[sig])
            new = (~sig) & (got != 0)
            assert np.all((got[new] & 0x7FFFFFFF) == (3 << (p - 2)))
            assert np.all((got[new] >> 31) == sign[new])
            assert np.all((mag[new] & 1) == 1)
    if mode in (0, 1, 4):
        assert n_members > 0


@needs_ref
def test_truncated_and_empty_refinement_segments():
    """lengths2 = 0 or one pass: nothing happens; a truncated segment reads zeros beyond its end -- as the reference does"""
    rng = np.random.default_rng(5)
    mag, sign = make_block(rng, 64, 64, 10, 1)
    cup, seg, _ = code_block(mag, sign, 10, 3)
    base = O.ht_decode_block(cup, 9, 64, 64)
    assert np.array_equal(O.ht_refine_decode(base, 9, b"", 3), base)
    for cut in (1, len(seg) // 3, len(seg) - 1):
        part = seg[:cut]
        ref = R.ht_decode_block_passes(cup + part, len(cup), len(part), 3, 9, 64, 64)
        assert np.array_equal(O.ht_refine_decode(base, 9, part, 3), ref)
