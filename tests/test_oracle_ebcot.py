"""CPU: the Part-1 (EBCOT/MQ) block-decoder oracle (oracle/ebcot_oracle.c) pinned
against the REAL reference T1 (oracle/_ref): blocks coded by Grok's own T1::compress_cblk are decoded by
both, sample for sample, and truncated pass sequences as well."""
import numpy as np
import pytest

import oracle as O
import refharness as R

pytestmark = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


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


@pytest.mark.parametrize("w,h,bits", [(64, 64, 8), (64, 64, 12), (32, 32, 10), (37, 3, 8), (1, 1, 5), (5, 64, 9), (64, 1, 7),
                                      (2, 2, 3), (63, 31, 14), (4, 4, 1), (13, 7, 16)])
@pytest.mark.parametrize("orient", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_t1_decode_equals_reference(w, h, bits, orient, mode):
    rng = np.random.default_rng(w * 31 + h * 5 + bits * 3 + orient + 7 * mode)
    coef = _block(rng, w, h, bits, mode)
    cb, npass, nbps = R.t1_encode_block(coef, orient)
    ref = R.t1_decode_block(cb, npass, nbps, orient, w, h)
    got = O.t1_decode_block(cb, npass, nbps, orient, w, h)
    assert np.array_equal(got, ref)
    assert np.array_equal(O.t1_dequant_rev(got), coef)            # all passes decoded: lossless


@pytest.mark.parametrize("keep", [1, 2, 3, 4, 7, 11])
def test_t1_decode_truncated_pass_sequence(keep):
    """Decoding fewer passes than were coded (what a quality-layer truncation leaves) matches too."""
    rng = np.random.default_rng(keep)
    coef = _block(rng, 64, 64, 10, 1)
    cb, npass, nbps = R.t1_encode_block(coef, 3)
    k = min(keep, npass)
    assert np.array_equal(O.t1_decode_block(cb, k, nbps, 3, 64, 64), R.t1_decode_block(cb, k, nbps, 3, 64, 64))
