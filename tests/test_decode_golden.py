"""Decode path against COMMITTED fixtures of the real reference (tests/golden/gen_golden_decode.py): no
oracle/_ref needed at run time.  CPU: oracle chain and Part-1 block decoder; GPU (-m gpu): the HIP path."""
import os

import numpy as np
import pytest

import grok_amd as G
import oracle as O
import j2kparse as J

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = np.load(os.path.join(GOLD, "decode_vectors.npz"))
T1V = np.load(os.path.join(GOLD, "t1_block_vectors.npz"))
STREAMS = ["dec_p1_irrev_3x96x160_r5", "dec_p1_irrev_3x128x128_p12_r6", "dec_p1_rev_3x100x77_r3", "dec_ht_rev_1x128x128_r4",
           "dec_p1_sty3f_irrev_3x96x128_p10_r4", "dec_p1_sty05_rev_1x128x96_p12_r3"]


@pytest.mark.parametrize("i", range(int(T1V["t1_count"][0])))
def test_t1_block_decoder_vs_golden(i):
    w, h, orient, npass, nbps = [int(v) for v in T1V["t1_%d_meta" % i]]
    got = O.t1_decode_block(T1V["t1_%d_coded" % i].tobytes(), npass, nbps, orient, w, h)
    assert np.array_equal(got, T1V["t1_%d_decoded" % i])
    assert np.array_equal(O.t1_dequant_rev(got), T1V["t1_%d_coef" % i])


@pytest.mark.parametrize("i", range(int(T1V["sty_count"][0])))
def test_t1_styled_block_decoder_vs_golden(i):
    w, h, orient, sty, nbps = [int(v) for v in T1V["sty_%d_meta" % i]]
    segs = [(int(a), int(b)) for a, b in T1V["sty_%d_segs" % i]]
    got, bad = O.t1_decode_block_sty(T1V["sty_%d_coded" % i].tobytes(), segs, nbps, orient, sty, w, h)
    assert bad == 0
    assert np.array_equal(got, T1V["sty_%d_decoded" % i])
    assert np.array_equal(O.t1_dequant_rev(got), T1V["sty_%d_coef" % i])


@pytest.mark.parametrize("name", STREAMS)
def test_oracle_chain_vs_golden_pixels(name):
    from test_oracle_ebcot import _oracle_decode_stream
    cs = open(os.path.join(GOLD, name + ".j2k"), "rb").read()
    got = _oracle_decode_stream(cs, part1="_p1_" in name)
    assert np.array_equal(got, VEC[name].astype(np.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", STREAMS)
def test_gpu_decode_vs_golden_pixels(name):
    from test_gpu_decode import _gpu_decode_reference_stream
    cs = open(os.path.join(GOLD, name + ".j2k"), "rb").read()
    got = _gpu_decode_reference_stream(cs, part1="_p1_" in name)
    assert np.array_equal(got.astype(np.int32), VEC[name].astype(np.int32))
