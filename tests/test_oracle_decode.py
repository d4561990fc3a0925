"""CPU: the decode-side oracle (oracle/j2k_decode_oracle.c) pinned against the REAL reference decoder
(oracle/_ref): per block (ojph_decode_codeblock) and whole image (our codestream -> grk_decompress)."""
import numpy as np
import pytest

import grok_amd as G
import oracle as O
import chain
import refharness as R
import synth

needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


def _random_block(rng, w, h, kmax, mode):
    # decodable range: U_q <= missing_msbs = kmax-1 (ojph_block_decoder.cpp:1194, defect D5), i.e.
    # bitlen(2*mag-1) <= kmax-1
    kmax = kmax - 2
    mag = rng.integers(0, (1 << kmax) + 1, size=(h, w))
    if mode == 1:
        mag = mag >> rng.integers(0, kmax + 1, size=(h, w))
    elif mode == 2:
        mag = np.where(rng.random((h, w)) < 0.93, 0, mag & 7)
    elif mode == 3:
        mag = np.zeros((h, w), np.int64)
    elif mode == 4:
        mag = np.full((h, w), (1 << kmax) - 1)
    sign = np.where(rng.random((h, w)) < 0.5, -1, 1)
    return (mag * sign).astype(np.int32)


@pytest.mark.parametrize("w,h,kmax", [(64, 64, 10), (64, 64, 16), (32, 32, 9), (37, 3, 8), (1, 1, 8), (5, 64, 12),
                                      (64, 1, 11), (2, 2, 3), (63, 31, 20)])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_ht_decode_inverts_encoder(w, h, kmax, mode):
    rng = np.random.default_rng(w * 131 + h * 7 + kmax + mode)
    coef = _random_block(rng, w, h, kmax, mode)
    cb = O.ht_encode_sm(O.signmag(coef, kmax), kmax)
    sm = O.ht_decode_block(cb, kmax - 1, w, h)
    assert sm is not None
    # decoded word = sign | (2*mu+1) << (30-kmax)   (bin centre, ojph_block_decoder.cpp:1238-1243)
    mu = np.abs(coef.astype(np.int64))
    want = np.where(mu > 0, np.where(coef < 0, 0x80000000, 0) | ((2 * mu + 1) << (30 - kmax)), 0).astype(np.uint32)
    assert np.array_equal(sm, want)
    assert np.array_equal(O.ht_dequant_rev(sm, kmax - 1), coef)
    if R.have_ref():
        assert np.array_equal(R.ht_decode_block(cb, kmax - 1, w, h), sm)


def test_ht_decode_rejects_what_the_reference_rejects():
    """Full-scale magnitudes (>= 2^(kmax-1)) make U_q exceed missing_msbs: both decoders refuse (D5)."""
    coef = np.full((8, 8), (1 << 10) - 1, np.int32)
    cb = O.ht_encode_sm(O.signmag(coef, 10), 10)
    assert O.ht_decode_block(cb, 9, 8, 8) is None
    if R.have_ref():
        with pytest.raises(RuntimeError):
            R.ht_decode_block(cb, 9, 8, 8)


def test_ht_decode_rejects_bad_scup():
    cb = bytearray(O.ht_encode_sm(O.signmag(np.ones((8, 8), np.int32), 8), 8))
    cb[-1] = 0xFF; cb[-2] |= 0x0F                       # Scup larger than the block
    assert O.ht_decode_block(bytes(cb), 7, 8, 8) is None
    assert O.ht_decode_block(b"\x00", 7, 8, 8) is None


@needs_ref
@pytest.mark.parametrize("Cn,H,W,prec,L,gen", [(1, 128, 128, 8, 3, "g2"), (3, 96, 160, 8, 4, "g2"), (3, 64, 64, 8, 5, "g0"),
                                                 (1, 70, 33, 12, 2, "g2"), (3, 128, 128, 16, 3, "g2")])
def test_lossless_chain_vs_reference_decoder(Cn, H, W, prec, L, gen):
    px = getattr(synth, gen)(Cn, H, W, prec)
    p, blocks, qcd, table, coded = chain.encode_tile_oracle(px, prec, L)
    cs = G.write_codestream(p, W, H, table, coded)
    ref = R.decode(cs, Cn, H, W)
    assert np.array_equal(ref, px.astype(np.int32))                    # the reference decodes our stream losslessly
    ours = chain.decode_tile_oracle(p, blocks, qcd, table, coded)
    assert np.array_equal(ours, ref)


@needs_ref
@pytest.mark.parametrize("Cn,H,W,prec,L", [(1, 128, 128, 8, 3), (3, 96, 160, 8, 4), (3, 128, 192, 12, 5), (1, 67, 45, 8, 2),
                                           (3, 128, 192, 16, 3), (3, 256, 256, 16, 5), (1, 100, 75, 16, 4), (3, 1024, 1024, 16, 5)])
def test_irreversible_chain_vs_reference_decoder(Cn, H, W, prec, L):
    """9/7 + ICT + dead-zone quantiser: our codestream through grk_decompress == the oracle's decode
    chain, pixel for pixel (both follow the same fp32 operation order)."""
    px = (synth.g2_mid if W >= 1024 else synth.g2)(Cn, H, W, prec)       # (defect D5: synth.g2_mid)
    p, blocks, qcd, table, coded = chain.encode_tile_oracle(px, prec, L, irrev=True)
    cs = G.write_codestream(p, W, H, table, coded)
    ref = R.decode(cs, Cn, H, W)
    ours = chain.decode_tile_oracle(p, blocks, qcd, table, coded)
    err = np.abs(ref.astype(np.int64) - px.astype(np.int64))
    assert err.max() <= max(2, (1 << prec) // 64), "reference decode of our lossy stream is far from the source"
    if prec == 16:          # cfg3's own bit depth: the default step sizes leave a few units of 65 536 (measured: 4, 98 dB)
        assert err.max() <= 8 and synth.psnr_db(ref, px, prec) >= 90.0
    assert np.array_equal(ours, ref)


@needs_ref
@pytest.mark.parametrize("Cn,prec", [(1, 8), (3, 8), (3, 12)])
def test_signed_samples_lossless_vs_reference_decoder(Cn, prec):
    """Signed components (SIZ Ssiz bit 7, DC shift 0, samples read as int8/int16, TileProcessor.cpp:1188-1212):
    our stream through grk_decompress returns the signed source; the oracle decode chain agrees."""
    H, W, L = 96, 128, 3
    u = synth.g2(Cn, H, W, prec).astype(np.int32)
    px = (u - (1 << (prec - 1))).astype(np.int8 if prec <= 8 else np.int16)
    p, blocks, qcd, table, coded = chain.encode_tile_oracle(px, prec, L, sgnd=True)
    cs = G.write_codestream(p, W, H, table, coded)
    ref = R.decode(cs, Cn, H, W)
    assert np.array_equal(ref, px.astype(np.int32))
    assert np.array_equal(chain.decode_tile_oracle(p, blocks, qcd, table, coded), ref)
