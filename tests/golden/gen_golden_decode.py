#!/usr/bin/env python3
"""Decode-side fixtures from the REAL reference (oracle/_ref): codestreams written by grk_compress and the
pixels grk_decompress makes of them.  Runs only in the build container; outputs are committed so that the
tests can check the decode path where /root/reference and oracle/_ref are absent.

    python tests/golden/gen_golden_decode.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refharness as R  # noqa: E402
import synth  # noqa: E402

CASES = [  # name, C, H, W, prec, numres, ht, irrev, code-block style
    ("dec_p1_irrev_3x96x160_r5", 3, 96, 160, 8, 5, 0, 1, 0),      # BASELINE configs[4] shape: Part-1 EBCOT + ICT + 9/7
    ("dec_p1_irrev_3x128x128_p12_r6", 3, 128, 128, 12, 6, 0, 1, 0),
    ("dec_p1_rev_3x100x77_r3", 3, 100, 77, 8, 3, 0, 0, 0),
    ("dec_ht_rev_1x128x128_r4", 1, 128, 128, 8, 4, 1, 0, 0),
    ("dec_p1_sty3f_irrev_3x96x128_p10_r4", 3, 96, 128, 10, 4, 0, 1, 0x3F),   # every Part-1 code-block style at once
    ("dec_p1_sty05_rev_1x128x96_p12_r3", 1, 128, 96, 12, 3, 0, 0, 0x05),      # LAZY + TERMALL: a segment per pass, raw passes
]


def main():
    out = {}
    for name, C, H, W, prec, numres, ht, irrev, sty in CASES:
        px = synth.g2(C, H, W, prec)
        cs, _ = R.encode(px, prec, numres=numres, mode=1, ht=ht, irrev=irrev, cblksty=sty)
        open(os.path.join(HERE, name + ".j2k"), "wb").write(cs)
        out[name] = R.decode(cs, C, H, W).astype(np.uint16 if prec > 8 else np.uint8)
        print(name, len(cs), "bytes")
    np.savez_compressed(os.path.join(HERE, "decode_vectors.npz"), **out)
    # Part-1 block vectors: blocks coded by the reference T1 and its own decode of them
    rng = np.random.default_rng(20260926)
    blk = {}
    for i, (w, h, bits, orient) in enumerate([(64, 64, 9, 0), (64, 64, 11, 3), (32, 32, 8, 1), (17, 5, 7, 2), (64, 3, 10, 3)]):
        coef = ((rng.integers(0, 1 << bits, size=(h, w)) >> rng.integers(0, bits + 1, size=(h, w))) *
                np.where(rng.random((h, w)) < 0.5, -1, 1)).astype(np.int32)
        cb, npass, nbps = R.t1_encode_block(coef, orient)
        blk["t1_%d_meta" % i] = np.array([w, h, orient, npass, nbps])
        blk["t1_%d_coded" % i] = np.frombuffer(cb, np.uint8)
        blk["t1_%d_decoded" % i] = R.t1_decode_block(cb, npass, nbps, orient, w, h)
        blk["t1_%d_coef" % i] = coef
    blk["t1_count"] = np.array([5])
    # the same with code-block styles: segments as (bytes, passes) pairs
    for i, (w, h, bits, orient, sty) in enumerate([(64, 64, 10, 1, 0x01), (64, 64, 9, 3, 0x04), (48, 20, 8, 2, 0x08 | 0x02),
                                                    (64, 64, 12, 0, 0x3F), (33, 64, 11, 3, 0x20 | 0x10)]):
        coef = ((rng.integers(0, 1 << bits, size=(h, w)) >> rng.integers(0, bits + 1, size=(h, w))) *
                np.where(rng.random((h, w)) < 0.5, -1, 1)).astype(np.int32)
        cb, segs, nbps = R.t1_encode_block_sty(coef, orient, sty)
        blk["sty_%d_meta" % i] = np.array([w, h, orient, sty, nbps])
        blk["sty_%d_segs" % i] = np.array(segs, np.uint32)
        blk["sty_%d_coded" % i] = np.frombuffer(cb, np.uint8)
        blk["sty_%d_decoded" % i] = R.t1_decode_block_sty(cb, segs, nbps, orient, sty, w, h)
        blk["sty_%d_coef" % i] = coef
    blk["sty_count"] = np.array([5])
    np.savez_compressed(os.path.join(HERE, "t1_block_vectors.npz"), **blk)


if __name__ == "__main__":
    main()
