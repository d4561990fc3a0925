#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz|*.j2k from the REAL reference (oracle/_ref, built by
oracle/Makefile from /root/reference).  Runs only in the build container; the outputs are
committed so that the GPU box (which has no /root/reference) can check against them.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refharness as R  # noqa: E402
import synth  # noqa: E402


def main():
    L = R.lib()
    rng = np.random.default_rng(20260925)
    out = {}
    # --- stage vectors: RCT / ICT
    r, g, b = [rng.integers(-128, 128, size=257).astype(np.int32) for _ in range(3)]
    rr = [v.copy() for v in (r, g, b)]
    L.ref_rct(rr[0].ctypes.data, rr[1].ctypes.data, rr[2].ctypes.data, r.size)
    out["rct_in"] = np.stack([r, g, b]); out["rct_out"] = np.stack(rr)
    r, g, b = [rng.integers(-32768, 32768, size=257).astype(np.int32) for _ in range(3)]
    rr = [v.copy() for v in (r, g, b)]
    L.ref_ict(rr[0].ctypes.data, rr[1].ctypes.data, rr[2].ctypes.data, r.size)
    out["ict_in"] = np.stack([r, g, b]); out["ict_out"] = np.stack(rr)
    # --- stage vectors: multi-level DWT on ragged sizes
    for i, (w, h, lv) in enumerate([(37, 23, 3), (64, 48, 2), (5, 1, 1), (1, 7, 2), (96, 80, 4)]):
        a = rng.integers(-255, 256, size=(h, w)).astype(np.int32)
        p = a.copy(); L.ref_dwt53_fwd(p.ctypes.data, w, h, w, lv)
        f = (rng.standard_normal((h, w)) * 100).astype(np.float32)
        q = f.copy(); L.ref_dwt97_fwd(q.ctypes.data, w, h, w, lv)
        out["dwt%d_meta" % i] = np.array([w, h, lv])
        out["dwt%d_in53" % i] = a; out["dwt%d_out53" % i] = p
        out["dwt%d_in97" % i] = f; out["dwt%d_out97" % i] = q.view(np.int32)
    # --- HT cleanup encoder: random blocks of assorted sizes / sparsity
    nb = 0
    for (w, h, kmax, mode) in [(64, 64, 10, 0), (64, 64, 12, 1), (32, 32, 9, 0), (64, 64, 11, 2), (17, 5, 8, 1),
                               (1, 1, 8, 0), (3, 64, 13, 1), (64, 2, 10, 0), (64, 64, 10, 3), (40, 33, 19, 1),
                               (64, 64, 8, 4)]:
        mag = rng.integers(0, 1 << kmax, size=(h, w))
        if mode == 1: mag = mag >> rng.integers(0, kmax + 1, size=(h, w))
        if mode == 2: mag = np.where(rng.random((h, w)) < 0.93, 0, mag & 7)
        if mode == 3: mag = np.zeros((h, w), np.int64)
        if mode == 4: mag = np.full((h, w), (1 << kmax) - 1)
        c = (mag * np.where(rng.random((h, w)) < 0.5, -1, 1)).astype(np.int32)
        sm = (np.where(c < 0, 0x80000000, 0) | (np.abs(c.astype(np.int64)) << (30 - kmax))).astype(np.uint32)
        coded = R.ht_encode_block(sm, kmax)
        out["ht%d_meta" % nb] = np.array([w, h, kmax]); out["ht%d_coeff" % nb] = c
        out["ht%d_coded" % nb] = np.frombuffer(coded, np.uint8)
        nb += 1
    out["ht_count"] = np.array([nb])
    np.savez_compressed(os.path.join(HERE, "stage_vectors.npz"), **out)
    # --- whole codestreams of the reference CPU encoder (HT, reversible)
    for name, px, numres, tile in [("g0_1x512x512_r4", synth.g0(1, 512, 512), 4, None),
                                   ("g2_1x256x256_r4", synth.g2(1, 256, 256), 4, None),
                                   ("g0_3x512x512_r6", synth.g0(3, 512, 512), 6, None),
                                   ("g2_3x192x160_r4", synth.g2(3, 160, 192), 4, None),
                                   ("g2_3x256x256_t128_r4", synth.g2(3, 256, 256), 4, 128),
                                   ("g2u16_1x128x128_r5", synth.g2(1, 128, 128, 12), 5, None)]:
        prec = 8 if px.dtype == np.uint8 else 12
        b, _ = R.encode(px, prec, TW=tile, TH=tile, numres=numres)
        open(os.path.join(HERE, name + ".j2k"), "wb").write(b)
        print(name, len(b))


if __name__ == "__main__":
    main()
