"""Synthetic tile generators shared by tests, bench.py and the golden-vector script.

G0 = the reference test's own ramp `data[i] = (uint8_t)i` (tests/test_tile_encoder.cpp:103-104).
G2 = bounded gradient + LCG noise, decodable by the reference (D5).
"""
import numpy as np


def g0(C, H, W, prec=8):
    n = C * H * W
    bps = (prec + 7) // 8
    raw = (np.arange(n * bps, dtype=np.uint64) & 0xFF).astype(np.uint8)
    if bps == 1:
        return raw.reshape(C, H, W)
    return raw.view(np.uint16).reshape(C, H, W)


def _lcg_all(n, seed=12345):
    """s_{i+1} = s_i*1664525 + 1013904223 mod 2^32, returns s_1..s_n (vectorised by doubling)."""
    a = np.uint64(1664525)
    c = np.uint64(1013904223)
    m = np.uint64(0xFFFFFFFF)
    out = np.empty(n, np.uint64)
    out[0] = (np.uint64(seed) * a + c) & m
    filled = 1
    ak, ck = a, c          # composition of `filled` steps: s -> ak*s + ck
    while filled < n:
        k = min(filled, n - filled)
        out[filled:filled + k] = (out[:k] * ak + ck) & m
        # square the step map
        ck = (ak * ck + ck) & m
        ak = (ak * ak) & m
        filled += k
    return out


def g2(C, H, W, prec=8, seed=12345):
    s = _lcg_all(C * H * W, seed).reshape(C, H, W)
    y, x = np.meshgrid(np.arange(H, dtype=np.uint64), np.arange(W, dtype=np.uint64), indexing="ij")
    c = np.arange(C, dtype=np.uint64).reshape(C, 1, 1)
    v = ((x + y + np.uint64(37) * c) * np.uint64((1 << prec) - 16)) // np.uint64(W + H + 74) + ((s >> np.uint64(24)) & np.uint64(7))
    return v.astype(np.uint8 if prec <= 8 else np.uint16)
