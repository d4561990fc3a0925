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
    """s_{i+1} = s_i*1664525 + 1013904223 mod 2^32, returns s_1..s_n as uint32 (vectorised by doubling;
    uint32 arithmetic wraps, which IS the mod 2^32)."""
    a = np.uint32(1664525)
    c = np.uint32(1013904223)
    out = np.empty(n, np.uint32)
    with np.errstate(over="ignore"):
        out[0] = np.uint32(seed & 0xFFFFFFFF) * a + c
        filled = 1
        ak, ck = a, c          # composition of `filled` steps: s -> ak*s + ck
        while filled < n:
            k = min(filled, n - filled)
            np.multiply(out[:k], ak, out=out[filled:filled + k])
            out[filled:filled + k] += ck
            # square the step map
            ck = ak * ck + ck
            ak = ak * ak
            filled += k
    return out


def g2(C, H, W, prec=8, seed=12345):
    s = _lcg_all(C * H * W, seed).reshape(C, H, W)
    s >>= np.uint32(24)
    s &= np.uint32(7)
    k, d = (1 << prec) - 16, W + H + 74
    wide = (W + H + 37 * C) * k >= 1 << 32       # (never for the sizes in use: 16384^2 x 16 bit is 2.2e9)
    t = np.uint64 if wide else np.uint32
    xy = np.arange(H, dtype=t)[:, None] + np.arange(W, dtype=t)[None, :]
    out = np.empty((C, H, W), np.uint8 if prec <= 8 else np.uint16)
    for c in range(C):
        v = (xy + t(37 * c)) * t(k) // t(d)
        v += s[c]
        out[c] = v
    return out


def psnr_db(a, b, prec):
    """PSNR of two sample arrays against the full scale of `prec` bits (inf when equal)."""
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean())
    return float("inf") if mse == 0.0 else float(10.0 * np.log10(float((1 << prec) - 1) ** 2 / mse))


def g2_mid(C, H, W, prec, seed=12345):
    """G2 clipped into the middle three quarters of the range.  The reference's HT DECODER refuses a block whose top magnitude
    needs all Kmax bit-planes (U_q > missing_msbs, ojph_block_decoder.cpp:1194: defect D5), and with the irreversible path's default
    step sizes the LL block under plain G2's darkest corner is such a block from 1024 x 1024 x 16-bit on; clipped, the stream
    is one grk_decompress accepts at every size (the coded size stays within 6 % of plain G2's)."""
    a = g2(C, H, W, prec, seed=seed)
    return np.clip(a, (1 << prec) // 8, 7 * (1 << prec) // 8).astype(a.dtype)
