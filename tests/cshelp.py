"""Test helper: oracle tile encode -> product Tier-2 writer -> whole codestream (CPU only)."""
import numpy as np

import grok_amd as G
import oracle as O
from grok_amd.capi import CODED_DTYPE


def oracle_codestream(px, prec, L, TW=None, TH=None, flags=0):
    C, H, W = px.shape
    TW = TW or W
    TH = TH or H
    p = G.TileParams.make(TW, TH, C, prec, L)
    tabs, chunks, off = [], [], 0
    for ty in range(H // TH):
        for tx in range(W // TW):
            tile = np.ascontiguousarray(px[:, ty * TH:(ty + 1) * TH, tx * TW:(tx + 1) * TW])
            blocks, lens, coded = O.encode_tile_rev(tile, prec, L)
            t = np.zeros(len(lens), CODED_DTYPE)
            t["length"] = lens
            t["offset"] = off + np.concatenate([[0], np.cumsum(lens)[:-1]])
            off += int(lens.sum())
            tabs.append(t)
            chunks.append(coded)
    return G.write_codestream(p, W, H, np.concatenate(tabs), np.concatenate(chunks), flags)
