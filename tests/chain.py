"""CPU encode/decode chains built from the oracle's stage functions and the product's host-only
geometry + codestream writer (no GPU needed).  Used to pin the decode oracle against the real
reference decoder and, on the GPU box, as the expected result of the HIP decode path."""
import ctypes as C
import numpy as np

import grok_amd as G
import grok_amd.capi
import oracle as O


def band_scale_dec(prec, qcd_word, kmax):
    """Decoder-side step of an irreversible HT band (codestream/Quantizer.cpp:41-63 with compress=false)."""
    expn, mant = qcd_word >> 11, qcd_word & 0x7FF
    step = np.float32((1.0 + mant / 2048.0) * 2.0 ** (prec - expn))
    return np.float32(step / np.float32(1 << (31 - kmax)))


def band_index(b):
    return 0 if b.res == 0 else 3 * b.res - 2 + (b.band - 1)


def encode_tile_oracle(px, prec, levels, irrev=False, mct=None, sgnd=False):
    """px: (C,H,W) pixels (int8/int16 when sgnd). Returns (params, blocks, qcd, table, coded bytes)."""
    Cn, H, W = px.shape
    if mct is None:
        mct = Cn >= 3
    p = G.TileParams.make(W, H, Cn, prec, levels, irreversible=irrev, mct=mct, sgnd=sgnd)
    blocks, qcd = G.tile_layout(p)
    planes = [px[c].astype(np.int32) - (0 if sgnd else 1 << (prec - 1)) for c in range(Cn)]
    if irrev:
        if mct:
            planes[:3] = [v.view(np.float32) for v in O.ict_fwd(*planes[:3])]
            planes[3:] = [v.astype(np.float32) for v in planes[3:]]
        else:
            planes = [v.astype(np.float32) for v in planes]
        mall = [O.dwt97_fwd(v, levels) for v in planes]
    else:
        if mct:
            planes[:3] = O.rct_fwd(*planes[:3])
        mall = [O.dwt53_fwd(v, levels) for v in planes]
    table = np.zeros(len(blocks), G.capi.CODED_DTYPE)
    chunks, off = [], 0
    L = O.lib()
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        sub = np.ascontiguousarray(mall[b.comp][b.py:b.py + bh, b.px:b.px + bw])
        if irrev:
            sm = np.zeros((bh, bw), np.uint32)
            L.orc_ht_signmag_irrev(sub.ctypes.data, bw, bw, bh, b.kmax, C.c_float(np.float32(1.0) / np.float32(b.stepsize)),
                                   sm.ctypes.data)
        else:
            sm = O.signmag(sub, b.kmax)
        cb = O.ht_encode_sm(sm, b.kmax)
        table["offset"][i] = off
        table["length"][i] = len(cb)
        chunks.append(cb)
        off += len(cb)
    return p, blocks, qcd, table, b"".join(chunks)


def decode_tile_oracle(p, blocks, qcd, table, coded, stop_after=None):
    """Inverse chain: HT block decode -> dequant -> inverse DWT -> inverse MCT + DC + clamp.
    Returns (C,H,W) int32 pixels (or the Mallat planes when stop_after == 'mallat')."""
    Cn, H, W, prec, levels = p.num_comps, p.tile_h, p.tile_w, p.prec, p.num_levels
    irrev = bool(p.irreversible)
    mall = [np.zeros((H, W), np.float32 if irrev else np.int32) for _ in range(Cn)]
    coded = np.frombuffer(coded, np.uint8)
    for i, b in enumerate(blocks):
        bw, bh = b.x1 - b.x0, b.y1 - b.y0
        o, n = int(table["offset"][i]), int(table["length"][i])
        sm = O.ht_decode_block(coded[o:o + n].tobytes(), b.kmax - 1, bw, bh)
        assert sm is not None, "block %d rejected" % i
        if irrev:
            v = O.ht_dequant_irrev(sm, band_scale_dec(prec, qcd[band_index(b)], b.kmax))
        else:
            v = O.ht_dequant_rev(sm, b.kmax - 1)
        mall[b.comp][b.py:b.py + bh, b.px:b.px + bw] = v
    if stop_after == "mallat":
        return mall
    planes = [O.dwt97_inv(m, levels) if irrev else O.dwt53_inv(m, levels) for m in mall]
    if stop_after == "idwt":
        return planes
    planes = [pl.view(np.int32) if irrev else pl for pl in planes]
    return np.stack(O.color_inv_store(planes, prec, irrev, bool(p.mct), sgnd=bool(p.sgnd)))
