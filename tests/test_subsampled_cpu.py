"""CPU: the codestream writer for images with sub-sampled components (grk_amd_write_codestream_subsampled: per-component
tile-components ceil(tile / dx), tile/TileProcessor.cpp:605-612; SIZ XRsiz / YRsiz) over the ORACLE's blocks == the file
grk_compress writes for the same planes.  (The GPU path of the same images: tests/test_gpu_subsampled.py.)"""
import ctypes as C

import numpy as np
import pytest

import grok_amd as G
import oracle as O
import refharness as R
import synth
from grok_amd.capi import CODED_DTYPE

pytestmark = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")


def _planes(W, H, sampling, prec):
    return [synth.g2(1, (H + dy - 1) // dy, (W + dx - 1) // dx, prec, seed=50 + 3 * c)[0] for c, (dx, dy) in enumerate(sampling)]


@pytest.mark.parametrize("W,H,sampling,prec,L,tile,order", [
    (256, 192, [(1, 1), (2, 2), (2, 2)], 8, 4, None, 0),
    (300, 201, [(1, 1), (2, 1), (2, 1)], 8, 3, None, 2),
    (320, 256, [(1, 1), (2, 2), (2, 2)], 8, 3, (128, 128), 3),
    (200, 150, [(1, 1), (2, 2), (2, 2), (1, 1)], 12, 2, (100, 75), 4),
])
def test_writer_over_oracle_blocks_equals_grk_compress(W, H, sampling, prec, L, tile, order, monkeypatch):
    monkeypatch.setenv("REF_PROG_ORDER", str(order))
    planes = _planes(W, H, sampling, prec)
    TW, TH = tile or (W, H)
    want = R.encode_planes(planes, sampling, prec, W, H, TW=TW, TH=TH, numres=L + 1, mct=0)
    Lh = G.lib()
    Lh.grk_amd_layout_tile_comp.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    Lh.grk_amd_write_codestream_subsampled.restype = C.c_int64
    Lh.grk_amd_write_codestream_subsampled.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_void_p, C.c_uint64]
    layout = G.ImageLayout.make(W, H, TW, TH)
    base = G.TileParams.make(1, 1, len(sampling), prec, L, mct=False)
    ntiles = Lh.grk_amd_layout_num_tiles(C.byref(layout))
    tabs, chunks, off = [], [], 0
    for t in range(ntiles):
        for c, (dx, dy) in enumerate(sampling):
            p = G.TileParams()
            assert Lh.grk_amd_layout_tile_comp(C.addressof(layout), C.addressof(base), dx, dy, t, C.addressof(p)) == 0
            sub = np.ascontiguousarray(planes[c][p.tile_y0:p.tile_y0 + p.tile_h, p.tile_x0:p.tile_x0 + p.tile_w])[None]
            _, lens, coded = O.encode_tile_rev(sub, prec, L, mct=False, origin=(p.tile_x0, p.tile_y0))
            tt = np.zeros(len(lens), CODED_DTYPE)
            tt["length"] = lens
            tt["offset"] = off + np.concatenate([[0], np.cumsum(lens)[:-1]])
            off += int(lens.sum())
            tabs.append(tt)
            chunks.append(coded)
    table, coded = np.concatenate(tabs), np.concatenate(chunks)
    dxs = (C.c_uint8 * len(sampling))(*[a for a, _ in sampling])
    dys = (C.c_uint8 * len(sampling))(*[b for _, b in sampling])
    out = np.empty(coded.size + len(table) * 8 + (1 << 20), np.uint8)
    n = Lh.grk_amd_write_codestream_subsampled(C.addressof(layout), C.addressof(base), C.addressof(dxs), C.addressof(dys), table.ctypes.data,
                                              coded.ctypes.data, G.CS_PROG(order), out.ctypes.data, out.size)
    assert n > 0
    assert out[:n].tobytes() == want
    back = R.decode_planes(want, sampling, W, H)
    for a, b in zip(back, planes):
        assert np.array_equal(a, b.astype(np.int32))
