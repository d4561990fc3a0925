"""ctypes wrapper around oracle/liboracle_j2k.so (the plain-C CPU restatement; test infra only)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ODIR = os.path.join(_HERE, "..", "oracle")
_lib = None


class Block(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
                ("res", C.c_uint8), ("band", C.c_uint8), ("kmax", C.c_uint8), ("comp", C.c_uint8),
                ("bx", C.c_uint32), ("by", C.c_uint32)]


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_ODIR, "liboracle_j2k.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _ODIR, "oracle"])
        L = C.CDLL(so)
        L.orc_ht_encode_sm.restype = C.c_int32
        L.orc_ht_encode_sm.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_ht_encode_block_rev.restype = C.c_int32
        L.orc_ht_encode_block_rev.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_enumerate_blocks.restype = C.c_uint32
        L.orc_enumerate_blocks.argtypes = [C.c_uint32] * 4 + [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_encode_tile_rev.restype = C.c_int32
        L.orc_encode_tile_rev.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32,
                                          C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        for f in ("orc_rct_fwd", "orc_ict_fwd", "orc_rct_inv"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
            getattr(L, f).restype = None
        for f in ("orc_dwt53_fwd", "orc_dwt97_fwd", "orc_dwt53_inv"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
            getattr(L, f).restype = None
        for f in ("orc_dwt53_fwd_1d", "orc_dwt97_fwd_1d"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint32]
            getattr(L, f).restype = None
        L.orc_ht_rev_exponents.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_ht_rev_exponents.restype = None
        L.orc_ht_irrev_stepsizes.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_ht_irrev_stepsizes.restype = None
        L.orc_ingest.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32]
        L.orc_ingest.restype = None
        L.orc_ht_signmag_rev.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_ht_signmag_rev.restype = None
        L.orc_ht_signmag_irrev.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p]
        L.orc_ht_signmag_irrev.restype = None
        L.orc_ht_block_distortion.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint32,
                                              C.c_double]
        L.orc_ht_block_distortion.restype = C.c_double
        L.orc_ht_decode_block.restype = C.c_int32
        L.orc_ht_decode_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_ht_dequant_rev.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_ht_dequant_rev.restype = None
        L.orc_ht_dequant_irrev.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]
        L.orc_ht_dequant_irrev.restype = None
        L.orc_dwt97_inv.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_dwt97_inv.restype = None
        for f in ("orc_rct_inv_store", "orc_ict_inv_store"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32]
            getattr(L, f).restype = None
        for f in ("orc_dc_store_rev", "orc_dc_store_irrev"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32]
            getattr(L, f).restype = None
        L.orc_t1_decode_block.restype = C.c_int32
        L.orc_t1_decode_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_void_p]
        L.orc_t1_decode_block_sty.restype = C.c_int32
        L.orc_t1_decode_block_sty.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_t1_dequant_rev.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_t1_dequant_rev.restype = None
        L.orc_t1_dequant_irrev.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]
        L.orc_t1_dequant_irrev.restype = None
        u32 = C.c_uint32
        for f in ("orc_dwt53_fwd_at", "orc_dwt97_fwd_at", "orc_dwt53_inv_at", "orc_dwt97_inv_at"):
            getattr(L, f).argtypes = [C.c_void_p] + [u32] * 6
            getattr(L, f).restype = None
        for f in ("orc_dwt53_fwd_1d_par", "orc_dwt97_fwd_1d_par"):
            getattr(L, f).argtypes = [C.c_void_p, u32, u32]
            getattr(L, f).restype = None
        L.orc_enumerate_blocks_at.restype = u32
        L.orc_enumerate_blocks_at.argtypes = [u32] * 6 + [C.c_void_p, C.c_void_p, u32]
        L.orc_enumerate_blocks_prc.restype = u32
        L.orc_enumerate_blocks_prc.argtypes = [u32] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, u32]
        L.orc_encode_tile_rev_prc.restype = C.c_int32
        L.orc_encode_tile_rev_prc.argtypes = [C.c_void_p, C.c_int, u32, u32, u32, u32, u32, C.c_int, u32, u32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, u32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.orc_encode_tile_rev_at.restype = C.c_int32
        L.orc_encode_tile_rev_at.argtypes = [C.c_void_p, C.c_int, u32, u32, u32, u32, u32, C.c_int, u32, u32, C.c_void_p,
                                             C.c_void_p, u32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def t1_decode_block_sty(coded, segs, numbps, orient, cblksty, w, h):
    """Part-1 block decode, general form: segs = [(length, passes), ...] -> ((h, w) int32, bad segmentation symbols)."""
    buf = np.frombuffer(bytes(coded) + b"\0" * 8, np.uint8).copy()
    sl = np.array([a for a, _ in segs], np.uint32)
    sp = np.array([b for _, b in segs], np.uint32)
    out = np.zeros((h, w), np.int32)
    rc = lib().orc_t1_decode_block_sty(buf.ctypes.data, len(segs), sl.ctypes.data, sp.ctypes.data, numbps, orient, cblksty, w, h,
                                       out.ctypes.data)
    return (out, rc) if rc >= 0 else (None, rc)


def t1_decode_block(coded, numpasses, numbps, orient, w, h):
    """Part-1 block decode -> (h, w) int32 in the decoder's representation, or None if rejected."""
    buf = np.frombuffer(bytes(coded) + b"\0\0", np.uint8).copy()
    out = np.zeros((h, w), np.int32)
    rc = lib().orc_t1_decode_block(buf.ctypes.data, len(coded), numpasses, numbps, orient, w, h, out.ctypes.data)
    return out if rc == 0 else None


def t1_dequant_rev(v):
    a = np.ascontiguousarray(v, np.int32)
    out = np.zeros(a.shape, np.int32)
    lib().orc_t1_dequant_rev(a.ctypes.data, a.size, out.ctypes.data)
    return out


def t1_dequant_irrev(v, stepsize):
    a = np.ascontiguousarray(v, np.int32)
    out = np.zeros(a.shape, np.float32)
    lib().orc_t1_dequant_irrev(a.ctypes.data, a.size, float(stepsize), out.ctypes.data)
    return out


def ht_decode_block(coded, missing_msbs, w, h):
    """-> (h, w) uint32 sign-magnitude words, or None for a stream the decoder rejects."""
    buf = np.frombuffer(bytes(coded), np.uint8).copy()
    out = np.zeros((h, w), np.uint32)
    rc = lib().orc_ht_decode_block(buf.ctypes.data, buf.size, missing_msbs, w, h, out.ctypes.data, w)
    return out if rc == 0 else None


def ht_refine_encode(mag, sign, npasses):
    """mag: (h, w) magnitudes whose LSB is bit-plane p - 1 (the cleanup pass codes mag >> 1), sign: (h, w) 0 / 1.
    -> (refinement segment bytes, length of its SigProp part)"""
    L = lib()
    L.orc_ht_refine_encode.restype = C.c_int32
    L.orc_ht_refine_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.POINTER(C.c_uint32)]
    m = np.ascontiguousarray(mag, np.uint32)
    sg = np.ascontiguousarray(sign, np.uint8)
    h, w = m.shape
    out = np.zeros(w * h + 64, np.uint8)
    spp = C.c_uint32(0)
    n = L.orc_ht_refine_encode(m.ctypes.data, sg.ctypes.data, w, h, npasses, out.ctypes.data, out.size, C.byref(spp))
    assert n >= 0
    return out[:n].tobytes(), spp.value


def ht_refine_decode(words, missing_msbs, seg, npasses):
    """words: (h, w) uint32 output of the cleanup pass; returns them refined by the passes in `seg`."""
    L = lib()
    L.orc_ht_refine_decode.restype = C.c_int32
    L.orc_ht_refine_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    o = np.ascontiguousarray(words, np.uint32).copy()
    h, w = o.shape
    buf = np.frombuffer(bytes(seg) + b"\0" * 8, np.uint8).copy()
    rc = L.orc_ht_refine_decode(o.ctypes.data, w, h, w, missing_msbs, buf.ctypes.data, len(seg), npasses)
    assert rc == 0
    return o


def ht_dequant_rev(sm, k_msbs):
    a = np.ascontiguousarray(sm, np.uint32)
    out = np.zeros(a.shape, np.int32)
    lib().orc_ht_dequant_rev(a.ctypes.data, a.size, k_msbs, out.ctypes.data)
    return out


def ht_dequant_irrev(sm, scale):
    a = np.ascontiguousarray(sm, np.uint32)
    out = np.zeros(a.shape, np.float32)
    lib().orc_ht_dequant_irrev(a.ctypes.data, a.size, float(scale), out.ctypes.data)
    return out


def dwt97_inv(plane, levels, origin=(0, 0)):
    p = np.ascontiguousarray(plane, np.float32).copy()
    h, w = p.shape
    lib().orc_dwt97_inv_at(p.ctypes.data, w, h, w, levels, origin[0], origin[1])
    return p


def color_inv_store(planes, prec, irrev, mct, sgnd=False):
    """planes: list of (H,W) int32 arrays (float bit patterns when irrev). Returns clamped int32 pixels."""
    shift = 0 if sgnd else 1 << (prec - 1)
    lo, hi = (-(1 << (prec - 1)), (1 << (prec - 1)) - 1) if sgnd else (0, (1 << prec) - 1)
    a = [np.ascontiguousarray(p, np.int32).copy() for p in planes]
    L = lib()
    k0 = 0
    if mct and len(a) >= 3:
        f = L.orc_ict_inv_store if irrev else L.orc_rct_inv_store
        f(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[0].size, shift, lo, hi)
        k0 = 3
    for k in range(k0, len(a)):
        (L.orc_dc_store_irrev if irrev else L.orc_dc_store_rev)(a[k].ctypes.data, a[k].size, shift, lo, hi)
    return a


def signmag(coeffs, kmax):
    c = np.asarray(coeffs, np.int64)
    return (np.where(c < 0, 0x80000000, 0) | (np.abs(c) << (30 - kmax))).astype(np.uint32)


def ht_block_distortion(sm, kmax, orient, level, reversible, mct, comp, stepsize):
    sm = np.ascontiguousarray(sm, np.uint32)
    return float(lib().orc_ht_block_distortion(sm.ctypes.data, sm.size, kmax, orient, level, int(reversible), int(mct), comp, float(stepsize)))


def ht_encode_sm(sm, kmax):
    a = np.ascontiguousarray(sm, np.uint32)
    h, w = a.shape
    out = np.zeros(w * h * 4 + 8192, np.uint8)
    n = lib().orc_ht_encode_sm(a.ctypes.data, kmax, w, h, out.ctypes.data, out.size)
    assert n >= 0
    return out[:n].tobytes()


def rev_exponents(prec, levels):
    e = np.zeros(3 * levels + 1, np.uint8)
    lib().orc_ht_rev_exponents(prec, levels, e.ctypes.data)
    return e


def irrev_stepsizes(prec, levels):
    q = np.zeros(3 * levels + 1, np.uint16)
    d = np.zeros(3 * levels + 1, np.float32)
    lib().orc_ht_irrev_stepsizes(prec, levels, q.ctypes.data, d.ctypes.data)
    return q, d


def _prc(precincts, levels):
    """[(PPx, PPy)] per resolution (0 = coarsest) -> the COD bytes, or None"""
    if precincts is None:
        return None
    assert len(precincts) == levels + 1
    return np.array([ppx | (ppy << 4) for ppx, ppy in precincts], np.uint8)


def enumerate_blocks(w, h, levels, expn=None, cblk_exp=6, origin=(0, 0), precincts=None):
    L = lib()
    e = None if expn is None else np.ascontiguousarray(expn, np.uint8)
    ep = e.ctypes.data if e is not None else None
    pr = _prc(precincts, levels)
    pp = pr.ctypes.data if pr is not None else None
    n = L.orc_enumerate_blocks_prc(w, h, levels, cblk_exp, origin[0], origin[1], pp, ep, None, 0)
    arr = (Block * n)()
    L.orc_enumerate_blocks_prc(w, h, levels, cblk_exp, origin[0], origin[1], pp, ep, arr, n)
    return list(arr)


def dwt53_fwd(plane, levels, origin=(0, 0)):
    p = np.ascontiguousarray(plane, np.int32).copy()
    h, w = p.shape
    lib().orc_dwt53_fwd_at(p.ctypes.data, w, h, w, levels, origin[0], origin[1])
    return p


def dwt97_fwd(plane, levels, origin=(0, 0)):
    p = np.ascontiguousarray(plane, np.float32).copy()
    h, w = p.shape
    lib().orc_dwt97_fwd_at(p.ctypes.data, w, h, w, levels, origin[0], origin[1])
    return p


def dwt53_inv(plane, levels, origin=(0, 0)):
    p = np.ascontiguousarray(plane, np.int32).copy()
    h, w = p.shape
    lib().orc_dwt53_inv_at(p.ctypes.data, w, h, w, levels, origin[0], origin[1])
    return p


def dwt_row(row, par, irrev=False):
    """One line of the forward transform whose first sample lies on a coordinate of parity `par`."""
    a = np.ascontiguousarray(row, np.float32 if irrev else np.int32).copy()
    (lib().orc_dwt97_fwd_1d_par if irrev else lib().orc_dwt53_fwd_1d_par)(a.ctypes.data, a.size, par)
    return a


def rct_fwd(r, g, b):
    a = [np.ascontiguousarray(v, np.int32).copy() for v in (r, g, b)]
    lib().orc_rct_fwd(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[0].size)
    return a


def ict_fwd(r, g, b):
    a = [np.ascontiguousarray(v, np.int32).copy() for v in (r, g, b)]
    lib().orc_ict_fwd(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[0].size)
    return [v.view(np.float32) for v in a]


def encode_tile_rev(pixels, prec, levels, mct=None, origin=(0, 0), precincts=None):
    """pixels (C,H,W) u8/u16 -> (blocks, lens, coded bytes) in reference enumeration order."""
    px = np.ascontiguousarray(pixels)
    Cn, H, W = px.shape
    if mct is None:
        mct = Cn >= 3
    L = lib()
    pr = _prc(precincts, levels)
    pp = pr.ctypes.data if pr is not None else None
    nb = L.orc_enumerate_blocks_prc(W, H, levels, 6, origin[0], origin[1], pp, None, None, 0) * Cn
    blocks = (Block * nb)()
    lens = np.zeros(nb, np.uint32)
    cap = px.size * 4 + nb * 64 + (1 << 16)
    coded = np.zeros(cap, np.uint8)
    tot = C.c_uint64(0)
    n = L.orc_encode_tile_rev_prc(px.ctypes.data, px.dtype.itemsize, Cn, W, H, prec, levels, int(mct), origin[0], origin[1], pp,
                                  blocks, lens.ctypes.data, nb, coded.ctypes.data, cap, C.byref(tot))
    assert n == nb, n
    return list(blocks), lens, coded[:tot.value]


def _bind_model():
    L = lib()
    vp, u32 = C.c_void_p, C.c_uint32
    L.orc_ht_raw_streams.restype = C.c_int32
    L.orc_ht_raw_streams.argtypes = [vp, u32, u32, u32, vp, u32, C.POINTER(u32), vp, u32, C.POINTER(u32), vp, vp]
    L.orc_ht_model_phase_b.restype = C.c_int32
    L.orc_ht_model_phase_b.argtypes = [vp, u32, vp, u32, vp, vp, vp]
    L.orc_ht_model_phase_b2.restype = C.c_int32
    L.orc_ht_model_phase_b2.argtypes = [vp, u32, vp, u32, vp, vp, vp]
    L.orc_ht_model_phase_b3.restype = C.c_int32
    L.orc_ht_model_phase_b3.argtypes = [vp, u32, vp, u32, vp, vp, vp]
    return L


def ht_wave_model(sm, kmax, form=1):
    """raw streams of the oracle encoder -> wave-parallel phase-B model -> bytes (form 1: walker + bitmaps, r01; form 2: speculative
    windows, r03 -- what kernels_ht.hip runs for deep content; form 3: the VLC windows run once, forwards, into a staging area, r05 --
    what it runs for packed 8-bit content)"""
    L = _bind_model()
    a = np.ascontiguousarray(sm, np.uint32)
    h, w = a.shape
    msw = w * h * (kmax + 2) // 32 + 8
    vw = 1024
    ms = np.zeros(msw, np.uint32)
    vl = np.zeros(vw, np.uint32)
    mel = np.zeros(512, np.uint8)
    st = (C.c_int * 4)()
    mb, vb = C.c_uint32(0), C.c_uint32(0)
    nref = L.orc_ht_raw_streams(a.ctypes.data, kmax, w, h, ms.ctypes.data, msw, C.byref(mb),
                                vl.ctypes.data, vw, C.byref(vb), mel.ctypes.data, st)
    out = np.zeros(nref + 64, np.uint8)
    nm = {1: L.orc_ht_model_phase_b, 2: L.orc_ht_model_phase_b2, 3: L.orc_ht_model_phase_b3}[form](ms.ctypes.data, mb.value, vl.ctypes.data, vb.value, mel.ctypes.data, st, out.ctypes.data)
    return out[:nm].tobytes()
