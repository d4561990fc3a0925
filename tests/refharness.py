"""ctypes loader for oracle/_ref/libref_harness.so (the REAL reference, test infrastructure only).

Available only where oracle/_ref was built (this container, or the prebuilt files shipped to the
GPU box).  Tests that need it call `have_ref()` and skip otherwise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REFDIR = os.path.join(_HERE, "..", "oracle", "_ref")
_lib = None


class EncCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("C", "W", "H", "TW", "TH", "prec", "irrev", "numres", "ht", "mode",
                 "rate_algo", "cblk_w", "cblk_h", "cblk_sty")]


def have_ref():
    return os.path.exists(os.path.join(_REFDIR, "libref_harness.so"))


def lib(threads=0):
    global _lib
    if _lib is None:
        C.CDLL(os.path.join(_REFDIR, "libgrokj2k_ref.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(os.path.join(_REFDIR, "libref_harness.so"))
        L.ref_init.restype = C.c_int
        L.ref_encode.restype = C.c_int64
        L.ref_encode.argtypes = [C.POINTER(EncCfg), C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_double), C.c_void_p]
        L.ref_decode.restype = C.c_int32
        L.ref_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.ref_ht_encode_block.restype = C.c_int32
        L.ref_ht_encode_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_uint32]
        L.ref_ht_decode_block.restype = C.c_int32
        L.ref_ht_decode_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        for f in ("ref_rct", "ref_ict"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
            getattr(L, f).restype = None
        for f in ("ref_dwt53_fwd", "ref_dwt97_fwd"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
            getattr(L, f).restype = None
        for f in ("ref_dwt53_row", "ref_dwt97_row"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint32, C.c_int]
            getattr(L, f).restype = None
        for f in ("ref_dwt53_fwd_at", "ref_dwt97_fwd_at"):
            getattr(L, f).argtypes = [C.c_void_p] + [C.c_uint32] * 6
            getattr(L, f).restype = None
        L.ref_abi_sizeof.restype = C.c_uint64
        L.ref_abi_sizeof.argtypes = [C.c_int]
        L.ref_init(threads, int(os.environ.get("REF_VERBOSE", "0")))
        _lib = L
    return _lib


def encode(pixels, prec, TW=None, TH=None, irrev=0, numres=6, ht=1, mode=0, rate_algo=0,
           plugin_tile=None, cblk=(0, 0), cblksty=0):
    """pixels: (C,H,W) uint8/uint16 array. Returns (bytes, seconds)."""
    L = lib()
    px = np.ascontiguousarray(pixels)
    Cn, H, W = px.shape
    cfg = EncCfg(Cn, W, H, TW or W, TH or H, prec, irrev, numres, ht, mode, rate_algo, cblk[0], cblk[1], cblksty)
    cap = px.size * 4 + (1 << 20)
    out = np.zeros(cap, np.uint8)
    secs = C.c_double(0)
    n = L.ref_encode(C.byref(cfg), px.ctypes.data, out.ctypes.data, cap, C.byref(secs),
                     plugin_tile)
    if n < 0:
        raise RuntimeError("ref_encode failed rc=%d" % n)
    return out[:n].tobytes(), secs.value


def encode_planes(planes, sampling, prec, W, H, TW=None, TH=None, numres=6, irrev=0, mct=0):
    """grk_compress of an image whose components are sub-sampled each in its own way (a raw / yuv image: grk_compress -F
    w,h,c,prec,u@1x1:2x2:2x2): planes = one 2-D array per component, sampling = [(dx, dy)]; W x H = the image area on the
    reference grid.  -> codestream bytes.  (Process-wide environment: REF_COMP_SUBSAMPLING / REF_TCP_MCT are set for the call.)"""
    L = lib()
    flat = np.concatenate([np.ascontiguousarray(p).reshape(-1) for p in planes])
    cfg = EncCfg(len(planes), W, H, TW or W, TH or H, prec, irrev, numres, 1, 1, 0, 0, 0, 0)
    cap = flat.size * flat.itemsize * 4 + (1 << 20)
    out = np.zeros(cap, np.uint8)
    secs = C.c_double(0)
    keep = {k: os.environ.get(k) for k in ("REF_COMP_SUBSAMPLING", "REF_TCP_MCT")}
    os.environ["REF_COMP_SUBSAMPLING"] = ",".join("%d,%d" % s for s in sampling)
    os.environ["REF_TCP_MCT"] = str(int(mct))
    try:
        n = L.ref_encode(C.byref(cfg), flat.ctypes.data, out.ctypes.data, cap, C.byref(secs), None)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if n < 0:
        raise RuntimeError("ref_encode failed rc=%d" % n)
    return out[:n].tobytes()


def decode_planes(j2k, sampling, W, H):
    """grk_decompress of such a stream -> one (h_c, w_c) int32 array per component (image area at the origin)."""
    L = lib()
    buf = np.frombuffer(j2k, np.uint8).copy()
    out = np.zeros((len(sampling), H, W), np.int32)
    keep = os.environ.get("REF_COMP_SUBSAMPLING")
    os.environ["REF_COMP_SUBSAMPLING"] = ",".join("%d,%d" % s for s in sampling)
    try:
        rc = L.ref_decode(buf.ctypes.data, buf.size, out.ctypes.data, len(sampling), W, H)
    finally:
        if keep is None:
            os.environ.pop("REF_COMP_SUBSAMPLING", None)
        else:
            os.environ["REF_COMP_SUBSAMPLING"] = keep
    if rc != 0:
        raise RuntimeError("ref_decode failed rc=%d" % rc)
    flat, res, at = out.reshape(-1), [], 0
    for dx, dy in sampling:
        w, h = (W + dx - 1) // dx, (H + dy - 1) // dy
        res.append(flat[at:at + w * h].reshape(h, w).copy())
        at += w * h
    return res


def decode(j2k, Cn, H, W):
    L = lib()
    buf = np.frombuffer(j2k, np.uint8).copy()
    out = np.zeros((Cn, H, W), np.int32)
    rc = L.ref_decode(buf.ctypes.data, buf.size, out.ctypes.data, Cn, W, H)
    if rc != 0:
        raise RuntimeError("ref_decode failed rc=%d" % rc)
    return out


def decode_window(j2k, Cn, x0, y0, x1, y1):
    """grk_decompress with grk_decompress_set_window(x0, y0, x1, y1) -> (C, y1 - y0, x1 - x0) int32."""
    L = lib()
    L.ref_decode_window.restype = C.c_int32
    L.ref_decode_window.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    buf = np.frombuffer(j2k, np.uint8).copy()
    out = np.zeros((Cn, y1 - y0, x1 - x0), np.int32)
    rc = L.ref_decode_window(buf.ctypes.data, buf.size, out.ctypes.data, Cn, x0, y0, x1, y1)
    if rc != 0:
        raise RuntimeError("ref_decode_window failed rc=%d" % rc)
    return out


def ht_encode_block(sm, kmax):
    """sm: (h,w) uint32 sign-magnitude MSB-aligned words."""
    L = lib()
    a = np.ascontiguousarray(sm, np.uint32)
    h, w = a.shape
    out = np.zeros(w * h * 4 + 4096, np.uint8)
    n = L.ref_ht_encode_block(a.ctypes.data, kmax, w, h, w, out.ctypes.data, out.size)
    assert n >= 0
    return out[:n].tobytes()


def ht_decode_block(coded, missing_msbs, w, h):
    L = lib()
    buf = np.frombuffer(coded, np.uint8).copy()
    out = np.zeros((h, w), np.uint32)
    rc = L.ref_ht_decode_block(buf.ctypes.data, buf.size, missing_msbs, w, h, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("ref_ht_decode_block failed")
    return out


def ht_decode_block_passes(coded, lengths1, lengths2, num_passes, missing_msbs, w, h):
    """ojph_decode_codeblock with the SigProp / MagRef segment: -> (h, w) uint32 words, or None when it rejects."""
    L = lib()
    L.ref_ht_decode_block_passes.restype = C.c_int32
    L.ref_ht_decode_block_passes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    buf = np.frombuffer(bytes(coded), np.uint8).copy()
    assert buf.size == lengths1 + lengths2
    out = np.zeros((h, w), np.uint32)
    rc = L.ref_ht_decode_block_passes(buf.ctypes.data, lengths1, lengths2, num_passes, missing_msbs, w, h, out.ctypes.data)
    return out if rc == 0 else None


def t1_encode_block(coef, orient):
    """Reference Part-1 (EBCOT) block encoder -> (bytes, numpasses, numbps)."""
    L = lib()
    L.ref_t1_encode_block.restype = C.c_int32
    L.ref_t1_encode_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    a = np.ascontiguousarray(coef, np.int32)
    h, w = a.shape
    out = np.zeros(w * h * 8 + 4096, np.uint8)
    npass, nbps = C.c_uint32(0), C.c_uint32(0)
    n = L.ref_t1_encode_block(a.ctypes.data, w, h, w, orient, out.ctypes.data, out.size, C.byref(npass), C.byref(nbps))
    assert n >= 0
    return out[:n].tobytes(), npass.value, nbps.value


def t1_decode_block(coded, numpasses, numbps, orient, w, h):
    L = lib()
    L.ref_t1_decode_block.restype = C.c_int32
    L.ref_t1_decode_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_void_p]
    buf = np.frombuffer(bytes(coded) + b"\0" * 8, np.uint8).copy()
    out = np.zeros((h, w), np.int32)
    rc = L.ref_t1_decode_block(buf.ctypes.data, len(coded), numpasses, numbps, orient, w, h, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("ref_t1_decode_block failed")
    return out


def plugin_dir():
    return os.path.abspath(os.path.join(_HERE, "..", "grok_amd", "lib"))


def plugin_load(threads=0):
    """Load OUR libgrokj2k_plugin.so through Grok's own minpf loader (grk_initialize(pluginPath))."""
    L = lib(threads)
    L.ref_plugin_load.restype = C.c_int
    L.ref_plugin_load.argtypes = [C.c_char_p, C.c_int]
    return L.ref_plugin_load(plugin_dir().encode(), threads)


def plugin_init(device=0, verbose=0):
    L = lib()
    L.ref_plugin_init.restype = C.c_int
    L.ref_plugin_init.argtypes = [C.c_int, C.c_int]
    return L.ref_plugin_init(device, verbose)


def warning_count(reset=True):
    """(warnings the reference has emitted since the last reset, text of the last one)"""
    L = lib()
    L.ref_warning_count.restype = C.c_int
    L.ref_warning_count.argtypes = [C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(256)
    n = L.ref_warning_count(int(reset), buf, 256)
    return n, buf.value.decode(errors="replace")


def plugin_debug_state():
    L = lib()
    L.ref_plugin_debug_state.restype = C.c_uint32
    return L.ref_plugin_debug_state()


def plugin_compress_file(pixels, prec, infile, numres=6, irrev=0, TW=None, TH=None):
    """grk_plugin_compress(params{infile}, host callback) -> bytes, or a negative refusal code."""
    L = lib()
    L.ref_plugin_compress_file.restype = C.c_int64
    L.ref_plugin_compress_file.argtypes = [C.POINTER(EncCfg), C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
    px = np.ascontiguousarray(pixels)
    Cn, H, W = px.shape
    cfg = EncCfg(Cn, W, H, TW or W, TH or H, prec, irrev, numres, 1, 1, 1, 0, 0)
    cap = px.size * 4 + (1 << 20)
    out = np.zeros(cap, np.uint8)
    n = L.ref_plugin_compress_file(C.byref(cfg), px.ctypes.data, infile.encode(), out.ctypes.data, cap)
    return out[:n].tobytes() if n >= 0 else int(n)


def plugin_batch_compress(in_dir, out_dir, numres=6, timeout_s=120):
    """grk_plugin_batch_compress over a directory of PNM files (what `grk_compress -y in -a out` does): -> number of files the
    host callback wrote, or a negative code."""
    L = lib()
    L.ref_plugin_batch_compress.restype = C.c_int32
    L.ref_plugin_batch_compress.argtypes = [C.POINTER(EncCfg), C.c_char_p, C.c_char_p, C.c_int]
    cfg = EncCfg(1, 1, 1, 1, 1, 8, 0, numres, 1, 1, 1, 0, 0)
    return int(L.ref_plugin_batch_compress(C.byref(cfg), in_dir.encode(), out_dir.encode(), timeout_s))


def plugin_decompress(j2k, Cn, H, W, as_file=True):
    """grk_plugin_decompress(params, host callback): Grok dlsym()s plugin_decompress in our .so and the two sides run the
    decode protocol (header -> T2 into the plugin's tile tree -> plugin decodes -> post-T1 -> clean).  as_file: the stream
    also lies in a file named by parameters->infile, as with `grk_decompress -i` (the plugin reads the main header's QCD
    and the file size from it; without a file it declines).
    -> ((C,H,W) int32 pixels, stage call counts) or (refusal code, stage call counts)."""
    import tempfile
    L = lib()
    L.ref_plugin_decompress.restype = C.c_int32
    L.ref_plugin_decompress.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_char_p]
    buf = np.frombuffer(j2k, np.uint8).copy()
    out = np.zeros((Cn, H, W), np.int32)
    stages = np.zeros(4, np.int32)
    path = None
    if as_file:
        fd, path = tempfile.mkstemp(suffix=".j2k")
        with os.fdopen(fd, "wb") as f:
            f.write(bytes(j2k))
    try:
        rc = L.ref_plugin_decompress(buf.ctypes.data, buf.size, out.ctypes.data, Cn, W, H, stages.ctypes.data,
                                     path.encode() if path else None)
    finally:
        if path:
            os.unlink(path)
    return (out if rc == 0 else int(rc)), [int(v) for v in stages]


def plugin_decompress_planes(j2k, sampling, W, H):
    """plugin_decompress for a stream whose components are sub-sampled each in its own way -> ([(h_c, w_c) int32], stages) or
    (refusal code, stages)."""
    keep = os.environ.get("REF_COMP_SUBSAMPLING")
    os.environ["REF_COMP_SUBSAMPLING"] = ",".join("%d,%d" % s for s in sampling)
    try:
        out, stages = plugin_decompress(j2k, len(sampling), H, W)
    finally:
        if keep is None:
            os.environ.pop("REF_COMP_SUBSAMPLING", None)
        else:
            os.environ["REF_COMP_SUBSAMPLING"] = keep
    if isinstance(out, int):
        return out, stages
    flat, res, at = out.reshape(-1), [], 0
    for dx, dy in sampling:
        w, h = (W + dx - 1) // dx, (H + dy - 1) // dy
        res.append(flat[at:at + w * h].reshape(h, w).copy())
        at += w * h
    return res, stages


def plugin_batch_decompress(in_dir, out_dir, timeout_s=60):
    L = lib()
    L.ref_plugin_batch_decompress.restype = C.c_int32
    L.ref_plugin_batch_decompress.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    return L.ref_plugin_batch_decompress(in_dir.encode(), out_dir.encode(), timeout_s)


def read_batch_output(path):
    """what the harness's batch callback writes: 'C W H\\n' + int32 planes -> (C, H, W) int32"""
    with open(path, "rb") as f:
        Cn, W, H = [int(v) for v in f.readline().split()]
        return np.frombuffer(f.read(), np.int32).reshape(Cn, H, W)


def write_pnm(path, px, prec):
    Cn, H, W = px.shape
    assert Cn in (1, 3)
    with open(path, "wb") as f:
        f.write(b"P%d\n%d %d\n%d\n" % (5 if Cn == 1 else 6, W, H, (1 << prec) - 1))
        inter = np.ascontiguousarray(np.moveaxis(px, 0, -1))
        f.write(inter.astype(">u2").tobytes() if prec > 8 else inter.astype(np.uint8).tobytes())


def t1_encode_block_sty(coef, orient, cblksty):
    """Reference Part-1 block encoder with a code-block style -> (bytes, [(segment length, passes)...], numbps)."""
    L = lib()
    L.ref_t1_encode_block_sty.restype = C.c_int32
    L.ref_t1_encode_block_sty.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                          C.c_uint32]
    a = np.ascontiguousarray(coef, np.int32)
    h, w = a.shape
    out = np.zeros(w * h * 8 + 4096, np.uint8)
    rate = np.zeros(128, np.uint32)
    term = np.zeros(128, np.uint8)
    npass, nbps = C.c_uint32(0), C.c_uint32(0)
    n = L.ref_t1_encode_block_sty(a.ctypes.data, w, h, w, orient, cblksty, out.ctypes.data, out.size, C.byref(npass),
                                  C.byref(nbps), rate.ctypes.data, term.ctypes.data, rate.size)
    assert n >= 0
    segs, start, first = [], 0, 0
    for i in range(npass.value):
        if term[i] or i == npass.value - 1:
            segs.append((int(rate[i]) - start, i + 1 - first))
            start, first = int(rate[i]), i + 1
    return out[:n].tobytes(), segs, nbps.value


def t1_decode_block_sty(coded, segs, numbps, orient, cblksty, w, h):
    L = lib()
    L.ref_t1_decode_block_sty.restype = C.c_int32
    L.ref_t1_decode_block_sty.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_void_p]
    buf = np.frombuffer(bytes(coded) + b"\0" * 8, np.uint8).copy()
    sl = np.array([a for a, _ in segs], np.uint32)
    sp = np.array([b for _, b in segs], np.uint32)
    out = np.zeros((h, w), np.int32)
    rc = L.ref_t1_decode_block_sty(buf.ctypes.data, len(segs), sl.ctypes.data, sp.ctypes.data, numbps, orient, cblksty, w, h,
                                   out.ctypes.data)
    if rc != 0:
        raise RuntimeError("ref_t1_decode_block_sty failed")
    return out
