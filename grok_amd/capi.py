"""ctypes binding of include/grok_amd.h (no compute in Python; fails loudly without the .so)."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS = 10


class NativeLibraryMissing(RuntimeError):
    pass


def lib_path():
    # (GRK_AMD_LIB: another build of the same library, for A/B timing on one box -- tools/build_variant.sh)
    return os.environ.get("GRK_AMD_LIB") or os.path.join(_HERE, "lib", "libgrok_amd.so")


class TileParams(C.Structure):
    _fields_ = [("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("num_comps", C.c_uint16),
                ("prec", C.c_uint8), ("sgnd", C.c_uint8), ("irreversible", C.c_uint8),
                ("mct", C.c_uint8), ("num_levels", C.c_uint8), ("cblk_w_exp", C.c_uint8),
                ("cblk_h_exp", C.c_uint8), ("reserved", C.c_uint8 * 3),
                ("tile_x0", C.c_uint32), ("tile_y0", C.c_uint32), ("precinct_exp", C.c_uint8 * 12)]

    @classmethod
    def make(cls, w, h, comps, prec, levels, irreversible=False, mct=None, sgnd=False, cblk=(6, 6), part1=False, cblksty=0,
             origin=(0, 0), precincts=None):
        """origin = (x0, y0): where the tile lies on the canonical grid (image offset / tile grid position).
        precincts = [(PPx, PPy), ...] for resolutions 0 (coarsest) .. levels, exponents as in the COD marker; None: one
        precinct per resolution."""
        if mct is None:
            mct = comps >= 3
        p = cls(w, h, comps, prec, int(sgnd), int(irreversible), int(mct), levels, cblk[0], cblk[1])
        p.reserved[0] = int(part1)
        p.reserved[1] = int(cblksty)       # Part-1 decode: LAZY 1, RESET 2, TERMALL 4, VSC 8, PTERM 16, SEGSYM 32
        p.tile_x0, p.tile_y0 = int(origin[0]), int(origin[1])
        if precincts is not None:
            assert len(precincts) == levels + 1
            for r, (ppx, ppy) in enumerate(precincts):
                p.precinct_exp[r] = int(ppx) | (int(ppy) << 4)
        return p


class ImageLayout(C.Structure):
    """grk_amd_image_layout: image area [x0, x1) x [y0, y1) on the canonical grid, tile grid anchored at (tx0, ty0)."""
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32),
                ("tx0", C.c_uint32), ("ty0", C.c_uint32), ("t_width", C.c_uint32), ("t_height", C.c_uint32)]

    @classmethod
    def make(cls, w, h, tile_w=None, tile_h=None, offset=(0, 0), tile_origin=(0, 0)):
        return cls(offset[0], offset[1], offset[0] + w, offset[1] + h, tile_origin[0], tile_origin[1],
                   tile_w or offset[0] + w - tile_origin[0], tile_h or offset[1] + h - tile_origin[1])


class Block(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32),
                ("px", C.c_uint32), ("py", C.c_uint32), ("comp", C.c_uint16), ("res", C.c_uint8),
                ("band", C.c_uint8), ("kmax", C.c_uint8), ("reserved", C.c_uint8 * 3),
                ("stepsize", C.c_float), ("precinct", C.c_uint32)]


class CodedBlock(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("length", C.c_uint32), ("missing_msbs", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise NativeLibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % p)
        L = C.CDLL(p)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        PP = C.POINTER(TileParams)
        L.grk_amd_version.restype = C.c_char_p
        L.grk_amd_last_error.restype = C.c_char_p
        L.grk_amd_last_error.argtypes = [vp]
        L.grk_amd_create.argtypes = [i32, i32, C.POINTER(vp)]
        L.grk_amd_destroy.argtypes = [vp]
        L.grk_amd_destroy.restype = None
        L.grk_amd_set_stream.argtypes = [vp, vp]
        L.grk_amd_tile_num_blocks.restype = C.c_int64
        L.grk_amd_tile_num_blocks.argtypes = [PP]
        L.grk_amd_tile_layout.restype = C.c_int64
        L.grk_amd_tile_layout.argtypes = [PP, vp, u64, vp]
        L.grk_amd_tile_precincts.argtypes = [PP, vp]
        L.grk_amd_plane_stride.restype = u32
        L.grk_amd_plane_stride.argtypes = [PP]
        L.grk_amd_plane_elems.restype = u64
        L.grk_amd_plane_elems.argtypes = [PP]
        L.grk_amd_encode_tiles.argtypes = [vp, PP, u32, vp, i32, vp, C.POINTER(u64)]
        L.grk_amd_fetch_table.argtypes = [vp, vp, C.POINTER(u64)]
        L.grk_amd_fetch_coded.argtypes = [vp, vp, u64]
        L.grk_amd_coded_device_ptr.restype = vp
        L.grk_amd_coded_device_ptr.argtypes = [vp]
        L.grk_amd_table_device_ptr.restype = vp
        L.grk_amd_table_device_ptr.argtypes = [vp, i32]
        L.grk_amd_plane_device_ptr.restype = vp
        L.grk_amd_plane_device_ptr.argtypes = [vp, i32]
        L.grk_amd_synchronize.argtypes = [vp]
        L.grk_amd_stage_ingest_mct.argtypes = [vp, PP, u32, vp, vp]
        L.grk_amd_stage_dwt_fwd.argtypes = [vp, PP, u32, vp, vp]
        L.grk_amd_stage_ht_encode.argtypes = [vp, PP, u32, vp]
        L.grk_amd_stage_dwt_inv.argtypes = [vp, PP, u32, vp, vp]
        L.grk_amd_stage_ht_decode.argtypes = [vp, PP, u32, vp, vp, u64, vp]
        L.grk_amd_decode_tiles.argtypes = [vp, PP, u32, vp, vp, u64, i32, vp, i32]
        L.grk_amd_decode_status.argtypes = [vp]
        L.grk_amd_set_decode_qcd.argtypes = [vp, vp, u32]
        L.grk_amd_set_decode_segments.argtypes = [vp, vp, vp, u32]
        L.grk_amd_set_decode_steps.argtypes = [vp, vp, u32]
        L.grk_amd_decode_region.argtypes = [vp, PP, vp, vp, u64, i32, u32, u32, u32, u32, vp, i32]
        L.grk_amd_set_overlap.argtypes = [vp, i32]
        L.grk_amd_set_decode_planes16.argtypes = [vp, i32]
        if hasattr(L, "grk_amd_plane_sample_bytes"):      # (absent from older builds loaded through GRK_AMD_LIB for A/B timing)
            L.grk_amd_plane_sample_bytes.argtypes = [vp, PP, i32, C.POINTER(u32)]
        L.grk_amd_set_pipelining.argtypes = [vp, i32]
        if hasattr(L, "grk_amd_set_decode_pipelining"):
            L.grk_amd_set_decode_pipelining.argtypes = [vp, i32]
        L.grk_amd_stream_wait_results.argtypes = [vp, vp]
        if hasattr(L, "grk_amd_block_distortion"):
            L.grk_amd_block_distortion.argtypes = [vp, vp, u64]
        if hasattr(L, "grk_amd_set_pixel_hold"):
            L.grk_amd_set_pixel_hold.argtypes = [vp, i32]
            L.grk_amd_stream_wait_pixels.argtypes = [vp, vp]
        if hasattr(L, "grk_amd_decode_stream_wait_slot"):
            L.grk_amd_decode_stream_wait_slot.argtypes = [vp, vp]
        L.grk_amd_stage_egress.argtypes = [vp, PP, u32, vp, vp]
        L.grk_amd_enable_timing.argtypes = [vp, i32]
        L.grk_amd_kernel_ms.restype = C.c_double
        L.grk_amd_kernel_ms.argtypes = [vp, i32, C.POINTER(u32)]
        L.grk_amd_write_codestream.restype = C.c_int64
        L.grk_amd_write_codestream.argtypes = [PP, u32, u32, vp, vp, vp, u64]
        L.grk_amd_write_codestream_ex.restype = C.c_int64
        L.grk_amd_write_codestream_ex.argtypes = [PP, u32, u32, vp, vp, u32, vp, u64]
        L.grk_amd_write_main_header.restype = C.c_int64
        L.grk_amd_write_main_header.argtypes = [PP, u32, u32, u32, vp, vp, u64]
        L.grk_amd_write_tile_part.restype = C.c_int64
        L.grk_amd_write_tile_part.argtypes = [PP, u32, u32, vp, vp, vp, u64]
        if hasattr(L, "grk_amd_node_create"):
            L.grk_amd_host_alloc.restype = vp
            L.grk_amd_host_alloc.argtypes = [vp, u64]
            L.grk_amd_host_free.argtypes = [vp, vp]
            L.grk_amd_node_create.argtypes = [vp, u32, i32, C.POINTER(vp)]
            L.grk_amd_node_destroy.argtypes = [vp]
            L.grk_amd_node_size.restype = u32
            L.grk_amd_node_size.argtypes = [vp]
            L.grk_amd_node_ctx.restype = vp
            L.grk_amd_node_ctx.argtypes = [vp, u32]
            L.grk_amd_node_last_error.restype = C.c_char_p
            L.grk_amd_node_last_error.argtypes = [vp]
            L.grk_amd_node_encode_image.restype = C.c_int64
            L.grk_amd_node_encode_image.argtypes = [vp, C.POINTER(ImageLayout), PP, vp, u32, vp, u64]
        L.grk_amd_locate_tile_parts.restype = C.c_int64
        L.grk_amd_locate_tile_parts.argtypes = [vp, u64, vp, vp, vp, u64, C.POINTER(i32)]
        PL = C.POINTER(ImageLayout)
        L.grk_amd_layout_num_tiles.restype = C.c_int64
        L.grk_amd_layout_num_tiles.argtypes = [PL]
        L.grk_amd_layout_tile.argtypes = [PL, PP, u32, PP]
        L.grk_amd_same_tile_geometry.argtypes = [PP, PP]
        L.grk_amd_write_codestream_layout.restype = C.c_int64
        L.grk_amd_write_codestream_layout.argtypes = [PL, PP, vp, vp, u32, vp, u64]
        L.grk_amd_write_main_header_layout.restype = C.c_int64
        L.grk_amd_write_main_header_layout.argtypes = [PL, PP, u32, vp, vp, u64]
        L.grk_amd_encode_image.restype = C.c_int64
        L.grk_amd_encode_image.argtypes = [vp, PL, PP, vp, u32, vp, u64]
        _lib = L
    return _lib


def tile_layout(params):
    """Host-only geometry: (list of Block, qcd words)."""
    L = lib()
    n = L.grk_amd_tile_num_blocks(C.byref(params))
    if n < 0:
        raise ValueError("grk_amd_tile_num_blocks failed: %d" % n)
    blocks = (Block * n)()
    qcd = (C.c_uint16 * (3 * MAX_LEVELS + 1))()
    rc = L.grk_amd_tile_layout(C.byref(params), blocks, n, qcd)
    if rc < 0:
        raise ValueError("grk_amd_tile_layout failed: %d" % rc)
    return list(blocks), list(qcd)[:3 * params.num_levels + 1]


CS_TLM, CS_PLT, CS_SOP, CS_EPH = 1, 2, 4, 8


def CS_PROG(order):
    """progression order for the writer's flags: 0 LRCP, 1 RLCP, 2 RPCL, 3 PCRL, 4 CPRL"""
    return int(order) << 8


def write_codestream(params, img_w, img_h, table, coded, flags=0):
    """table: ctypes array / numpy structured array of CodedBlock rows; coded: bytes-like; flags: CS_TLM | CS_PLT."""
    L = lib()
    cbuf = np.frombuffer(coded, np.uint8) if not isinstance(coded, np.ndarray) else coded
    cap = int(cbuf.size) + len(table) * 8 + (1 << 20)
    out = np.empty(cap, np.uint8)
    tptr = table.ctypes.data if isinstance(table, np.ndarray) else C.addressof(table)
    n = L.grk_amd_write_codestream_ex(C.byref(params), img_w, img_h, tptr, cbuf.ctypes.data, flags, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("grk_amd_write_codestream failed: %d" % n)
    return out[:n].tobytes()


def layout_tiles(layout, base):
    """The parameters of every tile of the layout (raster order): base with the tile's size and origin."""
    L = lib()
    n = L.grk_amd_layout_num_tiles(C.byref(layout))
    if n < 0:
        raise ValueError("grk_amd_layout_num_tiles failed: %d" % n)
    out = []
    for t in range(n):
        p = TileParams()
        rc = L.grk_amd_layout_tile(C.byref(layout), C.byref(base), t, C.byref(p))
        if rc:
            raise ValueError("grk_amd_layout_tile failed: %d" % rc)
        out.append(p)
    return out


def same_tile_geometry(a, b):
    rc = lib().grk_amd_same_tile_geometry(C.byref(a), C.byref(b))
    if rc < 0:
        raise ValueError("grk_amd_same_tile_geometry failed: %d" % rc)
    return bool(rc)


def write_codestream_layout(layout, base, table, coded, flags=0):
    """table: the tiles' rows one tile after the other (every tile with the rows of ITS geometry, layout_tiles())."""
    L = lib()
    cbuf = np.frombuffer(coded, np.uint8) if not isinstance(coded, np.ndarray) else coded
    cap = int(cbuf.size) + len(table) * 8 + (1 << 20)
    out = np.empty(cap, np.uint8)
    t = np.ascontiguousarray(table)
    n = L.grk_amd_write_codestream_layout(C.byref(layout), C.byref(base), t.ctypes.data, cbuf.ctypes.data, flags, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("grk_amd_write_codestream_layout failed: %d" % n)
    return out[:n].tobytes()


def write_tile_part(params, tile_index, tile_table, coded, flags=0, size_only=False):
    """One tile-part (SOT [PLT] SOD packets) as bytes, or only its length (no coded bytes needed)."""
    L = lib()
    t = np.ascontiguousarray(tile_table)
    if size_only:
        n = L.grk_amd_write_tile_part(C.byref(params), tile_index, flags, t.ctypes.data, None, None, 0)
        if n < 0:
            raise RuntimeError("grk_amd_write_tile_part failed: %d" % n)
        return int(n)
    cbuf = np.frombuffer(coded, np.uint8) if not isinstance(coded, np.ndarray) else coded
    cap = int(t["length"].sum()) + len(t) * 8 + 4096
    out = np.empty(cap, np.uint8)
    n = L.grk_amd_write_tile_part(C.byref(params), tile_index, flags, t.ctypes.data, cbuf.ctypes.data, out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("grk_amd_write_tile_part failed: %d" % n)
    return out[:n].tobytes()


def write_main_header(params, img_w, img_h, flags=0, tile_part_bytes=None):
    L = lib()
    out = np.empty(4096 + (5 * len(tile_part_bytes) if tile_part_bytes is not None else 0), np.uint8)
    tp = np.ascontiguousarray(tile_part_bytes, np.uint32) if tile_part_bytes is not None else None
    n = L.grk_amd_write_main_header(C.byref(params), img_w, img_h, flags, tp.ctypes.data if tp is not None else None,
                                    out.ctypes.data, out.size)
    if n < 0:
        raise RuntimeError("grk_amd_write_main_header failed: %d" % n)
    return out[:n].tobytes()


def locate_tile_parts(cs):
    """-> ([(offset, length, tile index)], used_tlm)"""
    L = lib()
    buf = np.frombuffer(cs, np.uint8)
    n = L.grk_amd_locate_tile_parts(buf.ctypes.data, buf.size, None, None, None, 0, None)
    if n < 0:
        raise RuntimeError("grk_amd_locate_tile_parts failed: %d" % n)
    off = np.zeros(n, np.uint64); ln = np.zeros(n, np.uint32); ti = np.zeros(n, np.uint16)
    used = C.c_int(0)
    L.grk_amd_locate_tile_parts(buf.ctypes.data, buf.size, off.ctypes.data, ln.ctypes.data, ti.ctypes.data, n, C.byref(used))
    return [(int(a), int(b), int(c)) for a, b, c in zip(off, ln, ti)], bool(used.value)


CODED_DTYPE = np.dtype([("offset", np.uint64), ("length", np.uint32), ("missing_msbs", np.uint32)])


class Context:
    """One GPU context (grk_amd_ctx)."""

    def __init__(self, device=0, verbose=False):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.grk_amd_create(device, int(verbose), C.byref(h))
        if rc != 0:
            raise RuntimeError("grk_amd_create(device=%d) failed: %d (no usable HIP device?)" % (device, rc))
        self._h = h

    def close(self):
        if self._h:
            self._L.grk_amd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %d (%s)" % (what, rc, self._L.grk_amd_last_error(self._h).decode()))

    def set_stream(self, stream_ptr):
        self._check(self._L.grk_amd_set_stream(self._h, stream_ptr), "set_stream")

    def synchronize(self):
        self._check(self._L.grk_amd_synchronize(self._h), "synchronize")

    def encode_tiles(self, params, ntiles, pixels_ptr, on_device, fetch=True):
        """Runs the hot path. Returns (table ndarray, total_bytes) when fetch else None."""
        if fetch:
            n = self._L.grk_amd_tile_num_blocks(C.byref(params)) * ntiles
            table = np.zeros(n, CODED_DTYPE)
            tot = C.c_uint64(0)
            self._check(self._L.grk_amd_encode_tiles(self._h, C.byref(params), ntiles, pixels_ptr, int(on_device),
                                                     table.ctypes.data, C.byref(tot)), "encode_tiles")
            return table, tot.value
        self._check(self._L.grk_amd_encode_tiles(self._h, C.byref(params), ntiles, pixels_ptr, int(on_device),
                                                 None, None), "encode_tiles")
        return None

    def encode_host(self, params, pixels, ntiles=1):
        """pixels: numpy array holding the tiles back to back (host memory)."""
        px = np.ascontiguousarray(pixels)
        table, tot = self.encode_tiles(params, ntiles, px.ctypes.data, False)
        coded = np.empty(tot, np.uint8)
        if tot:
            self._check(self._L.grk_amd_fetch_coded(self._h, coded.ctypes.data, tot), "fetch_coded")
        return table, coded

    def host_array(self, nbytes):
        """A uint8 numpy array over pinned host memory (grk_amd_host_alloc): crosses the link in one DMA.  Freed with the
        array (keep a reference while the context uses it)."""
        p = self._L.grk_amd_host_alloc(self._h, int(nbytes))
        if not p:
            raise MemoryError("grk_amd_host_alloc(%d) failed" % nbytes)
        buf = (C.c_uint8 * int(nbytes)).from_address(p)
        arr = np.frombuffer(buf, np.uint8)
        L = self._L
        import weakref
        weakref.finalize(buf, lambda: L.grk_amd_host_free(None, C.c_void_p(p)))     # (no context needed: it may be gone by then)
        return arr

    def encode_image(self, layout, base, pixels, flags=0):
        """Whole image (C, H, W) of any tile layout -> codestream bytes (grk_amd_encode_image)."""
        px = np.ascontiguousarray(pixels)
        cap = px.size * 4 + (1 << 20)
        out = np.empty(cap, np.uint8)
        n = self._L.grk_amd_encode_image(self._h, C.byref(layout), C.byref(base), px.ctypes.data, flags, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("encode_image failed: %d (%s)" % (n, self._L.grk_amd_last_error(self._h).decode()))
        return out[:n].tobytes()

    def encode_image_subsampled(self, layout, base, sampling, planes, flags=0):
        """Image with sub-sampled components -> codestream bytes (grk_amd_encode_image_subsampled).  sampling: [(dx, dy)] per
        component; planes: one 2-D array per component, component c of ceil(x1 / dx) - ceil(x0 / dx) columns."""
        dx = (C.c_uint8 * len(sampling))(*[int(a) for a, _ in sampling])
        dy = (C.c_uint8 * len(sampling))(*[int(b) for _, b in sampling])
        px = np.concatenate([np.ascontiguousarray(pl).reshape(-1) for pl in planes])
        cap = px.size * px.itemsize * 4 + (1 << 20)
        out = np.empty(cap, np.uint8)
        self._L.grk_amd_encode_image_subsampled.restype = C.c_int64
        self._L.grk_amd_encode_image_subsampled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                            C.c_uint32, C.c_void_p, C.c_uint64]
        n = self._L.grk_amd_encode_image_subsampled(self._h, C.addressof(layout), C.addressof(base), C.addressof(dx), C.addressof(dy),
                                                    px.ctypes.data, flags, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("encode_image_subsampled failed: %d (%s)" % (n, self._L.grk_amd_last_error(self._h).decode()))
        return out[:n].tobytes()

    def fetch_table(self, nblocks):
        table = np.zeros(nblocks, CODED_DTYPE)
        tot = C.c_uint64(0)
        self._check(self._L.grk_amd_fetch_table(self._h, table.ctypes.data, C.byref(tot)), "fetch_table")
        return table, tot.value

    def fetch_coded(self, nbytes):
        coded = np.empty(nbytes, np.uint8)
        if nbytes:
            self._check(self._L.grk_amd_fetch_coded(self._h, coded.ctypes.data, nbytes), "fetch_coded")
        return coded

    def assemble_device(self, params, tile_index, flags=0, dst_offset=0):
        """Tier-2 on the device for the latest encode_tiles call (grk_amd_assemble_device) -> (bytes assembled, [tile-part lengths])."""
        idx = np.ascontiguousarray(np.asarray(tile_index, np.uint32))
        lens = np.zeros(idx.size, np.uint32)
        self._L.grk_amd_assemble_device.restype = C.c_int64
        self._L.grk_amd_assemble_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        n = self._L.grk_amd_assemble_device(self._h, C.addressof(params), idx.size, idx.ctypes.data, flags, dst_offset, lens.ctypes.data)
        if n < 0:
            raise RuntimeError("assemble_device failed: %d (%s)" % (n, self._L.grk_amd_last_error(self._h).decode()))
        return int(n), lens

    def assemble_device_async(self, params, tile_index, flags, stream_ptr):
        """grk_amd_assemble_device_async: Tier-2 of the latest encode queued on `stream_ptr`; results stay on the device."""
        idx = np.ascontiguousarray(np.asarray(tile_index, np.uint32))
        self._L.grk_amd_assemble_device_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        self._check(self._L.grk_amd_assemble_device_async(self._h, C.addressof(params), idx.size, idx.ctypes.data, flags, stream_ptr), "assemble_device_async")

    def assembled_table_ptr(self, which):
        self._L.grk_amd_assembled_table_ptr.restype = C.c_void_p
        self._L.grk_amd_assembled_table_ptr.argtypes = [C.c_void_p, C.c_int]
        return self._L.grk_amd_assembled_table_ptr(self._h, which)

    def fetch_assembled(self, offset, nbytes):
        out = np.empty(nbytes, np.uint8)
        self._L.grk_amd_fetch_assembled.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        if nbytes:
            self._check(self._L.grk_amd_fetch_assembled(self._h, offset, nbytes, out.ctypes.data), "fetch_assembled")
        return out

    def assembled_device_ptr(self):
        self._L.grk_amd_assembled_device_ptr.restype = C.c_void_p
        self._L.grk_amd_assembled_device_ptr.argtypes = [C.c_void_p]
        return self._L.grk_amd_assembled_device_ptr(self._h)

    def coded_device_ptr(self):
        return self._L.grk_amd_coded_device_ptr(self._h)

    def table_device_ptr(self, which):
        return self._L.grk_amd_table_device_ptr(self._h, which)

    def plane_device_ptr(self, which):
        return self._L.grk_amd_plane_device_ptr(self._h, which)

    def stage_ingest_mct(self, params, ntiles, d_pixels, d_planes):
        self._check(self._L.grk_amd_stage_ingest_mct(self._h, C.byref(params), ntiles, d_pixels, d_planes), "stage_ingest_mct")

    def stage_dwt_fwd(self, params, nplanes, d_in, d_out):
        self._check(self._L.grk_amd_stage_dwt_fwd(self._h, C.byref(params), nplanes, d_in, d_out), "stage_dwt_fwd")

    def stage_ht_encode(self, params, ntiles, d_mallat):
        self._check(self._L.grk_amd_stage_ht_encode(self._h, C.byref(params), ntiles, d_mallat), "stage_ht_encode")

    def stage_ht_decode(self, params, ntiles, table, d_coded, coded_bytes, d_mallat):
        t = np.ascontiguousarray(table)
        self._check(self._L.grk_amd_stage_ht_decode(self._h, C.byref(params), ntiles, t.ctypes.data, d_coded, coded_bytes,
                                                    d_mallat),
                    "stage_ht_decode")

    def decode_host(self, params, table, coded, ntiles=1):
        """table: CODED_DTYPE rows, coded: bytes-like (host). Returns pixels (ntiles, C, H, W)."""
        t = np.ascontiguousarray(table)
        cb = np.frombuffer(coded, np.uint8) if not isinstance(coded, np.ndarray) else np.ascontiguousarray(coded)
        dt = np.uint8 if params.prec <= 8 else np.uint16
        out = np.zeros((ntiles, params.num_comps, params.tile_h, params.tile_w), dt)
        self._check(self._L.grk_amd_decode_tiles(self._h, C.byref(params), ntiles, t.ctypes.data, cb.ctypes.data, cb.size, 0,
                                                 out.ctypes.data, 0), "decode_tiles")
        return out

    def decode_region_host(self, params, table, coded, x0, y0, x1, y1):
        """Windowed decode of one tile -> pixels (C, y1 - y0, x1 - x0)."""
        t = np.ascontiguousarray(table)
        cb = np.frombuffer(coded, np.uint8) if not isinstance(coded, np.ndarray) else np.ascontiguousarray(coded)
        dt = np.uint8 if params.prec <= 8 else np.uint16
        out = np.zeros((params.num_comps, y1 - y0, x1 - x0), dt)
        self._check(self._L.grk_amd_decode_region(self._h, C.byref(params), t.ctypes.data, cb.ctypes.data, cb.size, 0,
                                                  x0, y0, x1, y1, out.ctypes.data, 0), "decode_region")
        return out

    def decode_region_device(self, params, table, d_coded, coded_bytes, x0, y0, x1, y1, d_pixels):
        t = np.ascontiguousarray(table)
        self._check(self._L.grk_amd_decode_region(self._h, C.byref(params), t.ctypes.data, d_coded, coded_bytes, 1,
                                                  x0, y0, x1, y1, d_pixels, 1), "decode_region")

    def decode_device(self, params, ntiles, table, d_coded, coded_bytes, d_pixels):
        t = np.ascontiguousarray(table)
        self._check(self._L.grk_amd_decode_tiles(self._h, C.byref(params), ntiles, t.ctypes.data, d_coded, coded_bytes, 1,
                                                 d_pixels, 1), "decode_tiles")

    def set_decode_steps(self, steps):
        """Band step sizes as the host's decoder holds them, [comp][band] (None / empty: back to the QCD words)."""
        a = np.ascontiguousarray(steps if steps is not None else [], np.float32).reshape(-1)
        self._check(self._L.grk_amd_set_decode_steps(self._h, a.ctypes.data if a.size else None, a.size), "set_decode_steps")

    def stream_wait_results(self, hip_stream):
        self._check(self._L.grk_amd_stream_wait_results(self._h, C.c_void_p(hip_stream)), "stream_wait_results")

    def block_distortion(self, nblocks):
        """Distortion decrease of every block of the latest encode (the rate-control hook, include/grok_amd.h)."""
        out = np.zeros(int(nblocks), np.float64)
        self._check(self._L.grk_amd_block_distortion(self._h, out.ctypes.data, int(nblocks)), "block_distortion")
        return out

    def probe_streams(self):
        """The context's stream probe now (grk_amd_probe_streams); returns the side streams replaced so far (-1: probe off)."""
        self._check(self._L.grk_amd_probe_streams(self._h), "probe_streams")
        return int(self._L.grk_amd_stream_probe_result(self._h))

    def internal_stream(self, which):
        self._L.grk_amd_internal_stream.restype = C.c_void_p
        self._L.grk_amd_internal_stream.argtypes = [C.c_void_p, C.c_int]
        return self._L.grk_amd_internal_stream(self._h, which)

    def streams_side_by_side(self, a, b):
        self._L.grk_amd_streams_side_by_side.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        r = self._L.grk_amd_streams_side_by_side(self._h, a, b)
        if r < 0:
            raise RuntimeError("streams_side_by_side failed: %d" % r)
        return bool(r)

    def set_pixel_hold(self, on):
        """True: the caller keeps a call's device pixels untouched until stream_wait_pixels / synchronize (include/grok_amd.h)."""
        if hasattr(self._L, "grk_amd_set_pixel_hold"):
            self._check(self._L.grk_amd_set_pixel_hold(self._h, int(bool(on))), "set_pixel_hold")

    def stream_wait_pixels(self, hip_stream):
        self._check(self._L.grk_amd_stream_wait_pixels(self._h, C.c_void_p(hip_stream)), "stream_wait_pixels")

    def set_decode_pipelining(self, frames_in_flight):
        """2..8: consecutive decode_device calls run on that many internal buffer / stream sets in turn (0 / 1: off)"""
        self._check(self._L.grk_amd_set_decode_pipelining(self._h, int(frames_in_flight)), "set_decode_pipelining")

    def set_pipelining(self, on):
        """False / True (two buffer sets) / 2 (three: results valid until the third next call)."""
        self._check(self._L.grk_amd_set_pipelining(self._h, int(on)), "set_pipelining")

    def decode_stream_wait_slot(self, stream_handle):
        """Makes that stream wait until the internal set the NEXT decode call uses has finished its last frame: the buffers handed
        over `frames in flight` calls ago may then be overwritten / read on it (include/grok_amd.h: buffer lifetime in a sequence)."""
        self._check(self._L.grk_amd_decode_stream_wait_slot(self._h, C.c_void_p(stream_handle)), "decode_stream_wait_slot")

    def set_decode_planes16(self, on):
        self._check(self._L.grk_amd_set_decode_planes16(self._h, int(on)), "set_decode_planes16")

    def plane_sample_bytes(self, params, decode=False):
        """(bytes per coefficient of the LL / Mallat planes the encode / decode path keeps for such tiles: 2 or 4, forward DWT
        levels that run on packed int16 pairs)"""
        n = C.c_uint32(0)
        b = self._L.grk_amd_plane_sample_bytes(self._h, C.byref(params), int(decode), C.byref(n))
        self._check(min(b, 0), "plane_sample_bytes")
        return b, n.value

    def set_overlap(self, on):
        self._check(self._L.grk_amd_set_overlap(self._h, int(bool(on))), "set_overlap")

    def set_decode_segments(self, per_block):
        """Part-1 blocks with several codeword segments (LAZY / TERMALL): per_block = [[(bytes, passes), ...], ...] in
        table order; None or [] returns to one segment per block."""
        if not per_block:
            self._check(self._L.grk_amd_set_decode_segments(self._h, None, None, 0), "set_decode_segments")
            return
        first = np.zeros(len(per_block) + 1, np.uint32)
        first[1:] = np.cumsum([len(b) for b in per_block])
        segs = np.array([v for b in per_block for s in b for v in s], np.uint32).reshape(-1, 2)
        self._check(self._L.grk_amd_set_decode_segments(self._h, first.ctypes.data, segs.ctypes.data if segs.size else None,
                                                        len(per_block)), "set_decode_segments")

    def set_decode_qcd(self, words):
        w = np.ascontiguousarray(words, np.uint16)
        self._check(self._L.grk_amd_set_decode_qcd(self._h, w.ctypes.data if w.size else None, w.size), "set_decode_qcd")

    def decode_status(self):
        self._check(self._L.grk_amd_decode_status(self._h), "decode_status")

    def stage_dwt_inv(self, params, nplanes, d_mallat, d_out):
        self._check(self._L.grk_amd_stage_dwt_inv(self._h, C.byref(params), nplanes, d_mallat, d_out), "stage_dwt_inv")

    def stage_egress(self, params, ntiles, d_planes, d_pixels):
        self._check(self._L.grk_amd_stage_egress(self._h, C.byref(params), ntiles, d_planes, d_pixels), "stage_egress")

    def enable_timing(self, on=True):
        self._check(self._L.grk_amd_enable_timing(self._h, int(on)), "enable_timing")

    def kernel_ms(self, which):
        n = C.c_uint32(0)
        ms = self._L.grk_amd_kernel_ms(self._h, which, C.byref(n))
        return ms, n.value


NODE_GATHER = 0x80000000


class Node:
    """grk_amd_node: one image over several GPUs natively -- a context + a host thread per entry of `devices` (None: all of
    the node; an entry may repeat: two contexts on one GPU), tiles t -> device t mod R, one codestream."""

    def __init__(self, devices=None, verbose=False):
        self._L = lib()
        h = C.c_void_p()
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            rc = self._L.grk_amd_node_create(arr, len(devices), int(verbose), C.byref(h))
        else:
            rc = self._L.grk_amd_node_create(None, 0, int(verbose), C.byref(h))
        if rc != 0:
            raise RuntimeError("grk_amd_node_create failed: %d" % rc)
        self._h = h

    @property
    def size(self):
        return int(self._L.grk_amd_node_size(self._h))

    def encode_image(self, layout, base, pixels, flags=0, out=None):
        px = pixels if isinstance(pixels, np.ndarray) and pixels.flags["C_CONTIGUOUS"] else np.ascontiguousarray(pixels)
        cap = px.size * px.itemsize * 2 + (1 << 20)
        if out is None:
            out = np.empty(cap, np.uint8)
        n = self._L.grk_amd_node_encode_image(self._h, C.byref(layout), C.byref(base), px.ctypes.data, flags, out.ctypes.data, out.size)
        if n < 0:
            raise RuntimeError("node_encode_image failed: %d (%s)" % (n, self._L.grk_amd_node_last_error(self._h).decode()))
        return out[:n]

    def encode_image_device(self, layout, base, pixels_ptr, nbytes, device, flags=0, out=None):
        """The image resident in the memory of HIP device `device` at `pixels_ptr` (nbytes of it): grk_amd_node_encode_image_device."""
        if out is None:
            out = np.empty(nbytes * 2 + (1 << 20), np.uint8)
        fn = self._L.grk_amd_node_encode_image_device
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_uint64]
        n = fn(self._h, C.cast(C.byref(layout), C.c_void_p), C.cast(C.byref(base), C.c_void_p), pixels_ptr, int(device), flags, out.ctypes.data, out.size)
        if n < 0:
            raise RuntimeError("node_encode_image_device failed: %d (%s)" % (n, self._L.grk_amd_node_last_error(self._h).decode()))
        return out[:n]

    def close(self):
        if self._h:
            self._L.grk_amd_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
