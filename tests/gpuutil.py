"""Helpers for the -m gpu parity tests: device buffers via torch, calls through the C-ABI."""
import numpy as np
import torch

import grok_amd as G

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        if not torch.cuda.is_available():
            raise RuntimeError("GPU test invoked without a GPU")
        _ctx = G.Context(0)
    return _ctx


def _settled(t):
    """torch fills / copies on ITS stream; the context works on its own -- finish torch's work before handing the buffer over."""
    torch.cuda.synchronize()
    return t


def to_dev(a):
    return _settled(torch.from_numpy(np.ascontiguousarray(a)).cuda())


def dev_planes(params, nplanes):
    n = G.lib().grk_amd_plane_elems(params) * nplanes
    return _settled(torch.zeros(int(n), dtype=torch.int32, device="cuda"))


def planes_to_numpy(t, params, nplanes):
    stride = G.lib().grk_amd_plane_stride(params)
    a = t.cpu().numpy().reshape(nplanes, params.tile_h, stride)
    return a[:, :, :params.tile_w]


def upload_planes(planes, params):
    """planes: (n, H, W) int32 -> device tensor in the padded plane layout."""
    stride = G.lib().grk_amd_plane_stride(params)
    n, H, W = planes.shape
    buf = np.zeros((n, H, stride), np.int32)
    buf[:, :, :W] = planes
    return to_dev(buf.reshape(-1))


def split_blocks(table, coded):
    return [bytes(coded[int(o):int(o) + int(l)]) for o, l in zip(table["offset"], table["length"])]


def from_dev_ptr(ptr, nbytes):
    """nbytes at a raw device pointer (context-owned memory) as a numpy uint8 array; everything queued so far is waited for."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    torch.cuda.synchronize()
    return torch.as_tensor(h, device="cuda").cpu().numpy().copy()
