"""CPU: the plugin shared object (drop-in boundary) loads, exports every symbol
include/grk_plugin_abi.h declares, its ABI mirror equals Grok's headers, Grok's own loader
accepts it, and without a GPU it declines so that the host takes its CPU path."""
import ctypes as C
import os

import pytest

import grok_amd as G
import refharness as R
from test_capi_host import declared_symbols

PLUGIN = os.path.join(os.path.dirname(G.lib_path()), "libgrokj2k_plugin.so")
needs_ref = pytest.mark.skipif(not R.have_ref(), reason="oracle/_ref not built")

STUB_SYMBOLS = ["minpf_post_load_plugin", "plugin_init", "plugin_encode", "plugin_batch_encode",
                "plugin_is_batch_complete", "plugin_stop_batch_encode", "plugin_decompress",
                "plugin_init_batch_decompress", "plugin_batch_decompress", "plugin_stop_batch_decompress",
                "plugin_get_debug_state", "plugin_debug_next_cxd", "plugin_debug_mqc_next_cxd",
                "plugin_debug_mqc_next_plane"]


def test_plugin_exports_reference_symbol_list():
    """the symbol list of the in-tree stub (src/lib/jp2_plugin/Plugin.cpp:19-125)"""
    assert os.path.exists(PLUGIN), "build() did not produce libgrokj2k_plugin.so"
    L = C.CDLL(PLUGIN)
    for s in STUB_SYMBOLS + ["grk_amd_plugin_tile_create", "grk_amd_plugin_tile_destroy"]:
        assert hasattr(L, s), s
    declared = [s for s in declared_symbols("grk_plugin_abi.h") if not s.startswith("grk_amd_") or "plugin" in s]
    for s in declared:
        assert hasattr(L, s), "declared in grk_plugin_abi.h but not exported: %s" % s


def test_plugin_declines_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")

    class Init(C.Structure):
        _fields_ = [("deviceId", C.c_int32), ("verbose", C.c_bool)]
    L = C.CDLL(PLUGIN)
    L.plugin_init.restype = C.c_bool
    L.plugin_init.argtypes = [Init]
    assert L.plugin_init(Init(0, False)) is False
    L.plugin_encode.restype = C.c_int32
    L.plugin_encode.argtypes = [C.c_void_p, C.c_void_p]
    assert L.plugin_encode(None, None) != 0
    L.plugin_get_debug_state.restype = C.c_uint32
    assert L.plugin_get_debug_state() == 0


@needs_ref
@pytest.mark.ref
def test_abi_mirror_compiled_against_grok_headers():
    """oracle/ref_harness/abi_check.cpp static_asserts sizeof/offsetof of every mirrored struct; its
    presence in the harness proves the asserts held at build time."""
    L = R.lib()
    L.ref_abi_mirror_checked.restype = C.c_int
    assert L.ref_abi_mirror_checked() == 1
    assert L.ref_abi_sizeof(1) == 1680 or L.ref_abi_sizeof(1) > 0     # grk_plugin_code_block


@needs_ref
@pytest.mark.ref
def test_decode_callback_record_layout_matches_reference():
    """plugin_decompress's callback record (PluginDecodeCallbackInfo, plugin/plugin_interface.h:86-130) has std::string
    members, so it is mirrored in C++ inside plugin.cpp: every offset and the size equal the reference's own."""
    P = C.CDLL(os.path.join(R.plugin_dir(), "libgrokj2k_plugin.so"))
    P.grk_amd_plugin_decode_info_layout.restype = C.c_size_t
    P.grk_amd_plugin_decode_info_layout.argtypes = [C.c_int]
    L = R.lib()
    L.ref_decode_info_layout.restype = C.c_size_t
    L.ref_decode_info_layout.argtypes = [C.c_int]
    for which in range(15):
        assert P.grk_amd_plugin_decode_info_layout(which) == L.ref_decode_info_layout(which), which
    assert L.ref_decode_info_layout(0) > 11000


@needs_ref
@pytest.mark.ref
def test_grok_loader_accepts_our_plugin():
    """grk_initialize(<dir with libgrokj2k_plugin.so>) -> minpf dlopen + registration succeed."""
    assert R.plugin_load() == 1
    import torch
    if not torch.cuda.is_available():
        assert R.plugin_init(0) == 0          # no MI355X here: plugin_init false -> host aborts plugin path
