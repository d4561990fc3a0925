"""grok_amd -- MI355X-native JPEG 2000 / HTJ2K tile processor behind Grok's plugin API.

The product is the C-ABI shared library grok_amd/lib/libgrok_amd.so (HIP kernels for gfx950 +
host geometry / Tier-2) and the plugin shim grok_amd/lib/libgrokj2k_plugin.so.  This Python
package is only the ctypes binding used by bench.py and the tests; it never computes anything
itself and raises loudly when the native library is missing.
"""
from .capi import (TileParams, Block, CodedBlock, Context, lib, lib_path, NativeLibraryMissing,
                   tile_layout, write_codestream, write_tile_part, write_main_header, locate_tile_parts,
                   CS_TLM, CS_PLT, CS_SOP, CS_EPH, CS_PROG, ImageLayout, layout_tiles, same_tile_geometry, write_codestream_layout, Node, NODE_GATHER)

__all__ = ["TileParams", "Block", "CodedBlock", "Context", "lib", "lib_path", "NativeLibraryMissing",
           "tile_layout", "write_codestream", "write_tile_part", "write_main_header", "locate_tile_parts", "CS_TLM", "CS_PLT", "CS_SOP", "CS_EPH", "CS_PROG",
           "ImageLayout", "layout_tiles", "same_tile_geometry", "write_codestream_layout", "Node", "NODE_GATHER"]
