"""PCIe-inclusive encode of the 8K workload: pixels in host memory (pageable / pinned by torch / grk_amd_host_alloc), table +
coded bytes fetched; and grk_amd_plugin_tile_create on the same frame (dev tool)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = 8192
px = synth.g2(3, H, W, 8); p = G.TileParams.make(W, H, 3, 8, 5)
ctx = G.Context(0)
nb = G.lib().grk_amd_tile_num_blocks(p)
pinned = torch.from_numpy(px.reshape(-1)).pin_memory()
ours = ctx.host_array(px.size); ours[...] = px.reshape(-1)
out_pinned = torch.empty(px.size, dtype=torch.uint8).pin_memory()
out_ours = ctx.host_array(px.size)
for name, ptr, outp in (("pageable", px.ctypes.data, None), ("torch-pinned", pinned.data_ptr(), out_pinned.data_ptr()),
                        ("host_alloc", ours.ctypes.data, out_ours.ctypes.data)):
    for rep in range(3):
        t0 = time.perf_counter()
        table, tot = ctx.encode_tiles(p, 1, ptr, False)          # upload + encode + table fetch
        t1 = time.perf_counter()
        if outp:
            G.lib().grk_amd_fetch_coded(ctx._h, C.c_void_p(outp), tot)
        else:
            coded = ctx.fetch_coded(tot)
        t2 = time.perf_counter()
    print(name, "upload+encode+table %.2f ms, coded download %.2f ms, total %.2f ms (%.0f MB in, %.0f MB out)" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, px.size / 1e6, tot / 1e6))
P = C.CDLL(os.path.join(os.path.dirname(G.lib_path()), "libgrokj2k_plugin.so"))
P.grk_amd_plugin_tile_create.restype = C.c_void_p
P.grk_amd_plugin_tile_create.argtypes = [C.c_void_p, C.POINTER(G.TileParams), C.c_void_p, C.c_int]
P.grk_amd_plugin_tile_destroy.argtypes = [C.c_void_p]
for name, ptr in (("pageable", px.ctypes.data), ("host_alloc", ours.ctypes.data)):
    for rep in range(4):
        t0 = time.perf_counter()
        t = P.grk_amd_plugin_tile_create(ctx._h, C.byref(p), C.c_void_p(ptr), 0)
        t1 = time.perf_counter()
        P.grk_amd_plugin_tile_destroy(t)
        print("plugin_tile_create", name, "rep", rep, "%.2f ms" % ((t1 - t0) * 1e3))
