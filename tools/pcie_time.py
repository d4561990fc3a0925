"""PCIe-inclusive encode of the 8K workload: pixels in host memory (pageable / pinned), table + coded bytes fetched (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = 8192
px = synth.g2(3, H, W, 8); p = G.TileParams.make(W, H, 3, 8, 5)
ctx = G.Context(0)
nb = G.lib().grk_amd_tile_num_blocks(p)
pinned = torch.from_numpy(px.reshape(-1)).pin_memory()
out_pinned = torch.empty(px.size, dtype=torch.uint8).pin_memory()
for name, src in (("pageable", px), ("pinned", pinned)):
    ptr = src.ctypes.data if name == "pageable" else src.data_ptr()
    for rep in range(3):
        t0 = time.perf_counter()
        table, tot = ctx.encode_tiles(p, 1, ptr, False)          # upload + encode + table fetch
        t1 = time.perf_counter()
        if name == "pinned":
            G.lib().grk_amd_fetch_coded(ctx._h, out_pinned.data_ptr(), tot)
        else:
            coded = ctx.fetch_coded(tot)
        t2 = time.perf_counter()
    print(name, "upload+encode+table %.2f ms, coded download %.2f ms, total %.2f ms (%.0f MB in, %.0f MB out)" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, px.size / 1e6, tot / 1e6))
