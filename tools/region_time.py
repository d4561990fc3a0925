"""Kernel-family times of a windowed decode at 8K (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = 8192
px = synth.g2(3, H, W, 8); p = G.TileParams.make(W, H, 3, 8, 5)
ctx = G.Context(0); d = torch.from_numpy(px.reshape(-1)).cuda()
table, tot = ctx.encode_tiles(p, 1, d.data_ptr(), True)
back = torch.empty_like(d)
for win in ((3584, 3584, 4608, 4608), (0, 0, 512, 512), (0, 0, 8192, 8192), (4000, 4000, 4064, 4064)):
    x0, y0, x1, y1 = win
    out = torch.empty((x1 - x0) * (y1 - y0) * 3, dtype=torch.uint8, device="cuda")
    for _ in range(2): ctx.decode_region_device(p, table, ctx.coded_device_ptr(), tot, x0, y0, x1, y1, out.data_ptr())
    ctx.synchronize(); ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(5): ctx.decode_region_device(p, table, ctx.coded_device_ptr(), tot, x0, y0, x1, y1, out.data_ptr())
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    print(win, "wall %.3f ms" % dt, "ht %.3f idwt %.3f whole %.3f" % (ctx.kernel_ms(5)[0], ctx.kernel_ms(6)[0], ctx.kernel_ms(3)[0]))
    ctx.enable_timing(False)
