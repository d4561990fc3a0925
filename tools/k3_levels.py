"""K3 alone by the number of DWT levels at small frame sizes (dev tool): which blocks set a small frame's K3 time?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
for S in (2048, 4096):
    px = synth.g2(3, S, S, 8)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    for L in (2, 3, 4, 5, 6):
        p = G.TileParams.make(S, S, 3, 8, L)
        nb = G.lib().grk_amd_tile_num_blocks(p)
        ctx = G.Context(0); ctx.set_overlap(False)
        for _ in range(5):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); ctx.enable_timing(True)
        for _ in range(30):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize()
        parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
        k3 = sum(m * c for m, c in parts) / max(max(x[1] for x in parts), 1)
        t, tot = ctx.fetch_table(nb)
        blocks, _ = G.tile_layout(p)
        i = int(np.argmax(t["length"]))
        print("%4d^2 L=%d: K3 %.4f ms, %5d blocks, longest block %4d bytes (res %d band %d, %dx%d)" % (
            S, L, k3, nb, int(t["length"][i]), blocks[i].res, blocks[i].band, blocks[i].x1 - blocks[i].x0, blocks[i].y1 - blocks[i].y0))
        ctx.close()
