"""K3 alone (HIP events) on small frames by content: is a 2048^2 frame's K3 slow because of its size or because of what is in it? (dev tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
big = synth.g2(3, 8192, 8192, 8)
rng = np.random.default_rng(1)
for S in (1024, 2048, 4096):
    cases = {"g2 at this size": synth.g2(3, S, S, 8), "crop of the 8192^2 g2": np.ascontiguousarray(big[:, 1000:1000 + S, 3000:3000 + S]),
             "zeros": np.zeros((3, S, S), np.uint8), "noise": rng.integers(0, 256, (3, S, S), dtype=np.uint8)}
    p = G.TileParams.make(S, S, 3, 8, 5)
    nb = G.lib().grk_amd_tile_num_blocks(p)
    for name, px in cases.items():
        ctx = G.Context(0)
        ctx.set_overlap(False)
        d = torch.from_numpy(px.reshape(-1)).cuda()
        for _ in range(5):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); ctx.enable_timing(True)
        for _ in range(30):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize()
        parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
        k3 = sum(m * c for m, c in parts) / max(max(x[1] for x in parts), 1)
        dwt = ctx.kernel_ms(1)[0]
        t, tot = ctx.fetch_table(nb)
        print("%4d^2 %-24s K3 %.4f ms  DWT %.4f ms  coded %.3f bytes/sample  longest block %d bytes" % (S, name, k3, dwt, tot / (3.0 * S * S), int(t["length"].max())))
        ctx.close()
