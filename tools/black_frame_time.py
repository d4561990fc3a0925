"""Pipelined 8K encode, ms per frame by content (dev tool): real content vs an all-zero (black) frame, whose LL band is K3's worst case"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
S = 8192
p = G.TileParams.make(S, S, 3, 8, 5)
stream = torch.cuda.Stream()
for name, px in (("G2", synth.g2(3, S, S, 8)), ("all zero", np.zeros((3, S, S), np.uint8)), ("all 255", np.full((3, S, S), 255, np.uint8)),
                 ("value 16 (video black)", np.full((3, S, S), 16, np.uint8))):
    ctx = G.Context(0); ctx.set_stream(stream.cuda_stream); ctx.set_pipelining(1)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    with torch.cuda.stream(stream):
        for _ in range(10):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(30):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    ctx.set_pipelining(False)
    nb = G.lib().grk_amd_tile_num_blocks(p)
    t, tot = ctx.fetch_table(nb)
    print("%-24s %.4f ms per frame, %d coded bytes, longest block %d" % (name, ms, tot, int(t["length"].max())))
    ctx.close()
