"""ms per pipelined 8K encode by the number of buffer sets in rotation (dev tool, GPU box): grk_amd_set_pipelining(ctx, n)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
S = int(os.environ.get("PD_SIZE", "8192"))
px = synth.g2(3, S, S, 8)
p = G.TileParams.make(S, S, 3, 8, 5)
ctx = G.Context(0)
stream = torch.cuda.Stream()
ctx.set_stream(stream.cuda_stream)
d = torch.from_numpy(px.reshape(-1)).cuda()
for n in (1, 2, 3, 4, 6, 7, 1):
    ctx.set_pipelining(n)
    with torch.cuda.stream(stream):
        for _ in range(30):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(40):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    torch.cuda.synchronize()
    print("pipelining %d (%d sets): %.4f ms per frame" % (n, n + 1, (time.perf_counter() - t0) / 40 * 1e3))
ctx.set_pipelining(False)
