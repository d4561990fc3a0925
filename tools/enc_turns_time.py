"""ms per frame when consecutive encodes of a sequence go to n contexts in turn (dev tool, GPU box): PD_SIZE, default 4096"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
S = int(os.environ.get("PD_SIZE", "4096"))
px = synth.g2(3, S, S, 8)
p = G.TileParams.make(S, S, 3, 8, 5)
d = torch.from_numpy(px.reshape(-1)).cuda()
ctxs = [G.Context(0) for _ in range(4)]
for c in ctxs:
    c.set_pipelining(True)
for n in (1, 2, 3, 4, 1):
    for k in range(40):
        ctxs[k % n].encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 120
    for k in range(N):
        ctxs[k % n].encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    for c in ctxs: c.synchronize()
    print("%d contexts in turn: %.4f ms per %dx%d frame = %.1f Gpixel/s" % (n, (time.perf_counter() - t0) / N * 1e3, S, S, S * S * N / (time.perf_counter() - t0) / 1e9))
