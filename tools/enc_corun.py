"""Pipelined 8K encodes of ONE context back to back against TWO (three) contexts fed in turn (dev tool): with two contexts the
DWT chain of one frame no longer waits behind the last small level of the frame before."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = 8192
px = synth.g2(3, H, W, 8, seed=12345)
p = G.TileParams.make(W, H, 3, 8, 5)
NC = int(os.environ.get("NCTX", "3"))
ctxs = [G.Context(0) for _ in range(NC)]
d = torch.from_numpy(px.reshape(-1)).cuda()
for c in ctxs:
    c.set_overlap(True); c.set_pipelining(True)
N = 24
def run(k):
    for _ in range(3):
        for i in range(k): ctxs[i].encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    for i in range(k): ctxs[i].synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        for i in range(k): ctxs[i].encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    for i in range(k): ctxs[i].synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (N * k) * 1e3
for k in range(1, NC + 1):
    print("%-16s %d context(s): %.4f ms per frame" % (os.environ.get("TAG", ""), k, run(k)))
