"""Two contexts decoding 8K frames alternately (dev tool): does the HT decode of one frame hide beside another's?  Frames per
second of ONE context back to back against TWO contexts fed in turn (their kernels co-run as the GPU lets them)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = 8192
px = synth.g2(3, H, W, 8, seed=12345)
p = G.TileParams.make(W, H, 3, 8, 5)
NC = int(os.environ.get("NCTX", "2"))
ctxs = [G.Context(0) for _ in range(NC)]
d = torch.from_numpy(px.reshape(-1)).cuda()
nb = G.lib().grk_amd_tile_num_blocks(p)
tabs, coded, backs = [], [], []
for c in ctxs:
    c.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    t, tot = c.fetch_table(nb)
    tabs.append((t, tot)); coded.append(c.coded_device_ptr()); backs.append(torch.empty_like(d))
N = 20
def run(k):
    for _ in range(3):
        for i in range(k): ctxs[i].decode_device(p, 1, tabs[i][0], coded[i], tabs[i][1], backs[i].data_ptr())
    for i in range(k): ctxs[i].decode_status()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        for i in range(k): ctxs[i].decode_device(p, 1, tabs[i][0], coded[i], tabs[i][1], backs[i].data_ptr())
    for i in range(k): ctxs[i].decode_status()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (N * k) * 1e3
for k in range(1, NC + 1):
    print("%-16s %d context(s): %.4f ms per frame   round trip %s" % (os.environ.get("TAG", ""), k, run(k), all(bool(torch.equal(b, d)) for b in backs[:k])))
