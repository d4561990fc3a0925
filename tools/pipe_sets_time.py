"""The pipelined 8K step by buffer sets in rotation (grk_amd_set_pipelining(ctx, n)): does the rotation's depth cost anything? (r05)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = 8192
rot = [torch.from_numpy(synth.g2(3, H, W, 8, seed=s).reshape(-1)).cuda() for s in (12345, 777, 424242)]
p = G.TileParams.make(W, H, 3, 8, 5)
ctx = G.Context(0)
def region(steps, warm):
    for f in range(warm): ctx.encode_tiles(p, 1, rot[f % 3].data_ptr(), True, fetch=False)
    ctx.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in range(steps): ctx.encode_tiles(p, 1, rot[(warm + f) % 3].data_ptr(), True, fetch=False)
    ctx.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for n in (1, 2, 4, 6, 7, 1):
    ctx.set_pipelining(n); region(40, 0)
    r = sorted(region(20, 3) for _ in range(5))
    print("set_pipelining(%d): %d buffer sets  step median %.4f  min %.4f  max %.4f ms" % (n, n + 1, r[2], r[0], r[-1]), flush=True)
    ctx.set_pipelining(False)
