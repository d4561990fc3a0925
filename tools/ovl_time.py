"""Encode step time of the 8K and cfg3 workloads with the K3 / DWT overlap off, on, and with consecutive encodes pipelined (dev tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
for name, prec, irrev in (("8k", 8, False), ("cfg3", 16, True)):
    W = H = 8192
    px = synth.g2(3, H, W, prec)
    p = G.TileParams.make(W, H, 3, prec, 5, irreversible=irrev)
    ctx = G.Context(0)
    d = torch.from_numpy(px.reshape(-1).view(np.uint8)).cuda()
    for ov, pipe in ((0, 0), (1, 0), (1, 1), (0, 0), (1, 1)):
        ctx.set_overlap(bool(ov)); ctx.set_pipelining(bool(pipe))
        for _ in range(4):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        print(name, "overlap", ov, "pipelined", pipe, "%.4f ms" % ms, flush=True)
    ctx.set_pipelining(False)
    ctx.close()
