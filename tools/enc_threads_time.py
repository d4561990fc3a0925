"""ms per frame when n host threads each drive a context of their own through a sequence of encodes (dev tool, GPU box)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
for S in [int(v) for v in os.environ.get("PD_SIZES", "2048,4096,8192").split(",")]:
    px = synth.g2(3, S, S, 8)
    p = G.TileParams.make(S, S, 3, 8, 5)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    ctxs = [G.Context(0) for _ in range(4)]
    for c in ctxs:
        c.set_pipelining(True)
        for _ in range(10): c.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        c.synchronize()
    for n in (1, 2, 3, 4):
        N = 200
        def work(c):
            for _ in range(N): c.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
            c.synchronize()
        ths = [threading.Thread(target=work, args=(ctxs[i],)) for i in range(n)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = (time.perf_counter() - t0) / (N * n)
        print("%5d^2: %d threads x contexts: %.4f ms per frame = %.1f Gpixel/s" % (S, n, dt * 1e3, S * S / dt / 1e9))
    del ctxs
