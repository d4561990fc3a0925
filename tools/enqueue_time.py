"""Host time per pipelined encode call against the GPU's period, by frame size: is a size bound by the host's runtime calls? (dev tool, GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
stream = torch.cuda.Stream()
for S in (2048, 4096, 8192):
    px = synth.g2(3, S, S, 8)
    p = G.TileParams.make(S, S, 3, 8, 5)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    ctx = G.Context(0)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_pixel_hold(os.environ.get("SF_HOLD", "0") == "1")
    ctx.set_pipelining(2)
    for _ in range(30):
        ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    t_host = (time.perf_counter() - t0) / N
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / N
    print("%d^2: host returns after %.4f ms per call, the GPU's period %.4f ms per frame" % (S, t_host * 1e3, t_all * 1e3))
    ctx.set_pipelining(False); ctx.close()
