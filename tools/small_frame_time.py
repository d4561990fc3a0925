"""ms per encode call by frame size, with and without the side-stream overlap / the buffer-set rotation (dev tool, GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
stream = torch.cuda.Stream()
sizes = [(int(v), 5 if int(v) >= 1024 else 3) for v in os.environ.get('SF_SIZES', '512,1024,2048,4096').split(',')]
for S, L in sizes:
    px = synth.g2(3, S, S, 8)
    p = G.TileParams.make(S, S, 3, 8, L)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    row = []
    for overlap, pipe in ((True, 1), (True, 0), (False, 0)):
        ctx = G.Context(0)
        ctx.set_stream(stream.cuda_stream)
        ctx.set_overlap(overlap)
        ctx.set_pipelining(pipe)
        ctx.set_pixel_hold(os.environ.get("SF_HOLD", "0") == "1")
        with torch.cuda.stream(stream):
            for _ in range(30):
                ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(100):
                ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 100 * 1e3)
        ctx.set_pipelining(False)
        ctx.close()
    print("%4d^2: overlap + rotation %.4f   overlap only %.4f   one stream %.4f ms per call" % (S, row[0], row[1], row[2]))
