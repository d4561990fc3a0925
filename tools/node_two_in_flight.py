"""The node's file route as a SEQUENCE: two grk_amd_node objects on one GPU, each driven by a host thread of its own (ctypes releases the GIL),
so that one frame's D2H runs beside the other's encode + Tier-2.  ms per frame, pinned and pageable output.  dev tool, GPU box"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
px = synth.g2(3, 8192, 8192, 8)
d_px = torch.from_numpy(px.reshape(-1)).cuda()
layout = G.ImageLayout.make(8192, 8192, 8192, 8192)
base = G.TileParams.make(1, 1, 3, 8, 5)
c = G.Context(0)
for name in ("pinned", "pageable"):
    for nthreads in (1, 2, 3):
        nodes = [G.Node([0]) for _ in range(nthreads)]
        outs = [c.host_array(px.size * 2 + (1 << 20)) if name == "pinned" else np.zeros(px.size * 2 + (1 << 20), np.uint8) for _ in range(nthreads)]
        N = 12
        def run(i, n):
            for _ in range(n):
                nodes[i].encode_image_device(layout, base, d_px.data_ptr(), d_px.numel(), 0, 0, out=outs[i])
        for i in range(nthreads):
            run(i, 2)
        th = [threading.Thread(target=run, args=(i, N)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = (time.perf_counter() - t0) / (N * nthreads)
        print("%s output, %d node(s) in flight: %.3f ms per frame" % (name, nthreads, dt * 1e3), flush=True)
        for n in nodes: n.close()
