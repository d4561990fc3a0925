"""Where the node's file route spends an 8K frame (GRK_AMD_NODE_TRACE=1; pinned and pageable output).  dev tool, GPU box"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
px = synth.g2(3, 8192, 8192, 8)
d_px = torch.from_numpy(px.reshape(-1)).cuda()
node = G.Node([0])
layout = G.ImageLayout.make(8192, 8192, 8192, 8192)
base = G.TileParams.make(1, 1, 3, 8, 5)
c = G.Context(0)
for name, out in (("pinned", c.host_array(px.size * 2 + (1 << 20))), ("pageable", np.zeros(px.size * 2 + (1 << 20), np.uint8))):
    for i in range(4):
        if i == 3:
            os.environ["GRK_AMD_NODE_TRACE"] = "1"
            sys.stderr.write("== %s output\n" % name)
        t0 = time.perf_counter()
        cs = node.encode_image_device(layout, base, d_px.data_ptr(), d_px.numel(), 0, 0, out=out)
        dt = time.perf_counter() - t0
    os.environ.pop("GRK_AMD_NODE_TRACE", None)
    sys.stderr.write("   whole call %.3f ms, %d bytes\n" % (dt * 1e3, len(cs)))
