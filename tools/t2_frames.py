"""Ten 8K frames (and ten frames of 64 tiles of 1024 x 1024) through encode_tiles + grk_amd_assemble_device: what tools/t2_route.sh traces."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import grok_amd as G  # noqa: E402
import synth  # noqa: E402

c = G.Context(0)
for T, n in ((8192, 1), (1024, 64)):
    px = synth.g2(3, T, T, 8, seed=3)
    d_px = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(px, (n,) + px.shape))).cuda()
    p = G.TileParams.make(T, T, 3, 8, 5)
    idx = list(range(n))
    for it in range(11):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        c.encode_tiles(p, n, d_px.data_ptr(), True, fetch=False)
        nb, lens = c.assemble_device(p, idx, 0)
    c.synchronize()
    print("%d tile(s) of %d^2: %.3f ms per frame (encode + assemble, nothing fetched), %d bytes" % (n, T, (time.perf_counter() - t0) / 10 * 1e3, nb))
