"""K3 alone by frame size (dev tool): the curve between 1024^2 and 4096^2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
big = synth.g2(3, 4096, 4096, 8)
import os
for S in [int(v) for v in os.environ.get('K3_SIZES', '1024,1280,1536,1792,2048,2304,2560,3072,3584,4096').split(',')]:
    px = np.ascontiguousarray(big[:, :S, :S])
    d = torch.from_numpy(px.reshape(-1)).cuda()
    p = G.TileParams.make(S, S, 3, 8, 5)
    nb = G.lib().grk_amd_tile_num_blocks(p)
    ctx = G.Context(0); ctx.set_overlap(False)
    for _ in range(5):
        ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize(); ctx.enable_timing(True)
    for _ in range(30):
        ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize()
    parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
    k3 = sum(m * c for m, c in parts) / max(max(x[1] for x in parts), 1)
    print("%4d^2: K3 %.4f ms, %5d blocks (%.1f per CU), parts %s" % (S, k3, nb, nb / 256.0, [(round(m, 4), c) for m, c in parts]))
    ctx.close()
