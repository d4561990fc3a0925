"""Small driver for rocprofv3 passes: N encodes of the 8K workload (dev tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = int(os.environ.get("PROF_SIZE", "8192"))
n = int(os.environ.get("PROF_N", "4"))
px = synth.g2(3, H, W, 8)
p = G.TileParams.make(W, H, 3, 8, 5)
ctx = G.Context(0)
d = torch.from_numpy(px.reshape(-1)).cuda()
for _ in range(n):
    ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
ctx.synchronize()
if os.environ.get("PROF_DECODE", "1") == "1":
    nb = G.lib().grk_amd_tile_num_blocks(p)
    table, tot = ctx.fetch_table(nb)
    back = torch.empty_like(d)
    for _ in range(n):
        ctx.decode_device(p, 1, table, ctx.coded_device_ptr(), tot, back.data_ptr())
    ctx.decode_status()
    assert torch.equal(back, d)
print("done")
