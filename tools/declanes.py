import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch, grok_amd as G, synth
px = synth.g2(3, 8192, 8192, 8); p = G.TileParams.make(8192, 8192, 3, 8, 5)
ctx = G.Context(0); d = torch.from_numpy(px.reshape(-1)).cuda()
table, tot = ctx.encode_tiles(p, 1, d.data_ptr(), True)
back = torch.empty_like(d)
for _ in range(2): ctx.decode_device(p, 1, table, ctx.coded_device_ptr(), tot, back.data_ptr())
ctx.decode_status(); ctx.enable_timing(True)
for _ in range(5): ctx.decode_device(p, 1, table, ctx.coded_device_ptr(), tot, back.data_ptr())
ctx.synchronize()
print(os.environ.get("GRK_AMD_VLC_LANES"), "ht_decode_ms", round(ctx.kernel_ms(5)[0], 4), "ok", bool(torch.equal(back, d)))
