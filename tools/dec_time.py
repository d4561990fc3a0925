"""HT decode at 8K on the GPU (dev tool): per step, the K5 family (HIP events), the inverse DWT family, the whole
decode_tiles call back to back, and a check that the pixels come back.  Environment switches of the build under test are
passed through (e.g. GRK_AMD_K5A_LANES)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
W = H = int(os.environ.get("PROF_SIZE", "8192"))
NT = int(os.environ.get("PROF_TILES", "1"))         # frames per decode call (a batch of tiles of one geometry)
px = np.stack([synth.g2(3, H, W, 8, seed=12345 + t) for t in range(NT)])
p = G.TileParams.make(W, H, 3, 8, 5)
ctx = G.Context(0)
d = torch.from_numpy(px.reshape(-1)).cuda()
ctx.encode_tiles(p, NT, d.data_ptr(), True, fetch=False)
nb = G.lib().grk_amd_tile_num_blocks(p) * NT
table, tot = ctx.fetch_table(nb)
back = torch.empty_like(d)
N = int(os.environ.get("PROF_N", "20"))
for _ in range(3):
    ctx.decode_device(p, NT, table, ctx.coded_device_ptr(), tot, back.data_ptr())
ctx.decode_status()
ctx.enable_timing(True)
for _ in range(N):
    ctx.decode_device(p, NT, table, ctx.coded_device_ptr(), tot, back.data_ptr())
ctx.decode_status()
k5, k6 = ctx.kernel_ms(5)[0], ctx.kernel_ms(6)[0]
ctx.enable_timing(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    ctx.decode_device(p, NT, table, ctx.coded_device_ptr(), tot, back.data_ptr())
ctx.decode_status()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
print("%-28s frames/call %d  K5 %.4f ms  idwt %.4f ms  decode back-to-back %.4f ms/call = %.4f ms/frame  round trip %s" % (
    os.environ.get("TAG", ""), NT, k5, k6, ms, ms / NT, bool(torch.equal(back, d))))
