"""8K encode of flat frames, kernels alone (HIP events, dev tool): which kernel is slow on content that has nothing in it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
S = int(os.environ.get("FF_SIZE", "8192"))
p = G.TileParams.make(S, S, 3, 8, 5)
nb = G.lib().grk_amd_tile_num_blocks(p)
def cases():
    g = synth.g2(3, S, S, 8)
    yield "G2", g
    yield "all zero", np.zeros((3, S, S), np.uint8)
    yield "all 128", np.full((3, S, S), 128, np.uint8)
    h = g.copy(); h[:, S // 2:, :] = 40
    yield "half flat", h
    sp = np.full((3, S, S), 90, np.uint8); sp[:, ::97, :] = 200; sp[:, :, ::131] = 30          # flat with thin lines
    yield "flat + lines", sp
    sm = (np.add.outer(np.arange(S), np.arange(S)) // 64 % 256).astype(np.uint8)                 # a slow ramp: smooth, no noise
    yield "ramp", np.ascontiguousarray(np.broadcast_to(sm, (3, S, S)))
    rng = np.random.default_rng(5)
    yield "noise +-2 on 100", (100 + rng.integers(-2, 3, (3, S, S))).astype(np.uint8)
    yield "noise full", rng.integers(0, 256, (3, S, S), dtype=np.uint8)


for name, px in cases():
    ctx = G.Context(0); ctx.set_overlap(False)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    for _ in range(3):
        ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize(); ctx.enable_timing(True)
    for _ in range(10):
        ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize()
    parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
    k3 = sum(m * c for m, c in parts) / max(max(x[1] for x in parts), 1)
    t, tot = ctx.fetch_table(nb)
    nz = int((t["length"] > 0).sum())
    print("%-18s K3 %.4f ms  DWT %.4f ms  %d of %d blocks carry bytes, %d coded bytes" % (name, k3, ctx.kernel_ms(1)[0], nz, nb, tot))
    ctx.close()
