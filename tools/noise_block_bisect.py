"""Which blocks of a 512^2 full-range-noise tile does the GPU HT decoder reject? (dev tool)  Each block decoded alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G
import oracle as O
rng = np.random.default_rng(6)
S = 512
px = np.ascontiguousarray(rng.integers(0, 256, (3, 8192, 8192), dtype=np.uint8)[:, :S, :S])
p = G.TileParams.make(S, S, 3, 8, 5)
ctx = G.Context(0)
table, coded = ctx.encode_host(p, px)
blocks, _ = G.tile_layout(p)
bad = []
for i in range(len(table)):
    t = table.copy()
    t["length"][:] = 0
    t["length"][i] = table["length"][i]
    try:
        ctx.decode_host(p, t, coded)
    except Exception:
        bad.append(i)
print("%d of %d blocks rejected when decoded alone: %s" % (len(bad), len(table), bad[:12]))
for i in bad[:6]:
    b = blocks[i]
    cb = bytes(coded[int(table["offset"][i]):int(table["offset"][i]) + int(table["length"][i])])
    w, h = b.x1 - b.x0, b.y1 - b.y0
    try:
        sm = O.ht_decode_block(cb, int(table["missing_msbs"][i]), w, h)
        orc = "oracle accepts it (max magnitude bits %d)" % int(np.max(sm & 0x7FFFFFFF)).bit_length()
    except Exception as e:
        orc = "oracle rejects it too: %s" % e
    lcup = len(cb); scup = ((cb[-1] << 4) | (cb[-2] & 0xF)) if lcup >= 2 else -1
    print("block %d: comp %d res %d band %d %dx%d kmax %d missing_msbs %d length %d Scup %d -- %s" % (
        i, b.comp, b.res, b.band, w, h, b.kmax, int(table["missing_msbs"][i]), lcup, scup, orc))
