"""Full-range noise through encode + decode at growing sizes (dev tool): where does the round trip break, and on whose side?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
import oracle as O
rng = np.random.default_rng(6)
big = rng.integers(0, 256, (3, 8192, 8192), dtype=np.uint8)
for S in (512, 1024, 2048, 4096, 8192):
    px = np.ascontiguousarray(big[:, :S, :S])
    p = G.TileParams.make(S, S, 3, 8, 5)
    ctx = G.Context(0)
    table, coded = ctx.encode_host(p, px)
    try:
        back = ctx.decode_host(p, table, coded)
        ok = np.array_equal(np.asarray(back).reshape(px.shape), px)
        print("%4d^2: round trip %s" % (S, ok))
    except Exception as e:
        print("%4d^2: decode failed: %s" % (S, e))
        if S <= 2048:
            blocks, lens, ocoded = O.encode_tile_rev(px, 8, 5)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            got = [bytes(coded[int(o):int(o) + int(l)]) for o, l in zip(table["offset"], table["length"])]
            bad = [i for i in range(len(blocks)) if got[i] != bytes(ocoded[off[i]:off[i + 1]])]
            print("       encoder blocks differing from the oracle encoder: %d of %d %s" % (len(bad), len(blocks), bad[:5]))
        break
    ctx.close()
