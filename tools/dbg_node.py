import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, grok_amd as G, synth
W,H,TW,TH,L,off,prec,flags = 700,530,200,150,3,(33,17),8,G.CS_TLM|G.CS_PLT
px = synth.g2(3,H,W,prec,seed=W+2)
layout = G.ImageLayout.make(W,H,TW,TH,offset=off)
base = G.TileParams.make(1,1,3,prec,L)
c = G.Context(0)
want = c.encode_image(layout, base, px, flags)
node = G.Node([0,0])
for it in range(3):
    got = bytes(node.encode_image(layout, base, px, flags | G.NODE_GATHER))
    print("iter", it, len(got), len(want), got == want)
    if got != want:
        a = G.locate_tile_parts(want); b = G.locate_tile_parts(got)
        for (oa, la, ta), (ob, lb, tb) in zip(zip(*a[:3]), zip(*b[:3])):
            same = want[oa:oa+la] == got[ob:ob+lb]
            if not same:
                d = [i for i in range(min(la,lb)) if want[oa+i] != got[ob+i]]
                print(" tile", ta, "len", la, lb, "first diff at", d[:3], "ndiff", len(d))
