import os, sys, time
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
S = int(sys.argv[1])
stream = torch.cuda.Stream()
px = synth.g2(3, S, S, 8)
p = G.TileParams.make(S, S, 3, 8, 5)
d = torch.from_numpy(px.reshape(-1)).cuda()
ctx = G.Context(0); ctx.set_stream(stream.cuda_stream); ctx.set_pipelining(int(os.environ.get('SF_PIPE', '1'))); ctx.set_overlap(os.environ.get('SF_OVERLAP', '1') != '0')
with torch.cuda.stream(stream):
    for _ in range(100):
        ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
torch.cuda.synchronize()
