"""What slows K3 beside another kernel (dev tool, run under rocprofv3 --kernel-trace): 8K encodes with every kernel alone on the
context's stream while another torch stream runs (CORUN=copy) a plain device copy loop -- memory contention only, few waves --,
(CORUN=alu) a compute-only elementwise loop on a small tensor, or nothing (CORUN=none)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
mode = os.environ.get("CORUN", "none")
S = 8192
ctx = G.Context(0)
px = synth.g2(3, S, S, 8)
p = G.TileParams.make(S, S, 3, 8, 5)
d = torch.from_numpy(px.reshape(-1).copy()).cuda()
side = torch.cuda.Stream()
src = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
small = torch.rand(1 << 22, device="cuda")
for _ in range(2):
    ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
ctx.synchronize(); torch.cuda.synchronize()
with torch.cuda.stream(side):
    for _ in range(40 if mode != "none" else 0):
        if mode == "copy": dst.copy_(src, non_blocking=True)
        else:
            for _ in range(4): small = torch.sin(small) * 1.0001
for _ in range(6):
    ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
ctx.synchronize(); torch.cuda.synchronize()
print("done", mode)
