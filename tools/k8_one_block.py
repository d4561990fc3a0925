import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import grok_amd as G, refharness as R
import gpuutil as U
W = H = 64; bits = 12
rng = np.random.default_rng(5)
p = G.TileParams.make(W, H, 1, 8, 0, part1=True)
blocks, _ = G.tile_layout(p)
mag = rng.integers(0, 1 << bits, size=(H, W))
coef = (mag * np.where(rng.random((H, W)) < 0.5, -1, 1)).astype(np.int32)
cb, npass, nbps = R.t1_encode_block(coef, blocks[0].band)
table = np.zeros(1, G.capi.CODED_DTYPE)
table["offset"][0] = 0; table["length"][0] = len(cb); table["missing_msbs"][0] = nbps | (npass << 8)
d_c = U.to_dev(np.frombuffer(cb + b"\0" * 64, np.uint8)); d_m = U.dev_planes(p, 1)
c = U.ctx()
c.stage_ht_decode(p, 1, table, d_c.data_ptr(), d_c.numel(), d_m.data_ptr()); c.synchronize()
