"""HT decode of flat 8K frames vs real content, ms per call (dev tool)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
S = 8192
p = G.TileParams.make(S, S, 3, 8, 5)
for name, px in (("G2", synth.g2(3, S, S, 8)), ("all zero", np.zeros((3, S, S), np.uint8)), ("all 128", np.full((3, S, S), 128, np.uint8)),
                 ("half flat", np.concatenate([synth.g2(3, S // 2, S, 8), np.full((3, S // 2, S), 40, np.uint8)], axis=1)),
                 ("noise +-2", (100 + np.random.default_rng(5).integers(-2, 3, (3, S, S))).astype(np.uint8))):
    # (full-range noise is outside the format's dynamic-range contract: its RCT chroma exceeds the bands' Kmax, the encoder's blocks
    #  equal the oracle encoder's and BOTH decoders -- ours and the reference's ojph_decode_codeblock -- reject them:
    #  tools/noise_block_bisect.py)
    px = np.ascontiguousarray(px)
    enc = G.Context(0)
    d_px = torch.from_numpy(px.reshape(-1)).cuda()
    table, tot = enc.encode_tiles(p, 1, d_px.data_ptr(), True)
    d_c = torch.from_numpy(np.frombuffer(bytes(enc.fetch_coded(tot)), np.uint8).copy()).cuda()
    d_out = torch.zeros(3 * S * S, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = G.Context(0)
    for _ in range(3):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
    ctx.synchronize(); ctx.decode_status()
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print("%-10s decode %.4f ms per call, round trip %s, sum of block lengths %d" % (name, ms, bool(torch.equal(d_out, d_px)), int(table["length"].sum())))
    ctx.close(); enc.close()
