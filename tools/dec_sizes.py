"""HT decode, ms per call by frame size (dev tool, GPU box): any size at which a decode costs more than a larger one?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
big = synth.g2(3, 4096, 4096, 8)
for S in [int(v) for v in os.environ.get("DEC_SIZES", "512,1024,1536,2048,2304,2560,3072,4096").split(",")]:
    px = np.ascontiguousarray(big[:, :S, :S])
    p = G.TileParams.make(S, S, 3, 8, 5)
    enc = G.Context(0)
    d_px = torch.from_numpy(px.reshape(-1)).cuda()
    table, tot = enc.encode_tiles(p, 1, d_px.data_ptr(), True)
    coded = enc.fetch_coded(tot)
    d_c = torch.from_numpy(np.frombuffer(bytes(coded), np.uint8).copy()).cuda()
    d_out = torch.zeros(3 * S * S, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx = G.Context(0)
    for _ in range(5):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
    ctx.synchronize(); ctx.decode_status()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(30):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    ok = bool(torch.equal(d_out, d_px))
    print("%4d^2: %.4f ms per decode call, K5 %.4f  IDWT %.4f  (%d blocks)  round trip %s" % (S, ms, ctx.kernel_ms(5)[0], ctx.kernel_ms(6)[0], len(table), ok))
    ctx.close(); enc.close()
