"""cfg5 (Part-1) decode of a SEQUENCE of frames: ms per frame by frames in flight (grk_amd_set_decode_pipelining); dev tool, GPU box"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth, j2kparse as J, refharness as R
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
path = "/tmp/cfg5_%d.j2k" % S
if not os.path.exists(path):
    R.lib(threads=os.cpu_count() or 1)
    cs, _ = R.encode(synth.g2(3, S, S, 12), 12, numres=6, mode=1, ht=0, irrev=1)
    open(path, "wb").write(cs)
cs = open(path, "rb").read()
R.lib(threads=os.cpu_count() or 1)
ref = R.decode(cs, 3, S, S)
info = J.parse(cs)
p = G.TileParams.make(S, S, 3, 12, info["levels"], irreversible=True, mct=True, part1=True)
blocks, _ = G.tile_layout(p)
rows, data = J.decode_table(info, blocks, True)
table = np.array(rows, dtype=G.capi.CODED_DTYPE)
ctx = G.Context(0)
ctx.set_decode_qcd([(e << 11) | m for e, m in info["qcd"]])
d_c = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).cuda()
outs = [torch.zeros(3 * S * S, dtype=torch.int16, device="cuda") for _ in range(8)]
for n in (1, 2, 3, 4, 6, 8):
    ctx.set_decode_pipelining(n if n > 1 else 0)
    for k in range(2 * n):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), outs[k % n].data_ptr())
    ctx.synchronize(); ctx.decode_status()
    N = 3 * n if n > 1 else 4
    t0 = time.perf_counter()
    for k in range(N):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), outs[k % n].data_ptr())
    t_enq = (time.perf_counter() - t0) / N
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / N
    ok = all(np.array_equal(o.cpu().numpy().view(np.uint16).reshape(3, S, S).astype(np.int32), ref) for o in outs[:min(n, 2)])
    print("%d frames in flight: %.3f ms per frame = %.2f Gpixel/s (host calls returned after %.3f ms per frame)  pixels == grk_decompress: %s"
          % (n, dt * 1e3, S * S / dt / 1e9, t_enq * 1e3, ok))
ctx.set_decode_pipelining(0)
