"""As tools/hwq_alias.py, for the DECODE side: ms per 8K HT frame decoded as a sequence (grk_amd_set_decode_pipelining(n)) and one frame at a
time, by the number of streams the process made before the context.  python tools/hwq_alias_dec.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, grok_amd as G, synth  # noqa: E401
    k = int(sys.argv[2])
    px = synth.g2(3, 8192, 8192, 8)
    p = G.TileParams.make(8192, 8192, 3, 8, 5)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    extra = []
    for _ in range(k):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(1024, device="cuda").add_(1)
        extra.append(s)
    torch.cuda.synchronize()
    ctx = G.Context(0)
    nb = G.lib().grk_amd_tile_num_blocks(p) if False else None
    import ctypes as C
    nb = G.lib().grk_amd_tile_num_blocks(C.byref(p))
    ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    table, total = ctx.fetch_table(nb)
    coded = torch.empty(total, dtype=torch.uint8, device="cuda")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy(ctypes.c_void_p(coded.data_ptr()), ctypes.c_void_p(ctx.coded_device_ptr()), ctypes.c_size_t(total), 3)
    outs = [torch.empty_like(d) for _ in range(4)]
    res = []
    for n in (0, 2, 3, 4):
        ctx.set_decode_pipelining(n)
        m = max(n, 1)
        for i in range(2 * m):
            ctx.decode_device(p, 1, table, coded.data_ptr(), total, outs[i % m].data_ptr())
        ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(12):
                ctx.decode_device(p, 1, table, coded.data_ptr(), total, outs[i % m].data_ptr())
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 12 * 1e3)
        ctx.decode_status()
        assert torch.equal(outs[0], d)
        res.append("%.3f" % best)
    ctx.set_decode_pipelining(0)
    print("/".join(res))
    sys.exit(0)

for q in (4, 8):
    row = []
    for k in (0, 1, 2, 3, 4, 5, 6, 7, 8):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=str(q))
        r = subprocess.run([sys.executable, __file__, "child", str(k)], env=env, capture_output=True, text=True, timeout=300)
        row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "fail:" + r.stderr[-200:])
    print("queues %d: ms per 8K HT frame (frames in flight 1/2/3/4) by earlier streams 0..8:\n   %s" % (q, "\n   ".join(row)), flush=True)
