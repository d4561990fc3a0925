"""A/B of library variants on the tiled workload (64 tiles of 1024^2) and the 8K one: DWT family time, overlap off (dev tool)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import grok_amd.capi as capi
    libp = sys.argv[2]
    capi.lib_path = lambda: libp
    import numpy as np, torch, grok_amd as G, synth
    out = {"lib": os.path.basename(os.path.dirname(libp))}
    for name, (W, H, nt) in (("tiles", (1024, 1024, 64)), ("8k", (8192, 8192, 1))):
        px = synth.g2(3, H, W, 8)
        p = G.TileParams.make(W, H, 3, 8, 5)
        ctx = G.Context(0); ctx.set_overlap(False)
        d = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(px, (nt,) + px.shape)).reshape(-1)).cuda()
        for _ in range(3): ctx.encode_tiles(p, nt, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); ctx.enable_timing(True)
        for _ in range(10): ctx.encode_tiles(p, nt, d.data_ptr(), True, fetch=False)
        ctx.synchronize()
        out[name] = {"dwt": round(ctx.kernel_ms(1)[0], 4), "all": round(ctx.kernel_ms(3)[0], 4)}
        ctx.close()
    print(json.dumps(out))
else:
    d = os.path.join(ROOT, "build", "abl")
    for rep in range(2):
        for name in sorted(os.listdir(d)):
            r = subprocess.run([sys.executable, __file__, "--one", os.path.join(d, name, "libgrok_amd.so")], capture_output=True, text=True)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
