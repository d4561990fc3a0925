"""Time the HT kernel of each ablation variant under build/abl/*/ (dev tool, timing only)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import grok_amd.capi as capi
    libp = sys.argv[2]
    capi.lib_path = lambda: libp
    import numpy as np, torch, grok_amd as G, synth
    W = H = int(os.environ.get("ABL_SIZE", "8192"))
    px = synth.g2(3, H, W, 8)
    p = G.TileParams.make(W, H, 3, 8, 5)
    ctx = G.Context(0)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    ctx.set_overlap(False)
    for _ in range(3): ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize(); ctx.enable_timing(True)
    for _ in range(10): ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize()
    enc = {"dwt": round(ctx.kernel_ms(1)[0], 4), "ht": round(ctx.kernel_ms(2)[0] + ctx.kernel_ms(4)[0] + ctx.kernel_ms(8)[0], 4), "all": round(ctx.kernel_ms(3)[0], 4)}
    nb = G.lib().grk_amd_tile_num_blocks(p)
    table, tot = ctx.fetch_table(nb)
    back = torch.empty_like(d)
    ctx.enable_timing(False)
    for _ in range(2): ctx.decode_device(p, 1, table, ctx.coded_device_ptr(), tot, back.data_ptr())
    ctx.synchronize(); ctx.enable_timing(True)
    for _ in range(8): ctx.decode_device(p, 1, table, ctx.coded_device_ptr(), tot, back.data_ptr())
    ctx.synchronize()
    print(json.dumps({"lib": os.path.basename(os.path.dirname(libp)), **enc, "dec_ht": round(ctx.kernel_ms(5)[0], 4), "idwt": round(ctx.kernel_ms(6)[0], 4), "dec_all": round(ctx.kernel_ms(3)[0], 4)}))
else:
    d = os.path.join(ROOT, "build", "abl")
    for name in sorted(os.listdir(d)):
        lp = os.path.join(d, name, "libgrok_amd.so")
        if os.path.exists(lp):
            r = subprocess.run([sys.executable, __file__, "--one", lp], capture_output=True, text=True)
            print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
