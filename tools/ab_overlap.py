"""A/B of GRK_AMD_OVERLAP (K3 of the top resolution beside DWT levels >= 1): ms per 8K encode step (dev tool)."""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import grok_amd.capi as capi
    if os.environ.get("AB_LIB"):
        capi.lib_path = lambda: os.environ["AB_LIB"]
    import numpy as np, torch, grok_amd as G, synth
    W = H = int(os.environ.get("ABL_SIZE", "8192"))
    px = synth.g2(3, H, W, 8)
    p = G.TileParams.make(W, H, 3, 8, 5)
    ctx = G.Context(0)
    ctx.set_pipelining(os.environ.get("AB_PIPE") == "1")
    d = torch.from_numpy(px.reshape(-1)).cuda()
    for _ in range(5): ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n): ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    ctx.synchronize()
    print(os.path.basename(os.path.dirname(os.environ.get("AB_LIB", "x/current/l"))), "pipe", os.environ.get("AB_PIPE"), "rest2", os.environ.get("GRK_AMD_REST_SIDE2"), "overlap", os.environ.get("GRK_AMD_OVERLAP"), "ms/step %.4f" % ((time.perf_counter() - t0) / n * 1e3))
else:
    variants = [("", "1", "1"), ("", "1", "2")]
    d = os.path.join(ROOT, "build", "abl")
    if os.path.isdir(d):
        variants = [(os.path.join(d, n, "libgrok_amd.so"), "0", "0") for n in sorted(os.listdir(d))] + variants
    for rep in range(2):
        for lib, ov, pipe in variants:
            env = dict(os.environ, GRK_AMD_OVERLAP=ov, AB_PIPE="1" if pipe != "0" else "0")
            if pipe == "2": env["GRK_AMD_REST_SIDE2"] = "1"
            if lib: env["AB_LIB"] = lib
            r = subprocess.run([sys.executable, __file__, "--one"], capture_output=True, text=True, env=env)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
