"""A/B of K3 builds on ONE box with bench.py's own timers, builds alternated:
    python tools/k3_ab.py [--rounds R] variant ...     variant = head | a name under build/abl/<name>/libgrok_amd.so
Per build and round: K3 alone (HIP events, grk_amd_set_overlap(0), 60 launches), the DWT family alone, the pipelined 8K step as
bench.py times it (three frames in rotation, 5 regions of 20 steps behind 40 pre-warm frames: median / min / max), cfg3's K3 alone,
and the md5 of the 8K codestream (must stay 7e5275ef...).  The pixels are generated once and shared through /tmp."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
PIX = "/tmp/k3_ab_pixels_%s.npy"


def pixels():
    import numpy as np, synth
    for name, prec, seed in (("8k_0", 8, 12345), ("8k_1", 8, 777), ("8k_2", 8, 424242), ("cfg3", 16, 12345)):
        if not os.path.exists(PIX % name):
            np.save(PIX % name, synth.g2(3, 8192, 8192, prec, seed=seed))


def one(lib, want_cfg3):
    import numpy as np, torch
    os.environ["GRK_AMD_LIB"] = lib
    import grok_amd as G
    out = {}
    W = H = 8192
    rot = [torch.from_numpy(np.load(PIX % ("8k_%d" % i)).reshape(-1)).cuda() for i in range(3)]
    p = G.TileParams.make(W, H, 3, 8, 5)
    ctx = G.Context(0)

    def region(steps, warm, base):
        for f in range(warm):
            ctx.encode_tiles(p, 1, rot[(base + f) % 3].data_ptr(), True, fetch=False)
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(steps):
            ctx.encode_tiles(p, 1, rot[(base + warm + f) % 3].data_ptr(), True, fetch=False)
        ctx.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    ctx.set_pipelining(True)
    region(40, 0, 0)
    regs = sorted(region(20, 3, i) for i in range(5))
    out["step"] = [round(regs[2], 4), round(regs[0], 4), round(regs[-1], 4)]
    ctx.set_pipelining(False)
    if want_cfg3:                                   # (first round only) BASELINE configs[1]: 4096^2, pipelined, on a crop of the same frames
        p2 = G.TileParams.make(4096, 4096, 3, 8, 5)
        r2 = [t.view(3, 8192, 8192)[:, :4096, :4096].contiguous().view(-1) for t in rot]
        ctx.set_pipelining(True)

        def region2(steps):
            for f in range(5):
                ctx.encode_tiles(p2, 1, r2[f % 3].data_ptr(), True, fetch=False)
            ctx.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in range(steps):
                ctx.encode_tiles(p2, 1, r2[f % 3].data_ptr(), True, fetch=False)
            ctx.synchronize(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        region2(40)
        out["cfg2_step"] = round(sorted(region2(40) for _ in range(5))[2], 4)
        ctx.set_pipelining(False)
        del r2

    def alone(prm, d, n):
        ctx.set_overlap(False)
        for _ in range(3):
            ctx.encode_tiles(prm, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); ctx.enable_timing(True)
        for _ in range(n):
            ctx.encode_tiles(prm, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize()
        parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
        k3 = sum(m * c for m, c in parts) / max(max(x[1] for x in parts), 1)
        dwt = ctx.kernel_ms(1)[0]
        ctx.enable_timing(False); ctx.set_overlap(True)
        return round(k3, 4), round(dwt, 4)
    out["k3"], out["dwt"] = alone(p, rot[0], 60)
    nb = G.lib().grk_amd_tile_num_blocks(p)
    table, tot = ctx.fetch_table(nb)
    out["md5"] = hashlib.md5(G.write_codestream(p, W, H, table, ctx.fetch_coded(tot))).hexdigest()[:8]
    if want_cfg3:
        d3 = torch.from_numpy(np.load(PIX % "cfg3").reshape(-1).view(np.uint8)).cuda()
        p3 = G.TileParams.make(W, H, 3, 16, 5, irreversible=True)
        out["cfg3_k3"], _ = alone(p3, d3, 30)
        t3, tot3 = ctx.fetch_table(nb)
        c3 = ctx.fetch_coded(tot3)
        out["cfg3_md5"] = hashlib.md5(b"".join(bytes(c3[int(o):int(o) + int(l)]) for o, l in zip(t3["offset"], t3["length"]))).hexdigest()[:8]
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2], sys.argv[3] == "1")
        sys.exit(0)
    args = sys.argv[1:]
    rounds = 2
    if args and args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    pixels()
    res = {}
    for r in range(rounds):
        for v in args or ["head"]:
            name, _, envs = v.partition("@")                 # build@ENV=VAL,ENV=VAL
            env = dict(os.environ)
            env.update(dict(kv.split("=", 1) for kv in envs.split(",") if kv))
            lib = os.path.join(ROOT, "grok_amd", "lib", "libgrok_amd.so") if name == "head" else os.path.join(ROOT, "build", "abl", name, "libgrok_amd.so")
            q = subprocess.run([sys.executable, __file__, "--one", lib, "1" if r == 0 else "0"], capture_output=True, text=True, env=env)
            line = (q.stdout.strip().splitlines() or ["FAILED " + q.stderr[-600:]])[-1]
            print("round %d %-28s %s" % (r, v, line), flush=True)
            try:
                res.setdefault(v, []).append(json.loads(line))
            except Exception:
                pass
    print("\n%-28s %-28s %-28s %s" % ("build", "K3 alone ms (per round)", "step median ms (per round)", "md5 / cfg3"))
    for v, rs in res.items():
        print("%-28s %-28s %-28s %s %s" % (v, " ".join("%.4f" % x["k3"] for x in rs), " ".join("%.4f" % x["step"][0] for x in rs),
                                           rs[0]["md5"], " ".join("%s=%s" % (k, rs[0][k]) for k in ("cfg2_step", "cfg3_k3", "cfg3_md5") if k in rs[0])))
