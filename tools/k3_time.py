"""K3 (HT cleanup encoder) alone on the GPU, for one or several builds of the library, on the same box:
    python tools/k3_time.py [variant ...]      variant = head | a name under build/abl/<name>/libgrok_amd.so
8192^2 x 3 8-bit G2 (the bench workload) and the cfg3 shape (16-bit ICT + 9/7); per build: K3 ms per frame (HIP events,
every kernel alone), DWT ms, pipelined ms per frame, and the md5 of the 8K codestream (must stay 7e5275ef...)."""
import hashlib, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def one(lib):
    import numpy as np, torch, time
    import grok_amd.capi as capi
    capi.lib_path = lambda: lib
    import grok_amd as G, synth
    out = {}
    for name, prec, irrev in (("8k", 8, False), ("cfg3", 16, True)):
        W = H = 8192
        px = synth.g2(3, H, W, prec)
        p = G.TileParams.make(W, H, 3, prec, 5, irreversible=irrev)
        ctx = G.Context(0)
        d = torch.from_numpy(px.reshape(-1).view(np.uint8)).cuda()
        ctx.set_overlap(False)
        for _ in range(3):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); ctx.enable_timing(True)
        for _ in range(20):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize()
        parts = [ctx.kernel_ms(i) for i in (2, 4, 8)]
        n = max(x[1] for x in parts)
        k3 = sum(m * c for m, c in parts) / max(n, 1)
        dwt = ctx.kernel_ms(1)[0]
        ctx.enable_timing(False)
        ctx.set_overlap(True); ctx.set_pipelining(True)
        for _ in range(3):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
        ctx.synchronize(); pipe = (time.perf_counter() - t0) / 20 * 1e3
        ctx.set_pipelining(False)
        nb = G.lib().grk_amd_tile_num_blocks(p)
        table, tot = ctx.fetch_table(nb)
        coded = ctx.fetch_coded(tot)
        if not irrev:
            md5 = hashlib.md5(G.write_codestream(p, W, H, table, coded)).hexdigest()
        else:
            md5 = hashlib.md5(b"".join(bytes(coded[int(o):int(o) + int(l)]) for o, l in zip(table["offset"], table["length"]))).hexdigest()
        class _H: pass
        hh = _H(); hh.__cuda_array_interface__ = {"shape": (24,), "typestr": "<i8", "data": (int(ctx.table_device_ptr(3)), False), "version": 2}
        handed = int(torch.as_tensor(hh, device="cuda").cpu().sum())
        out[name] = {"k3_ms": round(k3, 4), "fallback_blocks": handed, "dwt_ms": round(dwt, 4), "pipelined_ms": round(pipe, 4), "md5": md5[:12]}
        ctx.close()
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        for v in sys.argv[1:] or ["head"]:
            lib = os.path.join(ROOT, "grok_amd", "lib", "libgrok_amd.so") if v == "head" else os.path.join(ROOT, "build", "abl", v, "libgrok_amd.so")
            r = subprocess.run([sys.executable, __file__, "--one", lib], capture_output=True, text=True)
            print(v, (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-400:]])[-1], flush=True)
