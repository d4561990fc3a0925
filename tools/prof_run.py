"""Small driver for rocprofv3 passes (dev tool): PROF_N steps of one BASELINE workload, kernels as the bench runs them.
PROF_WORKLOAD = 8k (default) | cfg2 | cfg3 | cfg4tile | cfg5 ; PROF_DECODE=1 adds the HT decode of the 8k / cfg2 result.
cfg5 (Part-1 + ICT + 9/7 decode) needs a stream only the reference's encoder can write: made once per box with
oracle/_ref and cached in /tmp/cfg5_<S>.j2k (PROF_SIZE, default 8192)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, grok_amd as G, synth
wl = os.environ.get("PROF_WORKLOAD", "8k")
if wl == "8k_int32":          # the headline frame on the reference's int32 planes (bench.py workloads.8k_int32)
    os.environ["GRK_AMD_DWT_PK"] = "0"; os.environ["GRK_AMD_PLANES16"] = "0"
n = int(os.environ.get("PROF_N", "4"))
S = int(os.environ.get("PROF_SIZE", "8192"))
ctx = G.Context(0)
if wl == "cfg5":
    import j2kparse as J
    path = "/tmp/cfg5_%d.j2k" % S
    if not os.path.exists(path):
        import refharness as R
        R.lib(threads=os.cpu_count() or 1)
        cs, _ = R.encode(synth.g2(3, S, S, 12), 12, numres=6, mode=1, ht=0, irrev=1)
        open(path, "wb").write(cs)
    cs = open(path, "rb").read()
    info = J.parse(cs)
    p = G.TileParams.make(S, S, 3, 12, info["levels"], irreversible=True, mct=True, part1=True)
    blocks, _ = G.tile_layout(p)
    rows, data = J.decode_table(info, blocks, True)
    table = np.array(rows, dtype=G.capi.CODED_DTYPE)
    ctx.set_decode_qcd([(e << 11) | m for e, m in info["qcd"]])
    d_c = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).cuda()
    d_out = torch.zeros(3 * S * S, dtype=torch.int16, device="cuda")
    for _ in range(n):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
    ctx.decode_status()
    print("done cfg5")
    sys.exit(0)
shape = {"8k": (3, S, S, 8, 5, 1, False), "8k_int32": (3, S, S, 8, 5, 1, False), "cfg2": (3, 4096, 4096, 8, 5, 1, False),
         "cfg3": (3, 8192, 8192, 16, 5, 1, True), "cfg4tile": (3, 1024, 1024, 8, 5, 64, False), "cfg4": (3, 1024, 1024, 8, 5, 256, False)}[wl]
Cn, W, H, prec, L, nt, irrev = shape
px = synth.g2(Cn, H, W, prec)
if os.environ.get("PROF_FLAT"):                       # a flat frame of that value (K3's other extreme: nothing to code)
    px = np.full_like(px, int(os.environ["PROF_FLAT"]))
p = G.TileParams.make(W, H, Cn, prec, L, irreversible=irrev)
host = np.ascontiguousarray(np.broadcast_to(px.reshape(1, -1), (nt, px.size))).reshape(-1)
d = torch.from_numpy(host.view(np.uint8).copy()).cuda()
if os.environ.get("PROF_PIPELINE") == "1":          # as bench.py's timed region runs them: consecutive frames pipelined
    ctx.set_pipelining(True)
for _ in range(n):
    ctx.encode_tiles(p, nt, d.data_ptr(), True, fetch=False)
ctx.synchronize()
if os.environ.get("PROF_PIPELINE") == "1":
    ctx.set_pipelining(False)
if os.environ.get("PROF_DECODE", "1") == "1" and not irrev:
    nb = G.lib().grk_amd_tile_num_blocks(p) * nt
    table, tot = ctx.fetch_table(nb)
    back = torch.empty_like(d)
    for _ in range(n):
        ctx.decode_device(p, nt, table, ctx.coded_device_ptr(), tot, back.data_ptr())
    ctx.decode_status()
    assert torch.equal(back, d)
print("done", wl)
