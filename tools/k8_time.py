"""cfg5 (8192^2 x 3 12-bit Part-1 + ICT + 9/7) decode timing on the GPU box (dev tool): K8 / K8L variants through env knobs.
python tools/k8_time.py [S]   -- prints ms per frame, K8 family ms (timer 5), pixels == grk_decompress"""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
VARIANTS = [("lanes", {}), ("k8 only", {"GRK_AMD_T1_LANES": "0"})] + \
    [("share %s" % v, {"GRK_AMD_T1_TAIL_SHARE": v}) for v in ("0.2", "0.3", "0.4", "0.5", "0.6")]
if os.environ.get("K8_VARIANTS"):
    VARIANTS = [(v, dict(kv.split("=") for kv in v.split(",") if kv)) for v in os.environ["K8_VARIANTS"].split(";")]
if os.environ.get("K8_CHILD"):
    import numpy as np, torch, grok_amd as G, synth, j2kparse as J, refharness as R
    S = int(os.environ.get("K8_SIZE", "8192"))
    path = "/tmp/cfg5_%d.j2k" % S
    if not os.path.exists(path):
        R.lib(threads=os.cpu_count() or 1)
        cs, _ = R.encode(synth.g2(3, S, S, 12), 12, numres=6, mode=1, ht=0, irrev=1)
        open(path, "wb").write(cs)
    cs = open(path, "rb").read()
    refp = "/tmp/cfg5_%d.ref.npy" % S
    if not os.path.exists(refp):
        R.lib(threads=os.cpu_count() or 1)
        np.save(refp, R.decode(cs, 3, S, S))
    ref = np.load(refp)
    info = J.parse(cs)
    p = G.TileParams.make(S, S, 3, 12, info["levels"], irreversible=True, mct=True, part1=True)
    blocks, _ = G.tile_layout(p)
    rows, data = J.decode_table(info, blocks, True)
    table = np.array(rows, dtype=G.capi.CODED_DTYPE)
    ctx = G.Context(0)
    ctx.set_decode_qcd([(e << 11) | m for e, m in info["qcd"]])
    d_c = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).cuda()
    d_out = torch.zeros(3 * S * S, dtype=torch.int16, device="cuda")
    for _ in range(2):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
    torch.cuda.synchronize()
    ctx.decode_status()
    ctx.enable_timing(True)
    n = 4
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    got = d_out.cpu().numpy().view(np.uint16).reshape(3, S, S).astype(np.int32)
    print(json.dumps({"ms_per_frame": round(ms, 3), "k8_ms": round(ctx.kernel_ms(5)[0], 3), "equal": bool(np.array_equal(got, ref)),
                      "diff": int((got != ref).sum())}))
    sys.exit(0)
S = sys.argv[1] if len(sys.argv) > 1 else "8192"
for name, env in VARIANTS:
    e = dict(os.environ); e.update(env); e["K8_CHILD"] = "1"; e["K8_SIZE"] = S
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True)
    print("%-22s %s" % (name, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED rc %d: %s" % (r.returncode, r.stderr[-600:])))
