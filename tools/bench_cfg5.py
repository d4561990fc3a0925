"""BASELINE configs[4] shape: a 12-bit RGB image coded by the REFERENCE as classic Part-1 (EBCOT) + ICT + 9/7,
decoded on the GPU (K8 MQ decode + ScaleFilter, K6 inverse 9/7, K7 inverse ICT) and by grk_decompress on the
host cores; checks pixel equality and prints one JSON line.  Needs oracle/_ref (test infrastructure) for the
stream and the CPU side.   usage: python tools/bench_cfg5.py [size] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import grok_amd.capi as _capi
if os.environ.get("AB_LIB"):
    _capi.lib_path = lambda: os.environ["AB_LIB"]
import grok_amd as G, synth, refharness as R, j2kparse as J

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prec, C, numres = 12, 3, 6
R.lib(threads=os.cpu_count() or 1)
px = synth.g2(C, S, S, prec)
t0 = time.time(); cs, _ = R.encode(px, prec, numres=numres, mode=1, ht=0, irrev=1); t_enc = time.time() - t0
cpu = []
for _ in range(3):
    t0 = time.time(); ref = R.decode(cs, C, S, S); cpu.append(time.time() - t0)
t0 = time.time(); info = J.parse(cs); t_parse = time.time() - t0
p = G.TileParams.make(S, S, C, prec, info["levels"], irreversible=True, mct=True, part1=True)
blocks, _ = G.tile_layout(p)
rows, data = J.decode_table(info, blocks, True)
table = np.array(rows, dtype=G.capi.CODED_DTYPE)
ctx = G.Context(0)
ctx.set_decode_qcd([(e << 11) | m for e, m in info["qcd"]])
d_c = torch.from_numpy(np.frombuffer(data, np.uint8).copy()).cuda()
d_out = torch.zeros(C * S * S, dtype=torch.int16, device="cuda")
for _ in range(2):
    ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
ctx.decode_status(); ctx.enable_timing(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    ctx.decode_device(p, 1, table, d_c.data_ptr(), d_c.numel(), d_out.data_ptr())
ctx.synchronize(); dt = (time.perf_counter() - t0) / steps
got = d_out.cpu().numpy().view(np.uint16).reshape(C, S, S).astype(np.int32)
print(json.dumps({
    "workload": "%dx%dx3 12-bit, Part-1 EBCOT + ICT + 9/7 coded by grk_compress (%d bytes), decode" % (S, S, len(cs)),
    "gpu_ms": round(dt * 1e3, 3), "gpu_Mpixels_s": round(S * S / dt / 1e6, 1),
    "kernels_ms": {"t1_ebcot_decode": round(ctx.kernel_ms(5)[0], 3), "idwt97": round(ctx.kernel_ms(6)[0], 3),
                   "egress_ict": round(ctx.kernel_ms(7)[0], 3)},
    "cpu_grk_decompress_s": round(sorted(cpu)[1], 3), "cpu_Mpixels_s": round(S * S / sorted(cpu)[1] / 1e6, 1),
    "cpu_threads": os.cpu_count(), "pixels_equal_grk_decompress": bool(np.array_equal(got, ref)),
    "max_abs_err_vs_source": int(np.abs(ref - px.astype(np.int32)).max()),
    "host_side": {"ref_encode_s": round(t_enc, 2), "t2_reader_s": round(t_parse, 2)}}))
