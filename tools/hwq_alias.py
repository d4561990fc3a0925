"""Does the slowdown of the encode pipeline on GPU_MAX_HW_QUEUES != 4 beside an RCCL exchange come from WHICH hardware queues the encoder's
three streams get -- i.e. from how many other streams of the process were given a queue before them?  Child processes with the variable
set; each makes K extra streams (a small kernel on each: the runtime creates a stream's hardware queue at its first use), BEFORE (or AFTER)
the encoder's context, keeps them busy with a trickle of small copies (or not), and times pipelined 8K encodes.  python tools/hwq_alias.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch, grok_amd as G, synth  # noqa: E401
    k, when, trickle = int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    px = synth.g2(3, 8192, 8192, 8)
    p = G.TileParams.make(8192, 8192, 3, 8, 5)
    d = torch.from_numpy(px.reshape(-1)).cuda()
    extra = []

    def make_extra():
        for _ in range(k):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                torch.zeros(1024, device="cuda").add_(1)
            extra.append(s)
        torch.cuda.synchronize()
    if when == "before":
        make_extra()
    ctx = G.Context(0, verbose=bool(int(os.environ.get("HWQ_VERBOSE", "0"))))
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_pipelining(2)
    with torch.cuda.stream(stream):
        for _ in range(20):
            ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
    torch.cuda.synchronize()
    if when == "after":
        make_extra()
    small = [torch.zeros(4096, device="cuda") for _ in extra]
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for i in range(40):
                ctx.encode_tiles(p, 1, d.data_ptr(), True, fetch=False)
                if trickle:
                    for s, b in zip(extra, small):
                        with torch.cuda.stream(s):
                            b.add_(1)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    print("%.4f/%d" % (best, G.lib().grk_amd_stream_probe_result(ctx._h)))
    sys.exit(0)

for q in (4, 8):
    for when in ("before",):
        for trickle in (0,):
            row = []
            for k in (0, 1, 2, 3, 4, 5, 6, 7, 8, 12):
                env = dict(os.environ, GPU_MAX_HW_QUEUES=str(q))
                env.setdefault("GRK_AMD_STREAM_PROBE", "1")
                r = subprocess.run([sys.executable, __file__, "child", str(k), when, str(trickle)], env=env, capture_output=True, text=True, timeout=300)
                row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "fail")
            print("queues %d, probe %s, extra streams made %-6s the context, %s: ms per 8K frame / side streams replaced, by number of extra streams 0 1 2 3 4 5 6 7 8 12: %s"
                  % (q, os.environ.get("GRK_AMD_STREAM_PROBE", "1"), when, "a small kernel on each per frame" if trickle else "idle", " ".join(row)), flush=True)
