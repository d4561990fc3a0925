"""A/B of the pipelined 8K encode period and the kernel families between library builds on ONE box (dev tool):
python tools/whatif_time.py build/abl/<name>/libgrok_amd.so ...   (the working tree's own library first)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in [None] + sys.argv[1:]:
    env = dict(os.environ)
    if lib: env["GRK_AMD_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-workloads", "--no-host-boundary", "--no-live-pmc"],
                       env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-40s step %.4f ms (single buffer %.4f) | alone: dwt %.4f k3 %.4f | pipelined: dwt %.4f k3 %.4f | md5 ok %s" % (
            lib or "HEAD", d["ms_per_step"], d["config"]["single_input_buffer_ms_per_step"], d["kernels"]["dwt53_5levels"]["avg_ms"],
            d["kernels"]["ht_cleanup_encode"]["avg_ms"], d["kernels_overlapped"]["dwt53_5levels"]["avg_ms"],
            d["kernels_overlapped"]["ht_cleanup_encode"]["avg_ms"], (d.get("bit_exact") or {}).get("equals_grok_cpu_file_on_all_ranks")))
    except Exception as e:
        print(lib, "FAILED", r.returncode, r.stderr[-400:], e)
