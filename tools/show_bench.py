"""Prints the interesting fields of bench.py JSON lines: python tools/show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, "value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    mg = d.get("multi_gpu")
    if mg:
        for k, v in mg.items():
            print("   ", k, json.dumps(v))
        print("    assembled", d["config"].get("assembled_codestream_bytes"))
    for k, v in (d.get("workloads") or {}).items():
        print("    wl", k, v.get("ms_per_step"), v.get("value"), {kk: (vv.get("avg_ms"), vv.get("frac")) for kk, vv in (v.get("kernels") or {}).items()}, v.get("error"))
