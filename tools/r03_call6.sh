#!/bin/bash
mkdir -p gpurun_out/r3c6
cd /root/repo
timeout 600 python bench.py > gpurun_out/r3c6/bench.json 2> gpurun_out/r3c6/bench.err; echo "bench rc $?" >> gpurun_out/r3c6/bench.err
tail -2 gpurun_out/r3c6/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c6/bench.json'))
print(d["value"], d["ms_per_step"])
print(json.dumps(d["decode"])[:1800])
print(json.dumps(d["host_boundary"]["via_grok_plugin"])[:400])
for k,v in d["workloads"].items(): print(k, v.get("ms_per_step"), v.get("value"))
PY
