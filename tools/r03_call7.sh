#!/bin/bash
mkdir -p gpurun_out/r3c7
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_at_size.py tests/test_gpu_plugin.py -q -x > gpurun_out/r3c7/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3c7/pytest.log
timeout 600 python bench.py > gpurun_out/r3c7/bench.json 2> gpurun_out/r3c7/bench.err; echo "bench rc $?" >> gpurun_out/r3c7/bench.err
tail -4 gpurun_out/r3c7/pytest.log; tail -1 gpurun_out/r3c7/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c7/bench.json'))
print(d["value"], d["ms_per_step"], d["decode"]["ms_per_step"])
print(json.dumps(d["workloads"]["cfg5"]))
PY
