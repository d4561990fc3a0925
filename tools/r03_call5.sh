#!/bin/bash
mkdir -p gpurun_out/r3c5
cd /root/repo
timeout 300 python tools/dec_time.py > gpurun_out/r3c5/dec_time.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_decode.py tests/test_gpu_ht_refine.py tests/test_gpu_offgrid.py tests/test_gpu_plugin.py -q -x > gpurun_out/r3c5/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3c5/pytest.log
cat gpurun_out/r3c5/dec_time.txt; tail -3 gpurun_out/r3c5/pytest.log
