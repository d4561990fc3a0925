#!/bin/bash
mkdir -p gpurun_out/r3c4
cd /root/repo
timeout 300 python tools/pcie_time.py > gpurun_out/r3c4/pcie.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_node.py tests/test_gpu_plugin.py -q > gpurun_out/r3c4/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3c4/pytest.log
cat gpurun_out/r3c4/pcie.txt; tail -8 gpurun_out/r3c4/pytest.log
