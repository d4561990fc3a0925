#!/bin/bash
mkdir -p gpurun_out/r3c3
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r3c3/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3c3/pytest.log
tail -30 gpurun_out/r3c3/pytest.log
