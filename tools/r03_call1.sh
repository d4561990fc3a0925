#!/bin/bash
# round 3, GPU call 1: K3 A/B (r02 library, new default, one switch off each), the GPU test suite, the bench line
mkdir -p gpurun_out/r3c1
cd /root/repo
timeout 600 python tools/k3_time.py r02 head dpp1 dpp0 spec0 skip0 > gpurun_out/r3c1/k3_ab.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3c1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r3c1/pytest.log
timeout 600 python bench.py > gpurun_out/r3c1/bench.json 2> gpurun_out/r3c1/bench.err; echo "bench rc $?" >> gpurun_out/r3c1/bench.err
cat gpurun_out/r3c1/k3_ab.txt; tail -5 gpurun_out/r3c1/pytest.log; tail -c 1500 gpurun_out/r3c1/bench.json
