#!/bin/bash
mkdir -p gpurun_out/r3c2
cd /root/repo
timeout 900 python tools/k3_time.py base head dpp1 dpp0 spec0 skip0 base head > gpurun_out/r3c2/k3_ab.txt 2>&1
PROF_DECODE=0 timeout 600 bash tools/pmc_k3.sh > gpurun_out/r3c2/sq_k3.txt 2>&1
cat gpurun_out/r3c2/k3_ab.txt; grep "^ht " gpurun_out/r3c2/sq_k3.txt
