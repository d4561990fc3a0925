#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/tools/prof_run.py > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 $f | sed 's/grk_amd::(anonymous namespace):://' | cut -c1-110 | head -12 || tail -5 /tmp/kt.log
