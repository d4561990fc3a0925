#!/bin/bash
# per-kernel durations of the cfg5 decode (dev tool, GPU box): tools/k8_trace.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
env "$@" PROF_WORKLOAD=cfg5 PROF_N=4 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
echo "== $*"; grep -i "t1_\|idwt" $f | awk -F'","' '{gsub(/"/,"",$1); n=split($1,a,"::"); printf "%-60s calls %s avg %.3f ms\n", substr(a[n],1,60), $2, $4/1e6}'
