#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof8; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 10 > $O/bench_under_rocprof.json 2> /tmp/kt.log
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv || tail -5 /tmp/kt.log
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-overlap --steps 10 > $O/bench_under_rocprof_no_overlap.json 2> /tmp/kt2.log
f=$(find /tmp/kt2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_no_overlap.csv || tail -5 /tmp/kt2.log
ls -la $O
