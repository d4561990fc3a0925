#!/bin/bash
# Runs on the GPU box: kernel-trace stats of bench.py + HBM traffic PMC passes (FETCH_SIZE and
# WRITE_SIZE in SEPARATE runs, MI355X_MICROARCH.md) of the 8K encode+decode; results -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 10 > $O/bench_under_rocprof.json 2> /tmp/kt.log
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv || tail -5 /tmp/kt.log
# the same with every kernel alone on the GPU: the durations bench.py's roofline figures are about
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-overlap --steps 10 > $O/bench_under_rocprof_no_overlap.json 2> /tmp/kt2.log
f=$(find /tmp/kt2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_no_overlap.csv || tail -5 /tmp/kt2.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_$c.csv || tail -5 /tmp/pmc_$c.log
done
ls -la $O; tail -1 $O/bench_under_rocprof.json | cut -c1-300
