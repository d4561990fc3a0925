#!/bin/bash
# Runs on the GPU box: the two HBM-traffic PMC passes of tools/profile_round.sh alone (FETCH_SIZE and WRITE_SIZE in SEPARATE runs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof8; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_$c.csv || tail -5 /tmp/pmc_$c.log
done
ls -la $O
