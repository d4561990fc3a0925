#!/bin/bash
# The round's profile set, on the GPU box: tools/prof_round.sh r02 [stats] [pmc] [sq]  -> gpurun_out/prof/<round>_*
# (copy what is to be judged into profiles/).  Counter passes are separate runs with nothing but --pmc.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
WHAT="$*"; [ -z "$WHAT" ] && WHAT="stats pmc sq"
OUT=$R/gpurun_out/prof; mkdir -p $OUT
if [[ "$WHAT" == *stats* ]]; then
  # per-kernel durations of the bench command itself: as the timed region runs (with the other workloads: cfg2, cfg3, cfg4tile,
  # cfg5 -- and the CPU baseline, which writes the cfg5 stream), and every kernel alone (what the roofline figures are quoted on)
  rm -rf /tmp/ks1 /tmp/ks2
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/ks1 -o p --output-format csv -- python $R/bench.py --steps 20 --no-host-boundary > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/ks1.err
  cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
  # (the 8K workload only: every ht_encode_kernel launch of this run is the one-launch-of-all-blocks form the roofline is quoted on,
  #  so the csv's average for that kernel IS roofline.avg_launch_ms of the line beside it)
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o p --output-format csv -- python $R/bench.py --steps 20 --no-overlap --no-host-boundary --no-workloads > $OUT/${TAG}_bench_under_rocprof_no_overlap.json 2> /tmp/ks2.err
  cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_no_overlap.csv
  tail -n 2 /tmp/ks1.err; tail -n 2 /tmp/ks2.err
fi
if [[ "$WHAT" == *pmc* ]]; then
  export GRK_AMD_OVERLAP=0 PROF_N=4
  for WL in ${PROF_WLS:-8k cfg3 cfg5}; do
    export PROF_WORKLOAD=$WL PROF_DECODE=1
    for C in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pm_$C
      timeout 400 rocprofv3 --pmc $C -d /tmp/pm_$C -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pm_$C.log 2>&1 || tail -3 /tmp/pm_$C.log
    done
    python3 $R/profiles/summarize_pmc.py $(find /tmp/pm_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
        $(find /tmp/pm_WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/${TAG}_pmc_traffic_$WL.json > $OUT/${TAG}_pmc_traffic_$WL.txt 2>&1
  done
fi
if [[ "$WHAT" == *sq* ]]; then
  unset PROF_WORKLOAD
  PROF_DECODE=1 bash $R/tools/pmc_k3.sh > $OUT/${TAG}_sq_summary_8k.txt 2>&1
fi
ls -la $OUT
if [[ "$WHAT" == *decstats* ]]; then
  # the decode kernels alone (GRK_AMD_OVERLAP=0, one frame at a time): the csv K5's and the inverse DWT's fractions can be recomputed from
  rm -rf /tmp/ks3
  GRK_AMD_OVERLAP=0 PROF_N=10 PROF_DECODE=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/ks3 -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/ks3.log 2>&1
  cp $(find /tmp/ks3 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats_decode_no_overlap.csv
  tail -n 2 /tmp/ks3.log
fi
