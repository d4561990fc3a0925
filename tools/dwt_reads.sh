#!/bin/bash
# the DWT's read amplification by row-segment length (GRK_AMD_DWT_MIN_WGS picks the segments: run_dwt, context.hip).
# FETCH_SIZE / WRITE_SIZE of the 8K encode's DWT launches, kernels alone, per minimum-workgroup setting.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GRK_AMD_OVERLAP=0 PROF_DECODE=0 PROF_N=4 PROF_WORKLOAD=8k
for m in 4096 2048 1024; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p1
    GRK_AMD_DWT_MIN_WGS=$m timeout 150 rocprofv3 --pmc $c -d /tmp/p1 -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/p1.log 2>&1
    f=$(find /tmp/p1 -name "*counter_collection.csv" | head -1)
    [ -z "$f" ] && { tail -3 /tmp/p1.log; continue; }
    python3 - $f $m $c <<'PY'
import csv, collections, sys
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if r["Counter_Name"] != sys.argv[3] or "idwt" in k or not ("dwt53_pk_kernel" in k or "dwt_level_kernel" in k): continue
    fam = "level0" if ("dwt53_pk_kernel<3" in k or "dwt_level_kernel<false, 3" in k) else "levels>=1"
    acc[fam] += float(r["Counter_Value"]); n[fam] += 1
for fam in sorted(acc):
    mb = acc[fam] / 4 * 1024 / 1e6 * (2 if sys.argv[3] == "FETCH_SIZE" else 1)
    print("min_wgs %-5s %-10s %-10s launches/step %.1f  %8.1f MB per step%s" % (sys.argv[2], sys.argv[3], fam, n[fam] / 4, mb, " (x2)" if sys.argv[3] == "FETCH_SIZE" else ""))
PY
  done
done
