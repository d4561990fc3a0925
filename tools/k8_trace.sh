#!/bin/bash
# per-kernel durations of the cfg5 decode (dev tool, GPU box): tools/k8_trace.sh [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
env "$@" PROF_WORKLOAD=cfg5 PROF_N=4 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
echo "== cfg5 decode, rocprofv3 --kernel-trace --stats, 4 frames; env: $*"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("t1_", "idwt", "dec_upload")):
        import re
        m = re.search(r"(t1_\w+|idwt\w+|dec_upload_kernel)(<[^>]*>)?", n)
        short = (m.group(1) + (m.group(2) or "")) if m else n[:34]
        print("%-34s calls %4s  avg %10.3f ms  min %10.3f  max %10.3f" % (short, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
