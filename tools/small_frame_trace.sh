#!/bin/bash
# per-kernel durations of 100 pipelined encodes of an S x S x 3 frame (dev tool, GPU box): tools/small_frame_trace.sh S
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/sft
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/sft -o p --output-format csv -- python $R/tools/small_frame_trace.py $1 > /tmp/sft.log 2>&1
python3 - "$(find /tmp/sft -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = 0
for r in rows[:12]:
    m = re.search(r"(\w+_kernel\w*)(<[^>]*>)?", r["Name"]); name = (m.group(1) + (m.group(2) or "")) if m else r["Name"][:40]
    print("  %-44s calls %4s avg %8.1f us  min %8.1f  max %8.1f" % (name[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)); tot += float(r["TotalDurationNs"])
print("  sum of kernel time per frame: %.1f us" % (tot / 100 / 1e3))
PY
