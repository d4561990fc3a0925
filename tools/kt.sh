#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $R/tools/prof_run.py > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' || tail -5 /tmp/kt.log
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("grk_amd::(anonymous namespace)::", "").replace("grk_amd::", "")
    if "at::native" in n or "rocclr" in n: continue
    print("%-70s calls %4s avg_us %10.1f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
