#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
env PROF_N=4 PROF_DECODE=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/kt -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/kt.log 2>&1 || tail -5 /tmp/kt.log
ls /tmp/kt/*/ | head
python3 $R/tools/timeline.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) 14
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*memory_copy_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows[-12:]:
        print(r.get("Direction"), r.get("Bytes") or r.get("Size"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us  start", int(r["Start_Timestamp"]) % 10**9 / 1e3)
PY
