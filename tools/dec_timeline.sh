#!/bin/bash
# Timeline of back-to-back 8K decode calls (dev tool, GPU box): kernels and memory copies of the last calls of tools/dec_time.py,
# with the gaps between them: tools/dec_timeline.sh [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/dtl
env PROF_N=${PROF_N:-8} "$@" timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/dtl -o p --output-format csv -- python $R/tools/dec_time.py > /tmp/dtl.log 2>&1 || tail -5 /tmp/dtl.log
grep "ms/frame" /tmp/dtl.log
python3 - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/dtl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ("q%s " % r.get("Queue_Id", "?")) + r["Kernel_Name"].replace("void ", "").replace("grk_amd::(anonymous namespace)::", "").split("(")[0][:40]))
for f in glob.glob("/tmp/dtl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "?"))))
ev.sort()
# the last two decode calls: from the second-last ht_dec_prep_kernel on
starts = [i for i, e in enumerate(ev) if "ht_dec_prep" in e[2]]
i0 = starts[-2] - 4 if len(starts) >= 2 else 0
t0, prev = ev[i0][0], ev[i0][0]
for s, e, n in ev[i0:]:
    print("%-50s start %9.1f  dur %8.1f  gap %7.1f us" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = max(prev, e)
PY
