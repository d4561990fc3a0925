#!/bin/bash
# tools/corun.sh: K3 / DWT durations with a copy loop, an ALU loop or nothing beside them (see corun.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in none copy alu; do
  rm -rf /tmp/cr
  CORUN=$m GRK_AMD_OVERLAP=0 timeout 200 rocprofv3 --kernel-trace -d /tmp/cr -o p --output-format csv -- python $R/tools/corun.py > /tmp/cr.log 2>&1 || tail -3 /tmp/cr.log
  python3 - $m <<'PY'
import csv, glob, sys, collections
f = glob.glob("/tmp/cr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for r in rows[len(rows) // 3:]:
    k = r["Kernel_Name"]
    n = "K3" if "ht_encode_kernel" in k else "dwt L0" if "dwt53_pk_kernel<3" in k else "dwt L1+" if "dwt" in k else "other:" + k.split("<")[0][-28:]
    acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[1], {k: (len(v), round(sorted(v)[len(v) // 2], 1)) for k, v in acc.items() if len(v) >= 3})
PY
done
