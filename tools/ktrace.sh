#!/bin/bash
# Per-launch kernel durations of one step (dev tool, GPU box): tools/ktrace.sh [env assignments...] -> the launches of the last
# encode (and decode) step of tools/prof_run.py, kernels one at a time (GRK_AMD_OVERLAP=0), in issue order.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt
env GRK_AMD_OVERLAP=0 PROF_N=${PROF_N:-6} "$@" timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/kt.log 2>&1 || tail -5 /tmp/kt.log
python3 - <<'PY'
import csv, glob, re, collections
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(k):
    m = re.search(r"(\w+)<([^>]*)>", k)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else k.replace("grk_amd::(anonymous namespace)::", "").split("(")[0][:40]
seq = [(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
# mean duration per (kernel, grid) over all steps, listed in first-occurrence order
acc = collections.OrderedDict()
for k, gx, gy, gz, us in seq:
    acc.setdefault((k, gx, gy, gz), []).append(us)
for (k, gx, gy, gz), v in acc.items():
    v = sorted(v)
    print("%-58s grid %5d x %4d x %3d  n %3d  median %8.1f us  min %8.1f" % (k[:58], gx, gy, gz, len(v), v[len(v) // 2], v[0]))
PY
