#!/bin/bash
# Kernel timeline of the pipelined 8K encode as bench.py runs it (dev tool, GPU box): tools/timeline.sh [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl
env "$@" timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o p --output-format csv -- python $R/bench.py --steps 12 --no-cpu-baseline --no-host-boundary --no-workloads > /tmp/tl.log 2>&1 || tail -5 /tmp/tl.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the longest run of back-to-back steps; take the encode launches before the decode kernels start
enc = [r for r in rows if "dwt" in r["Kernel_Name"] and "idwt" not in r["Kernel_Name"] or "ht_encode" in r["Kernel_Name"]]
# find the last level-0 launches (grid y largest) and print three steps in the middle of the timed run
l0 = [i for i, r in enumerate(enc) if "<3, 1" in r["Kernel_Name"] or "false, 3, 1" in r["Kernel_Name"]]
mid = l0[int(__import__('os').environ.get('TL_FRAME', '8'))] if len(l0) > 8 else 0       # (frames 3 .. 14 are bench.py's timed, pipelined region)
sel = enc[mid:mid + 30]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    name = r["Kernel_Name"].split("(")[0].replace("grk_amd::(anonymous namespace)::", "")[:40]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-40s q%-3s grid %6d x %4d  start %8.1f  end %8.1f  dur %7.1f us" % (name, r.get("Queue_Id", "?"), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
          int(r["Grid_Size_Y"]), s / 1e3, e / 1e3, (e - s) / 1e3))
PY
