#!/bin/bash
# Runs on the GPU box: kernel trace of a short bench run, then the start/end of every kernel of the last frames
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/timeline; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace -d /tmp/tl -o tl --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 8 --warmup 3 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
[ -z "$f" ] && { tail -5 /tmp/tl.log; exit 1; }
python3 - "$f" <<'PY' | tee $O/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the timed encode region: the densest run of dwt level-0 kernels; print the 3 frames before the last ingest-free gap
def short(n):
    for k in ("dwt_level_kernel", "ht_encode_fallback", "ht_encode_kernel", "ht_alloc", "ingest", "ht_dec_vlc", "ht_dec_ms", "idwt_level", "egress"):
        if k in n: 
            tag = k
            if k == "dwt_level_kernel": tag += "<PX>" if "true" in n.split("dwt_level_kernel")[1][:40] else ""
            return tag
    return n[:40]
enc = [i for i, r in enumerate(rows) if "ht_alloc" in r["Kernel_Name"]]
# the pipelined frames: K3 launches on a queue other than the one ht_alloc runs on; take three frames from their middle
mainq = rows[enc[0]]["Queue_Id"]
side = [i for i, r in enumerate(rows) if "ht_encode_kernel" in r["Kernel_Name"] and r["Queue_Id"] != mainq]
mid = side[len(side) // 2] if side else enc[len(enc) // 2]
k = max(j for j, i in enumerate(enc) if i <= mid)
a, b = enc[max(k - 1, 0)], enc[min(k + 2, len(enc) - 1)]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    g = r.get("Grid_Size", r.get("Grid_Size_X", ""))
    print("%9.1f %9.1f %8.1f  q%-3s %-28s grid %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"]), g))
PY
