# The file route of one 8K frame with Tier-2 on the device and on the host (bench.py's host_boundary.node_native), and the two
# Tier-2 kernels' durations from a kernel trace.  Run on the GPU box: gpurun -- 'bash tools/t2_route.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for r in device host; do
GRK_AMD_NODE_T2=$r timeout 600 python $R/bench.py --no-cpu-baseline --no-workloads --no-live-pmc --steps 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('== Tier-2 on the $r'); print(json.dumps(d['host_boundary']['node_native'], indent=1))"
done
rm -rf /tmp/t2prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/t2prof -o p --output-format csv -- python $R/tools/t2_frames.py > /tmp/t2prof.log 2>&1
tail -3 /tmp/t2prof.log
f=$(find /tmp/t2prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/t2_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("  %-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
