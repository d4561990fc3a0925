# The encode pipeline beside an RCCL exchange (bench.py at world size 1 over RCCL) on 4 and on 8 hardware queues: per-kernel stats and the
# tail of the kernel trace (queue ids, start / end) of both into gpurun_out/ -- what profiles/r06_hw_queues.txt is made from.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for q in 4 8; do
rm -rf /tmp/qt$q
GPU_MAX_HW_QUEUES=$q GROK_AMD_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/qt$q -o p --output-format csv -- python $R/bench.py --exchange counts --no-cpu-baseline --no-workloads --no-host-boundary --no-live-pmc --steps 40 > /tmp/qt$q.log 2>&1
echo "== queues $q: $(grep -h '"metric"' /tmp/qt$q.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d.get("exchange"))')"
f=$(find /tmp/qt$q -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/qt$q -name "*kernel_trace.csv" | head -1)
(head -1 $t; tail -4000 $t) > $R/gpurun_out/queue_trace_q$q.csv
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:9]:
    print("  %-60s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
