cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for q in 4 8; do
rm -rf /tmp/qt$q
GPU_MAX_HW_QUEUES=$q GROK_AMD_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/qt$q -o p --output-format csv -- python $R/bench.py --exchange counts --no-cpu-baseline --no-workloads --no-host-boundary --no-live-pmc --steps 40 > /tmp/qt$q.log 2>&1
echo "== queues $q: $(tail -1 /tmp/qt$q.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d.get("exchange"))')"
f=$(find /tmp/qt$q -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:9]:
    print("  %-60s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
