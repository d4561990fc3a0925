cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in 0 4 6; do
rm -rf /tmp/hq$k
GPU_MAX_HW_QUEUES=4 timeout 300 rocprofv3 --kernel-trace -d /tmp/hq$k -o p --output-format csv -- python $R/tools/hwq_alias.py child $k before 0 > /tmp/hq$k.log 2>&1
echo "== k=$k: $(grep -E '^[0-9.]+$' /tmp/hq$k.log | tail -1) ms"
t=$(find /tmp/hq$k -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
sq=collections.defaultdict(set); names=collections.defaultdict(collections.Counter)
for r in rows:
    sq[r['Stream_Id']].add(r['Queue_Id']); names[r['Stream_Id']][r['Kernel_Name'].split('(')[0][-28:]]+=1
for s in sorted(sq, key=int):
    print("   stream", s, "queue", sorted(sq[s]), dict(names[s].most_common(2)))
PY
done
