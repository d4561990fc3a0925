cd $GRAFT_REPO_ROOT
true
for q in 4 8; do for pr in 1 0; do
echo "forced dist, queues $q, probe $pr: $(GRK_AMD_STREAM_PROBE=$pr GPU_MAX_HW_QUEUES=$q GROK_AMD_FORCE_DIST=1 timeout 300 python bench.py --exchange counts --no-cpu-baseline --no-workloads --no-host-boundary --no-live-pmc --steps 40 2>/dev/null | grep -h '"metric"' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); e=d.get("exchange"); print(d["ms_per_step"], "counts", e["counts"]["ms_per_step"], "gather", e["gather"]["ms_per_step"])')"
done; done
for q in 4 8; do
echo "plain bench, queues $q: $(GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-workloads --no-host-boundary --no-live-pmc --steps 40 2>/dev/null | grep -h '"metric"' | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"])')"
done
