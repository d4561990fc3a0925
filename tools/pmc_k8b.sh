#!/bin/bash
# second counter pass over the Part-1 block decoder (dev tool, GPU box): instruction cache, waits, issue
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pk8b
PROF_WORKLOAD=cfg5 PROF_N=2 timeout 300 rocprofv3 --pmc ${1:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY} -d /tmp/pk8b -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pk8b.log 2>&1 || tail -3 /tmp/pk8b.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pk8b/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "t1_dec_kernel" in k: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()): print("t1_dec_kernel %-28s mean %16.1f  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
