#!/bin/bash
# SQ counters of the Part-1 block decoder on the cfg5 stream (dev tool, GPU box): instruction mix per kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pk8
PROF_WORKLOAD=cfg5 PROF_N=2 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU -d /tmp/pk8 -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pk8.log 2>&1 || tail -3 /tmp/pk8.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pk8/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    for fam in ("t1_dec_kernel", "t1_lanes_kernel", "t1_recon_kernel"):
        if fam in k: acc[(fam, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (f, c), v in sorted(acc.items()): print("%-16s %-24s mean %16.1f  (%d launches)" % (f, c, sum(v) / len(v), len(v)))
PY
