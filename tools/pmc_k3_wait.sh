#!/bin/bash
# one SQ pass (wait / LDS counters) over the 8K encode, kernels alone, for the library in GRK_AMD_LIB: per-block means of K3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GRK_AMD_OVERLAP=0 PROF_DECODE=0 PROF_N=3
rm -rf /tmp/pkw
timeout 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pkw -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pkw.log 2>&1
f=$(find /tmp/pkw -name "*counter_collection.csv" | head -1)
[ -z "$f" ] && { tail -3 /tmp/pkw.log; exit 1; }
python3 - $f <<'PY'
import csv, collections, sys
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ht_encode_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join("%s %.0f" % (c.replace("SQ_",""), sum(v)/3/49152) for c,v in sorted(acc.items())))
PY
