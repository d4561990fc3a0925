#!/bin/bash
# one SQ pass (instruction mix + busy cycles) over the 8K encode, kernels alone, for the library in GRK_AMD_LIB: per-block means of K3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GRK_AMD_OVERLAP=0 PROF_DECODE=0 PROF_N=3
for P in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_IFETCH"; do
  rm -rf /tmp/pks
  timeout 150 rocprofv3 --pmc $P -d /tmp/pks -o p --output-format csv -- python $R/tools/prof_run.py > /tmp/pks.log 2>&1
  f=$(find /tmp/pks -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { tail -3 /tmp/pks.log; exit 1; }
  python3 - $f <<'PY'
import csv, collections, sys
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ht_encode_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join("%s %.0f" % (c.replace("SQ_",""), sum(v)/3/49152) for c,v in sorted(acc.items())))
PY
done
